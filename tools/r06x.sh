#!/bin/bash
# round 6: does the register-light tile kernel (-DLIG_K2_CB_GLOBAL: 113 VGPRs) unlock real concurrency between a stage 1 and the other proof's stage 2
# (hashes on one lane, samplers on another; fewer sampler workgroups)?
O=gpurun_out/r06x; mkdir -p $O
one() { tag=$1; shift
  env "$@" GPU_MAX_HW_QUEUES=8 timeout 150 python bench.py --no-cpu-baseline --no-verify --no-h2d --quad-mix 0 2>/dev/null | tail -1 > $O/$tag.json
  python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); print("%-34s value %.4e  one proof %.3f ms  pin %s" % ("$tag", d["value"], d["proof_wall_ms"], d["config"].get("proof_equals_oracle_pin")))
except Exception as e: print("%-34s FAILED" % "$tag")
PY
}
CB=LIG_HIP_LIB=$PWD/tools/ab/liblig_hip_exp_cb.so
EX=LIG_HIP_LIB=$PWD/tools/ab/liblig_hip_exp.so
for i in 1 2; do
  one k2_158_one_lane_$i $EX LIG_STREAM_MAP=a01012121
  one k2_113_one_lane_$i $CB LIG_STREAM_MAP=a01012121
  one k2_113_two_lanes_512_$i $CB LIG_STREAM_MAP=a01032123
  one k2_113_two_lanes_256_$i $CB LIG_STREAM_MAP=a01032123 LIG_AES_BLOCKS=256
  one k2_113_all_apart_256_$i $CB LIG_STREAM_MAP=a01032425 LIG_AES_BLOCKS=256
  one k2_113_two_lanes_128_$i $CB LIG_STREAM_MAP=a01032123 LIG_AES_BLOCKS=128
done | tee $O/ab.txt
