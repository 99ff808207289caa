#!/bin/bash
# round 6: hashes on one shared lane, samplers on another (hash next to sampler allowed) with fewer persistent sampler workgroups (LDS room for the tiles)?
O=gpurun_out/r06v; mkdir -p $O
export LIG_HIP_LIB=$PWD/tools/ab/liblig_hip_exp.so
one() { tag=$1; shift
  env "$@" GPU_MAX_HW_QUEUES=8 timeout 150 python bench.py --no-cpu-baseline --no-verify --no-h2d --quad-mix 0 2>/dev/null | tail -1 > $O/$tag.json
  python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); print("%-34s value %.4e  one proof %.3f ms  pin %s" % ("$tag", d["value"], d["proof_wall_ms"], d["config"].get("proof_equals_oracle_pin")))
except Exception as e: print("%-34s FAILED" % "$tag")
PY
}
for i in 1 2; do
  one one_lane_$i LIG_STREAM_MAP=a01012121
  one two_lanes_512_$i LIG_STREAM_MAP=a01032123
  one two_lanes_256_$i LIG_STREAM_MAP=a01032123 LIG_AES_BLOCKS=256
  one two_lanes_128_$i LIG_STREAM_MAP=a01032123 LIG_AES_BLOCKS=128
  one one_lane_256_$i LIG_STREAM_MAP=a01012121 LIG_AES_BLOCKS=256
done | tee $O/ab.txt
