#!/bin/bash
# round 6: stages run one at a time (shared side stream) -- do larger launch groups (fewer kernel boundaries per stage) help now?
O=gpurun_out/r06y; mkdir -p $O
one() { tag=$1; shift
  env "$@" timeout 150 python bench.py --no-cpu-baseline --no-verify --no-h2d --quad-mix 0 2>/dev/null | tail -1 > $O/$tag.json
  python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); print("%-30s value %.4e  one proof %.3f ms  stages %s pin %s" % ("$tag", d["value"], d["proof_wall_ms"], [round(x,2) for x in d["config"]["stage_ms"].values()], d["config"].get("proof_equals_oracle_pin")))
except Exception as e: print("%-30s FAILED" % "$tag")
PY
}
for i in 1 2; do
  one chunk512_$i A=1
  one chunk768_$i LIG_HIP_LIB=$PWD/tools/ab/liblig_hip_chunk768.so LIG_ENCODE_CHUNK=768
  one chunk1024_$i LIG_HIP_LIB=$PWD/tools/ab/liblig_hip_chunk1024.so LIG_ENCODE_CHUNK=1024
  one chunk1024_head256_$i LIG_HIP_LIB=$PWD/tools/ab/liblig_hip_chunk1024.so LIG_ENCODE_CHUNK=1024 LIG_S1_HEAD=256 LIG_S2_HEAD=384
  one chunk2048_$i LIG_HIP_LIB=$PWD/tools/ab/liblig_hip_chunk2048.so LIG_ENCODE_CHUNK=2048
done | tee $O/ab.txt
