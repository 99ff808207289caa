#!/bin/bash
# round 6: the side work of a proof split in two roles -- stage-1 column hash / stage-2 sampler -- which of them must not run next to the other proof's?
O=gpurun_out/r06t; mkdir -p $O
export LIG_HIP_LIB=$PWD/tools/ab/liblig_hip_exp.so
one() { # tag map queues inflight
  tag=$1; map=$2; q=$3; inf=$4
  if [ -n "$map" ]; then export LIG_STREAM_MAP=$map; else unset LIG_STREAM_MAP; fi
  GPU_MAX_HW_QUEUES=$q timeout 150 python bench.py --inflight $inf --no-cpu-baseline --no-verify --no-h2d --quad-mix 0 2>/dev/null | tail -1 > $O/$tag.json
  python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json"))
    print("%-30s map %-14s queues $q inflight $inf  value %.4e  one proof %.3f ms  pin %s" % ("$tag", "$map" or "-", d["value"], d["proof_wall_ms"], d["config"].get("proof_equals_oracle_pin")))
except Exception as e:
    print("%-30s map %-14s queues $q inflight $inf  FAILED / timed out" % ("$tag", "$map" or "-"))
PY
}
for i in 1 2 3; do
  one default_$i "" 4 2
  one all_side_shared_$i a01012121 8 2
  one smp_shared_hash_shared_$i a01032123 8 2
  one smp_shared_hash_separate_$i a01032124 8 2
  one hash_shared_smp_separate_$i a01032423 8 2
  one cross_$i a01032321 8 2
  one three_split_$i a010321234153 8 3
done | tee $O/ab.txt
