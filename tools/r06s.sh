#!/bin/bash
# round 6: explicit stream placements with the EXPERIMENTS build: is it the NUMBER of hardware queues or WHICH streams share?
O=gpurun_out/r06s; mkdir -p $O
export LIG_HIP_LIB=$PWD/tools/ab/liblig_hip_exp.so
one() { # tag map queues inflight
  tag=$1; map=$2; q=$3; inf=$4
  if [ -n "$map" ]; then export LIG_STREAM_MAP=$map; else unset LIG_STREAM_MAP; fi
  GPU_MAX_HW_QUEUES=$q timeout 150 python bench.py --inflight $inf --no-cpu-baseline --no-verify --no-h2d --quad-mix 0 2>/dev/null | tail -1 > $O/$tag.json
  python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json"))
    print("%-26s map %-13s queues $q inflight $inf  value %.4e  one proof %.3f ms  pin %s" % ("$tag", "$map" or "-", d["value"], d["proof_wall_ms"], d["config"].get("proof_equals_oracle_pin")))
except Exception as e:
    print("%-26s map %-13s queues $q inflight $inf  FAILED / timed out" % ("$tag", "$map" or "-"))
PY
}
for i in 1 2; do
  one default_$i "" 4 2
  one sides_share_q4_$i 010212 4 2
  one sides_share_q8_$i 010212 8 2
  one four_separate_q4_$i 010232 4 2
  one four_separate_q8_$i 010232 8 2
  one three_default_$i "" 4 3
  one three_sides_share_q4_$i 010212313 4 3
  one three_sides_share_q8_$i 010212313 8 3
  one four_sides_share_q8_$i 010212313414 8 4
done | tee $O/ab.txt
