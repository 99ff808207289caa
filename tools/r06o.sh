#!/bin/bash
# round 6: which streams of the two proofs in flight should share a hardware queue?  LIG_STREAM_MAP = physical stream of
# [A.main A.side A.copy B.main B.side B.copy]; GPU_MAX_HW_QUEUES=8 so that every physical stream has a queue of its own
O=gpurun_out/r06o; mkdir -p $O
one() { # tag, map ('' = the runtime's own mapping), queues
  tag=$1; map=$2; q=$3
  if [ -n "$map" ]; then export LIG_STREAM_MAP=$map; else unset LIG_STREAM_MAP; fi
  GPU_MAX_HW_QUEUES=$q timeout 150 python bench.py --no-cpu-baseline --no-verify --no-h2d --quad-mix 0 2>/dev/null | tail -1 > $O/$tag.json
  python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json"))
    print("%-22s map %-7s queues $q  value %.4e  one proof %.3f ms  pin %s" % ("$tag", "$map" or "-", d["value"], d["proof_wall_ms"], d["config"].get("proof_equals_oracle_pin")))
except Exception as e:
    print("%-22s map %-7s queues $q  FAILED / timed out (%r)" % ("$tag", "$map" or "-", e))
PY
}
for i in 1 2; do
  one default4_$i "" 4
  one separate_$i 012345 8
  one mains_$i 012045 8
  one sides_$i 012315 8
  one Amain_Bside_$i 012305 8
  one Aside_Bmain_$i 012145 8
  one mains_and_sides_$i 012015 8
  one cross_both_$i 012105 8
done | tee $O/ab.txt
