#!/bin/bash
# round 6: LIG_SHA_GATE=3 -- K1 of chunk b+1 on a second stream next to K3 of chunk b (both between two tile launches), against gate 1 (default) and 2
O=gpurun_out/r06ad; mkdir -p $O
one() { tag=$1; shift
  env "$@" timeout 150 python bench.py --no-cpu-baseline --no-verify --no-h2d --quad-mix 0 2>/dev/null | tail -1 > $O/$tag.json
  python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); print("%-12s value %.4e  one proof %.3f ms  K2 512-row %.0f us  pin %s" % ("$tag", d["value"], d["proof_wall_ms"], 1e3*d["roofline"]["launches_of_512_rows"]["avg_launch_ms"], d["config"].get("proof_equals_oracle_pin")))
except Exception as e: print("%-12s FAILED" % "$tag")
PY
}
for i in 1 2 3; do
  one gate1_$i LIG_SHA_GATE=1
  one gate3_$i LIG_SHA_GATE=3
  one gate2_$i LIG_SHA_GATE=2
done | tee $O/ab.txt
