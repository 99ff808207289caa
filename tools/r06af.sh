#!/bin/bash
O=gpurun_out/r06af; mkdir -p $O
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
  one base_$i LIG_K1_FOLD=0
  one fold2_$i LIG_K1_FOLD=2
done | tee $O/ab.txt
