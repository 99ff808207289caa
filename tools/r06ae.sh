#!/bin/bash
# round 6: LIG_K1_FOLD=1 -- K1 folded into the tile kernel's load: parity (every transform size, proofs, pins), then the A/B
O=gpurun_out/r06ae; mkdir -p $O
LIG_K1_FOLD=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_zres.py tests/test_gpu_rows_api.py -q -m gpu -x > $O/parity.log 2>&1; echo "parity rc $?" | tee -a $O/parity.log
tail -3 $O/parity.log
one() { tag=$1; shift
  env "$@" timeout 150 python bench.py --no-cpu-baseline --no-verify --no-h2d --quad-mix 0 2>/dev/null | tail -1 > $O/$tag.json
  python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); print("%-12s value %.4e  one proof %.3f ms  K2 512-row %.0f us  stages %s pin %s" % ("$tag", d["value"], d["proof_wall_ms"], 1e3*d["roofline"]["launches_of_512_rows"]["avg_launch_ms"], [round(x,2) for x in d["config"]["stage_ms"].values()], d["config"].get("proof_equals_oracle_pin")))
except Exception as e: print("%-12s FAILED" % "$tag")
PY
}
for i in 1 2 3; do
  one base_$i LIG_K1_FOLD=0
  one fold_$i LIG_K1_FOLD=1
done | tee $O/ab.txt
