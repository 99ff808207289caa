#!/bin/bash
# round 6: blind search over the runtime's stream -> hardware-queue mapping (GPU_MAX_HW_QUEUES = 4, the default): dummy streams before the
# even / odd context's streams and the creation order of {main, side, copy} (LIG_STREAM_PAD = "even,odd,order")
O=gpurun_out/r06p; mkdir -p $O
one() {
  tag=$1; pad=$2
  if [ -n "$pad" ]; then export LIG_STREAM_PAD=$pad; else unset LIG_STREAM_PAD; fi
  timeout 150 python bench.py --no-cpu-baseline --no-verify --no-h2d --quad-mix 0 2>/dev/null | tail -1 > $O/$tag.json
  python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json"))
    print("%-14s value %.4e  one proof %.3f ms  pin %s" % ("$pad" or "default", d["value"], d["proof_wall_ms"], d["config"].get("proof_equals_oracle_pin")))
except Exception as e:
    print("%-14s FAILED / timed out" % ("$pad" or "default"))
PY
}
one default_a ""
for ord in 012 102 201 021; do for a in 0 1 2 3; do for b in 0 1 2 3; do
  one p${a}_${b}_$ord "$a,$b,$ord"
done; done; done | tee $O/grid.txt
one default_b "" | tee -a $O/grid.txt
sort -k3 -g -r $O/grid.txt | head -12
