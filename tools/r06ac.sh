#!/bin/bash
# round 6: bench.py now asks for 8 hardware queues by default -- the full default line (all legs) twice, against GPU_MAX_HW_QUEUES=4
O=gpurun_out/r06ac; mkdir -p $O
for i in 1 2; do
  for q in 8 4; do
    GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/q${q}_$i.json
    python - <<PY
import json
d=json.load(open("$O/q${q}_$i.json"))
print("queues $q run $i value %.4e one proof %.2f ms  incl_h2d %.4e  caller_rands %.4e  quad %.4e  verify %.2f ms pin %s" % (d["value"], d["proof_wall_ms"], d["value_incl_h2d"], d["incl_h2d"]["caller_rands"]["value"], d["quad_mix"]["value"], d["config"]["verify_ms"], d["config"]["proof_equals_oracle_pin"]))
PY
  done
done | tee $O/ab.txt
