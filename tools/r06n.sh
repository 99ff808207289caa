#!/bin/bash
# round 6: how many hardware queues should the process's streams be spread over?  (default 4; 8 measured -12 % in r06k)
O=gpurun_out/r06n; mkdir -p $O
for i in 1 2; do for q in 4 2 3 5 6; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --no-verify --no-h2d --quad-mix 0 2>/dev/null | tail -1 > $O/q${q}_$i.json
  python - <<PY
import json
d=json.load(open("$O/q${q}_$i.json"))
print("queues $q run $i value %.4e  one proof %.3f ms  K2 512-row %.1f us" % (d["value"], d["proof_wall_ms"], 1e3*d["roofline"]["launches_of_512_rows"]["avg_launch_ms"]))
PY
done; done | tee $O/ab.txt
