#!/bin/bash
# round 6: GPU slots A/B -- default bench (two proofs in flight) against three / four in flight with LIG_GPU_SLOTS = 2 / 3 (and without), alternating
O=gpurun_out/r06k; mkdir -p $O
run() { # name, env..., -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-cpu-baseline --no-verify --no-h2d --quad-mix 0 "$@" 2>/dev/null | tail -1 > $O/$name.json
  python - <<PY
import json
d=json.load(open("$O/$name.json"))
print("%-28s value %.4e  ms/step %.3f  inflight %s  K2 avg launch ms %.4f frac %.4f" % ("$name", d["value"], d["ms_per_step"], d["config"]["proofs_in_flight"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
PY
}
for i in 1 2 3; do
  run base2_$i LIG_GPU_SLOTS=0 -- --inflight 2
  run slots2_of3_$i LIG_GPU_SLOTS=2 -- --inflight 3
  run noslots_3_$i LIG_GPU_SLOTS=0 -- --inflight 3
  run slots2_of4_$i LIG_GPU_SLOTS=2 -- --inflight 4
  run slots3_of4_$i LIG_GPU_SLOTS=3 -- --inflight 4
done | tee $O/ab.txt
run slots2_of3_q8 LIG_GPU_SLOTS=2 GPU_MAX_HW_QUEUES=8 -- --inflight 3 | tee -a $O/ab.txt
run base2_q8 LIG_GPU_SLOTS=0 GPU_MAX_HW_QUEUES=8 -- --inflight 2 | tee -a $O/ab.txt
run slots1_of2 LIG_GPU_SLOTS=1 -- --inflight 2 | tee -a $O/ab.txt
