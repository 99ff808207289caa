#!/bin/bash
# round 6: stage 1 software-pipelined over its chunks (LIG_S1_PIPE=1): K1 of chunk b+1 / K3 of chunk b-1 on a second stream next to K2 of chunk b
O=gpurun_out/r06z; mkdir -p $O
one() { tag=$1; shift
  env "$@" timeout 150 python bench.py --no-cpu-baseline --no-verify --no-h2d --quad-mix 0 2>/dev/null | tail -1 > $O/$tag.json
  python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); print("%-26s value %.4e  one proof %.3f ms  K2 512-row %.0f us  stages %s pin %s" % ("$tag", d["value"], d["proof_wall_ms"], 1e3*d["roofline"]["launches_of_512_rows"]["avg_launch_ms"], [round(x,2) for x in d["config"]["stage_ms"].values()], d["config"].get("proof_equals_oracle_pin")))
except Exception as e: print("%-26s FAILED" % "$tag")
PY
}
for i in 1 2 3; do
  one base_$i A=1
  one base_q8_$i GPU_MAX_HW_QUEUES=8
  one pipe_q8_$i LIG_S1_PIPE=1 GPU_MAX_HW_QUEUES=8
  one pipe_q4_$i LIG_S1_PIPE=1
done | tee $O/ab.txt
