#!/bin/bash
# round 6: LIG_JIT_SIDE=1 -- the shared side stream never holds an entry that still has long to wait (the host waits instead): do the stages of the
# two proofs then interleave chunk by chunk, and is that better than taking turns stage by stage?
O=gpurun_out/r06aa; mkdir -p $O
one() { tag=$1; shift
  env "$@" timeout 150 python bench.py --no-cpu-baseline --no-verify --no-h2d --quad-mix 0 2>/dev/null | tail -1 > $O/$tag.json
  python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); print("%-22s value %.4e  one proof %.3f ms  K2 512-row %.0f us  stages %s pin %s" % ("$tag", d["value"], d["proof_wall_ms"], 1e3*d["roofline"]["launches_of_512_rows"]["avg_launch_ms"], [round(x,2) for x in d["config"]["stage_ms"].values()], d["config"].get("proof_equals_oracle_pin")))
except Exception as e: print("%-22s FAILED" % "$tag")
PY
}
for i in 1 2 3; do
  one base_$i A=1
  one jit_$i LIG_JIT_SIDE=1
  one jit_3inflight_$i LIG_JIT_SIDE=1 GPU_MAX_HW_QUEUES=8
done | tee $O/ab.txt
env LIG_JIT_SIDE=1 GPU_MAX_HW_QUEUES=8 timeout 150 python bench.py --inflight 3 --no-cpu-baseline --no-verify --no-h2d --quad-mix 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('jit inflight 3 value %.4e pin %s' % (d['value'], d['config'].get('proof_equals_oracle_pin')))" | tee -a $O/ab.txt
