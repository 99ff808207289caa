#!/bin/bash
# round 6: do several PROCESSES on one GPU (own hardware queues each) get more out of the chip than several contexts of one process?
O=gpurun_out/r06q; mkdir -p $O
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-verify --no-h2d --quad-mix 0 --no-sharded-leg 2>/dev/null | tail -1 > $O/p1_i2_$i.json
  for n in 2 4; do for inf in 1 2; do
    LIG_BENCH_SHARE_GPU=1 LIG_COMM=ipc LIG_COMM_TAG=q${n}_${inf}_$i HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python bench.py --gpus $n --backend gloo --inflight $inf --no-cpu-baseline --no-verify --no-h2d --quad-mix 0 --no-sharded-leg 2>/dev/null | tail -1 > $O/p${n}_i${inf}_$i.json
  done; done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], "n_procs", d["n_gpus"], "inflight", d["config"]["proofs_in_flight"], "value %.4e" % d["value"], "ms/step %.2f" % d["ms_per_step"])
    except Exception as e: print(f, "failed", e)
PY
