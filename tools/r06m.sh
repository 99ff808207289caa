#!/bin/bash
# round 6: measurements of the build with the entry-major sampler tables -- per-row drop-in split, GPU suite, kernel tables, VALU budget, traffic,
# the default bench line, the shared-device sweep, the upload-fault soak
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06m; mkdir -p $O
g++ -std=c++17 -O2 -Iinclude tests/cpp/stage123_rows.cpp -Lligero-prover_amd -llig_hip -Loracle -llig_oracle -Wl,-rpath,$PWD/ligero-prover_amd -Wl,-rpath,$PWD/oracle -o tests/cpp/stage123_rows || exit 1
E=tests/cpp/stage123_rows
for lg in 20 24; do for d in 0 512; do
  p=3; [ $d = 0 ] && [ $lg = 24 ] && p=2
  timeout 600 $E $lg $d 0 8192 $p | tail -1 | tee -a $O/per_row.jsonl
done; done
python -m pytest tests -q -m gpu -x > $O/suite.log 2>&1; echo "suite rc $?" | tee -a $O/suite.log
tail -3 $O/suite.log
bash tools/profile_round.sh > $O/profile_round.log 2>&1
bash tools/valu_budget.sh > $O/valu_budget.log 2>&1
bash tools/pmc_round.sh > $O/pmc_round.log 2>&1
python bench.py 2> $O/bench_default.err | tail -1 > $O/bench_default.json
python -c "
import json; d=json.load(open('$O/bench_default.json')); print('default bench', d['value'], d['proof_wall_ms'], d['roofline']['frac'], d['roofline'].get('launches_of_512_rows'), d['cpu_baseline']['value'])"
LIG_BENCH_SHARE_GPU=1 python bench.py --gpus-sweep 1,2,4,8 --steps 3 --warmup 1 --sharded-leg --sharded-steps 2 --no-cpu-baseline --no-verify --no-h2d --quad-mix 0 --sharded-timeout 400 > $O/gpus_sweep_shared_device.jsonl 2> $O/gpus_sweep.err
tail -1 $O/gpus_sweep_shared_device.jsonl | cut -c1-600
python tools/soak_upload_fault.py 5 4 > $O/soak_upload_fault.log 2>&1; tail -2 $O/soak_upload_fault.log
