#!/bin/bash
# round 6, HEAD: the round-end sequence as the driver runs it (GPU suite, smoke, default bench) + the profiled kernel tables
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06final; mkdir -p $O
python -m pytest tests -q -m gpu -x > $O/suite.log 2>&1; echo "suite rc $?" | tee -a $O/suite.log; tail -2 $O/suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.log
python bench.py --gpus 1 2> $O/bench_default.err | tail -1 > $O/bench_default.json
bash tools/profile_round.sh > $O/profile_round.log 2>&1
python -c "
import json
d=json.load(open('$O/bench_default.json')); print('default', '%.4e'%d['value'], d['proof_wall_ms'], d['roofline']['frac'], d['roofline']['launches_of_512_rows']['avg_launch_ms'], d['value_incl_h2d'], d['incl_h2d']['caller_rands']['value'], d['cpu_baseline']['value'], d['config']['proof_equals_oracle_pin'], d['config']['verifier_accepts'])
for n in (1,2):
    p=json.load(open('gpurun_out/prof/bench_inflight%d.json'%n)); print('profiled inflight',n, p['roofline']['launches_of_512_rows'])"
grep -n "4096 |" gpurun_out/prof/inflight1_kernel_stats.md gpurun_out/prof/inflight2_kernel_stats.md | head -4
