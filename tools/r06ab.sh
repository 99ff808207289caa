#!/bin/bash
# round 6, last call: kernel tables of runs whose dominant-kernel launches are exactly the timed region's (--warmup 0 --no-latency-probe, every context bracketed),
# then the default bench line of the final build
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06ab; mkdir -p $O
bash tools/profile_round.sh > $O/profile_round.log 2>&1
python bench.py 2> $O/bench_default.err | tail -1 > $O/bench_default.json
python -c "
import json
for f in ['gpurun_out/prof/bench_inflight1.json','gpurun_out/prof/bench_inflight2.json','$O/bench_default.json']:
    d=json.load(open(f)); print(f, '%.4e'%d['value'], d['proof_wall_ms'], d['roofline']['launches'], d['roofline']['avg_launch_ms'], d['roofline']['launches_of_512_rows'])"
grep -n "4096 |" gpurun_out/prof/inflight1_kernel_stats.md gpurun_out/prof/inflight2_kernel_stats.md | head -4
