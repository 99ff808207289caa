#!/bin/bash
# Idle VALU issue slots of the benched configuration by running kernel mix (tools/issue_timeline.py):
#   gpurun -- 'bash tools/issue_timeline.sh'  ->  gpurun_out/timeline/{valu_budget.json, issue_timeline.json, bench_under_trace.json}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/timeline
mkdir -p $O
rm -rf /tmp/tl_pmc /tmp/tl_tr
rocprofv3 --pmc VALUBusy --kernel-trace --output-format csv -d /tmp/tl_pmc -o pmc -- python bench.py --no-cpu-baseline --no-h2d --no-verify --no-sharded-leg --quad-mix 0 --inflight 1 --steps 3 --warmup 1 > /dev/null 2> $O/pmc.err || true
c=$(find /tmp/tl_pmc -name "*counter_collection.csv" | head -1); t=$(find /tmp/tl_pmc -name "*kernel_trace.csv" | head -1)
python tools/valu_budget.py "$c" "$t" > $O/valu_budget.json
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_tr -o tr -- python bench.py --no-cpu-baseline --no-h2d --no-verify --no-sharded-leg --quad-mix 0 --inflight 2 --steps 20 --warmup 3 > $O/bench_under_trace.json 2> $O/trace.err || true
t2=$(find /tmp/tl_tr -name "*kernel_trace.csv" | head -1)
gzip -c "$t2" > $O/kernel_trace.csv.gz
python tools/issue_timeline.py "$t2" $O/valu_budget.json 0.25 0.85 > $O/issue_timeline.json
python -c "
import json; d=json.load(open('$O/issue_timeline.json'))
print({k: v for k, v in d.items() if k != 'mixes'})
for m in d['mixes']: print(m)"
