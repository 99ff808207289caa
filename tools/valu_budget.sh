#!/bin/bash
# VALU issue budget of one proof (tools/valu_budget.py):   gpurun -- 'bash tools/valu_budget.sh'  ->  gpurun_out/pmc/valu_budget.json
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
rm -rf /tmp/pmcb
rocprofv3 --pmc VALUBusy --kernel-trace --output-format csv -d /tmp/pmcb -o pmc -- python bench.py --no-cpu-baseline --no-h2d --no-verify --no-sharded-leg --quad-mix 0 --inflight 1 --steps 3 --warmup 1 > /dev/null 2> gpurun_out/pmc/valu_budget.err || true
c=$(find /tmp/pmcb -name "*counter_collection.csv" | head -1); t=$(find /tmp/pmcb -name "*kernel_trace.csv" | head -1)
head -2 "$c" > gpurun_out/pmc/valu_budget_head.txt; head -2 "$t" >> gpurun_out/pmc/valu_budget_head.txt
python tools/valu_budget.py "$c" "$t" > gpurun_out/pmc/valu_budget.json
python -c "
import json; d=json.load(open('gpurun_out/pmc/valu_budget.json')); print('proofs', d['proofs'], 'standalone', d['standalone_ms_per_proof'], 'busy', d['valu_busy_ms_per_proof'])
for k,v in list(d['kernels'].items())[:12]: print(k, v)"
