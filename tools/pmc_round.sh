#!/bin/bash
# whole-proof HBM traffic of the final build: one FETCH_SIZE and one WRITE_SIZE pass (each with --kernel-trace only)
#   gpurun -- 'bash tools/pmc_round.sh'  ->  gpurun_out/pmc/whole_proof_traffic.json
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
CMD="python bench.py --no-cpu-baseline --no-h2d --no-verify --no-sharded-leg --quad-mix 0 --inflight 1 --steps 3 --warmup 1"
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmcw_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcw_$c -o pmc -- $CMD > /dev/null 2> gpurun_out/pmc/whole_$c.err || true
    f=$(find /tmp/pmcw_$c -name "*counter_collection.csv" | head -1)
    cp "$f" gpurun_out/pmc/whole_$c.csv
done
python tools/pmc_whole_proof.py gpurun_out/pmc/whole_FETCH_SIZE.csv gpurun_out/pmc/whole_WRITE_SIZE.csv > gpurun_out/pmc/whole_proof_traffic.json
python tools/pmc_traffic.py gpurun_out/pmc/whole_FETCH_SIZE.csv gpurun_out/pmc/whole_WRITE_SIZE.csv > gpurun_out/pmc/pmc_traffic.json
python -c "
import json; d=json.load(open('gpurun_out/pmc/whole_proof_traffic.json'))
print('total MB/proof', d['total_MB_per_proof'], 'bytes/row', d['total_bytes_per_committed_row'], 'ratio', d['ratio_to_algorithmic'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['MB_per_proof'])[:12]: print(k, round(v['MB_per_proof'],1), 'MB', round(v['bytes_per_row']), 'B/row')
"
rm -f gpurun_out/pmc/whole_*.csv
