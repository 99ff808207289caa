# round 6, call b: kernel tables (one / two proofs in flight) and whole-proof HBM traffic of the default bench command with
# LIG_ZRES=0 (codeword planes, K3 a kernel of its own) and LIG_ZRES=1 (Z tiles resident, K3 inside the column hash)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06b; mkdir -p $O
for z in 0 1; do
  export LIG_ZRES=$z
  for n in 1 2; do
    rm -rf /tmp/prof_$n
    rocprofv3 --kernel-trace --stats -d /tmp/prof_$n -o p -- python bench.py --no-cpu-baseline --no-h2d --no-verify --no-sharded-leg --quad-mix 0 --inflight $n > $O/bench_zres${z}_inflight$n.json 2> $O/bench_zres${z}_inflight$n.err
    db=$(find /tmp/prof_$n -name "*.db" | head -1)
    {
      echo "# LIG_ZRES=$z rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-h2d --no-verify --no-sharded-leg --quad-mix 0 --inflight $n   ($n proof(s) in flight; tools/rocpd_summary.py)"
      echo
      python tools/rocpd_summary.py "$db" "k_encode_tiles<10, true>;k_sha_update_rows"
    } > $O/zres${z}_inflight${n}_kernel_stats.md
  done
  CMD="python bench.py --no-cpu-baseline --no-h2d --no-verify --no-sharded-leg --quad-mix 0 --inflight 1 --steps 3 --warmup 1"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmcw_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcw_$c -o pmc -- $CMD > /dev/null 2> $O/whole_$c.err || true
    f=$(find /tmp/pmcw_$c -name "*counter_collection.csv" | head -1)
    cp "$f" $O/whole_$c.csv
  done
  python tools/pmc_whole_proof.py $O/whole_FETCH_SIZE.csv $O/whole_WRITE_SIZE.csv > $O/zres${z}_whole_proof_traffic.json
  rm -f $O/whole_*.csv
  python -c "
import json; d=json.load(open('$O/zres${z}_whole_proof_traffic.json'))
print('LIG_ZRES=$z total MB/proof', d['total_MB_per_proof'], 'bytes/row', d['total_bytes_per_committed_row'], 'ratio', d['ratio_to_algorithmic'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['MB_per_proof'])[:10]: print('  ', k, round(v['MB_per_proof'],1), 'MB', round(v['bytes_per_row']), 'B/row')
"
  head -n 16 $O/zres${z}_inflight1_kernel_stats.md
done
