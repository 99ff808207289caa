#!/bin/bash
# rocprofv3 kernel statistics of the default bench command with one and two proofs in flight (the tables under profiles/):
#   gpurun -- 'bash tools/profile_round.sh'   ->  gpurun_out/prof/inflight{1,2}_kernel_stats.md + the bench lines of the same runs
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
for n in 1 2; do
    rm -rf /tmp/prof_$n
    rocprofv3 --kernel-trace --stats -d /tmp/prof_$n -o p -- python bench.py --warmup 0 --no-latency-probe --no-cpu-baseline --no-h2d --no-verify --no-sharded-leg --quad-mix 0 --inflight $n > gpurun_out/prof/bench_inflight$n.json 2> gpurun_out/prof/bench_inflight$n.err
    db=$(find /tmp/prof_$n -name "*.db" | head -1)
    {
        echo "# rocprofv3 --kernel-trace --stats -- python bench.py --warmup 0 --no-latency-probe --no-cpu-baseline --no-h2d --no-verify --no-sharded-leg --quad-mix 0 --inflight $n   ($n proof(s) in flight; tools/rocpd_summary.py)"
        echo
        python tools/rocpd_summary.py "$db" "k_encode_tiles<10, true>;k_encode_in<10>;k_rand_rlc<4, 1>;k_encode_tiles<10, false>;k_sha_update_rows"
    } > gpurun_out/prof/inflight${n}_kernel_stats.md
done
python bench.py --no-cpu-baseline --no-h2d --quad-mix 0 > gpurun_out/prof/bench_unprofiled.json 2>/dev/null
python bench.py --no-cpu-baseline --no-h2d --quad-mix 0 --inflight 1 > gpurun_out/prof/bench_unprofiled_inflight1.json 2>/dev/null
ls -la gpurun_out/prof
