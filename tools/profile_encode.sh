#!/bin/bash
# rocprofv3 kernel statistics of the configs[1] command (RS-encode only):  gpurun -- 'bash tools/profile_encode.sh'
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
rm -rf /tmp/prof_enc
rocprofv3 --kernel-trace --stats -d /tmp/prof_enc -o p -- python bench.py --workload encode --no-cpu-baseline --steps 50 --warmup 5 > gpurun_out/prof/bench_encode_under_rocprof.json 2> gpurun_out/prof/bench_encode.err
db=$(find /tmp/prof_enc -name "*.db" | head -1)
{
    echo "# rocprofv3 --kernel-trace --stats -- python bench.py --workload encode --no-cpu-baseline --steps 50 --warmup 5   (configs[1]: 2^20 constraints = 132 rows, INTT_k + NTT_4k; tools/rocpd_summary.py)"
    echo
    python tools/rocpd_summary.py "$db" "k_encode_tiles<10, true>"
} > gpurun_out/prof/encode_kernel_stats.md
cat gpurun_out/prof/encode_kernel_stats.md | head -20
