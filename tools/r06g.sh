# round 6, call g: per-row drop-in after the ring-pool fix: constraints/s (3 proofs, best), and the GPU side of the deferred mode (kernel table)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06g; mkdir -p $O
g++ -std=c++17 -O2 -Iinclude tests/cpp/stage123_rows.cpp -Lligero-prover_amd -llig_hip -Loracle -llig_oracle -Wl,-rpath,$PWD/ligero-prover_amd -Wl,-rpath,$PWD/oracle -o tests/cpp/stage123_rows || exit 1
E=tests/cpp/stage123_rows
for lg in 20 24; do for d in 0 512; do
  p=3; [ $d = 0 ] && [ $lg = 24 ] && p=2
  timeout 600 $E $lg $d 0 8192 $p | tail -1 | tee -a $O/per_row.jsonl
done; done
rm -rf /tmp/prof_d
rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o p -- $E 24 512 0 8192 2 > $O/deferred_profiled.json 2>/dev/null
python tools/rocpd_summary.py $(find /tmp/prof_d -name "*.db" | head -1) > $O/deferred_2p24_kernel_stats.md
head -n 14 $O/deferred_2p24_kernel_stats.md
