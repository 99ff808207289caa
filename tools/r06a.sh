# round 6, call a: LIG_ZRES (K3 inside the column hash) -- parity tests, then same-box alternating A/B of the default bench
O=gpurun_out/r06a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_zres.py -x -q -m gpu --durations=5 > $O/pytest_zres.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_zres.txt
tail -n 15 $O/pytest_zres.txt
bash tools/ab_env.sh $O/ab_inflight2.txt 3 "planar:LIG_ZRES=0" "zres:LIG_ZRES=1"
for z in 0 1; do
  LIG_ZRES=$z python bench.py --no-cpu-baseline --no-h2d --no-verify --no-sharded-leg --quad-mix 0 --inflight 1 --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/bench_inflight1_zres$z.json
  python tools/pick.py value ms_per_step proof_wall_ms config.stage_ms < $O/bench_inflight1_zres$z.json | sed "s/^/inflight1 zres=$z /" | tee -a $O/ab_inflight1.txt
done
