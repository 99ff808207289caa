# round 6, call d: LIG_SHA_GATE=2 (K1 of the next chunk ahead of the hash) x LIG_ZRES: parity, then alternating A/B with one and two proofs in flight
O=gpurun_out/r06d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_zres.py -x -q -m gpu --durations=5 > $O/pytest_zres.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_zres.txt
tail -n 6 $O/pytest_zres.txt
LIG_SHA_GATE=2 timeout 900 python -m pytest tests/test_gpu_rows_api.py -x -q -m gpu > $O/pytest_rows_gate2.txt 2>&1; echo "pytest rows gate2 rc=$?" >> $O/pytest_rows_gate2.txt
tail -n 3 $O/pytest_rows_gate2.txt
bash tools/ab_env.sh $O/ab_inflight2.txt 3 "planar:LIG_ZRES=0" "planar-k1:LIG_ZRES=0 LIG_SHA_GATE=2" "zres:LIG_ZRES=1" "zres-k1:LIG_ZRES=1 LIG_SHA_GATE=2"
for r in 1 2; do for v in "LIG_ZRES=0" "LIG_ZRES=0 LIG_SHA_GATE=2" "LIG_ZRES=1" "LIG_ZRES=1 LIG_SHA_GATE=2"; do
  env $v python bench.py --no-cpu-baseline --no-h2d --no-verify --no-sharded-leg --quad-mix 0 --inflight 1 --steps 20 --warmup 3 2>/dev/null | tail -1 | python tools/pick.py value ms_per_step proof_wall_ms config.stage_ms config.proof_equals_oracle_pin | sed "s/^/inflight1 [$v] /" | tee -a $O/ab_inflight1.txt
done; done
