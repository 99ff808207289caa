# round 5, GPU call b: reference-backend vectors on the HIP path, the wave-specialised column hash (parity, chain time, proofs)
mkdir -p gpurun_out/r05b
O=gpurun_out/r05b
python -m pytest tests/test_gpu_ref_backend.py tests/test_gpu_parity.py tests/test_context_cpp.py -m gpu -x -q > $O/tests1.txt 2>&1; echo "rc $?" >> $O/tests1.txt
python tools/sha_chain_bench.py --reps 5 > $O/sha_chain.txt 2>&1
for ws in 0 2 1 4; do
  LIG_SHA_WS=$ws python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-h2d --no-h2d-rands --quad-mix 0 > $O/bench_ws$ws.json 2> $O/bench_ws$ws.err
done
python -m pytest tests -m gpu -x -q > $O/suite.txt 2>&1; echo "suite rc $?" >> $O/suite.txt
tail -3 $O/tests1.txt $O/suite.txt; cat $O/sha_chain.txt | tail -40
for ws in 0 2 1 4; do python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/bench_ws$ws.json") if l.startswith("{")][-1])
    print("ws$ws value %.4g ms/step %.3f proof_wall_ms %s pin %s" % (d["value"], d["ms_per_step"], d.get("proof_wall_ms"), d.get("proof_equals_oracle_pin")))
except Exception as e: print("ws$ws", e)
PY
done
