O=gpurun_out/r05h
mkdir -p $O
python -m pytest tests/test_gpu_ref_backend.py tests/test_gpu_rows_api.py -m gpu -x -q > $O/tests.txt 2>&1; echo "rc $?" >> $O/tests.txt; tail -n 3 $O/tests.txt
for rep in 1 2 3; do for sw in 0 1; do
  LIG_SPIN_WAIT=$sw python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-h2d --no-h2d-rands --quad-mix 0 --no-verify > $O/bench_spin${sw}_$rep.json 2>/dev/null
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05h/bench_spin*.json")):
    d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f.split("/")[-1], "value %.4g" % d["value"], "ms/step %.3f" % d["ms_per_step"], "wall %.3f" % d["proof_wall_ms"])
PY
