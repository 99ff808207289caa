# sampler launch shape: persistent workgroups of the big sampler launches (LIG_AES_BLOCKS; default 512 = two per CU, each holding 64 KiB of LDS,
# which leaves no LDS for a tile workgroup on that CU while the sampler runs)
O=gpurun_out/r05s
mkdir -p $O
for rep in 1 2 3; do for v in 0 256 384 768; do
  LIG_AES_BLOCKS=$v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-h2d --no-h2d-rands --quad-mix 0 --no-verify > $O/b${v}_$rep.json 2>/dev/null
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05s/*.json")):
    try: d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e: print(f, "no line"); continue
    print("%-10s value %.4g  ms/step %.3f  wall %.3f  pin %s" % (f.split("/")[-1][:-5], d["value"], d["ms_per_step"], d["proof_wall_ms"], d.get("proof_equals_oracle_pin")))
PY
