# tile kernel with the coefficients through its own Y tile instead of 36 registers (112 VGPRs, 4 waves per SIMD): A/B
O=gpurun_out/r05u
mkdir -p $O
run() { # name lib dynlds extra
  LIG_HIP_LIB=$2 LIG_K2_DYN_LDS=$3 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-h2d --no-h2d-rands --quad-mix 0 $4 > $O/$1.json 2>$O/$1.err
}
run pin_base "" 0 ""
run pin_cbg tools/ab/liblig_hip_cbg.so 0 ""
for rep in 1 2 3; do
  run base_$rep "" 0 --no-verify
  run cbg_$rep tools/ab/liblig_hip_cbg.so 0 --no-verify
  run cbg3_$rep tools/ab/liblig_hip_cbg.so 12288 --no-verify
done
for i in 1 3; do LIG_HIP_LIB=tools/ab/liblig_hip_cbg.so timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-h2d --no-h2d-rands --quad-mix 0 --no-verify --inflight $i > $O/cbg_inflight$i.json 2>/dev/null
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-h2d --no-h2d-rands --quad-mix 0 --no-verify --inflight $i > $O/base_inflight$i.json 2>/dev/null; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05u/*.json")):
    try: d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e: print(f, "no line"); continue
    r=d["roofline"]
    print("%-16s value %.4g  ms/step %.3f  wall %.3f  K2 avg %.3f / alone %.3f  pin %s acc %s" % (f.split("/")[-1][:-5], d["value"], d["ms_per_step"], d["proof_wall_ms"], r["avg_launch_ms"], r["one_proof_in_flight"]["avg_launch_ms"], d["config"].get("proof_equals_oracle_pin"), d["config"].get("verifier_accepts")))
PY
