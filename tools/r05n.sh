# re-tune of the stage-1 chunk schedule knobs for the wave-specialised hash (alternating, two repetitions)
O=gpurun_out/r05n
mkdir -p $O
run() { name=$1; shift; env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-h2d --no-h2d-rands --quad-mix 0 --no-verify > $O/$name.json 2>/dev/null; }
for rep in 1 2; do
  run base_$rep LIG_X=0
  run head64_$rep LIG_S1_HEAD=64
  run head256_$rep LIG_S1_HEAD=256
  run tail48_$rep LIG_S1_TAIL=48
  run tail192_$rep LIG_S1_TAIL=192
  run gate0_$rep LIG_SHA_GATE=0
  run gaterows8_$rep LIG_SHA_GATE_ROWS=8
  run gaterows32_$rep LIG_SHA_GATE_ROWS=32
  run s2head128_$rep LIG_S2_HEAD=128
  run s2head320_$rep LIG_S2_HEAD=320
  run chunk384_$rep LIG_ENCODE_CHUNK=384
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05n/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print("%-16s value %.4g  ms/step %.3f  wall %.3f" % (f.split("/")[-1][:-5], d["value"], d["ms_per_step"], d["proof_wall_ms"]))
    except Exception as e: print(f, e)
PY
