# tile-kernel wave priority re-measured with the wave-specialised hash (r04_tile_setprio_ab.md had the one-wave hash)
O=gpurun_out/r05r
mkdir -p $O
for rep in 1 2 3; do for v in base prio2 prio3; do
  lib=""; [ $v != base ] && lib=tools/ab/liblig_hip_$v.so
  LIG_HIP_LIB=$lib python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-h2d --no-h2d-rands --quad-mix 0 --no-verify > $O/${v}_$rep.json 2>/dev/null
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05r/*.json")):
    d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    print("%-10s value %.4g  ms/step %.3f  wall %.3f  K2 avg %.3f / one-in-flight %.3f  pin %s" % (f.split("/")[-1][:-5], d["value"], d["ms_per_step"], d["proof_wall_ms"], d["roofline"]["avg_launch_ms"], d["roofline"]["one_proof_in_flight"]["avg_launch_ms"], d.get("proof_equals_oracle_pin")))
PY
