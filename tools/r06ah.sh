#!/bin/bash
# round 6: a few knobs re-checked on the final build (shared side stream, entry-major sampler): sampler launch shape, stage-2 head, chunk head / tail
O=gpurun_out/r06ah; mkdir -p $O
one() { tag=$1; shift
  env "$@" timeout 150 python bench.py --no-cpu-baseline --no-verify --no-h2d --quad-mix 0 2>/dev/null | tail -1 > $O/$tag.json
  python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); print("%-22s value %.4e  one proof %.3f ms pin %s" % ("$tag", d["value"], d["proof_wall_ms"], d["config"].get("proof_equals_oracle_pin")))
except Exception as e: print("%-22s FAILED" % "$tag")
PY
}
for i in 1 2; do
  one default_$i A=1
  one aes256_$i LIG_AES_BLOCKS=256
  one aes384_$i LIG_AES_BLOCKS=384
  one aes768_$i LIG_AES_BLOCKS=768
  one s2head128_$i LIG_S2_HEAD=128
  one s2head320_$i LIG_S2_HEAD=320
  one s1head256_$i LIG_S1_HEAD=256
  one s1tail192_$i LIG_S1_TAIL=192
  one gate_rows8_$i LIG_SHA_GATE_ROWS=8
  one sha_ws4_$i LIG_SHA_WS=4
done | tee $O/ab.txt
