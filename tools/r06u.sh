#!/bin/bash
# round 6: the shared side stream as the product's default -- against LIG_SHARED_SIDE=0, under other queue counts and creation orders (robustness),
# then the GPU suite
O=gpurun_out/r06u; mkdir -p $O
one() { # tag, env...
  tag=$1; shift
  env "$@" timeout 150 python bench.py --no-cpu-baseline --no-verify --no-h2d --quad-mix 0 2>/dev/null | tail -1 > $O/$tag.json
  python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json"))
    print("%-34s value %.4e  one proof %.3f ms  pin %s" % ("$tag", d["value"], d["proof_wall_ms"], d["config"].get("proof_equals_oracle_pin")))
except Exception as e:
    print("%-34s FAILED / timed out" % "$tag")
PY
}
EXP=LIG_HIP_LIB=$PWD/tools/ab/liblig_hip_exp.so
for i in 1 2 3; do
  one shared_$i A=1
  one own_side_$i LIG_SHARED_SIDE=0
  one shared_q8_$i GPU_MAX_HW_QUEUES=8
  one shared_q2_$i GPU_MAX_HW_QUEUES=2
  one shared_q3_$i GPU_MAX_HW_QUEUES=3
  one shared_q6_$i GPU_MAX_HW_QUEUES=6
done | tee $O/ab.txt
for pad in 0,3,102 3,0,102 2,1,201 3,3,012 1,0,012 0,0,021; do
  one shared_pad_$pad $EXP LIG_STREAM_PAD=$pad
done | tee -a $O/ab.txt
python -m pytest tests -q -m gpu -x > $O/suite.log 2>&1; echo "suite rc $?" | tee -a $O/suite.log
tail -4 $O/suite.log
