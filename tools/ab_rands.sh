#!/bin/bash
# A/B of the caller-randomness-rows upload (profiles/r04_caller_rands_ab.md): the incl_h2d.caller_rands leg of bench.py with the
# uploader thread (default) and with round 3's event-chained copies on the side stream, alternating on one box
out=$1; rounds=${2:-2}; : > "$out"
for r in $(seq 1 "$rounds"); do
  for mode in 2 1; do
    line=$(LIG_RANDS_UPLOAD_MODE=$mode timeout 300 python bench.py --no-cpu-baseline --no-verify --quad-mix 0 --h2d-narrow 0 --steps 20 2>/dev/null | tail -1 |
           python tools/pick.py value incl_h2d.value incl_h2d.ms_per_step incl_h2d.caller_rands.value incl_h2d.caller_rands.ms_per_proof incl_h2d.caller_rands.frac_of_link_bound incl_h2d.caller_rands.same_proof_bytes)
    echo "round $r LIG_RANDS_UPLOAD_MODE=$mode $line" | tee -a "$out"
  done
done
