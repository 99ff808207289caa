#!/bin/bash
# Same-box alternating A/B of environment variants of the default bench (resident witness, two proofs in flight):
#   tools/ab_env.sh OUT.txt ROUNDS "NAME1:ENV=.. ENV=.." "NAME2:..." ...
# every round runs every variant once, in order; one line per run: variant, constraints/s, ms per step, one-proof wall, stage ms
out=$1; rounds=$2; shift 2
mkdir -p "$(dirname "$out")"; : > "$out"
for r in $(seq 1 "$rounds"); do
  for v in "$@"; do
    name=${v%%:*}; envs=${v#*:}
    line=$(env $envs python bench.py --no-cpu-baseline --no-h2d --no-verify --quad-mix 0 --steps 20 --warmup 3 2>/dev/null | tail -1 |
           python tools/pick.py value ms_per_step proof_wall_ms config.stage_ms roofline.avg_launch_ms config.proof_equals_oracle_pin)
    echo "round $r $name [$envs] $line" | tee -a "$out"
  done
done
