# round 5, final measurement call: suite, profiles of the final build, bench lines
O=gpurun_out/r05g
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/suite.txt 2>&1; echo "suite rc $?" >> $O/suite.txt
tail -n 3 $O/suite.txt
bash tools/issue_timeline.sh > $O/timeline_stdout.txt 2>&1
bash tools/profile_round.sh > $O/profile_round.txt 2>&1
bash tools/pmc_round.sh > $O/pmc_round.txt 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
for n in 1 3 4; do python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-h2d --no-h2d-rands --quad-mix 0 --no-verify --inflight $n > $O/bench_inflight$n.json 2>/dev/null; done
python bench.py --steps 6 --warmup 2 --sharded-leg --no-cpu-baseline --no-h2d --no-h2d-rands --quad-mix 0 --no-verify > $O/bench_sharded_leg_n1.json 2> $O/bench_sharded_leg_n1.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05g/bench_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        sh=d.get("sharded") or {}
        print(f.split("/")[-1], "value %.4g" % d["value"], "ms/step %.3f" % d["ms_per_step"], "wall", d.get("proof_wall_ms"), "pin", d.get("proof_equals_oracle_pin"), "sharded ms", sh.get("ms_per_proof"), sh.get("stage_ms"))
    except Exception as e: print(f, e)
PY
tail -n 25 $O/timeline_stdout.txt
