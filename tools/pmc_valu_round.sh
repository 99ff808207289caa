#!/bin/bash
# VALUBusy / LDSBankConflict / MemUnitStalled per kernel over a FULL proof (one counter per pass, --kernel-trace only)
#   gpurun -- 'bash tools/pmc_valu_round.sh'  ->  gpurun_out/pmc/full_<COUNTER>.json
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
CMD="python bench.py --no-cpu-baseline --no-h2d --no-verify --no-sharded-leg --quad-mix 0 --inflight 1 --steps 2 --warmup 1"
for c in VALUBusy LDSBankConflict MemUnitStalled; do
    rm -rf /tmp/pmcv_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcv_$c -o pmc -- $CMD > /dev/null 2> gpurun_out/pmc/full_$c.err || true
    f=$(find /tmp/pmcv_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py "$f" > gpurun_out/pmc/full_$c.json
done
python - <<'PY'
import json
out = {"note": "rocprofv3 --pmc <counter> --kernel-trace, one counter per pass, over python bench.py --no-cpu-baseline --no-h2d --no-verify "
               "--inflight 1 --steps 2 --warmup 1 (full 2^24 proofs, ONE in flight: kernels of the two streams of a proof still overlap); "
               "per kernel the average over its launches at the largest grid", "kernels": {}}
for c in ("VALUBusy", "LDSBankConflict", "MemUnitStalled"):
    try:
        d = json.load(open("gpurun_out/pmc/full_%s.json" % c))
    except Exception as e:
        continue
    for k, v in d["kernels"].items():
        out["kernels"].setdefault(k, {})[c] = round(v["avg_at_largest_grid"], 2)
json.dump(out, open("gpurun_out/pmc/full_proof_valu_lds.json", "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("VALUBusy", 0))[:14]:
    print(k, v)
PY
