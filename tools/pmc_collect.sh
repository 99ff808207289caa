#!/bin/bash
# Collects per-kernel PMC counters with rocprofv3, ONE counter per pass (with --kernel-trace only, as the microarch guide
# prescribes), for two commands:
#   full  : python bench.py --no-cpu-baseline --no-h2d --inflight 1 --steps 2 --warmup 1     (FETCH_SIZE, WRITE_SIZE)
#   encode: python bench.py --workload encode --log2-constraints 23 --steps 3 --warmup 1 --no-cpu-baseline
#           (VALUBusy, LDSBankConflict, MemUnitStalled: stand-alone encode launches)
#   sampler: python tools/time_aes.py  (VALUBusy, LDSBankConflict of the dense AES fill)
# CSVs land in gpurun_out/pmc/<tag>_<COUNTER>.csv; summarise with tools/pmc_traffic.py / tools/pmc_summary.py.
set -e
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
run() {   # tag counter cmd...
    tag=$1; ctr=$2; shift 2
    rm -rf /tmp/pmc_$tag_$ctr
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$ctr -o pmc -- "$@" > /dev/null 2> gpurun_out/pmc/${tag}_$ctr.err || true
    f=$(find /tmp/pmc_${tag}_$ctr -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python tools/pmc_summary.py "$f" > gpurun_out/pmc/${tag}_$ctr.json; cp "$f" gpurun_out/pmc/${tag}_$ctr.csv; fi
}
for c in FETCH_SIZE WRITE_SIZE; do run full $c python bench.py --no-cpu-baseline --no-h2d --inflight 1 --steps 2 --warmup 1; done
for c in VALUBusy LDSBankConflict MemUnitStalled; do run encode $c python bench.py --workload encode --log2-constraints 23 --steps 3 --warmup 1 --no-cpu-baseline; done
for c in VALUBusy LDSBankConflict; do run sampler $c python tools/time_aes.py; done
ls -la gpurun_out/pmc | head -30
