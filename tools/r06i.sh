#!/bin/bash
# round 6: AES layout test + failing-peer variants, then the sampler A/B (LIG_AES_LAYOUT 0 / 1): default bench alternating, LDS counters of k_rand_rlc
mkdir -p gpurun_out/r06i
O=gpurun_out/r06i
python -m pytest tests/test_gpu_aes_layout.py tests/test_gpu_sharded.py -q -m gpu -x -k "aes or failing_peer" > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log
for i in 1 2 3; do
  for L in 0 1; do
    LIG_AES_LAYOUT=$L python bench.py --no-cpu-baseline --no-verify --no-h2d --quad-mix 0 2>/dev/null | tail -1 > $O/bench_layout${L}_$i.json
    python - <<PY
import json
d=json.load(open("$O/bench_layout${L}_$i.json"))
print("layout $L run $i value %.4e one_proof_ms %.3f stage_ms %s" % (d["value"], d["proof_wall_ms"], d["config"]["stage_ms"]))
PY
  done
done | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
for L in 0 1; do
  for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "VALUBusy SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"; do
    tag=$(echo $C | tr ' ' '_')
    LIG_AES_LAYOUT=$L rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_${L}_$tag -o out --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --no-verify --no-h2d --quad-mix 0 > /dev/null 2>&1
    f=$(find /tmp/pmc_${L}_$tag -name "*counter_collection.csv" | head -1)
    python - <<PY >> $GRAFT_REPO_ROOT/$O/counters.txt
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open("$f")):
    k=r["Kernel_Name"]
    if "k_rand_rlc" not in k: continue
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
for k,v in acc.items():
    for c,val in v.items():
        print("layout $L", k[:40], c, "sum %.4g" % val, "launches", n[(k,c)], "per launch %.4g" % (val/n[(k,c)]))
PY
  done
done
cd $GRAFT_REPO_ROOT
cat $O/counters.txt | tail -40
tail -4 $O/tests.log
