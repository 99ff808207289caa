#!/bin/bash
# round 6: why is the folded tile kernel 300 us slower per 512-row launch?  counters of k_encode_tiles<10, true, {false,true}>
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06ag; mkdir -p $O; rm -f $O/counters.txt
for F in 0 2; do
  for C in "VALUBusy SQ_INSTS_VALU" "MemUnitBusy MemUnitStalled" "FETCH_SIZE" "SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
    tag=$(echo $C | tr ' ' '_')
    rm -rf /tmp/pm
    LIG_K1_FOLD=$F rocprofv3 --pmc $C --kernel-trace -d /tmp/pm -o out --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --inflight 1 --no-latency-probe --no-cpu-baseline --no-verify --no-h2d --quad-mix 0 > /dev/null 2>&1
    f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
    python - <<PY >> $O/counters.txt
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
try:
    for r in csv.DictReader(open("$f")):
        k=r["Kernel_Name"]
        if "k_encode_tiles<10, true" not in k or int(r["Grid_Size"]) != 1048576 and int(r.get("Grid_Size_X", r["Grid_Size"])) != 1048576: continue
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
    for k,v in acc.items():
        for c,val in v.items():
            print("fold $F", k.split("lig::")[-1][:34], c, "launches", n[(k,c)], "per launch %.5g" % (val/n[(k,c)]))
except Exception as e:
    print("fold $F $C failed", e)
PY
  done
done
cat $O/counters.txt
