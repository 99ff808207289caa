#!/bin/bash
# round 6: kernel trace (with stream ids) of the default bench, shared side stream -> gpurun_out/r06w/kt.csv.gz for the two-proof timeline
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06w; mkdir -p $O
rm -rf /tmp/kt
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-verify --no-h2d --quad-mix 0 > $O/bench.json 2>/dev/null
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv, gzip
rows=list(csv.DictReader(open("$f")))
with gzip.open("$O/kt.csv.gz","wt") as g:
    w=csv.writer(g); w.writerow(["stream","queue","name","start","end","grid"])
    for r in rows:
        w.writerow([r["Stream_Id"], r["Queue_Id"], r["Kernel_Name"].split("(")[0].replace("void lig::","")[:40], r["Start_Timestamp"], r["End_Timestamp"], r["Grid_Size_X"]])
print(len(rows), "kernels")
PY
tail -c 300 $O/bench.json
