#!/bin/bash
# round 6: which hardware queue does each stream of the two proofs in flight use (default placement)?  kernel trace -> (stream, queue) pairs
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06r; mkdir -p $O
for tag in default pad10; do
  rm -rf /tmp/kt_$tag
  if [ $tag = pad10 ]; then export LIG_HIP_LIB=; fi
  rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$tag -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-h2d --quad-mix 0 > $O/bench_$tag.json 2>/dev/null
  f=$(find /tmp/kt_$tag -name "*kernel_trace.csv" | head -1)
  head -1 $f > $O/header_$tag.txt
  python - <<PY > $O/pairs_$tag.txt
import csv, collections
rows=list(csv.DictReader(open("$f")))
print("columns:", list(rows[0].keys()))
pairs=collections.defaultdict(collections.Counter)
for r in rows:
    key=(r.get("Stream_Id"), r.get("Queue_Id"))
    pairs[key][r["Kernel_Name"].split("(")[0][-40:]]+=1
for k,v in sorted(pairs.items()):
    print("stream", k[0], "queue", k[1], "kernels", sum(v.values()), dict(v.most_common(5)))
PY
  break
done
cat $O/pairs_default.txt
