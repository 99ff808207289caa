# round 6, call e: which counters exist; instruction-cache behaviour of the tile kernel (encode-only workload, stand-alone launches)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06e; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
grep -i -o "SQC\?_[A-Z_0-9]*\(ICACHE\|IFETCH\|INST_LEVEL\|WAIT_INST\|WAIT_ANY\|BUSY_CYCLES\|ACTIVE_INST\|INSTS_VALU\|INST_CYCLES\|LDS_BANK\|LDS_IDX\|LDS_ADDR\|LDS_DATA\|WAVE_CYCLES\|WAIT_INST_LDS\|INSTS_LDS\)[A-Z_0-9]*" $O/counters.txt | sort -u | tr '\n' ' '
echo
CMD="python bench.py --workload encode --no-cpu-baseline --steps 5 --warmup 1"
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_IFETCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $set | tr ' ' '+')
  rm -rf /tmp/pmce
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmce -o pmc -- $CMD > /dev/null 2> $O/$tag.err || true
  f=$(find /tmp/pmce -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" <<'PY' | tee -a $O/summary.txt
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_[A-Za-z_0-9]+(?:<[^>]*>)?)", row["Kernel_Name"]); k = m.group(1) if m else row["Kernel_Name"][:40]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k in sorted(agg):
    if "encode" in k or "sha" in k:
        print(k, {c: round(v / max(1, cnt[(k, c)])) for c, v in agg[k].items()}, "launches", max(cnt[(k, c)] for c in agg[k]))
PY
  else echo "no csv for $tag: $(tail -n 2 $O/$tag.err)" | tee -a $O/summary.txt; fi
done
