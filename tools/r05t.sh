# the driver's round-end sequence, three times on one box: GPU suite, smoke, default bench
O=gpurun_out/r05t
mkdir -p $O
rm -f $O/summary.txt
for i in 1 2 3; do
  t0=$(date +%s); timeout 900 python -m pytest tests/ -x -q -m gpu > $O/suite_$i.txt 2>&1; echo "suite $i rc=$? in $(( $(date +%s) - t0 )) s" >> $O/summary.txt; tail -n 2 $O/suite_$i.txt >> $O/summary.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $O/summary.txt 2>&1
  timeout 600 python bench.py > $O/bench_$i.json 2> $O/bench_$i.err; echo "bench $i rc=$?" >> $O/summary.txt
done
cat $O/summary.txt
