# tests/test_gpu_sharded.py (every multi-rank test of the suite) eight times in a row on one box
O=gpurun_out/r05v
mkdir -p $O; rm -f $O/summary.txt
for i in 1 2 3 4 5 6 7 8; do
  t0=$(date +%s); timeout 600 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu > $O/run_$i.txt 2>&1; echo "run $i rc=$? in $(( $(date +%s) - t0 )) s: $(tail -n 1 $O/run_$i.txt)" >> $O/summary.txt
done
cat $O/summary.txt
