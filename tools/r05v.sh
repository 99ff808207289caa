# tests/test_gpu_sharded.py (every multi-rank test of the suite) in a loop on one box, slowest tests recorded per run
O=gpurun_out/r05v
N=${1:-5}
mkdir -p $O; rm -f $O/summary.txt
for i in $(seq 1 $N); do
  t0=$(date +%s); timeout 600 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu --durations=8 > $O/run_$i.txt 2>&1; echo "run $i rc=$? in $(( $(date +%s) - t0 )) s: $(tail -n 1 $O/run_$i.txt)" >> $O/summary.txt
  grep -A3 "slowest" $O/run_$i.txt | tail -n 3 >> $O/summary.txt
done
cat $O/summary.txt
