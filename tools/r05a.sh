mkdir -p gpurun_out/r05a
python -m pytest tests/test_gpu_sharded.py -m gpu -x -q -k "failing_peer or rows_entry" > gpurun_out/r05a/first.txt 2>&1
echo "first rc $?" >> gpurun_out/r05a/first.txt
timeout 1500 python tools/soak_sharded.py --rows-entry --iters 150 --log gpurun_out/r05a/soak_rows_entry.log > gpurun_out/r05a/soak_stdout.txt 2>&1
python -m pytest tests -m gpu -x -q > gpurun_out/r05a/suite.txt 2>&1
echo "suite rc $?" >> gpurun_out/r05a/suite.txt
tail -3 gpurun_out/r05a/first.txt gpurun_out/r05a/soak_stdout.txt gpurun_out/r05a/suite.txt
