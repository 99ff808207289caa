#!/bin/bash
# round 6, second session: the new GPU tests (upload retry, LIG_SHA_WS sweep, W = 8, --gpus-sweep), then the whole GPU suite
mkdir -p gpurun_out/r06h
python -m pytest tests/test_gpu_rows_api.py tests/test_gpu_sha_ws.py -q -m gpu -x -k "retried or kernel_variant" > gpurun_out/r06h/new_tests.log 2>&1
echo "new tests rc $?" >> gpurun_out/r06h/new_tests.log
python -m pytest tests/test_gpu_sharded.py -q -m gpu -x -k "sweep or (configs3 and 8) or (rows_entry and 8-)" > gpurun_out/r06h/w8.log 2>&1
echo "w8 rc $?" >> gpurun_out/r06h/w8.log
python -m pytest tests -q -m gpu -x > gpurun_out/r06h/suite.log 2>&1
echo "suite rc $?" >> gpurun_out/r06h/suite.log
tail -3 gpurun_out/r06h/new_tests.log gpurun_out/r06h/w8.log gpurun_out/r06h/suite.log
