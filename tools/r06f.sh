# round 6, call f: the literal three-line drop-in -- a proof driven row by row through ligero::hip_context (tests/cpp/stage123_rows.cpp),
# eager and with set_deferred_rows(512): parity tests, then constraints/s at 2^20 and 2^24
O=gpurun_out/r06f; mkdir -p $O
timeout 900 python -m pytest tests/test_context_cpp.py -x -q -m gpu -k "three_stage or stage1_rows" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -n 5 $O/pytest.txt
E=tests/cpp/stage123_rows
for lg in 20 24; do
  for d in 0 512; do
    p=3; [ $d = 0 ] && [ $lg = 24 ] && p=1
    timeout 600 $E $lg $d 0 8192 $p | tail -1 | tee -a $O/per_row.jsonl
  done
done
