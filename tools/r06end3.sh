#!/bin/bash
# round 6: the round-end sequence (GPU suite, smoke, default bench) three times on one box -- flakiness check of the final build
O=gpurun_out/r06end3; mkdir -p $O
for i in ${RUNS:-1 2 3}; do
  python -m pytest tests -q -m gpu -x > $O/suite_$i.log 2>&1; echo "run $i suite rc $? $(tail -1 $O/suite_$i.log)" | tee -a $O/summary.txt
  python -c "import __graft_entry__ as g; g.smoke(); print('run $i smoke ok')" 2>&1 | tail -1 | tee -a $O/summary.txt
  python bench.py --gpus 1 2>/dev/null | tail -1 > $O/bench_$i.json
  python -c "
import json; d=json.load(open('$O/bench_$i.json')); print('run $i bench %.4e constraints/s, one proof %.2f ms, pin %s, verifier %s' % (d['value'], d['proof_wall_ms'], d['config']['proof_equals_oracle_pin'], d['config']['verifier_accepts']))" | tee -a $O/summary.txt
done
