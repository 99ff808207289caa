#!/bin/bash
# Shader clock and socket power WHILE the default bench runs (rocm-smi polled every 0.2 s): is the chip throttling under the
# multiply chains?   gpurun -- 'bash tools/clock_under_load.sh'   (round 3: 2.31-2.32 GHz of 2.4, 1.25-1.28 kW)
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed -E 's/.*\(([0-9]+)Mhz\).*/sclk \1/; s/.*\(W\): ([0-9.]+).*/W \1/' | tr '\n' ' '; echo; sleep 0.2; done ) > /tmp/clk.log &
MP=$!
python bench.py --no-cpu-baseline --no-h2d --no-verify --no-sharded-leg --steps 600 --warmup 5 > /tmp/b.json 2>/dev/null
kill $MP
sort /tmp/clk.log | uniq -c | sort -rn | head -25
python -c "
import json; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
