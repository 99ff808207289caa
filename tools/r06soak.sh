#!/bin/bash
# round 6: fuzz + soak of the final build (one shared side stream per device, entry-major sampler tables, upload retry)
O=gpurun_out/r06soak; mkdir -p $O
L=$O/fuzz_soak.log
{
echo "# python tools/fuzz_parity.py 40 5"; timeout 900 python tools/fuzz_parity.py 40 5 2>&1 | tail -3
echo "# LIG_FUZZ_K=8192 python tools/fuzz_parity.py 8 6"; LIG_FUZZ_K=8192 timeout 900 python tools/fuzz_parity.py 8 6 2>&1 | tail -3
echo "# python tools/soak.py 1000   (two contexts, 2^24, every envelope against the oracle pin)"; timeout 1200 python tools/soak.py 1000 2>&1 | tail -2
echo "# python tools/soak_rands.py"; timeout 900 python tools/soak_rands.py 2>&1 | tail -2
echo "# python tools/soak_rows.py"; timeout 900 python tools/soak_rows.py 2>&1 | tail -2
echo "# python tools/soak_sharded.py --world 2 --iters 21"; timeout 900 python tools/soak_sharded.py --world 2 --iters 21 2>&1 | tail -2
echo "# python tools/soak_sharded.py --world 8 --iters 8"; timeout 900 python tools/soak_sharded.py --world 8 --iters 8 2>&1 | tail -2
echo "# python tools/soak_upload_fault.py 3 4"; timeout 900 python tools/soak_upload_fault.py 3 4 2>&1 | tail -1
} > $L 2>&1
cat $L
