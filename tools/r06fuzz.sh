#!/bin/bash
# round 6: a longer differential fuzz of the final build against the oracle (random trace shapes and batch programs; other seeds than before)
O=gpurun_out/r06fuzz; mkdir -p $O
{
echo "# python tools/fuzz_parity.py 200 61"; timeout 1500 python tools/fuzz_parity.py 200 61 2>&1 | tail -2
echo "# LIG_FUZZ_K=8192 python tools/fuzz_parity.py 24 62"; LIG_FUZZ_K=8192 timeout 1500 python tools/fuzz_parity.py 24 62 2>&1 | tail -2
echo "# LIG_FUZZ_K=2048 python tools/fuzz_parity.py 40 63"; LIG_FUZZ_K=2048 timeout 1500 python tools/fuzz_parity.py 40 63 2>&1 | tail -2
echo "# LIG_ZRES=1 python tools/fuzz_parity.py 60 64"; LIG_ZRES=1 timeout 1500 python tools/fuzz_parity.py 60 64 2>&1 | tail -2
echo "# LIG_SHARED_SIDE=0 LIG_AES_LAYOUT=0 python tools/fuzz_parity.py 40 65"; LIG_SHARED_SIDE=0 LIG_AES_LAYOUT=0 timeout 1500 python tools/fuzz_parity.py 40 65 2>&1 | tail -2
echo "# python tools/fuzz_sharded.py 12 71   (W in {2, 4, 8}, gloo or comm_ipc)"; timeout 2400 python tools/fuzz_sharded.py 12 71 2>&1 | tail -13
} > $O/fuzz.log 2>&1
cat $O/fuzz.log
