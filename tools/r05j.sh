# round 5: differential fuzz and soaks of the final build (wave-specialised hash, polling waits, bounded uploader)
O=gpurun_out/r05j
mkdir -p $O
{
echo "# python tools/fuzz_parity.py 40 5"; python tools/fuzz_parity.py 40 5 2>&1 | tail -n 42
echo "# LIG_FUZZ_K=8192 python tools/fuzz_parity.py 8 6"; LIG_FUZZ_K=8192 python tools/fuzz_parity.py 8 6 2>&1 | tail -n 10
echo "# python tools/soak.py 1000   (two contexts, 2^24, every envelope against the oracle pin)"; python tools/soak.py 1000 2>&1 | tail -n 2
echo "# python tools/soak_rands.py  (host randomness rows through the uploader thread)"; python tools/soak_rands.py 2>&1 | tail -n 3
echo "# python tools/soak_rows.py"; python tools/soak_rows.py 2>&1 | tail -n 3
echo "# python tools/soak_sharded.py --world 2 --iters 21"; python tools/soak_sharded.py --world 2 --iters 21 2>&1 | tail -n 3
echo "# python tools/soak_sharded.py --world 4 --iters 14"; python tools/soak_sharded.py --world 4 --iters 14 2>&1 | tail -n 3
} > $O/fuzz_soak.log 2>&1
tail -n 30 $O/fuzz_soak.log
