"""One-off: random trace shapes through the sharded prover (W processes sharing the one GPU, gloo collectives) against the
single-GPU prover.  python tools/fuzz_sharded.py [cases] [seed]"""
import pathlib
import random
import sys
import tempfile
sys.path.insert(0, "tests")
import test_gpu_sharded as ts

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
for i in range(cases):
    world = rng.choice([2, 4, 8])
    n_lin = rng.choice([0, 1, rng.randint(0, 3000), rng.randint(0, 200000)])
    n_quad = rng.choice([0, rng.randint(0, 2000), rng.randint(0, 60000)])
    batch = rng.random() < 0.4
    with tempfile.TemporaryDirectory() as d:
        outs = ts.run_world(pathlib.Path(d), world, 320, 512, 2048, n_lin, n_quad, batch=batch, comm=rng.choice([None, "ipc"]))
    ok = all(o["again"] and o["all_equal"] for o in outs) and outs[0]["ref_sha"] == outs[0]["sha"]
    print("case %2d W=%d lin %6d quad %6d batch %-5s rows %5d -> %s" % (i, world, n_lin, n_quad, batch, outs[0]["rows"], "ok" if ok else "MISMATCH"), flush=True)
    if not ok:
        sys.exit(1)
print("all", cases, "sharded cases identical to the single-GPU proof")
