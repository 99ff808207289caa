#!/usr/bin/env python3
"""Soak of the sharded prover over the stream-ordered process-to-process communicator (csrc/comm_ipc.hip): W processes on
the one GPU keep ONE communicator and alternate trace shapes (1 .. 6 exchange rounds, quadratic rows, a rank without rows),
so that buffers are freed and re-allocated under recycled addresses between proofs; every envelope of every rank is
compared with the unsharded prover's.  Not part of the product path.

    python tools/soak_sharded.py --world 2 --iters 40        (launcher; spawns the ranks)
    python tools/soak_sharded.py --rows-entry --iters 200    (round 5: FRESH processes per iteration through the lig_shard_rows_* entry --
                                                             the test that hung once in the round-4 driver run -- every rank's stderr kept)
"""
import argparse
import hashlib
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(2000, 900), (320 * 1500 + 7, 330), (700, 0), (320 * 5200 + 3, 960), (320 * 2600 + 1, 0), (0, 320 * 40 + 3), (320 * 1100, 0)]


def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "ligero-prover_amd", rel))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def worker(iters):
    pkg, dist = load("ligero_prover_amd", "__init__.py"), load("lig_dist", "dist.py")
    g = dist.Group("gloo")
    ctx = pkg.Context(320, 512, 2048, device=0)
    comm = g.make_comm(pkg, ctx)
    bad = 0
    for it in range(iters):
        nl, nq = SHAPES[it % len(SHAPES)]
        job = pkg.Context.make_job(nl, nq, generated_at=it)
        sh = ctx.shard_prepare(job, g.rank, g.world, comm)
        outs = [ctx.shard_prove(sh) for _ in range(1 + it % 3)]
        proofs = [o[0] for o in outs]
        ctx.shard_destroy(sh)
        ref = rinfo = None
        if g.rank == 0:
            tr = ctx.synth_prepare_job(job)
            ref, rinfo = ctx.synth_prove(tr)
            ctx.trace_destroy(tr)
        digs = g.gather_digests(hashlib.sha256(proofs[-1]).digest())
        ok = len(set(digs)) == 1 and all(p == proofs[0] for p in proofs) and (ref is None or ref == proofs[0])
        bad += not ok
        if not ok:
            def parts(info):
                return "root %s seed2 %s const %s valid %d%d%d" % (bytes(info.root).hex()[:8], bytes(info.stage2_seed).hex()[:8], bytes(info.const_sum).hex()[:8],
                                                                    info.valid_code, info.valid_linear, info.valid_quad)
            print("rank %d: MISMATCH at iteration %d (shape %r): ranks agree %s, repeated proofs %s, vs unsharded %s\n    sharded: %s\n    unsharded: %s"
                  % (g.rank, it, (nl, nq), len(set(digs)) == 1, [p == proofs[0] for p in proofs], None if ref is None else ref == proofs[0],
                     " | ".join(parts(o[1]) for o in outs), parts(rinfo) if rinfo is not None else "-"), flush=True)
    print(json.dumps({"rank": g.rank, "iters": iters, "mismatches": bad}), flush=True)
    g.close()
    ctx.close()
    return bad


def rows_entry_soak(iters, only_hung, log, extra_env=None):
    """tests/test_gpu_sharded.py::test_sharded_rows_entry_equals_rows_prove_and_oracle, its parameters in a loop, every iteration in
    fresh processes (as under pytest); the parameter that hung in the round-4 driver run ([2-900-330-True-own_pads-ipc]) every other
    iteration.  Failures (with every rank's stderr) go to `log`; the run goes on."""
    import tempfile
    import time
    import pathlib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import multirank as mr
    import test_gpu_sharded as tgs
    hung = (2, 900, 330, True, "own_pads", "ipc")
    others = [(2, 2000, 900, False, "own_pads", "ipc"), (4, 700, 0, False, "library_pads", "ipc"), (2, 320 * 1500 + 7, 330, False, "device_rows", "ipc"),
              (4, 320 * 4300 + 1, 0, False, "library_pads", "ipc"), (2, 320 * 700 + 9, 335, False, "dense_rands", "ipc"), (4, 900, 330, True, "own_pads", "ipc"),
              (2, 900, 330, True, "library_pads", "ipc"), (8, 900, 330, True, "own_pads", "ipc")]
    tmp = pathlib.Path(tempfile.mkdtemp(prefix="lig_soak_"))
    bad, t_all, worst = 0, time.time(), 0.0
    with open(log, "a") as f:
        for it in range(iters):
            par = hung if (only_hung or it % 2 == 0) else others[(it // 2) % len(others)]
            world, n_lin, n_quad, batch, mode, comm = par
            t0 = time.time()
            try:
                outs = tgs.run_rows_world(tmp, world, 320, 512, 2048, n_lin, n_quad, batch, mode, comm, timeout=90, **dict({"LIG_IPC_STALL_S": 20}, **(extra_env or {})))
                ok = all(o["valid"] == [1, 1, 1] and o["again"] and o["const"] and o["all_equal"] for o in outs) and outs[0]["equals_rows_prove"] and outs[0]["equals_oracle"]
                err = "" if ok else "MISMATCH %r" % (outs,)
            except AssertionError as e:
                ok, err = False, str(e)
            dt = time.time() - t0
            worst = max(worst, dt)
            bad += not ok
            f.write("iter %d %r: %s in %.1f s\n" % (it, par, "ok" if ok else "FAILED", dt))
            if not ok:
                f.write(err + "\n")
            f.flush()
        f.write("rows-entry soak: %d iterations, %d failures, slowest %.1f s, total %.0f s\n" % (iters, bad, worst, time.time() - t_all))
    print("rows-entry soak: %d iterations, %d failures, slowest iteration %.1f s" % (iters, bad, worst))
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--iters", type=int, default=21)
    ap.add_argument("--comm", default="ipc")
    ap.add_argument("--rows-entry", action="store_true")
    ap.add_argument("--only-hung", action="store_true")
    ap.add_argument("--log", default="soak_rows_entry.log")
    ap.add_argument("--env", action="append", default=[], help="KEY=VALUE for the ranks (repeatable)")
    a = ap.parse_args()
    if a.rows_entry:
        sys.exit(1 if rows_entry_soak(a.iters, a.only_hung, a.log, dict(e.split("=", 1) for e in a.env)) else 0)
    if "RANK" in os.environ:
        sys.exit(1 if worker(a.iters) else 0)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29931", WORLD_SIZE=str(a.world), LIG_COMM_TAG=str(os.getpid()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    if a.comm == "ipc":
        env["LIG_COMM"] = "ipc"
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--iters", str(a.iters)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(a.world)]
    rc = max(p.wait() for p in procs)
    print("soak_sharded: world %d, %d iterations, %s" % (a.world, a.iters, "OK" if rc == 0 else "FAILED"))
    sys.exit(rc)


if __name__ == "__main__":
    main()
