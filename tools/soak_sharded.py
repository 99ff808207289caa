#!/usr/bin/env python3
"""Soak of the sharded prover over the stream-ordered process-to-process communicator (csrc/comm_ipc.hip): W processes on
the one GPU keep ONE communicator and alternate trace shapes (1 .. 6 exchange rounds, quadratic rows, a rank without rows),
so that buffers are freed and re-allocated under recycled addresses between proofs; every envelope of every rank is
compared with the unsharded prover's.  Not part of the product path.

    python tools/soak_sharded.py --world 2 --iters 40        (launcher; spawns the ranks)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--iters", type=int, default=21)
    ap.add_argument("--comm", default="ipc")
    a = ap.parse_args()
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
