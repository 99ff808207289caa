#!/usr/bin/env python3
"""Stand-alone time of the column-hash chain (csrc/sha.hip) at the two geometries that matter:
  production  32768 columns x 2101 rows  (one 2^24-constraint proof on one GPU; chunks of 512 rows as the prover launches them)
  W = 8       4096 columns x 8392 rows   (what ONE rank of configs[3] hashes: 2^26 constraints over 8 GPUs)
for every setting of LIG_SHA_WS (0: one wave per 64 columns, the round 1-4 kernel; 1 / 2 / 4: wave-specialised, groups per workgroup).
The knob is read once per process, so every setting runs in a child.   python tools/sha_chain_bench.py [--reps 5]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(reps):
    import hashlib
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hip_lib
    amd = hip_lib.load()
    out = {}
    for name, n_inst, rows, chunk in (("production_32768x2101", 32768, 2101, 512), ("w8_4096x8392", 4096, 8392, 512)):
        c = amd.Context(8000, 8192, 32768)
        buf_rows = min(rows, 1024)
        d = c.malloc(buf_rows * n_inst * 32)
        c.rng_fill(bytes(range(32)), 0, d, buf_rows * n_inst)
        st, dl = c.sha_state(n_inst), c.malloc(32 * n_inst)
        L = c.L
        import ctypes as C
        best = None
        for rep in range(reps + 1):
            c.check(L.lig_sha_init(c.h, st, n_inst))
            c.sync()
            t0 = time.perf_counter()
            done = 0
            while done < rows:
                nb = min(chunk, rows - done, buf_rows)
                c.check(L.lig_sha_update_rows(c.h, st, d, nb))      # (the same rows again: the chain's time does not depend on the data)
                done += nb
            c.sync()
            dt = (time.perf_counter() - t0) * 1e3
            if rep and (best is None or dt < best):
                best = dt
        c.sha_final(st, dl)
        digest = hashlib.sha256(c.download(dl, (n_inst, 8)).tobytes()).hexdigest()
        out[name] = {"ms": round(best, 3), "us_per_compression": round(best * 1e3 / (rows / 2), 3), "leaves_sha256": digest[:16]}
        c.close()
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        return child(a.reps)
    res = {}
    for ws in ("0", "1", "2", "4"):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--reps", str(a.reps)], env=dict(os.environ, LIG_SHA_WS=ws), capture_output=True, text=True)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        res["LIG_SHA_WS=" + ws] = json.loads(line[-1]) if line else {"error": p.stderr[-500:]}
    print(json.dumps(res, indent=1))
    leaves = {k: tuple(v[g]["leaves_sha256"] for g in sorted(v) if "leaves_sha256" in v[g]) for k, v in res.items() if "error" not in v}
    print("all settings give the same leaves:", len(set(leaves.values())) == 1)


if __name__ == "__main__":
    main()
