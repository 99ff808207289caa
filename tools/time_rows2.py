"""two contexts feeding traces from pinned host memory at once (value_incl_h2d with --h2d-inflight 2): per-call wall times"""
import ctypes as C
import hashlib
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, "tests")
import hip_lib

amd = hip_lib.load()
L_, K_, N_ = 8000, 8192, 32768
R = 2098
NCTX = int(sys.argv[1]) if len(sys.argv) > 1 else 2
STAGGER = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
ctxs = [amd.Context(L_, K_, N_) for _ in range(NCTX)]
per_row = np.full(R, L_, dtype=np.uint32)
per_row[-1] = (1 << 24) % L_
host = torch.empty((R, K_, 8), dtype=torch.int32, pin_memory=True)
c = ctxs[0]
d = c.malloc(R * K_ * 32)
c.rng_fill_rows(hashlib.sha256(b"lig-synth" + (1).to_bytes(8, "little")).digest(), 0, per_row, d)
c.check(c.L.lig_read(c.h, C.c_void_p(host.data_ptr()), d, R * K_ * 32))
c.free(d)
kinds = np.full(R, amd.ROW_KINDS["LINEAR"] | amd.ROW_DRAW_PAD, dtype=np.uint8)
hp = C.c_void_p(host.data_ptr())
traces, keep = [], []
for c in ctxs:
    job = amd.RowsJob()
    job.rows = R; job.kinds = kinds.ctypes.data; job.msgs = host.data_ptr(); job.msgs_on_device = 0
    for i in range(32):
        job.encoding_seed[i] = i
    job.version = b"1.5.0"; job.set_public_args(None); job.dense_rands_per_row = per_row.ctypes.data
    tr = C.c_void_p()
    c.check(c.L.lig_rows_begin(c.h, C.byref(job), C.byref(tr)))
    traces.append(tr); keep.append(job)
T0 = time.perf_counter()
lines = []

def loop(i, steps):
    c, tr = ctxs[i], traces[i]
    if STAGGER and i:
        time.sleep(STAGGER * 1e-3 * i)
    loaded = True
    for s_ in range(steps):
        t0 = time.perf_counter()
        if not loaded:
            c.check(c.L.lig_rows_restart(tr, hp, 0))
        c.rows_commit(tr)
        t2 = time.perf_counter()
        loaded = s_ + 1 < steps
        if loaded:
            c.check(c.L.lig_rows_restart(tr, hp, 0))
        t3 = time.perf_counter()
        (addr, ln), info = c.rows_prove(tr, None, None, copy=False)
        t4 = time.perf_counter()
        lines.append("ctx %d step %d @%.1f: commit %.2f restart %.2f prove %.2f total %.2f | lib stage1 %.2f stage2 %.2f stage3 %.2f" %
                     (i, s_, 1e3 * (t0 - T0), 1e3 * (t2 - t0), 1e3 * (t3 - t2), 1e3 * (t4 - t3), 1e3 * (t4 - t0), info.ms_stage1, info.ms_stage2, info.ms_stage3))

for rep in range(2):
    for i in range(NCTX):                         # every trace starts loaded
        pass
    T0 = time.perf_counter()
    th = [threading.Thread(target=loop, args=(i, 6)) for i in range(NCTX)]
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - T0
    print("rep %d: %d proofs in %.2f ms = %.2f ms per proof" % (rep, 6 * NCTX, 1e3 * dt, 1e3 * dt / (6 * NCTX)))
    for i in range(NCTX):
        ctxs[i].check(ctxs[i].L.lig_rows_restart(traces[i], hp, 0))
    for c in ctxs: c.sync()
print("\n".join(lines))
