"""timing probe of the caller-rows entry with the witness in pinned host memory (one context): per-call wall times"""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "tests")
import hip_lib

amd = hip_lib.load()
L_, K_, N_ = 8000, 8192, 32768
R = 2098
c = amd.Context(L_, K_, N_)
per_row = np.full(R, L_, dtype=np.uint32)
per_row[-1] = (1 << 24) % L_
host = torch.empty((R, K_, 8), dtype=torch.int32, pin_memory=True)
d = c.malloc(R * K_ * 32)
import hashlib
c.rng_fill_rows(hashlib.sha256(b"lig-synth" + (1).to_bytes(8, "little")).digest(), 0, per_row, d)
c.check(c.L.lig_read(c.h, C.c_void_p(host.data_ptr()), d, R * K_ * 32))
kinds = np.full(R, amd.ROW_KINDS["LINEAR"] | amd.ROW_DRAW_PAD, dtype=np.uint8)
hp = C.c_void_p(host.data_ptr())
t0 = time.perf_counter()
tr, keep = c.rows_begin(kinds, host.data_ptr(), on_device=True if False else False, dense_rands_per_row=per_row) if False else (None, None)
job = amd.RowsJob()
job.rows = R; job.kinds = kinds.ctypes.data; job.msgs = host.data_ptr(); job.msgs_on_device = 0
for i in range(32):
    job.encoding_seed[i] = i
job.version = b"1.5.0"; job.set_public_args(None); job.dense_rands_per_row = per_row.ctypes.data
tr = C.c_void_p()
c.check(c.L.lig_rows_begin(c.h, C.byref(job), C.byref(tr)))
print("begin %.2f ms" % (1e3 * (time.perf_counter() - t0)))
mode = sys.argv[1] if len(sys.argv) > 1 else "pipelined"
for it in range(6):
    t0 = time.perf_counter()
    if it and mode != "pipelined":
        c.check(c.L.lig_rows_restart(tr, hp, 0))
    t1 = time.perf_counter()
    c.rows_commit(tr)
    t2 = time.perf_counter()
    if mode == "pipelined":
        c.check(c.L.lig_rows_restart(tr, hp, 0))
    t3 = time.perf_counter()
    (addr, ln), info = c.rows_prove(tr, None, None, copy=False)
    t4 = time.perf_counter()
    print("%s it %d: restart %.2f commit %.2f restart %.2f prove %.2f total %.2f | lib stage1 %.2f stage2 %.2f stage3 %.2f" %
          (mode, it, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), 1e3 * (t4 - t0), info.ms_stage1, info.ms_stage2, info.ms_stage3))

# bench-style: a run() of 5 steps starting with nothing loaded
def loop(steps, tag):
    loaded = False
    for s_ in range(steps):
        t0 = time.perf_counter()
        if not loaded:
            c.check(c.L.lig_rows_restart(tr, hp, 0))
        t1 = time.perf_counter()
        c.rows_commit(tr)
        t2 = time.perf_counter()
        loaded = s_ + 1 < steps
        if loaded:
            c.check(c.L.lig_rows_restart(tr, hp, 0))
        t3 = time.perf_counter()
        (addr, ln), info = c.rows_prove(tr, None, None, copy=False)
        t4 = time.perf_counter()
        print("%s step %d: restart %.2f commit %.2f restart %.2f prove %.2f | stage2 %.2f stage3 %.2f" % (tag, s_, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), info.ms_stage2, info.ms_stage3))
for rep in range(2):
    t0 = time.perf_counter()
    loop(5, "run%d" % rep)
    print("run(5): %.2f ms per proof" % (1e3 * (time.perf_counter() - t0) / 5))
from concurrent.futures import ThreadPoolExecutor
pool = ThreadPoolExecutor(max_workers=1)
for rep in range(2):
    t0 = time.perf_counter()
    list(pool.map(lambda r: loop(5, "thr%d" % r), [rep]))
    print("threaded run(5): %.2f ms per proof" % (1e3 * (time.perf_counter() - t0) / 5))
t0 = time.perf_counter()
loop(5, "main-after")
print("main after threaded run(5): %.2f ms per proof" % (1e3 * (time.perf_counter() - t0) / 5))
torch.cuda.synchronize()
t0 = time.perf_counter()
loop(5, "after-torch-sync")
print("after torch.cuda.synchronize run(5): %.2f ms per proof" % (1e3 * (time.perf_counter() - t0) / 5))
