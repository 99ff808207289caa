"""soak of the real-driver pipeline (round 4): witness rows AND the caller's randomness rows in pinned host memory, two contexts on
two host threads, commit(i) -> restart(i+1) -> prove(i): even iterations hand the whole randomness matrix to lig_rows_prove (the
uploader thread fills the double buffer), odd iterations push it in three pieces (lig_rows_push_rands) and prove with NULL.
EVERY proof is compared with the oracle pin.      python tools/soak_rands.py [proofs per context]"""
import ctypes as C, hashlib, json, sys, threading
import numpy as np, torch
sys.path.insert(0, "tests")
import hip_lib
amd = hip_lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
L_, K_, N_ = 8000, 8192, 32768
R = 2098
pin = json.load(open("tests/golden/full_pin_2p24.json"))["proof_sha256"]
per_row = np.full(R, L_, dtype=np.uint32); per_row[-1] = (1 << 24) % L_
kinds = np.full(R, amd.ROW_KINDS["LINEAR"] | amd.ROW_DRAW_PAD, dtype=np.uint8)
c0 = amd.Context(L_, K_, N_)
host = torch.empty((R, K_, 8), dtype=torch.int32, pin_memory=True)
d = c0.malloc(R * K_ * 32)
c0.rng_fill_rows(hashlib.sha256(b"lig-synth" + (1).to_bytes(8, "little")).digest(), 0, per_row, d)
c0.check(c0.L.lig_read(c0.h, C.c_void_p(host.data_ptr()), d, R * K_ * 32))
rands = torch.empty((R, K_, 8), dtype=torch.int32, pin_memory=True)
bad, lock = [], threading.Lock()


def begin(c):
    job = amd.RowsJob()
    job.rows = R; job.kinds = kinds.ctypes.data; job.msgs = host.data_ptr(); job.msgs_on_device = 0
    for i in range(32): job.encoding_seed[i] = i
    job.version = b"1.5.0"; job.set_public_args(None)
    tr = C.c_void_p(); c.check(c.L.lig_rows_begin(c.h, C.byref(job), C.byref(tr)))
    return tr, job


tr0, keep0 = begin(c0)
_, seed = c0.rows_commit(tr0)                         # the same trace every time => the same seed => the same randomness rows
c0.rng_fill_rows(seed, 0, per_row, d)
c0.check(c0.L.lig_read(c0.h, C.c_void_p(rands.data_ptr()), d, R * K_ * 32)); c0.free(d)
c0.rows_prove(tr0, rands.data_ptr(), None, copy=False)
hp = C.c_void_p(host.data_ptr())
row_bytes = K_ * 32


def run(i, c, tr, first_loaded):
    loaded = first_loaded
    for it in range(N):
        if not loaded: c.check(c.L.lig_rows_restart(tr, hp, 0))
        c.rows_commit(tr)
        loaded = it + 1 < N
        if loaded: c.check(c.L.lig_rows_restart(tr, hp, 0))
        if it & 1:
            cuts = [0, 700, 1500, R]
            for a, b in zip(cuts[:-1], cuts[1:]):
                c.rows_push_rands(tr, a, b - a, rands.data_ptr() + a * row_bytes)
            (addr, ln), info = c.rows_prove(tr, None, None, copy=False)
        else:
            (addr, ln), info = c.rows_prove(tr, rands.data_ptr(), None, copy=False)
        h = hashlib.sha256(C.string_at(addr, ln)).hexdigest()
        if h != pin or not (info.valid_code and info.valid_linear and info.valid_quad):
            with lock: bad.append((i, it, h[:12]))


c1 = amd.Context(L_, K_, N_)
tr1, keep1 = begin(c1)
th = [threading.Thread(target=run, args=(0, c0, tr0, False)), threading.Thread(target=run, args=(1, c1, tr1, True))]
[t.start() for t in th]; [t.join() for t in th]
print("soak_rands: %d proofs (host randomness rows: whole matrix / pushed in pieces, alternating), %d mismatches" % (2 * N, len(bad)), bad[:3])
sys.exit(1 if bad else 0)
