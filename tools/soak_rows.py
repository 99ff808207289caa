"""soak of the caller-rows pipeline: witness in pinned host memory, commit(i) -> restart(i+1) -> prove(i) so that the upload of
the next trace overlaps the proof of the current one; EVERY proof is compared with the oracle pin.  python tools/soak_rows.py [n]"""
import ctypes as C, hashlib, json, sys
import numpy as np, torch
sys.path.insert(0, "tests")
import hip_lib
amd = hip_lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
L_, K_, N_ = 8000, 8192, 32768
R = 2098
pin = json.load(open("tests/golden/full_pin_2p24.json"))["proof_sha256"]
c = amd.Context(L_, K_, N_)
per_row = np.full(R, L_, dtype=np.uint32); per_row[-1] = (1 << 24) % L_
host = torch.empty((R, K_, 8), dtype=torch.int32, pin_memory=True)
d = c.malloc(R * K_ * 32)
c.rng_fill_rows(hashlib.sha256(b"lig-synth" + (1).to_bytes(8, "little")).digest(), 0, per_row, d)
c.check(c.L.lig_read(c.h, C.c_void_p(host.data_ptr()), d, R * K_ * 32)); c.free(d)
kinds = np.full(R, amd.ROW_KINDS["LINEAR"] | amd.ROW_DRAW_PAD, dtype=np.uint8)
job = amd.RowsJob()
job.rows = R; job.kinds = kinds.ctypes.data; job.msgs = host.data_ptr(); job.msgs_on_device = 0
for i in range(32): job.encoding_seed[i] = i
job.version = b"1.5.0"; job.set_public_args(None); job.dense_rands_per_row = per_row.ctypes.data
tr = C.c_void_p(); c.check(c.L.lig_rows_begin(c.h, C.byref(job), C.byref(tr)))
hp = C.c_void_p(host.data_ptr()); bad = 0
for it in range(N):
    c.rows_commit(tr)
    if it + 1 < N: c.check(c.L.lig_rows_restart(tr, hp, 0))
    (addr, ln), info = c.rows_prove(tr, None, None, copy=False)
    if hashlib.sha256(C.string_at(addr, ln)).hexdigest() != pin or not (info.valid_code and info.valid_linear and info.valid_quad): bad += 1
print("soak_rows: %d proofs, %d mismatches" % (N, bad))
sys.exit(1 if bad else 0)
