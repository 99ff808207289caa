"""Soak of the upload retry at configs[2]'s size (round 6): FRESH processes, each with one transfer of the uploader
thread made to "never complete" (LIG_FAULT_UPLOAD = 1: a chunk of the 550 MB witness matrix; 2: a chunk of the caller's randomness rows),
LIG_UPLOAD_TIMEOUT_S = 1; the pipelined caller-rows entry (commit(i) -> restart(i+1) -> prove(i)) must return the oracle pin's proof for
EVERY trace -- the one that hit the fault included -- and lig_upload_health must report one retry and nothing pending.
    python tools/soak_upload_fault.py [processes per fault kind] [proofs per process]        (prints one line per process + a summary)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes as C, hashlib, json, sys, time
import numpy as np, torch
sys.path.insert(0, "tests")
import hip_lib
amd = hip_lib.load()
N, with_rands = int(sys.argv[1]), sys.argv[2] == "2"
L_, K_, N_ = 8000, 8192, 32768
R = 2098
pin = json.load(open("tests/golden/full_pin_2p24.json"))
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
job.version = b"1.5.0"; job.set_public_args(None)
if not with_rands: job.dense_rands_per_row = per_row.ctypes.data          # (the library samples the synthetic stream's dense rows itself)
tr = C.c_void_p(); c.check(c.L.lig_rows_begin(c.h, C.byref(job), C.byref(tr)))
hp = C.c_void_p(host.data_ptr()); bad = 0; times = []
rands = torch.empty((R, K_, 8), dtype=torch.int32, pin_memory=True) if with_rands else None
for it in range(N):
    t0 = time.time()
    root, seed1 = c.rows_commit(tr)
    if it + 1 < N: c.check(c.L.lig_rows_restart(tr, hp, 0))
    if with_rands:                       # the synthetic stream's dense rows, made on the device, shipped back from HOST memory like a constraint generator's
        if it == 0:
            dr = c.malloc(R * K_ * 32)
            c.rng_fill_rows(bytes(seed1), 0, per_row, dr)
            c.check(c.L.lig_read(c.h, C.c_void_p(rands.data_ptr()), dr, R * K_ * 32)); c.free(dr)
        (addr, ln), info = c.rows_prove(tr, rands.data_ptr(), None, copy=False)
    else:
        (addr, ln), info = c.rows_prove(tr, None, None, copy=False)
    ok = hashlib.sha256(C.string_at(addr, ln)).hexdigest() == pin["proof_sha256"] and bytes(info.root).hex() == pin["root"] and info.valid_code and info.valid_linear and info.valid_quad
    bad += 0 if ok else 1
    times.append(round(time.time() - t0, 3))
print(json.dumps(dict(proofs=N, mismatches=bad, health=list(c.upload_health()), seconds=times)))
'''


def main():
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    proofs = sys.argv[2] if len(sys.argv) > 2 else "4"
    script = "/tmp/soak_upload_fault_child.py"
    with open(script, "w") as f:
        f.write(CHILD)
    total, bad, retried = 0, 0, 0
    for fault in ("1", "2"):
        for i in range(procs):
            p = subprocess.run([sys.executable, script, proofs, fault], cwd=ROOT, env=dict(os.environ, LIG_FAULT_UPLOAD=fault, LIG_UPLOAD_TIMEOUT_S="1"),
                               capture_output=True, timeout=600)
            lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not lines:
                print("fault %s process %d: rc %d %s" % (fault, i, p.returncode, p.stderr.decode()[-600:]))
                bad += 1
                continue
            out = json.loads(lines[-1])
            print("fault %s process %d: %s" % (fault, i, json.dumps(out)))
            total += out["proofs"]; bad += out["mismatches"] + (0 if out["health"] == [1, 0] else 1); retried += out["health"][0]
    print("soak_upload_fault: %d proofs in %d processes, %d retries, %d mismatches / unhealthy processes" % (total, 2 * procs, retried, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
