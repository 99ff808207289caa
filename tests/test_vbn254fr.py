"""The guest-visible batch ("vbn254fr") layer on ligero::hip_context (include/lig_hip_vbn254fr.hpp, mirror of
include/host_modules/vbn254fr.hpp): CPU test = the header compiles and links; GPU test = a batch program's committed rows
(every on_batch_* hook, in program order) equal an independent replay of the program with Python integers."""
import os
import subprocess

import numpy as np
import pytest

import hip_lib
import oracle_lib as ol

ROOT = hip_lib.ROOT
SRC = os.path.join(ROOT, "tests", "cpp", "vbn254fr_prog.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "vbn254fr_prog")
P = ol.P
K_, L_ = 512, 320
R256 = (1 << 256) % P


def build_exe():
    mod = hip_lib.load()
    if not os.path.exists(mod.LIB_PATH):
        mod.build()
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), SRC,
                           "-L" + os.path.dirname(mod.LIB_PATH), "-llig_hip",
                           "-Wl,-rpath," + os.path.dirname(mod.LIB_PATH), "-o", EXE])
    return EXE


def test_vbn254fr_header_compiles_and_links():
    subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-fsyntax-only", SRC])
    assert os.path.exists(build_exe())


class Model:
    """the same program on Python integers; a variable is a list of k values (l data slots + k - l padding slots)"""

    def __init__(self):
        self.log, self.inits = [], 0

    def init(self, vals, keep=None):
        """keep: the variable's previous content -- the write_limbs family (vbn254fr_set_str / _set_bytes, vbn254fr.hpp:200,251)
        clears nothing, slots beyond the written ones stay; None: write_buffer_clear (vbn254fr_set_ui, :154) zeroes the rest"""
        x = list(vals) + (list(keep[len(vals):]) if keep is not None else [0] * (K_ - len(vals)))
        for j in range(K_ - L_):
            x[L_ + j] = 1000 * self.inits + j
        self.inits += 1
        self.log.append(("I", [list(x)]))
        return x

    def quad(self, x, y, z): self.log.append(("Q", [list(x), list(y), list(z)]))
    def equal(self, x, y): self.log.append(("E", [list(x), list(y)]))
    def bit(self, x): self.log.append(("B", [list(x)]))


def replay():
    m = Model()
    Kc = 0x123456789abcdef0 | (0x0fedcba987654321 << 128)
    inv = lambda v: pow(v, P - 2, P)
    a = m.init([3 + 2 * i for i in range(10)])
    b = m.init([9] * L_)
    c = [x * y % P for x, y in zip(a, b)]; m.quad(a, b, c)
    d = [x * inv(y) % P for x, y in zip(c, b)]; m.quad(d, b, c)
    m.equal(d, a)
    e = [(x + y) % P for x, y in zip(a, b)]
    e = [(x + Kc) % P for x in e]
    e = [(x - y) % P for x, y in zip(e, a)]
    e = [(x - 77) % P for x in e]
    e = [(Kc - x) % P for x in e]
    e = [x * Kc % P for x in e]
    e = [x * Kc * inv(R256) % P for x in e]                # EltwiseMontMultMod: x * k / 2^256
    d = list(e); m.equal(d, e)
    m.equal(e, e)
    z = [x * x % P for x in e]; m.quad(e, e, z); e = z
    c = m.init([(0xffffffffffffffff | (0x1fffffffffffffff << 128)) % P, 1, 0], keep=c)      # vbn254fr_set over the product a*b
    d = m.init([Kc] * L_)
    c = [(x + y) % P for x, y in zip(c, d)]
    q = [x * inv(y) % P if y else 0 for x, y in zip(d, c)]; m.quad(q, c, d); c = q
    for i in range(254):
        m.bit([(x >> i) & 1 for x in d])
    return m


@pytest.mark.gpu
def test_vbn254fr_program_rows_match_integer_replay(tmp_path):
    exe = build_exe()
    log = tmp_path / "rows.bin"
    out = subprocess.check_output([exe, str(log)]).decode()
    lines = out.strip().splitlines()
    assert lines[0] == "handles 0 512 1024 1536 2048 size 320"          # element offsets i * k, FIFO free list
    assert lines[1] == "realloc 2560 free 507"                         # a freed variable goes to the back of the list
    assert lines[2] == "inits 4"
    raw = log.read_bytes()
    want = replay().log
    pos = 0
    for i, (kind, rows) in enumerate(want):
        assert chr(raw[pos]) == kind and raw[pos + 1] == len(rows), "record %d" % i
        pos += 2
        for r, row in enumerate(rows):
            got = np.frombuffer(raw, dtype=np.uint32, count=K_ * 8, offset=pos).reshape(K_, 8)
            pos += K_ * 32
            assert ol.from_limbs(got) == row, "record %d (%s) row %d" % (i, kind, r)
    assert pos == len(raw)


@pytest.mark.gpu
def test_recorded_program_proves_like_the_oracle(tmp_path):
    """the operations hip_vbn254fr records while it runs are a lig_batch_op program; proving it (HIP) gives the envelope the
    oracle gives for the same program, valid, and the committed batch rows are the rows the hooks saw (data slots)"""
    import ctypes as C

    import batch_prog
    exe = build_exe()
    log, prog_path = tmp_path / "rows.bin", tmp_path / "prog.bin"
    subprocess.check_output([exe, str(log), str(prog_path)])
    raw = prog_path.read_bytes()
    n_ops = int.from_bytes(raw[:8], "little")
    ops_b = raw[8:8 + 32 * n_ops]
    n_bytes = int.from_bytes(raw[8 + 32 * n_ops:16 + 32 * n_ops], "little")
    data_b = raw[16 + 32 * n_ops:16 + 32 * n_ops + n_bytes]
    assert n_ops == 21 and len(data_b) == n_bytes

    def attach(job):
        ops = (batch_prog.BatchOp * n_ops).from_buffer_copy(ops_b)
        data = (C.c_uint8 * n_bytes).from_buffer_copy(data_b)
        job.batch_ops, job.n_batch_ops = C.cast(ops, C.c_void_p), n_ops
        job.batch_data, job.batch_data_bytes = C.cast(data, C.c_void_p), n_bytes
        job._keep = (ops, data)
        return job

    amd = hip_lib.load()
    ojob = attach(ol.make_job(L_, K_, 2048, 192, 500, 0, generated_at=9, threads=4))
    pr = ol.Proof()
    assert ol.lib().lo_prove(C.byref(ojob), C.byref(pr)) == 0
    c = amd.Context(L_, K_, 2048)
    try:
        job = attach(amd.Context.make_job(500, 0, generated_at=9))
        tr = c.synth_prepare_job(job)
        proof, info = c.synth_prove(tr)
        c.trace_destroy(tr)
        assert (info.valid_code, info.valid_linear, info.valid_quad) == (1, 1, 1)
        assert (pr.valid_code, pr.valid_linear, pr.valid_quad) == (1, 1, 1)
        assert proof == bytes(pr.proof[:pr.proof_len])
        assert c.synth_verify(job, bytes(info.const_sum), proof).accept == 1
    finally:
        ol.lib().lo_proof_free(C.byref(pr))
        c.close()
