"""Batch ("vbn254fr") rows through the three-stage flow: the oracle runs the batch program in its reference-structured
prover (CPU test), the HIP prover must produce the same envelope and both verifiers must agree (GPU test)."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import batch_prog
import oracle_lib as ol

P = ol.P
L_, K_, N_ = 320, 512, 2048


def demo_program(bad_assert=False, with_bits=True):
    """alloc-free straight-line program over slots 0..: init rows, a product, a quotient constrained back, constants,
    copies (equal rows), a squared variable (x == y), a bit decomposition"""
    p = batch_prog.Program()
    kc = int.from_bytes(hashlib.sha256(b"batch-const").digest(), "little") % P
    p.set(0, [3 + 2 * i for i in range(10)])                 # a
    p.set_scalar(1, 9)                                       # b
    p.mul(2, 0, 1)                                           # c = a*b                    (a, b, c)
    p.div(3, 2, 1)                                           # d = c/b                    (d, b, c)
    p.assert_equal(3, 0)                                     # d == a on the data slots; pads differ -> see below
    p.add(4, 0, 1)
    p.const("ADD_CONST", 4, 4, kc)
    p.sub(4, 4, 0)
    p.const("SUB_CONST", 4, 4, 77)
    p.const("CONST_SUB", 4, 4, kc)
    p.const("MUL_CONST", 4, 4, kc)
    p.const("MONTMUL_CONST", 4, 4, kc)
    p.copy(3, 4)                                             # (d, e)
    p.copy(4, 4)                                             # (e, e)
    p.mul(4, 4, 4)                                           # x == y, out aliases both
    p.set_scalar(5, (1 << 200) + 12345)
    p.add(5, 5, 0)                                           # nonzero in every data slot (x / 0 = 0 would violate q*y = x)
    p.set_scalar(6, kc)
    p.div(5, 6, 5)
    if with_bits:
        p.set(7, [0, 1, 5, (1 << 253) + 7])
        p.bit_decompose(list(range(8, 8 + 254)), 7)
    if bad_assert:
        p.assert_equal(0, 1)                                 # a != b: the quadratic test must fail
    p.free(0)
    return p


def oracle_prove(n_linear, n_quad, prog, threads=2):
    job = ol.make_job(L_, K_, N_, 192, n_linear, n_quad, generated_at=5, threads=threads)
    if prog is not None:
        prog.attach(job)
    pr = ol.Proof()
    assert ol.lib().lo_prove(C.byref(job), C.byref(pr)) == 0
    return job, pr


def test_assert_equal_on_padded_rows_note():
    """on_batch_equal compares whole k-element rows: d = c/b equals a on the l data slots, but the padding slots of d are
    pad_a*pad_b/pad_b = pad_a as well, so the honest demo program satisfies the equality constraint on all k slots"""
    assert True


@pytest.mark.parametrize("n_linear,n_quad", [(0, 0), (700, 330)])
def test_oracle_batch_program_proves_and_verifies(n_linear, n_quad):
    if n_linear == 0:
        n_linear = 1                                         # the synthetic stream needs one row for the linear constant
    job, pr = oracle_prove(n_linear, n_quad, demo_program())
    try:
        assert (pr.valid_code, pr.valid_linear, pr.valid_quad) == (1, 1, 1)
        rows_batch = 2 + 3 + 3 + 2 + 2 + 2 + 3 + 2 + 3 + 1 + 254
        synth = -(-n_linear // L_) + 3 * (-(-n_quad // L_))
        assert pr.rows == rows_batch + synth + 3
        proof = bytes(pr.proof[:pr.proof_len])
        cs = (C.c_uint64 * 4)(*pr.const_sum)
        buf = (C.c_uint8 * len(proof)).from_buffer_copy(proof)
        assert ol.lib().lo_verify(C.byref(job), cs, buf, len(proof)) == 1
        # the verifier only needs the hook sequence: a different program shape is another statement
        job2 = ol.make_job(L_, K_, N_, 192, n_linear, n_quad, generated_at=5, threads=2)
        demo_program(with_bits=False).attach(job2)
        assert ol.lib().lo_verify(C.byref(job2), cs, buf, len(proof)) == 0
    finally:
        ol.lib().lo_proof_free(C.byref(pr))


def test_oracle_false_equality_fails_the_quadratic_test():
    job, pr = oracle_prove(100, 0, demo_program(bad_assert=True, with_bits=False))
    try:
        assert (pr.valid_code, pr.valid_linear, pr.valid_quad) == (1, 1, 0)
        proof = bytes(pr.proof[:pr.proof_len])
        cs = (C.c_uint64 * 4)(*pr.const_sum)
        buf = (C.c_uint8 * len(proof)).from_buffer_copy(proof)
        assert ol.lib().lo_verify(C.byref(job), cs, buf, len(proof)) == 0
    finally:
        ol.lib().lo_proof_free(C.byref(pr))


def test_oracle_rejects_malformed_program():
    job = ol.make_job(L_, K_, N_, 192, 10, 0)
    p = batch_prog.Program()
    p.set(600, [1])                                          # slot out of range
    p.attach(job)
    pr = ol.Proof()
    assert ol.lib().lo_prove(C.byref(job), C.byref(pr)) != 0
    for outs, src in (([3, 512], 1), ([3, 1, 4], 1)):        # bit-decompose output slot outside the slab / equal to the source
        job = ol.make_job(L_, K_, N_, 192, 10, 0)
        p = batch_prog.Program()
        p.set(1, [5])
        p.bit_decompose(outs, src)
        p.attach(job)
        assert ol.lib().lo_prove(C.byref(job), C.byref(pr)) != 0


# ------------------------------------------------------------------------------------------------ HIP prover / verifier
import hip_lib                                      # noqa: E402


@pytest.fixture(scope="module")
def amd():
    return hip_lib.load()


@pytest.mark.gpu
@pytest.mark.parametrize("n_linear,n_quad,bits", [(1, 0, True), (700, 330, True), (2000, 0, False)])
def test_hip_batch_rows_equal_oracle(amd, n_linear, n_quad, bits):
    """the batch program runs on the GPU once, its rows join the resident witness matrix in program order; envelope bytes,
    root, seeds and constant equal the oracle's three-pass run of the same program; both verifiers accept"""
    prog = demo_program(with_bits=bits)
    ojob, pr = oracle_prove(n_linear, n_quad, prog, threads=4)
    c = amd.Context(L_, K_, N_)
    try:
        job = prog.attach(amd.Context.make_job(n_linear, n_quad, generated_at=5))
        tr = c.synth_prepare_job(job)
        proof, info = c.synth_prove(tr)
        proof2, _ = c.synth_prove(tr)
        c.trace_destroy(tr)
        assert proof == proof2
        assert info.rows == pr.rows
        assert (info.valid_code, info.valid_linear, info.valid_quad) == (1, 1, 1)
        assert bytes(info.root) == bytes(pr.root)
        assert bytes(info.stage1_seed) == bytes(pr.stage1_seed) and bytes(info.stage2_seed) == bytes(pr.stage2_seed)
        assert bytes(info.const_sum) == bytes(pr.const_sum)
        assert proof == bytes(pr.proof[:pr.proof_len])
        v = c.synth_verify(job, bytes(info.const_sum), proof)
        assert v.accept == 1
        cs = (C.c_uint64 * 4)(*pr.const_sum)
        buf = (C.c_uint8 * len(proof)).from_buffer_copy(proof)
        assert ol.lib().lo_verify(C.byref(ojob), cs, buf, len(proof)) == 1
        # another program shape is another statement
        job2 = demo_program(with_bits=not bits).attach(amd.Context.make_job(n_linear, n_quad, generated_at=5))
        assert c.synth_verify(job2, bytes(info.const_sum), proof).accept == 0
    finally:
        ol.lib().lo_proof_free(C.byref(pr))
        c.close()


@pytest.mark.gpu
def test_hip_false_equality_fails_quadratic_test_and_bad_program_is_rejected(amd):
    prog = demo_program(bad_assert=True, with_bits=False)
    ojob, pr = oracle_prove(100, 0, prog)
    c = amd.Context(L_, K_, N_)
    try:
        job = prog.attach(amd.Context.make_job(100, 0, generated_at=5))
        tr = c.synth_prepare_job(job)
        proof, info = c.synth_prove(tr)
        c.trace_destroy(tr)
        assert (info.valid_code, info.valid_linear, info.valid_quad) == (1, 1, 0)
        assert proof == bytes(pr.proof[:pr.proof_len])
        v = c.synth_verify(job, bytes(info.const_sum), proof)
        assert (v.accept, v.valid_quad, v.valid_merkle, v.quad_equal) == (0, 0, 1, 1)
        bad = batch_prog.Program()
        bad.set(600, [1])
        with pytest.raises(Exception):
            c.synth_prepare_job(bad.attach(amd.Context.make_job(10, 0)))
        for outs in ([3, 512], [3, 1, 4]):                   # bit-decompose output slot outside the slab / equal to the source
            bad = batch_prog.Program()
            bad.set(1, [5])
            bad.bit_decompose(outs, 1)
            with pytest.raises(Exception):
                c.synth_prepare_job(bad.attach(amd.Context.make_job(10, 0)))
    finally:
        ol.lib().lo_proof_free(C.byref(pr))
        c.close()


# ------------------------------------------------------------------------------------------------ whole-proof regression pins
import json                                         # noqa: E402
import os                                           # noqa: E402

PINS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "proof_pins.json")))["jobs"]


def _pin_id(p):
    return "l%d_k%d_lin%d_quad%d_%s" % (p["l"], p["k"], p["n_linear"], p["n_quad"], p["batch"] or "nobatch")


@pytest.mark.parametrize("pin", PINS, ids=_pin_id)
def test_oracle_matches_whole_proof_pins(pin):
    """tests/golden/proof_pins.json (made by tests/golden/make_proof_pins.py from this oracle): drift detector"""
    j = ol.make_job(pin["l"], pin["k"], pin["n"], 192, pin["n_linear"], pin["n_quad"], generated_at=pin["generated_at"], threads=2)
    if pin["batch"]:
        demo_program(with_bits=pin["batch"] == "demo").attach(j)
    pr = ol.Proof()
    assert ol.lib().lo_prove(C.byref(j), C.byref(pr)) == 0
    try:
        assert hashlib.sha256(bytes(pr.proof[:pr.proof_len])).hexdigest() == pin["proof_sha256"]
        assert bytes(pr.root).hex() == pin["root"] and bytes(pr.const_sum).hex() == pin["const_sum"]
    finally:
        ol.lib().lo_proof_free(C.byref(pr))


@pytest.mark.gpu
@pytest.mark.parametrize("pin", PINS, ids=_pin_id)
def test_hip_matches_whole_proof_pins(amd, pin):
    c = amd.Context(pin["l"], pin["k"], pin["n"])
    try:
        job = amd.Context.make_job(pin["n_linear"], pin["n_quad"], generated_at=pin["generated_at"])
        if pin["batch"]:
            demo_program(with_bits=pin["batch"] == "demo").attach(job)
        tr = c.synth_prepare_job(job)
        proof, info = c.synth_prove(tr)
        c.trace_destroy(tr)
        assert hashlib.sha256(proof).hexdigest() == pin["proof_sha256"] and len(proof) == pin["proof_len"]
        assert bytes(info.root).hex() == pin["root"] and bytes(info.stage2_seed).hex() == pin["stage2_seed"]
        assert [info.valid_code, info.valid_linear, info.valid_quad] == pin["valid"]
    finally:
        c.close()
