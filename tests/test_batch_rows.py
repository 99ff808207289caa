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


def slicing_program(upstream):
    """a program on which the two slicing semantics of buffer_view (include/lig_hip.h, LIG_BOP_UPSTREAM_COMPAT) differ and which is a
    TRUE statement under both: variable 0 through write_buffer_clear (the same either way), variables written by the write_limbs
    family (data intact either way; their on_batch_init pads go to their own slots as declared / to variable 0's as upstream
    defines slice_bytes), and one write_buffer_clear on variable 6 (declared: clears the rest of variable 6; upstream: wipes the
    slab from variable 0's slot l to the end of variable 6 -- products of zeros still satisfy x*y = z)"""
    p = batch_prog.Program()
    if upstream:
        p.upstream_compat()
    p.set(0, [3 + 2 * i for i in range(10)])
    p.set_scalar(1, 9, limbs=True)
    p.mul(2, 0, 1)
    p.set(3, [5, 6, 7], limbs=True)                          # slots 3.. keep what variable 3 held (zeros)
    p.add(4, 2, 3)
    p.copy(5, 4)
    p.mul(4, 4, 4)
    p.set(2, [11, 12], limbs=True)                           # over the product: slots 2.. keep the product's values
    p.mul(7, 2, 5)
    p.set_scalar(6, 77)
    p.mul(7, 6, 6)
    p.mul(7, 7, 2)
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


@pytest.mark.parametrize("upstream", [False, True])
def test_oracle_slicing_semantics_declared_and_upstream(upstream):
    """both semantics give a valid, verifying proof of the slicing program -- and different ones: the upstream definition of
    buffer_view::slice_bytes (src/webgpu/buffer_view.cpp:91-95) puts every on_batch_init pad into variable 0 and lets
    write_buffer_clear wipe the slab up to the end of the variable"""
    job, pr = oracle_prove(100, 0, slicing_program(upstream))
    try:
        assert (pr.valid_code, pr.valid_linear, pr.valid_quad) == (1, 1, 1)
        proof = bytes(pr.proof[:pr.proof_len])
        cs = (C.c_uint64 * 4)(*pr.const_sum)
        buf = (C.c_uint8 * len(proof)).from_buffer_copy(proof)
        assert ol.lib().lo_verify(C.byref(job), cs, buf, len(proof)) == 1
        job2, pr2 = oracle_prove(100, 0, slicing_program(not upstream))
        try:
            assert pr2.rows == pr.rows and bytes(pr2.root) != bytes(pr.root)
        finally:
            ol.lib().lo_proof_free(C.byref(pr2))
    finally:
        ol.lib().lo_proof_free(C.byref(pr))


def test_oracle_upstream_slicing_rows_are_what_buffer_view_cpp_defines():
    """row by row against a direct model of the reference's calls: views = (offset, size) pairs sliced as
    buffer_view.cpp:91-95 defines slice_bytes, write_buffer / clear_buffer / copy on a byte array"""
    l, k = L_, K_
    job = ol.make_job(l, k, N_, 192, 1, 0, generated_at=5, threads=1)
    slicing_program(True).attach(job)
    rows, _, _, _ = ol.form_rows(job)
    kinds = ol.row_kinds(job)
    # the model: a slab of 8 variables as python ints, upstream's slice_bytes
    slab = [0] * (8 * k)
    enc = ol.rng_fill(bytes(job.encoding_seed), 0, 192 * 6).reshape(-1, 8)     # pads of the 6 init rows, in program order
    to_int = lambda e: int.from_bytes(np.asarray(e, dtype=np.uint32).tobytes(), "little")
    n_init = [0]

    def slice_up(view, begin):                               # buffer_view::slice<u8>(begin) -> slice_bytes(begin, size - begin), as DEFINED
        off, size = view
        return (begin, off + (size - begin))

    def init(x_elem):                                        # on_batch_init: write 192 pads at x.slice(l * 32), then commit x
        off, _ = slice_up((x_elem * 32, k * 32), l * 32)
        for i in range(192):
            slab[off // 32 + i] = to_int(enc[192 * n_init[0] + i])
        n_init[0] += 1
        return slab[x_elem:x_elem + k]

    def write_clear(x_elem, vals):                           # write_buffer_clear: write, then clear_buffer(x.slice(len * 32))
        for i, v in enumerate(vals):
            slab[x_elem + i] = v
        off, size = slice_up((x_elem * 32, k * 32), len(vals) * 32)
        for e in range(off // 32, (off + size) // 32):
            slab[e] = 0

    def write(x_elem, vals):
        for i, v in enumerate(vals):
            slab[x_elem + i] = v

    want = []
    V = lambda i: i * k
    mul = lambda a, b: [(x * y) % P for x, y in zip(slab[a:a + k], slab[b:b + k])]
    write_clear(V(0), [3 + 2 * i for i in range(10)]); want.append(init(V(0)))
    write(V(1), [9] * l); want.append(init(V(1)))
    t = mul(V(0), V(1)); want += [slab[V(0):V(0) + k], slab[V(1):V(1) + k], t]; slab[V(2):V(2) + k] = t
    write(V(3), [5, 6, 7]); want.append(init(V(3)))
    slab[V(4):V(4) + k] = [(x + y) % P for x, y in zip(slab[V(2):V(2) + k], slab[V(3):V(3) + k])]
    slab[V(5):V(5) + k] = slab[V(4):V(4) + k]; want += [slab[V(5):V(5) + k], slab[V(4):V(4) + k]]
    t = mul(V(4), V(4)); want += [slab[V(4):V(4) + k], slab[V(4):V(4) + k], t]; slab[V(4):V(4) + k] = t
    write(V(2), [11, 12]); want.append(init(V(2)))
    t = mul(V(2), V(5)); want += [slab[V(2):V(2) + k], slab[V(5):V(5) + k], t]; slab[V(7):V(7) + k] = t
    write_clear(V(6), [77] * l); want.append(init(V(6)))
    t = mul(V(6), V(6)); want += [slab[V(6):V(6) + k], slab[V(6):V(6) + k], t]; slab[V(7):V(7) + k] = t
    t = mul(V(7), V(2)); want += [slab[V(7):V(7) + k], slab[V(2):V(2) + k], t]; slab[V(7):V(7) + k] = t
    nb = int((np.asarray(kinds) >= 4).sum())
    assert nb == len(want) == 5 + 3 + 2 + 3 + 3 + 3 + 3
    for r in range(nb):
        got = [to_int(e) for e in rows[r]]
        assert got == want[r], "batch row %d" % r
    assert all(v == 0 for v in want[-1][:l]) and any(want[0])          # variable 6's write_buffer_clear wiped the slab: zeros from there on


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
@pytest.mark.parametrize("upstream", [False, True])
def test_hip_slicing_semantics_equal_oracle(amd, upstream):
    """LIG_BOP_UPSTREAM_COMPAT / LIG_BOP_F_WRITE_LIMBS on the device interpreter (csrc/prover.hip, lig_run_batch_program): the
    envelope equals the oracle's under the declared and under the upstream slicing semantics"""
    prog = slicing_program(upstream)
    ojob, pr = oracle_prove(700, 330, prog, threads=4)
    c = amd.Context(L_, K_, N_)
    try:
        job = prog.attach(amd.Context.make_job(700, 330, generated_at=5))
        tr = c.synth_prepare_job(job)
        proof, info = c.synth_prove(tr)
        c.trace_destroy(tr)
        assert (info.valid_code, info.valid_linear, info.valid_quad) == (1, 1, 1)
        assert bytes(info.root) == bytes(pr.root) and proof == bytes(pr.proof[:pr.proof_len])
        assert c.synth_verify(job, bytes(info.const_sum), proof).accept == 1
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


def pin_program(name):
    return {"demo": lambda: demo_program(with_bits=True), "demo_no_bits": lambda: demo_program(with_bits=False),
            "slicing_declared": lambda: slicing_program(False), "slicing_upstream": lambda: slicing_program(True)}[name]()


def _pin_id(p):
    return "l%d_k%d_lin%d_quad%d_%s" % (p["l"], p["k"], p["n_linear"], p["n_quad"], p["batch"] or "nobatch")


@pytest.mark.parametrize("pin", PINS, ids=_pin_id)
def test_oracle_matches_whole_proof_pins(pin):
    """tests/golden/proof_pins.json (made by tests/golden/make_proof_pins.py from this oracle): drift detector"""
    j = ol.make_job(pin["l"], pin["k"], pin["n"], 192, pin["n_linear"], pin["n_quad"], generated_at=pin["generated_at"], threads=2)
    if pin["batch"]:
        pin_program(pin["batch"]).attach(j)
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
            pin_program(pin["batch"]).attach(job)
        tr = c.synth_prepare_job(job)
        proof, info = c.synth_prove(tr)
        c.trace_destroy(tr)
        assert hashlib.sha256(proof).hexdigest() == pin["proof_sha256"] and len(proof) == pin["proof_len"]
        assert bytes(info.root).hex() == pin["root"] and bytes(info.stage2_seed).hex() == pin["stage2_seed"]
        assert [info.valid_code, info.valid_linear, info.valid_quad] == pin["valid"]
    finally:
        c.close()
