"""GPU: the caller-rows prover entry (lig_rows_begin / lig_rows_commit / lig_rows_prove), public arguments, the
verifier's derived linear constant, and configs[2] at full size.

A row-batching driver of the reference's constraint generator hands the backend (a) the rows witness_manager forms
(include/zkp/backend/witness_manager.hpp:200-269), (b) after the stage-1 seed is known, the per-witness randomness rows
and the public constant of the linear test (include/zkp/nonbatch_context.hpp:654-780, src/webgpu_prover.cpp:307).
Here the oracle plays that driver: lo_form_rows / lo_rand_rows are fed through the rows entry and the envelope must be
byte-identical to the oracle's reference-structured prover.
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import hip_lib
import oracle_lib as ol
import test_batch_rows as tb

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def amd():
    return hip_lib.load()


def oracle_prove(job):
    pr = ol.Proof()
    assert ol.lib().lo_prove(C.byref(job), C.byref(pr)) == 0
    out = dict(proof=bytes(pr.proof[:pr.proof_len]), root=bytes(pr.root), seed1=bytes(pr.stage1_seed), seed2=bytes(pr.stage2_seed),
               const_sum=bytes(pr.const_sum), rows=pr.rows, valid=(pr.valid_code, pr.valid_linear, pr.valid_quad))
    ol.lib().lo_proof_free(C.byref(pr))
    return out


PUB = [(42).to_bytes(8, "little"), b"statement\0", bytes.fromhex("00ff10")]


@pytest.mark.parametrize("l,k,n,n_linear,n_quad,prog,pub", [
    (320, 512, 2048, 700, 0, None, None),
    (320, 512, 2048, 640, 330, None, PUB),
    (320, 512, 2048, 0, 0, None, None),                       # masks only
    (320, 512, 2048, 320 * 600 + 5, 320 + 9, None, None),     # > one 512-row chunk, a triple across the boundary region
    (320, 512, 2048, 100, 0, "demo", PUB),                    # batch rows (init / equal / product / bit rows) ahead of the stream
    (8000, 8192, 32768, 3 * 8000 + 123, 8000 + 5, None, None),
])
@pytest.mark.parametrize("mode", ["host_rows_library_pads", "device_rows_own_pads"])
def test_rows_entry_equals_oracle(amd, l, k, n, n_linear, n_quad, prog, pub, mode):
    job = ol.make_job(l, k, n, 192, n_linear, n_quad, generated_at=77, threads=8, public_args=pub)
    if prog:
        tb.demo_program().attach(job)
    want = oracle_prove(job)
    rows, _, _, _ = ol.form_rows(job)
    kinds = ol.row_kinds(job).copy()
    R = rows.shape[0]
    c = amd.Context(l, k, n)
    try:
        if mode == "host_rows_library_pads":
            # the driver leaves the padding slots of every row that draws padding upstream to the library
            msgs = rows.copy()
            draws = (kinds <= 3) | (kinds == amd.ROW_KINDS["INIT"])
            msgs[draws, l:] = 0xDEADBEEF                       # whatever is there must be overwritten
            kinds[draws] |= amd.ROW_DRAW_PAD
            tr, keep = c.rows_begin(kinds, msgs, on_device=False, generated_at=77, public_args=pub)
        else:
            d_msgs = c.upload(rows) if R else c.malloc(32)
            tr, keep = c.rows_begin(kinds, d_msgs, on_device=True, generated_at=77, public_args=pub)
        root, seed1 = c.rows_commit(tr)
        assert root == want["root"] and seed1 == want["seed1"]
        rands, const_sum = ol.rand_rows(job, seed1)            # the driver's second pass: randomness rows + public constant
        assert const_sum == want["const_sum"]
        if mode == "host_rows_library_pads":
            proof, info = c.rows_prove(tr, rands, const_sum, on_device=False)
        else:
            d_r = c.upload(rands) if R else c.malloc(32)
            proof, info = c.rows_prove(tr, d_r, const_sum, on_device=True)
        assert info.rows == want["rows"]
        assert (info.valid_code, info.valid_linear, info.valid_quad) == (1, 1, 1)
        assert bytes(info.stage2_seed) == want["seed2"]
        assert proof == want["proof"]
        # the next trace of the same shape reuses every buffer (lig_rows_restart); with const_sum = NULL the library reports
        # the constant that fits the rows, which for an honest driver is the public one
        if mode == "host_rows_library_pads":
            c.rows_restart(tr, msgs, on_device=False)
        else:
            c.rows_restart(tr, d_msgs, on_device=True)
        assert c.rows_commit(tr) == (root, seed1)
        proof2, info2 = c.rows_prove(tr, rands if mode == "host_rows_library_pads" else d_r, None, on_device=mode != "host_rows_library_pads")
        assert proof2 == proof and bytes(info2.const_sum) == want["const_sum"]
        c.trace_destroy(tr)
        # the verifier's side of the rows job: kinds + public data + the proof -> stage-1 seed -> the driver's randomness rows
        vt, vseed, vinfo = c.rows_verify_begin(ol.row_kinds(job), proof, public_args=pub)
        assert vt is not None and vseed == seed1 and vinfo.parsed == 1 and vinfo.indices_match == 1
        v = c.rows_verify_finish(vt, rands, const_sum)
        assert [v.valid_merkle, v.valid_code, v.valid_linear, v.valid_quad, v.code_equal, v.linear_equal, v.quad_equal, v.accept] == [1] * 8
        if R:
            vt, _, _ = c.rows_verify_begin(ol.row_kinds(job), proof, public_args=pub)
            wrong = ((int.from_bytes(const_sum, "little") + 1) % ol.P).to_bytes(32, "little")
            v = c.rows_verify_finish(vt, d_r if mode != "host_rows_library_pads" else rands, wrong, on_device=mode != "host_rows_library_pads")
            assert (v.accept, v.valid_linear, v.valid_merkle, v.code_equal, v.linear_equal) == (0, 0, 1, 1, 1)
            bad = bytearray(proof)
            bad[len(bad) - 40] ^= 1                              # an opened mask element
            vt, _, _ = c.rows_verify_begin(ol.row_kinds(job), bytes(bad), public_args=pub)
            assert c.rows_verify_finish(vt, rands, const_sum).valid_merkle == 0
            vt, _, vi = c.rows_verify_begin(ol.row_kinds(job), proof[:len(proof) // 2], public_args=pub)
            assert vt is None and vi.accept == 0
            other_rands = rands.copy()
            other_rands[0, 0, 0] ^= 1                            # another public constraint stream
            vt, _, _ = c.rows_verify_begin(ol.row_kinds(job), proof, public_args=pub)
            assert c.rows_verify_finish(vt, other_rands, const_sum).accept == 0
            vt, _, _ = c.rows_verify_begin(ol.row_kinds(job), proof, public_args=pub)
            c.vtrace_destroy(vt)                                 # a verifier that gives up between begin and finish
            c.vtrace_destroy(None)
        # both verifiers accept it, deriving the constant from the public statement themselves
        hjob = amd.Context.make_job(n_linear, n_quad, generated_at=77, public_args=pub)
        if prog:
            tb.demo_program().attach(hjob)
        assert c.synth_verify(hjob, None, proof).accept == 1
        buf = (C.c_uint8 * len(proof)).from_buffer_copy(proof)
        assert ol.lib().lo_verify(C.byref(job), None, buf, len(proof)) == 1
    finally:
        c.close()


def test_rows_entry_pipelined_restart_and_dense_rands(amd):
    """commit(i) -> restart(i+1) -> prove(i): the next trace is uploaded into the second message matrix while the current
    one is proved; the dense randomness rows of the synthetic stream are generated on the device (dense_rands_per_row).
    Two different traces alternate; every envelope equals the oracle's for its trace."""
    l, k, n = 320, 512, 2048
    nl = 320 * 700 + 11                                        # 701 rows: two stage-1 chunks
    jobs = [ol.make_job(l, k, n, 192, nl, 0, synth_seed=s, generated_at=3, threads=8) for s in (1, 2)]
    want = [oracle_prove(j) for j in jobs]
    rows = [ol.form_rows(j)[0].copy() for j in jobs]
    kinds = ol.row_kinds(jobs[0]) | amd.ROW_DRAW_PAD
    per_row = np.full(len(kinds), l, dtype=np.uint32)
    per_row[-1] = nl % l
    c = amd.Context(l, k, n)
    try:
        tr, keep = c.rows_begin(kinds, rows[0], generated_at=3, dense_rands_per_row=per_row)
        for i in range(5):
            root, seed1 = c.rows_commit(tr)
            assert root == want[i % 2]["root"]
            if i < 4:
                c.rows_restart(tr, rows[(i + 1) % 2])          # prefetch under the proof of trace i
            proof, info = c.rows_prove(tr, None, None)
            assert proof == want[i % 2]["proof"] and bytes(info.const_sum) == want[i % 2]["const_sum"]
        with pytest.raises(amd.LigError):                      # nothing loaded any more
            c.rows_commit(tr)
        c.trace_destroy(tr)
    finally:
        c.close()


def test_rows_entry_rejects_malformed_jobs(amd):
    l, k, n = 320, 512, 2048
    c = amd.Context(l, k, n)
    try:
        msgs = np.zeros((3, k, 8), dtype=np.uint32)
        for kinds in ([1, 2], [2, 3, 0], [0, 3], [6], [7, 0], [8, 9], [11], [5 | 0x80], [6 | 0x80, 7]):
            with pytest.raises(amd.LigError):
                c.rows_begin(np.array(kinds, dtype=np.uint8), msgs[:len(kinds)])
        tr, keep = c.rows_begin(np.array([0], dtype=np.uint8), msgs[:1])
        with pytest.raises(amd.LigError):                      # prove before commit
            c.rows_prove(tr, msgs[:1], bytes(32))
        c.rows_commit(tr)
        with pytest.raises(amd.LigError):                      # constant not reduced
            c.rows_prove(tr, msgs[:1], b"\xff" * 32)
        with pytest.raises(amd.LigError):                      # commit twice
            c.rows_commit(tr)
        with pytest.raises(amd.LigError):                      # no randomness rows and no dense counts
            c.rows_prove(tr, None, bytes(32))
        # a wrong public constant is not an error: the prover's self-check reports it
        proof, info = c.rows_prove(tr, msgs[:1], (5).to_bytes(32, "little"))
        assert info.valid_linear == 0 and info.valid_code == 1
        c.trace_destroy(tr)
    finally:
        c.close()
    with pytest.raises(amd.LigError):                          # k - l < 192: fewer random pads than opened columns
        c2 = amd.Context(400, 512, 2048)
        try:
            c2.synth_prepare(100)
        finally:
            c2.close()


def test_public_args_and_derived_linear_constant(amd):
    """instance_hash with public arguments enters both seeds; the verifier derives the constant of the linear test from the
    public statement (witness_key stream) -- a proof for another statement fails exactly the linear predicate"""
    l, k, n, nl, nq = 320, 512, 2048, 2000, 400
    c = amd.Context(l, k, n)
    try:
        job = amd.Context.make_job(nl, nq, generated_at=9, public_args=PUB)
        tr = c.synth_prepare_job(job)
        proof, info = c.synth_prove(tr)
        c.trace_destroy(tr)
        want = oracle_prove(ol.make_job(l, k, n, 192, nl, nq, generated_at=9, threads=8, public_args=PUB))
        assert proof == want["proof"]
        plain = oracle_prove(ol.make_job(l, k, n, 192, nl, nq, generated_at=9, threads=8))
        assert plain["root"] == want["root"] and plain["seed1"] != want["seed1"]        # args enter after the commitment
        v = c.synth_verify(job, None, proof)
        assert v.accept == 1
        assert c.synth_verify(job, bytes(info.const_sum), proof).accept == 1          # caller-supplied constant still works
        # same rows, other public arguments: the code / linear / quadratic coefficients differ -> not accepted
        assert c.synth_verify(amd.Context.make_job(nl, nq, generated_at=9), None, proof).accept == 0
        # another public statement (witness_key): everything but the linear predicate still holds
        other = amd.Context.make_job(nl, nq, synth_seed=2, generated_at=9, public_args=PUB)
        vo = c.synth_verify(other, None, proof)
        assert (vo.accept, vo.valid_linear, vo.valid_merkle, vo.valid_code, vo.valid_quad, vo.code_equal, vo.linear_equal, vo.quad_equal) == \
            (0, 0, 1, 1, 1, 1, 1, 1)
        ojob = ol.make_job(l, k, n, 192, nl, nq, synth_seed=2, generated_at=9, threads=8, public_args=PUB)
        buf = (C.c_uint8 * len(proof)).from_buffer_copy(proof)
        assert ol.lib().lo_verify(C.byref(ojob), None, buf, len(proof)) == 0
    finally:
        c.close()


def test_full_size_600_rows_against_oracle_on_host_cores(amd):
    """k = 8192 beyond one 512-row launch chunk: multi-chunk encode, the randomness double-buffer ring, the hash gate
    (nb > 4) and the 96-row schedule tails, all compared byte for byte with the oracle run on this box's host cores"""
    l, k, n = 8000, 8192, 32768
    nl, nq = 600 * 8000 + 77, 8 * 8000 + 5            # 601 linear rows + 27 quadratic rows = 628 rows + 3 masks
    c = amd.Context(l, k, n)
    try:
        tr = c.synth_prepare(nl, nq, generated_at=5)
        proof, info = c.synth_prove(tr)
        c.trace_destroy(tr)
        job = amd.Context.make_job(nl, nq, generated_at=5)
        assert c.synth_verify(job, None, proof).accept == 1
    finally:
        c.close()
    want = oracle_prove(ol.make_job(l, k, n, 192, nl, nq, generated_at=5, threads=os.cpu_count() or 8))
    assert info.rows == want["rows"] == 631
    assert hashlib.sha256(proof).hexdigest() == hashlib.sha256(want["proof"]).hexdigest()
    assert proof == want["proof"]


@pytest.mark.parametrize("lg", [24, 26, "24_q50"])
def test_full_size_proof_equals_oracle_pin(amd, lg):
    """the exact bench.py jobs -- configs[2]: 2^24 linear constraints, and the configs[3] trace of 2^26 constraints on one
    GPU (k = 8192, synthetic seed 1, encoding seed 0..31, generated_at 0): the envelope's SHA-256, root, seeds, constant and
    sample indices equal tests/golden/full_pin_2p<lg>.json, which the oracle's reference-structured prover produced
    (tests/golden/make_full_pin.py: 68 s / 277 s on the build container's cores)"""
    # "24_q50": the quadratic mix of SURVEY.md 8(d) -- 2^23 linear + 2^23 quadratic constraints (bench.py's `quad_mix` leg)
    with open(os.path.join(GOLD, "full_pin_2p%s.json" % lg)) as f:
        pin = json.load(f)
    c = amd.Context(pin["l"], pin["k"], pin["n"])
    try:
        tr = c.synth_prepare(pin["n_linear"], pin["n_quad"], synth_seed=pin["synth_seed"], generated_at=pin["generated_at"])
        proof, info = c.synth_prove(tr)
        c.trace_destroy(tr)
        assert info.rows == pin["rows"] and len(proof) == pin["proof_len"]
        assert bytes(info.root).hex() == pin["root"]
        assert bytes(info.stage1_seed).hex() == pin["stage1_seed"]
        assert bytes(info.stage2_seed).hex() == pin["stage2_seed"]
        assert bytes(info.const_sum).hex() == pin["const_sum"]
        assert hashlib.sha256(proof).hexdigest() == pin["proof_sha256"]
        assert list(amd.sample_columns(bytes(info.stage2_seed), pin["n"])) == pin["sample_idx"]
        job = amd.Context.make_job(pin["n_linear"], pin["n_quad"], synth_seed=pin["synth_seed"], generated_at=pin["generated_at"])
        assert c.synth_verify(job, None, proof).accept == 1
    finally:
        c.close()


@pytest.mark.parametrize("bits,n_linear,n_quad,where", [
    (32, 320 * 5 + 17, 330, "host"),
    (64, 320 * 600 + 5, 320 + 9, "host"),            # > one 512-row chunk: chunk byte ranges of mixed-width rows
    (32, 320 * 3, 0, "device"),
    (64, 8000 * 3 + 11, 8000, "host8192"),
])
def test_rows_entry_narrow_row_format_equals_oracle(amd, bits, n_linear, n_quad, where):
    """lig_rows_job.elem_bytes: witness rows shipped as 4- / 8-byte integers (only the l data slots; the library draws the pads)
    and expanded on the device give the envelope of the oracle's prover on the same small-witness trace (lo_job.witness_bits);
    rows that do not fit (the z rows of x*y = z, every 7th linear row here) stay 32 bytes wide in the same matrix"""
    l, k, n = (8000, 8192, 32768) if where == "host8192" else (320, 512, 2048)
    job = ol.make_job(l, k, n, 192, n_linear, n_quad, generated_at=31, threads=8)
    job.witness_bits = bits
    want = oracle_prove(job)
    rows, _, _, _ = ol.form_rows(job)
    kinds = ol.row_kinds(job).copy()
    R = rows.shape[0]
    assert not rows[kinds <= 2][:, :l, bits // 32:].any(), "the oracle's small-witness rows are small"
    widths = np.full(R, 32, dtype=np.uint8)
    for r in range(R):
        if kinds[r] <= 2 and not (kinds[r] == 0 and r % 7 == 3):
            widths[r] = bits // 8
    kk = kinds.copy()
    kk[kinds <= 3] |= amd.ROW_DRAW_PAD
    wide = rows.copy()
    wide[kinds <= 3, l:] = 0x5A5A5A5A                              # pads come from the library in every case
    packed = amd.pack_rows(wide, widths, l)
    assert len(packed) < 0.6 * wide.nbytes
    c = amd.Context(l, k, n)
    try:
        if where == "device":
            d_packed = c.upload(packed)
            tr, keep = c.rows_begin(kk, d_packed, on_device=True, generated_at=31, elem_bytes=widths)
        else:
            tr, keep = c.rows_begin(kk, packed, generated_at=31, elem_bytes=widths)
        root, seed1 = c.rows_commit(tr)
        assert root == want["root"] and seed1 == want["seed1"]
        rands, const_sum = ol.rand_rows(job, seed1)
        assert const_sum == want["const_sum"]
        c.rows_restart(tr, d_packed if where == "device" else packed.ctypes.data, on_device=where == "device")     # the next trace arrives under the proof
        proof, info = c.rows_prove(tr, rands, const_sum)
        assert (info.valid_code, info.valid_linear, info.valid_quad) == (1, 1, 1)
        assert proof == want["proof"]
        assert c.rows_commit(tr) == (root, seed1)
        proof2, _ = c.rows_prove(tr, rands, None)
        assert proof2 == proof
        # the same job fed from the other side (device rows after host rows and vice versa)
        if where == "device":
            c.rows_restart(tr, packed.ctypes.data, on_device=False)
        else:
            d_alt = c.upload(packed)
            c.rows_restart(tr, d_alt, on_device=True)
        assert c.rows_commit(tr) == (root, seed1)
        proof3, _ = c.rows_prove(tr, rands, const_sum)
        assert proof3 == proof
        c.trace_destroy(tr)
        # a narrow row without LIG_ROW_DRAW_PAD, a width that does not exist
        bad = kk.copy(); bad[0] &= 0x7f
        with pytest.raises(amd.LigError):
            c.rows_begin(bad, packed, generated_at=31, elem_bytes=widths)
        w2 = widths.copy(); w2[0] = 16
        with pytest.raises(amd.LigError):
            c.rows_begin(kk, packed, generated_at=31, elem_bytes=w2)
    finally:
        c.close()


def test_rows_push_rands_in_pieces_equals_rows_prove_with_all_rows(amd):
    """lig_rows_push_rands (the stage-2 callbacks of a constraint generator deliver their rows one at a time, nonbatch_context.hpp:
    654-712): randomness rows handed over in three pieces from page-locked memory (lig_host_alloc) while "the guest runs", then
    lig_rows_prove(rands = NULL) -- the envelope of lig_rows_prove with the whole matrix, which is the oracle's; gaps, disorder and a
    proof before the last piece are refused"""
    l, k, n = 320, 512, 2048
    job = ol.make_job(l, k, n, 192, 320 * 700 + 5, 330, generated_at=12, threads=8)          # 701 + 6 rows: two stage-2 chunks
    want = oracle_prove(job)
    rows, _, _, _ = ol.form_rows(job)
    kinds = ol.row_kinds(job)
    R = rows.shape[0]
    c = amd.Context(l, k, n)
    try:
        tr, keep = c.rows_begin(kinds, rows, generated_at=12)
        root, seed1 = c.rows_commit(tr)
        rands, const_sum = ol.rand_rows(job, seed1)
        pinned, ptr = c.host_alloc(R * k * 32)
        pinned[:] = np.ascontiguousarray(rands, dtype=np.uint32).view(np.uint8).reshape(-1)
        base = ptr.value
        with pytest.raises(amd.LigError):
            c.rows_push_rands(tr, 5, 10, base + 5 * k * 32)                      # not from row 0
        cuts = [0, 200, 513, R]
        for a, b in zip(cuts[:-2], cuts[1:-1]):
            c.rows_push_rands(tr, a, b - a, base + a * k * 32)
        with pytest.raises(amd.LigError):
            c.rows_push_rands(tr, cuts[-2] + 1, 1, base)                           # a gap
        with pytest.raises(amd.LigError):
            c.rows_prove(tr, None, const_sum)                                      # not every row delivered yet
        c.rows_push_rands(tr, cuts[-2], R - cuts[-2], base + cuts[-2] * k * 32)   # (the refused prove dropped nothing: the last piece completes the matrix)
        proof, info = c.rows_prove(tr, None, const_sum)
        assert (info.valid_code, info.valid_linear, info.valid_quad) == (1, 1, 1)
        assert proof == want["proof"]
        # the same trace again with the whole matrix in one call: same bytes
        c.rows_restart(tr, np.ascontiguousarray(rows, dtype=np.uint32).ctypes.data)
        assert c.rows_commit(tr) == (root, seed1)
        proof2, _ = c.rows_prove(tr, rands, const_sum)
        assert proof2 == proof
        c.trace_destroy(tr)
        c.host_free(ptr)
    finally:
        c.close()


def test_rows_push_rands_sparse_ships_only_rows_that_have_randomness(amd):
    """lig_rows_push_rands_sparse: a trace that starts with the rows of a batch program (their randomness rows are zero: the vbn254fr
    hooks carry none, nonbatch_context.hpp:782-850) -- only the rows that HAVE a randomness row are in the pinned staging, packed, the
    library zero-fills the others on the device; the envelope is the oracle's"""
    import test_batch_rows
    l, k, n = 320, 512, 2048
    job = ol.make_job(l, k, n, 192, 320 * 300 + 9, 330, generated_at=5, threads=8)
    test_batch_rows.demo_program(with_bits=False).attach(job)
    want = oracle_prove(job)
    rows, _, _, _ = ol.form_rows(job)
    kinds = ol.row_kinds(job)
    R = rows.shape[0]
    c = amd.Context(l, k, n)
    try:
        tr, keep = c.rows_begin(kinds, rows, generated_at=5)
        root, seed1 = c.rows_commit(tr)
        assert root == want["root"]
        rands, const_sum = ol.rand_rows(job, seed1)
        rands = np.ascontiguousarray(rands, dtype=np.uint32).reshape(R, -1)
        present = rands.any(axis=1).astype(np.uint8)
        assert 0 < present.sum() < R and not present[kinds >= 4].any()           # the batch rows are the absent ones
        packed = rands[present != 0]
        pinned, ptr = c.host_alloc(packed.nbytes)
        pinned[:] = packed.view(np.uint8).reshape(-1)
        cut = 23                                                                  # first push ends inside the batch rows, the second holds the rest
        off = int(present[:cut].sum()) * k * 32
        c.rows_push_rands_sparse(tr, 0, present[:cut], ptr.value)
        c.rows_push_rands_sparse(tr, cut, present[cut:], ptr.value + off)
        proof, info = c.rows_prove(tr, None, const_sum)
        assert (info.valid_code, info.valid_linear, info.valid_quad) == (1, 1, 1)
        assert proof == want["proof"]
        c.trace_destroy(tr)
        c.host_free(ptr)
    finally:
        c.close()


@pytest.mark.parametrize("fault,how", [("1", "host"), ("2", "host"), ("2", "push")])
def test_a_transfer_that_never_completes_is_retried_not_an_error(tmp_path, fault, how):
    """round 6: the uploader thread bounds its wait for a host-to-device transfer (LIG_UPLOAD_TIMEOUT_S).  With one
    transfer made to "never complete" (LIG_FAULT_UPLOAD: 1 = witness rows, 2 = randomness rows -- handed to lig_rows_prove as host rows or
    pushed row by row) the streams that wait for the arrival word are released, the stage that ran over the missing rows is discarded,
    the rows are brought again by stream-ordered copies and the call returns the ORACLE's proof, not an error; lig_upload_health counts
    the retry and reports nothing pending; the next proof of the process (uploads on stream-ordered copies from then on) is right too"""
    import subprocess
    import sys
    import textwrap
    script = tmp_path / "stuck_upload.py"
    script.write_text(textwrap.dedent('''
        import ctypes as C, json, os, sys, time
        import numpy as np
        root, how = sys.argv[1], sys.argv[2]
        sys.path.insert(0, os.path.join(root, "tests"))
        import hip_lib, oracle_lib as ol
        amd = hip_lib.load()
        l, k, n = 320, 512, 2048
        job = ol.make_job(l, k, n, 192, 320 * 300 + 7, 330, generated_at=9, threads=4)
        pr = ol.Proof()
        assert ol.lib().lo_prove(C.byref(job), C.byref(pr)) == 0
        want = bytes(pr.proof[:pr.proof_len])
        rows, _, _, _ = ol.form_rows(job)
        kinds = ol.row_kinds(job).copy()
        c = amd.Context(l, k, n)
        out = {"proofs": []}
        for it in range(2):
            t0 = time.time()
            tr, keep = c.rows_begin(kinds, rows, generated_at=9)
            root_, seed1 = c.rows_commit(tr)
            rands, cs = ol.rand_rows(job, seed1)
            rands = np.ascontiguousarray(rands, dtype=np.uint32).reshape(len(kinds), -1)
            if how == "push":
                pinned, ptr = c.host_alloc(rands.nbytes)
                pinned[:] = rands.view(np.uint8).reshape(-1)
                cut = 100
                c.rows_push_rands(tr, 0, cut, ptr.value)
                c.rows_push_rands(tr, cut, len(kinds) - cut, ptr.value + cut * k * 32)
                proof, info = c.rows_prove(tr, None, cs)
                c.host_free(ptr)
            else:
                proof, info = c.rows_prove(tr, rands, cs)
            out["proofs"].append(proof == want)
            out.setdefault("seconds", []).append(time.time() - t0)
            c.trace_destroy(tr)
        out["health"] = list(c.upload_health())
        c.close()
        print(json.dumps(out))
    '''))
    p = subprocess.run([sys.executable, str(script), os.path.dirname(os.path.dirname(GOLD)), how], env=dict(os.environ, LIG_FAULT_UPLOAD=fault, LIG_UPLOAD_TIMEOUT_S="2"),
                       capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    out = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert out["proofs"] == [True, True], out
    assert out["health"] == [1, 0], out                       # one call needed the second attempt; no abandoned transfer is still pending
    assert 1.5 < out["seconds"][0] < 20 and out["seconds"][1] < 5, out
