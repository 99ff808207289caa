"""GPU parity tests (-m gpu): every C-ABI entry point of liblig_hip.so against the CPU oracle on the same
seeded inputs, against the committed golden fixtures, and -- at the full k=8192/n=32768 size -- through
size-independent properties (decode(encode(m)) = m, linearity, chunk-invariance of the column hash).
Bit-exact everywhere: the path is integer arithmetic mod p and SHA-256."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import hip_lib
import oracle_lib as ol
import pydef

pytestmark = pytest.mark.gpu
P = pydef.P
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def amd():
    return hip_lib.load()


@pytest.fixture(scope="module")
def ctx512(amd):
    c = amd.Context(320, 512, 2048)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx8192(amd):
    c = amd.Context(8000, 8192, 32768)
    yield c
    c.close()


def edge_mix(rng, count):
    x = ol.rand_field(rng, count)
    edge = ol.to_limbs([0, 1, 2, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, 1 << 253, (1 << 253) + 12345, 0xFFFFFFFF, 1 << 32])
    x[:len(edge)] = edge
    return x


def test_eltwise_all_ops(ctx512):
    c = ctx512
    rng = np.random.default_rng(11)
    N = 4099                                   # ragged: not a multiple of the block size
    x, y, o = edge_mix(rng, N), edge_mix(rng, N)[::-1].copy(), edge_mix(rng, N)
    y[5] = 0                                   # division by zero -> 0
    dx, dy = c.upload(x), c.upload(y)
    scalar = int.from_bytes(hashlib.sha256(b"c").digest(), "little") % P
    for op in range(13):
        do = c.upload(o)
        want = o.copy()
        ol.eltwise(op, x, y, want, scalar=scalar, bit=200)
        c.eltwise(op, dx, dy, do, N, scalar=scalar, bit=200)
        got = c.download(do, (N, 8))
        assert np.array_equal(got, want), "op %d" % op
        c.free(do)
    # aliasing: out == x
    want = x.copy()
    ol.eltwise(6, x, y, want)
    c.eltwise("MUL", dx, dy, dx, N)
    assert np.array_equal(c.download(dx, (N, 8)), want)
    # empty input is a no-op; non-canonical scalar is rejected
    c.eltwise("ADD", dx, dy, dx, 0)
    with pytest.raises(Exception):
        c.eltwise("MUL_CONST", dx, dy, dx, N, scalar=P)


def test_division_batched_inversion(ctx512):
    """EltwiseDivMod shares one inversion among M denominators (Montgomery's trick): exact quotients vs the oracle's
    per-element inverse for a row with zeros in every position class, out aliasing y, and -- for the large-batch (M = 16)
    variant -- the defining property out * y == x (y != 0), out == 0 (y == 0) on 2^20 + 37 elements"""
    c = ctx512
    rng = np.random.default_rng(12)
    N = 8192 + 5
    x, y = ol.rand_field(rng, N), edge_mix(rng, N)
    y[[0, 1, 2, 3, 1024, 1025, 4099, N - 1]] = 0          # neighbours in one thread's group, group boundaries, the tail
    dx, dy, do = c.upload(x), c.upload(y), c.malloc(32 * N)
    want = np.zeros_like(x)
    ol.eltwise(11, x, y, want)
    c.eltwise("DIV", dx, dy, do, N)
    assert np.array_equal(c.download(do, (N, 8)), want)
    c.eltwise("DIV", dx, dy, dy, N)                         # out == y
    assert np.array_equal(c.download(dy, (N, 8)), want)
    big = (1 << 20) + 37
    xb, yb = ol.rand_field(rng, big), ol.rand_field(rng, big)
    zpos = rng.integers(0, big, 500)
    yb[zpos] = 0
    dxb, dyb, dob, dchk = c.upload(xb), c.upload(yb), c.malloc(32 * big), c.malloc(32 * big)
    c.eltwise("DIV", dxb, dyb, dob, big)
    c.eltwise("MUL", dob, dyb, dchk, big)
    q, chk = c.download(dob, (big, 8)), c.download(dchk, (big, 8))
    nz = np.ones(big, dtype=bool)
    nz[zpos] = False
    assert np.array_equal(chk[nz], xb[nz])
    assert not q[~nz].any()
    for p in (dx, dy, do, dxb, dyb, dob, dchk):
        c.free(p)


def test_powmod_reference_kats(ctx512):
    """tests/webgpu/test_powmod.cpp:58-197 on the HIP backend (N = 8192)"""
    c = ctx512
    N = 8192
    exp = np.arange(N, dtype=np.uint32)
    one, zero = ol.to_limbs([1] * N), ol.to_limbs([0] * N)
    dexp, done, dzero, dout = c.upload(exp), c.upload(one), c.upload(zero), c.malloc(32 * N)
    c.powmod(7, dexp, dzero, dout, N)
    assert not c.download(dout, (N, 8)).any()
    c.powmod(1, dexp, done, dout, N)
    assert ol.from_limbs(c.download(dout, (N, 8))) == [1] * N
    c.powmod(7, dexp, done, dout, N)
    assert ol.from_limbs(c.download(dout, (N, 8))) == [pow(7, i, P) for i in range(N)]
    dexp2 = c.upload((np.arange(N, dtype=np.uint64) + (1 << 16)).astype(np.uint32))
    dcoef = c.upload(ol.to_limbs([P - 1] * N))
    c.powmod(P - 1, dexp2, dcoef, dout, N)
    assert ol.from_limbs(c.download(dout, (N, 8))) == [pow(P - 1, (1 << 16) + i + 1, P) for i in range(N)]
    c.powmod(7, dexp, done, dout, N)
    for _ in range(9):
        c.powmod(7, dexp, done, dout, N, add=True)
    assert ol.from_limbs(c.download(dout, (N, 8))) == [10 * pow(7, i, P) % P for i in range(N)]


def test_encode_golden_k512(ctx512):
    g = gold("encode_k512.json")
    c = ctx512
    k, n = 512, 2048
    rows = ol.rng_fill(bytes.fromhex(g["key"]), 0, 3 * k).reshape(3, k, 8)
    h256 = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    # single-buffer API, in place
    buf = c.malloc(32 * n)
    for r in range(3):
        c.L.lig_write_clear(c.h, buf, 32 * n, rows[r].ctypes.data_as(C.c_void_p), 32 * k)
        c.encode(buf)
        assert h256(c.download(buf, (n, 8))) == g["cw_sha256"][r]
    # batched API
    dm, dc = c.upload(rows), c.malloc(3 * 32 * n)
    c.encode_rows(dm, dc, 3)
    cws = c.download(dc, (3, n, 8))
    assert [h256(cws[r]) for r in range(3)] == g["cw_sha256"]
    # mask-row encode and decode
    m2 = np.zeros((n, 8), dtype=np.uint32); m2[:2 * k] = np.concatenate([rows[0], rows[1]])
    c.write(buf, m2); c.encode_2k(buf)
    assert h256(c.download(buf, (n, 8))) == g["cw2k_sha256"]
    c.write(buf, cws[0]); c.decode(buf)
    assert h256(c.download(buf, (n, 8))) == g["decode0_sha256"]
    # column hash + merkle
    st, dl = c.sha_state(n), c.malloc(32 * n)
    c.sha_update_rows(st, dc, 3)
    c.sha_final(st, dl)
    leaves = c.download(dl, (n, 32), dtype=np.uint8)
    assert leaves[0].tobytes().hex() == g["leaf0"] and hashlib.sha256(leaves.tobytes()).hexdigest() == g["leaves_sha256"]
    nodes = c.merkle_build(dl, n)
    assert c.download(nodes, (32,), dtype=np.uint8).tobytes().hex() == g["root"]


@pytest.mark.parametrize("k", [512, 1024, 2048, 4096, 8192, 32768])
def test_transforms_vs_oracle(amd, k):
    """fast path (k = 512 ... 8192; tile lengths 64 ... 1024, odd log2 lengths take one extra radix-2 stage) and the generic
    radix-2 path (k = 32768: four times the production packing) against the oracle"""
    n, l = 4 * k, k - 192
    c = amd.Context(l, k, n)
    o = ol.Ctx(l, k, n)
    rng = np.random.default_rng(k)
    try:
        rows = 5
        msgs = np.stack([edge_mix(rng, k) for _ in range(rows)])
        msgs[1] = 0                                      # all-zero row
        msgs[2] = ol.to_limbs([P - 1] * k)               # all-max row
        dm, dc = c.upload(msgs), c.malloc(rows * 32 * n)
        c.encode_rows(dm, dc, rows)
        got = c.download(dc, (rows, n, 8))
        want = o.encode_rows(msgs, threads=4)
        assert np.array_equal(got, want)
        assert not got[1].any()
        buf = c.malloc(32 * n)
        # every transform size/direction of the executor surface
        for which, size in ((0, k), (1, 2 * k), (2, n)):
            for inverse in (False, True):
                x = np.zeros((n, 8), dtype=np.uint32); x[:size] = edge_mix(rng, size)
                c.write(buf, x); c.ntt(buf, which, inverse)
                res = c.download(buf, (n, 8))
                assert np.array_equal(res[:size], o.ntt(which, inverse, x[:size])) and np.array_equal(res[size:], x[size:])
        m2 = np.zeros((n, 8), dtype=np.uint32); m2[:2 * k] = edge_mix(rng, 2 * k)
        c.write(buf, m2); c.encode_2k(buf)
        assert np.array_equal(c.download(buf, (n, 8)), o.encode_2k(m2[:2 * k]))
        c.write(buf, want[0]); c.decode(buf)
        dec = c.download(buf, (n, 8))
        assert np.array_equal(dec, o.decode(want[0]))
        assert np.array_equal(dec[:k], msgs[0]) and not dec[k:].any()
        # decode of a non-codeword (accumulator-like): raw coefficients stay in [k,n)
        x = edge_mix(rng, n)
        c.write(buf, x); c.decode(buf)
        assert np.array_equal(c.download(buf, (n, 8)), o.decode(x))
    finally:
        c.close()


def test_full_size_properties(ctx8192):
    """k=8192: round trip, linearity and batch-size invariance on 300 rows (sizes the oracle does not sweep)"""
    c = ctx8192
    k, n, rows = 8192, 32768, 300
    key = hashlib.sha256(b"props").digest()
    dm, dc = c.malloc(rows * 32 * k), c.malloc(rows * 32 * n)
    c.rng_fill(key, 0, dm, rows * k)
    c.encode_rows(dm, dc, rows)
    # (a) rows 0..2 against the oracle
    msgs = c.download(dm, (3, k, 8))
    assert np.array_equal(msgs, ol.rng_fill(key, 0, 3 * k).reshape(3, k, 8))
    o = ol.Ctx(8000, k, n)
    assert np.array_equal(c.download(dc, (3, n, 8)), o.encode_rows(msgs, threads=3))
    # (b) row 299 encoded alone (different launch geometry) equals its batched encoding
    buf = c.malloc(32 * n)
    c.L.lig_clear(c.h, buf, 32 * n)
    c.L.lig_copy(c.h, buf, C.c_void_p(dm.value + 299 * 32 * k), 32 * k)
    c.encode(buf)
    last = c.download(dc, (n, 8), offset=299 * 32 * n)
    assert np.array_equal(c.download(buf, (n, 8)), last)
    # (c) decode(encode(m)) = m || 0
    c.decode(buf)
    dec = c.download(buf, (n, 8))
    assert np.array_equal(dec[:k], c.download(dm, (k, 8), offset=299 * 32 * k)) and not dec[k:].any()
    # (d) linearity: Enc(m0 + m1) = Enc(m0) + Enc(m1)
    ds, dsum = c.malloc(32 * n), c.malloc(32 * n)
    c.L.lig_clear(c.h, ds, 32 * n)
    c.eltwise("ADD", dm, C.c_void_p(dm.value + 32 * k), ds, k)
    c.encode(ds)
    c.eltwise("ADD", dc, C.c_void_p(dc.value + 32 * n), dsum, n)
    assert np.array_equal(c.download(ds, (n, 8)), c.download(dsum, (n, 8)))
    # (e) column hash is invariant to how the rows are chunked (odd/even splits exercise the pending half block)
    leaves = []
    for split in ([300], [1, 299], [7, 2, 1, 290], [150, 150]):
        st, dl = c.sha_state(n), c.malloc(32 * n)
        off = 0
        for cnt in split:
            c.sha_update_rows(st, C.c_void_p(dc.value + off * 32 * n), cnt)
            off += cnt
        c.sha_final(st, dl)
        c.sha_final(st, dl)                      # final does not consume the state
        leaves.append(c.download(dl, (n, 32), dtype=np.uint8))
        c.free(st); c.free(dl)
    assert all(np.array_equal(leaves[0], x) for x in leaves[1:])
    # spot-check 64 leaves against hashlib over the downloaded columns
    cols = list(range(0, n, n // 64))
    cw_host = c.download(dc, (rows, n, 8))
    for j in cols:
        assert leaves[0][j].tobytes() == pydef.leaf(ol.from_limbs(cw_host[:, j]))
    # Merkle root vs the oracle tree over the same leaves
    dl = c.upload(leaves[0])
    nodes = c.merkle_build(dl, n)
    got_nodes = c.download(nodes, (2 * n - 1, 32), dtype=np.uint8)
    assert np.array_equal(got_nodes, ol.merkle_build(leaves[0]))


def test_sha_small_and_ragged(ctx512):
    c = ctx512
    rng = np.random.default_rng(5)
    for ninst, rows in ((1, 1), (37, 2), (192, 5), (2048, 1)):
        data = np.stack([ol.rand_field(rng, ninst) for _ in range(rows)])
        dd, st, dl = c.upload(data), c.sha_state(ninst), c.malloc(32 * ninst)
        for r in range(rows):                    # one row at a time: the reference's sha256_digest_update
            c.check(c.L.lig_sha_update(c.h, st, C.c_void_p(dd.value + r * 32 * ninst)))
        c.sha_final(st, dl)
        assert np.array_equal(c.download(dl, (ninst, 32), dtype=np.uint8), ol.colsha(data))
        # zero rows absorbed: SHA-256 of the empty string, word-swapped
        st0 = c.sha_state(ninst)
        c.sha_final(st0, dl)
        e = hashlib.sha256(b"").digest()
        assert c.download(dl, (ninst, 32), dtype=np.uint8)[0].tobytes() == b"".join(e[4 * i:4 * i + 4][::-1] for i in range(8))
    # non power-of-two leaf count is zero padded
    leaves = rng.integers(0, 256, size=(37, 32), dtype=np.uint8)
    nodes = c.merkle_build(c.upload(leaves), 37)
    assert np.array_equal(c.download(nodes, (127, 32), dtype=np.uint8), ol.merkle_build(leaves))
    # unknown state is an error, not a crash
    with pytest.raises(Exception):
        c.sha_update_rows(c.malloc(64), c.malloc(64), 1)


def test_rng_fill_vs_oracle_and_openssl(ctx512):
    c = ctx512
    for v in gold("aes_ctr.json")["vectors"]:
        key = bytes.fromhex(v["key"])
        out = c.malloc(32 * 600)
        c.rng_fill(key, 0, out, 600)
        got = c.download(out, (600, 8))
        assert [hex(e) for e in ol.from_limbs(got[:8])] == v["field_first8"]
        assert [hex(e) for e in ol.from_limbs(got[511:514])] == v["field_511_512_513"]
        assert np.array_equal(got, ol.rng_fill(key, 0, 600))
        c.rng_fill(key, (1 << 33) + 5, out, 100)            # block counter beyond 32 bits
        assert np.array_equal(c.download(out, (100, 8)), ol.rng_fill(key, (1 << 33) + 5, 100))


def test_rlc_and_gather_vs_oracle(ctx512):
    c = ctx512
    n, rows, t = 2048, 9, 192
    rng = np.random.default_rng(9)
    U = np.stack([edge_mix(rng, n) for _ in range(rows)])
    Rn = np.stack([edge_mix(rng, n) for _ in range(rows)])
    acc = [edge_mix(rng, n) for _ in range(3)]
    rc = [int(v) for v in ol.from_limbs(ol.rand_field(rng, rows))]
    triples = [(0, 1, 2), (4, 5, 6)]
    rq = [int(v) for v in ol.from_limbs(ol.rand_field(rng, 2))]
    code, lin, quad = (a.copy() for a in acc)
    for r in range(rows):                     # the reference's per-row sequence (nonbatch_context.hpp:756-780)
        ol.eltwise(10, U[r], None, code, scalar=rc[r])
        ol.eltwise(9, U[r], Rn[r], lin)
    for (x, y, z), q in zip(triples, rq):
        t1, t2 = np.zeros((n, 8), np.uint32), np.zeros((n, 8), np.uint32)
        ol.eltwise(6, U[x], U[y], t1); ol.eltwise(1, t1, U[z], t2); ol.eltwise(10, t2, None, quad, scalar=q)
    dU, dR = c.upload(U), c.upload(Rn)
    dc, dl, dq = (c.upload(a) for a in acc)
    c.rlc_rows(dU, dR, rows, rc, dc, dl, triples, rq, dq)
    assert np.array_equal(c.download(dc, (n, 8)), code)
    assert np.array_equal(c.download(dl, (n, 8)), lin)
    assert np.array_equal(c.download(dq, (n, 8)), quad)
    idx = np.sort(rng.choice(n, t, replace=False)).astype(np.uint32)
    c.sample_init(idx)
    out = c.malloc(rows * t * 32)
    c.gather_rows(dU, rows, out)
    assert np.array_equal(c.download(out, (rows, t, 8)), U[:, idx])
    c.check(c.L.lig_sample_gather(c.h, C.c_void_p(dU.value + 3 * 32 * n), out, 2))   # reference-style: one row into slot 2
    assert np.array_equal(c.download(out, (t, 8), offset=2 * t * 32), U[3, idx])
    with pytest.raises(Exception):
        c.sample_init(np.array([n], dtype=np.uint32))


@pytest.mark.parametrize("l,k,n,n_linear,n_quad", [
    (320, 512, 2048, 700, 0), (320, 512, 2048, 640, 330), (320, 512, 2048, 100, 700), (320, 512, 2048, 1, 0),
    (320, 512, 2048, 0, 0),              # empty statement: only the three mask rows are committed
    (320, 512, 2048, 0, 321),            # quadratic constraints only (one full + one partial triple)
    (832, 1024, 4096, 2000, 900),        # tile length 128 = 2 * 4^3: fast path with the extra radix-2 stage
    (16192, 16384, 65536, 20000, 16200), # twice the production packing: tile length 2048 (72 KiB of LDS per workgroup)
    (32576, 32768, 131072, 40000, 0),    # four times: tile length 4096 (144 KiB, one workgroup per CU)
    (8000, 8192, 32768, 3 * 8000 + 123, 8000 + 5),
])
def test_batched_prover_equals_reference_structured_oracle(amd, l, k, n, n_linear, n_quad):
    """the whole three-stage flow: proof envelope bytes, root, seeds and the linear constant of the HIP prover
    (rows encoded once, codewords resident) equal the oracle's reference-structured prover (every row re-encoded
    per stage, one executor call per row); the restated verifier accepts the HIP proof."""
    c = amd.Context(l, k, n)
    try:
        tr = c.synth_prepare(n_linear, n_quad, generated_at=1234567)
        proof, info = c.synth_prove(tr)
        proof2, info2 = c.synth_prove(tr)            # proving twice from the same resident witness is deterministic
        c.trace_destroy(tr)
    finally:
        c.close()
    assert proof == proof2 and bytes(info.root) == bytes(info2.root)
    job = ol.make_job(l, k, n, 192, n_linear, n_quad, generated_at=1234567, threads=8)
    pr = ol.Proof()
    assert ol.lib().lo_prove(C.byref(job), C.byref(pr)) == 0
    try:
        assert info.rows == pr.rows
        assert bytes(info.root) == bytes(pr.root)
        assert bytes(info.stage1_seed) == bytes(pr.stage1_seed)
        assert bytes(info.stage2_seed) == bytes(pr.stage2_seed)
        assert bytes(info.const_sum) == bytes(pr.const_sum)
        assert (info.valid_code, info.valid_linear, info.valid_quad) == (1, 1, 1)
        want = bytes(pr.proof[:pr.proof_len])
        assert len(proof) == len(want)
        assert proof == want
        cs = (C.c_uint64 * 4)(*pr.const_sum)
        buf = (C.c_uint8 * len(proof)).from_buffer_copy(proof)
        assert ol.lib().lo_verify(C.byref(job), cs, buf, len(proof)) == 1
    finally:
        ol.lib().lo_proof_free(C.byref(pr))


@pytest.mark.parametrize("l,k,n,n_linear,n_quad", [(320, 512, 2048, 640, 330), (832, 1024, 4096, 1500, 0)])
def test_generic_row_path_end_to_end(amd, monkeypatch, l, k, n, n_linear, n_quad):
    """contexts without the tiled encoder (k > 32768) take the generic radix-2 row path: planar codewords through strided
    copies, coset-2 values stored and accumulated by the separate pass.  Forced here at small k (LIG_ENCODE_GENERIC=1)."""
    monkeypatch.setenv("LIG_ENCODE_GENERIC", "1")
    c = amd.Context(l, k, n)
    try:
        tr = c.synth_prepare(n_linear, n_quad, generated_at=77)
        proof, info = c.synth_prove(tr)
        c.trace_destroy(tr)
    finally:
        c.close()
    job = ol.make_job(l, k, n, 192, n_linear, n_quad, generated_at=77, threads=4)
    pr = ol.Proof()
    assert ol.lib().lo_prove(C.byref(job), C.byref(pr)) == 0
    try:
        assert proof == bytes(pr.proof[:pr.proof_len])
    finally:
        ol.lib().lo_proof_free(C.byref(pr))


@pytest.mark.parametrize("l,k,n,n_linear,n_quad", [
    (320, 512, 2048, 640, 330), (832, 1024, 4096, 2000, 900), (8000, 8192, 32768, 2 * 8000 + 123, 8000 + 5),
])
def test_hip_verifier_agrees_with_restated_verifier(amd, l, k, n, n_linear, n_quad):
    """lig_synth_verify (verifier context on the 192 opened columns, recommit, decode, seven predicates) accepts the
    prover's envelope and rejects exactly what the oracle's restated verifier rejects: a flipped opened element, a
    flipped accumulator element, a flipped sibling hash, a wrong public constant, a truncated envelope."""
    c = amd.Context(l, k, n)
    job_o = ol.make_job(l, k, n, 192, n_linear, n_quad, generated_at=99, threads=8)

    def oracle_accepts(p, const_sum):
        cs = (C.c_uint64 * 4).from_buffer_copy(bytes(const_sum))
        buf = (C.c_uint8 * len(p)).from_buffer_copy(p)
        return ol.lib().lo_verify(C.byref(job_o), cs, buf, len(p))

    try:
        job = amd.Context.make_job(n_linear, n_quad, generated_at=99)
        tr = c.synth_prepare(n_linear, n_quad, generated_at=99)
        proof, info = c.synth_prove(tr)
        c.trace_destroy(tr)
        cs = bytes(info.const_sum)
        v = c.synth_verify(job, cs, proof)
        assert [v.parsed, v.indices_match, v.valid_merkle, v.valid_code, v.valid_linear, v.valid_quad, v.code_equal,
                v.linear_equal, v.quad_equal, v.accept] == [1] * 10
        assert oracle_accepts(proof, cs) == 1
        rows = info.rows
        smp_off = len(proof) - (rows + 3) * 192 * 32          # the opened columns are the last field of the envelope
        acc_off = proof.index(bytes(info.root)) + 32
        cases = {
            "opened element": (smp_off + 32 * 5 + 1, "valid_merkle"),
            "opened mask element": (len(proof) - 32 * 100 + 2, "valid_merkle"),
        }
        for name, (pos, flag) in cases.items():
            bad = bytearray(proof)
            bad[pos] ^= 1
            vb = c.synth_verify(job, cs, bytes(bad))
            assert vb.accept == 0 and getattr(vb, flag) == 0, name
            assert oracle_accepts(bytes(bad), cs) == 0, name
        # a flipped byte inside the encoded code-test polynomial changes the stage-2 seed -> other columns are sampled
        bad = bytearray(proof)
        pos = smp_off - 2 * n * 32 - 1000
        bad[pos] ^= 1
        vb = c.synth_verify(job, cs, bytes(bad))
        assert vb.accept == 0
        assert oracle_accepts(bytes(bad), cs) == 0
        # wrong public constant: only the linear predicate fails
        wrong = (int.from_bytes(cs, "little") + 1) % ol.P
        vb = c.synth_verify(job, wrong.to_bytes(32, "little"), proof)
        assert (vb.accept, vb.valid_linear, vb.valid_merkle, vb.valid_code, vb.valid_quad, vb.code_equal, vb.linear_equal,
                vb.quad_equal) == (0, 0, 1, 1, 1, 1, 1, 1)
        assert oracle_accepts(proof, wrong.to_bytes(32, "little")) == 0
        # truncated / empty envelopes are rejected, not errors
        for cut in (0, 10, len(proof) // 2, len(proof) - 1):
            vb = c.synth_verify(job, cs, proof[:cut])
            assert vb.accept == 0
        # a different statement (one more linear constraint) does not verify
        job2 = amd.Context.make_job(n_linear + l, n_quad, generated_at=99)
        assert c.synth_verify(job2, cs, proof).accept == 0
        # the verifier's device workspace can be given back and comes back on demand; derive / given-constant calls share it
        c.verify_release()
        assert c.synth_verify(job, None, proof).accept == 1
        assert c.synth_verify(job, cs, proof).accept == 1
        c.verify_release()
    finally:
        c.close()


def test_concurrent_contexts_prove_independently(amd):
    """two contexts (= two sets of streams) proving from two host threads at the same time, as bench.py does with
    --inflight 2, give the envelopes they give alone: no state is shared between contexts"""
    import threading
    jobs = [(320, 512, 2048, 5000, 700, 11), (320, 512, 2048, 3000, 0, 12)]

    def prove(job, out, idx, reps):
        l, k, n, nl, nq, ts = job
        c = amd.Context(l, k, n)
        try:
            tr = c.synth_prepare(nl, nq, generated_at=ts)
            proofs = [c.synth_prove(tr)[0] for _ in range(reps)]
            c.trace_destroy(tr)
            out[idx] = proofs
        finally:
            c.close()

    alone = [None, None]
    for i, j in enumerate(jobs):
        prove(j, alone, i, 1)
    both = [None, None]
    th = [threading.Thread(target=prove, args=(j, both, i, 6)) for i, j in enumerate(jobs)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(2):
        assert both[i] is not None and all(p == alone[i][0] for p in both[i])


@pytest.mark.parametrize("n_linear,n_quad", [
    (320 * 509, 320),            # 509 linear rows + one triple = 512 rows: exactly one chunk
    (320 * 510, 320 + 7),        # the full triple occupies rows 510-512: it straddles the 512-row chunk boundary
    (320 * 511 + 3, 0),          # 511 full rows + partial = 512
    (320 * 512 + 1, 5),          # 513 linear rows (last partial) + a partial triple
    (320 * 1020, 320 * 2 + 1),   # 1020 + 6 + 3 rows: two chunk boundaries, triples at 1020-1028
])
def test_batched_prover_chunk_boundaries(amd, n_linear, n_quad):
    """row counts around the 512-row launch chunks (and the 64 / 16-row accumulation groups) with quadratic triples
    straddling a chunk boundary: envelope equal to the oracle's"""
    l, k, n = 320, 512, 2048
    c = amd.Context(l, k, n)
    try:
        tr = c.synth_prepare(n_linear, n_quad, generated_at=3)
        proof, info = c.synth_prove(tr)
        c.trace_destroy(tr)
    finally:
        c.close()
    job = ol.make_job(l, k, n, 192, n_linear, n_quad, generated_at=3, threads=8)
    pr = ol.Proof()
    assert ol.lib().lo_prove(C.byref(job), C.byref(pr)) == 0
    try:
        assert info.rows == pr.rows
        assert (info.valid_code, info.valid_linear, info.valid_quad) == (1, 1, 1)
        assert proof == bytes(pr.proof[:pr.proof_len])
    finally:
        ol.lib().lo_proof_free(C.byref(pr))


def test_hip_encode_and_leaves_equal_the_wgsl_kernels_executing(ctx512):
    """ingest hook (round 6): when tests/golden/wgsl_encode_k512.json exists (dumped from the reference's webgpu_context,
    tools/make_wgsl_pin.py) the HIP encoder and column hash must reproduce the WGSL kernels' codewords and leaves bit for bit"""
    import test_ref_pins as trp
    g, msgs, cws, leaves = trp._wgsl_fixture()
    c = ctx512
    n, k = g["n"], g["k"]
    dm, dc = c.upload(msgs), c.malloc(32 * n * g["rows"])
    c.encode_rows(dm, dc, g["rows"])
    got = c.download(dc, (g["rows"], n, 8))
    for r in range(g["rows"]):
        assert np.array_equal(got[r], cws[r])
    st, dl = c.sha_state(n), c.malloc(32 * n)
    c.sha_update_rows(st, dc, g["rows"])
    c.sha_final(st, dl)
    assert np.array_equal(c.download(dl, (n, 32), dtype=np.uint8), leaves)
