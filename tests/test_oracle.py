"""CPU-only tests: pin the C oracle (oracle/) against independent definitions and the golden fixtures.

The reference ships no known-answer vectors for this path except tests/webgpu/test_powmod.cpp
(restated in test_powmod_kats).  Everything else is pinned against tests/pydef.py (Python big
ints, hashlib) and tests/golden/*.json (OpenSSL CLI keystreams, definition-computed codewords).
"""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
import pydef

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
P = pydef.P


def gold(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def h256(vals):
    return hashlib.sha256(b"".join(int(v).to_bytes(32, "little") for v in vals)).hexdigest()


def test_constants_and_roots():
    # shader/bn254fr.wgsl.in:19-45 and src/bn254.cpp:36-49
    assert pydef.ROOT2 == pydef.ROOT2_DEC
    assert pow(pydef.ROOT1, 1 << 28, P) == 1 and pow(pydef.ROOT1, 1 << 27, P) != 1
    w = np.zeros((3, 8), dtype=np.uint32)
    for k in (512, 8192):
        ol.lib().lo_omegas(k, ol.ptr(w[0:1]), ol.ptr(w[1:2]), ol.ptr(w[2:3]))
        assert tuple(ol.from_limbs(w)) == pydef.omegas(k)


def test_field_mul_montmul():
    rng = np.random.default_rng(1)
    a = ol.rand_field(rng, 256)
    b = ol.rand_field(rng, 256)
    # edge values
    edge = ol.to_limbs([0, 1, P - 1, P - 2, (P - 1) // 2, 2 ** 253])
    a[:6] = edge
    b[:6] = edge[::-1]
    out = np.zeros_like(a)
    mo = np.zeros_like(a)
    for i in range(a.shape[0]):
        ol.lib().lo_fr_mul(ol.ptr(out[i:i + 1]), ol.ptr(a[i:i + 1]), ol.ptr(b[i:i + 1]))
        ol.lib().lo_fr_montmul(ol.ptr(mo[i:i + 1]), ol.ptr(a[i:i + 1]), ol.ptr(b[i:i + 1]))
    A, B = ol.from_limbs(a), ol.from_limbs(b)
    Rinv = pow(1 << 256, -1, P)
    assert ol.from_limbs(out) == [x * y % P for x, y in zip(A, B)]
    assert ol.from_limbs(mo) == [x * y * Rinv % P for x, y in zip(A, B)]


def test_eltwise_ops():
    rng = np.random.default_rng(2)
    x, y, o = ol.rand_field(rng, 64), ol.rand_field(rng, 64), ol.rand_field(rng, 64)
    X, Y, O = ol.from_limbs(x), ol.from_limbs(y), ol.from_limbs(o)
    c = 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % P
    Rinv = pow(1 << 256, -1, P)
    cases = {
        0: [(a + b) % P for a, b in zip(X, Y)], 1: [(a - b) % P for a, b in zip(X, Y)],
        2: [(a + b) % P for a, b in zip(O, X)], 3: [(a + c) % P for a in X], 4: [(a - c) % P for a in X],
        5: [(c - a) % P for a in X], 6: [a * b % P for a, b in zip(X, Y)], 7: [a * c % P for a in X],
        8: [a * c * Rinv % P for a in X], 9: [(q + a * b) % P for q, a, b in zip(O, X, Y)],
        10: [(q + a * c) % P for q, a in zip(O, X)], 11: [a * pow(b, -1, P) % P for a, b in zip(X, Y)],
    }
    for op, want in cases.items():
        out = o.copy()
        ol.eltwise(op, x, y, out, scalar=c)
        assert ol.from_limbs(out) == want, op
    out = o.copy()
    ol.eltwise(12, x, None, out, scalar=0, bit=77)
    assert ol.from_limbs(out) == [(a >> 77) & 1 for a in X]


def test_powmod_kats():
    """tests/webgpu/test_powmod.cpp:58-197 -- the reference's only device known-answer tests (N = 8192)."""
    N = 8192
    lib = ol.lib()
    one = ol.to_limbs([1] * N)
    zero = ol.to_limbs([0] * N)
    exp = np.arange(N, dtype=np.uint32)
    out = np.zeros((N, 8), dtype=np.uint32)
    kat = gold("powmod_kat.json")
    # zero coefficient -> 0
    lib.lo_powmod(ol.ptr(ol.to_limbs([7])), ol.ptr(exp), ol.ptr(zero), ol.ptr(out), N, 0)
    assert not out.any()
    # base 1 -> 1
    lib.lo_powmod(ol.ptr(ol.to_limbs([1])), ol.ptr(exp), ol.ptr(one), ol.ptr(out), N, 0)
    assert ol.from_limbs(out) == [1] * N
    # 7^i
    lib.lo_powmod(ol.ptr(ol.to_limbs([7])), ol.ptr(exp), ol.ptr(one), ol.ptr(out), N, 0)
    got = ol.from_limbs(out)
    assert got == [pow(7, i, P) for i in range(N)]
    assert [hex(v) for v in got[:64]] == kat["generator"]
    # (p-1) * (p-1)^(2^16 + i)
    exp2 = (np.arange(N, dtype=np.uint64) + (1 << 16)).astype(np.uint32)
    lib.lo_powmod(ol.ptr(ol.to_limbs([P - 1])), ol.ptr(exp2), ol.ptr(ol.to_limbs([P - 1] * N)), ol.ptr(out), N, 0)
    got = ol.from_limbs(out)
    assert got == [pow(P - 1, (1 << 16) + i + 1, P) for i in range(N)]
    assert [hex(v) for v in got[:64]] == kat["minus"]
    # powmod + 9 x powmod_add = 10 * 7^i
    lib.lo_powmod(ol.ptr(ol.to_limbs([7])), ol.ptr(exp), ol.ptr(one), ol.ptr(out), N, 0)
    for _ in range(9):
        lib.lo_powmod(ol.ptr(ol.to_limbs([7])), ol.ptr(exp), ol.ptr(one), ol.ptr(out), N, 1)
    got = ol.from_limbs(out)
    assert got == [10 * pow(7, i, P) % P for i in range(N)]
    assert [hex(v) for v in got[:64]] == kat["powmod_add"]


def test_aes_fips197():
    key = bytes(range(32))
    pt = bytes.fromhex("00112233445566778899aabbccddeeff")
    rk = np.zeros(60, dtype=np.uint32)
    out = np.zeros(16, dtype=np.uint8)
    ol.lib().lo_aes256_expand(ol.ptr(np.frombuffer(key, dtype=np.uint8).copy()), ol.ptr(rk))
    ol.lib().lo_aes256_encrypt_block(ol.ptr(rk), ol.ptr(np.frombuffer(pt, dtype=np.uint8).copy()), ol.ptr(out))
    assert out.tobytes().hex() == "8ea2b7ca516745bfeafc49904b496089"      # FIPS-197 C.3


def test_aes_ctr_sampler_golden():
    for v in gold("aes_ctr.json")["vectors"]:
        key = bytes.fromhex(v["key"])
        nblk = (16384 + 64) // 16
        ks = ol.keystream(key, 0, nblk)
        assert ks[:64].hex() == v["keystream_first64"]
        assert hashlib.sha256(ks).hexdigest() == v["keystream_sha256"]
        assert [hex(e) for e in ol.from_limbs(ol.rng_fill(key, 0, 8))] == v["field_first8"]
        assert [hex(e) for e in ol.from_limbs(ol.rng_fill(key, 511, 3))] == v["field_511_512_513"]


def test_encode_decode_golden_k512():
    g = gold("encode_k512.json")
    k, n = g["k"], g["n"]
    key = bytes.fromhex(g["key"])
    rows = ol.rng_fill(key, 0, 3 * k).reshape(3, k, 8)
    assert [hex(v) for v in ol.from_limbs(rows[0][:4])] == g["row0_first4"]
    ctx = ol.Ctx(320, k, n)
    cws = [ctx.encode(rows[r]) for r in range(3)]
    assert [hex(v) for v in ol.from_limbs(cws[0][:4])] == g["cw0_first4"]
    assert [hex(v) for v in ol.from_limbs(cws[0][-2:])] == g["cw0_last2"]
    assert [h256(ol.from_limbs(c)) for c in cws] == g["cw_sha256"]
    msg2k = np.concatenate([rows[0], rows[1]])
    assert h256(ol.from_limbs(ctx.encode_2k(msg2k))) == g["cw2k_sha256"]
    assert h256(ol.from_limbs(ctx.decode(cws[0]))) == g["decode0_sha256"]
    # batched entry agrees
    assert np.array_equal(ctx.encode_rows(rows, threads=2), np.stack(cws))
    # leaves + merkle
    leaves = ol.colsha(np.stack(cws))
    assert leaves[0].tobytes().hex() == g["leaf0"] and leaves[-1].tobytes().hex() == g["leaf_last"]
    assert hashlib.sha256(leaves.tobytes()).hexdigest() == g["leaves_sha256"]
    nodes = ol.merkle_build(leaves)
    assert nodes[0].tobytes().hex() == g["root"]


def test_encode_by_definition_fresh_random():
    """same check on inputs that are not in the fixture (definition evaluated live, k=512)"""
    k, n = 512, 2048
    rng = np.random.default_rng(7)
    msg = ol.rand_field(rng, k)
    msg[0] = ol.to_limbs([P - 1])[0]
    msg[1] = 0
    ctx = ol.Ctx(320, k, n)
    cw = ctx.encode(msg)
    assert ol.from_limbs(cw) == pydef.encode(ol.from_limbs(msg), k, n)
    # decode(encode(m)) returns m on [0,k) and zeros above (degree < k)
    dec = ol.from_limbs(ctx.decode(cw))
    assert dec[:k] == ol.from_limbs(msg) and not any(dec[k:])
    # forward/inverse transforms are inverse permutations of each other for every size
    for which, size in ((0, k), (1, 2 * k), (2, n)):
        x = ol.rand_field(rng, size)
        assert np.array_equal(ctx.ntt(which, True, ctx.ntt(which, False, x)), x)
    # w_4k is derived from root2 = root1^(2^61 - 1), so w_4k^4 = w_k^-1: coset 0 of the codeword is the message reversed
    # (the HIP encoder copies it instead of computing it)
    assert np.array_equal(cw[0::4], msg[(k - np.arange(k)) % k])
    # linearity of the code
    m2 = ol.rand_field(rng, k)
    s = ol.to_limbs([(a + b) % P for a, b in zip(ol.from_limbs(msg), ol.from_limbs(m2))])
    assert ol.from_limbs(ctx.encode(s)) == [(a + b) % P for a, b in zip(ol.from_limbs(cw), ol.from_limbs(ctx.encode(m2)))]


def test_column_hash_and_merkle_vs_hashlib():
    rng = np.random.default_rng(3)
    for rows in (1, 2, 5):
        ncols = 37
        data = np.stack([ol.rand_field(rng, ncols) for _ in range(rows)])
        leaves = ol.colsha(data)
        for j in range(ncols):
            assert leaves[j].tobytes() == pydef.leaf(ol.from_limbs(data[:, j]))
    leaves = rng.integers(0, 256, size=(37, 32), dtype=np.uint8)      # non power of two -> zero padded
    nodes = ol.merkle_build(leaves)
    assert [n.tobytes() for n in nodes] == pydef.merkle_nodes([l.tobytes() for l in leaves])


def test_merkle_decommit_recommit():
    rng = np.random.default_rng(4)
    nleaves = 256
    leaves = rng.integers(0, 256, size=(nleaves, 32), dtype=np.uint8)
    nodes = ol.merkle_build(leaves)
    for idx in ([0], [255], [3, 4, 5, 200], sorted(rng.choice(nleaves, 40, replace=False).tolist()), list(range(nleaves))):
        sib = ol.merkle_decommit(nodes, nleaves, idx)
        pos = pydef.sibling_positions(idx, 2 * nleaves - 1)
        assert [s.tobytes() for s in sib] == [nodes[p].tobytes() for p in pos]
        root = np.zeros(32, dtype=np.uint8)
        ia = np.array(idx, dtype=np.uint32)
        ok = ol.lib().lo_merkle_recommit(nleaves, ol.ptr(ia), len(idx), ol.ptr(np.ascontiguousarray(leaves[idx])),
                                         ol.ptr(sib), len(sib), ol.ptr(root))
        assert ok == 1 and root.tobytes() == nodes[0].tobytes()
        if len(sib):
            bad = sib.copy(); bad[0, 0] ^= 1
            ok = ol.lib().lo_merkle_recommit(nleaves, ol.ptr(ia), len(idx), ol.ptr(np.ascontiguousarray(leaves[idx])),
                                             ol.ptr(bad), len(bad), ol.ptr(root))
            assert not (ok == 1 and root.tobytes() == nodes[0].tobytes())


def test_sampling_golden():
    for c in gold("sampling.json")["cases"]:
        got = ol.sample_indices(bytes.fromhex(c["seed"]), c["n"], c["t"])
        assert got.tolist() == c["indices"]
        assert len(set(got.tolist())) == c["t"] and got.max() < c["n"]


def test_seeds():
    root = bytes(range(32))
    ih = np.zeros(32, dtype=np.uint8)
    ol.lib().lo_instance_hash_default(ol.ptr(ih))
    assert ih.tobytes() == hashlib.sha256(bytes(32) + b"Ligero\0").digest()
    s1 = np.zeros(32, dtype=np.uint8)
    r = np.frombuffer(root, dtype=np.uint8).copy()
    ol.lib().lo_stage1_seed(ol.ptr(r), ol.ptr(ih), ol.ptr(s1))
    assert s1.tobytes() == hashlib.sha256(b"LigetronStage1\0" + root + ih.tobytes()).digest()
    rng = np.random.default_rng(5)
    a, b, c = (ol.rand_field(rng, 16) for _ in range(3))
    s2 = np.zeros(32, dtype=np.uint8)
    ol.lib().lo_stage2_seed(ol.ptr(r), ol.ptr(a), ol.ptr(b), ol.ptr(c), 16, ol.ptr(s2))
    assert s2.tobytes() == hashlib.sha256(b"LigetronStage2\0" + root + a.tobytes() + b.tobytes() + c.tobytes()).digest()


@pytest.mark.parametrize("n_linear,n_quad", [(700, 0), (640, 330), (100, 700), (0, 0), (0, 321)])
def test_reference_structured_prover_and_verifier(n_linear, n_quad):
    """config-1 counterpart: a tiny row stream through the whole 3-stage flow; the restated verifier accepts."""
    import ctypes as C
    l, k, n, t = 320, 512, 2048, 192
    job = ol.make_job(l, k, n, t, n_linear, n_quad, generated_at=1234567)
    pr = ol.Proof()
    assert ol.lib().lo_prove(C.byref(job), C.byref(pr)) == 0
    try:
        assert (pr.valid_code, pr.valid_linear, pr.valid_quad) == (1, 1, 1)
        rows = pr.rows
        assert rows == ol.lib().lo_job_rows(C.byref(job))
        proof = bytes(pr.proof[:pr.proof_len])
        # independent wire-format walk
        top = pydef.pb_fields(proof)
        assert [f for f, _, _ in top] == [1, 2]
        meta = dict((f, v) for f, _, v in pydef.pb_fields(top[0][2]))
        assert meta[1] == b"1.5.0" and meta[2] == 1 and meta[3] == 1 and meta[6] == k and meta[7] == n and meta[8] == t and meta[9] == 128
        assert pydef.pb_fields(meta[5]) == [(1, 0, 1234567)]
        body = pydef.pb_fields(top[1][2])
        assert [f for f, _, _ in body] == [1, 2, 3, 4, 5]
        md = pydef.pb_fields(body[0][2])
        assert md[0] == (1, 0, 1) and pydef.pb_fields(md[1][2])[0][2] == bytes(pr.root)
        idx = [pr.sample_idx[i] for i in range(t)]
        packed = [v for f, _, v in md if f == 4][0]
        dec, i = [], 0
        while i < len(packed):
            v, i = pydef._varint(packed, i); dec.append(v)
        assert dec == idx == sorted(idx)
        nsib = sum(1 for f, _, _ in md if f == 3)
        assert nsib == len(pydef.sibling_positions(idx, 2 * n - 1))
        for fnum in (2, 3, 4):
            assert len(pydef.pb_fields(body[fnum - 1][2])[0][2]) == 32 * n
        assert len(pydef.pb_fields(body[4][2])[0][2]) == 32 * rows * t
        cs = (C.c_uint64 * 4)(*pr.const_sum)
        assert ol.lib().lo_verify(C.byref(job), cs, pr.proof, pr.proof_len) == 1
        # tamper: one sampled element, one accumulator limb, the constant sum
        buf = bytearray(proof)
        buf[-5] ^= 1
        tb = (C.c_uint8 * len(buf)).from_buffer(buf)
        assert ol.lib().lo_verify(C.byref(job), cs, tb, len(buf)) == 0
        cs2 = (C.c_uint64 * 4)(*pr.const_sum); cs2[0] ^= 1
        assert ol.lib().lo_verify(C.byref(job), cs2, pr.proof, pr.proof_len) == 0
    finally:
        ol.lib().lo_proof_free(C.byref(pr))
