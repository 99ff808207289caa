"""CPU-only: the GMP side of the oracle pinned to the REFERENCE's own code.

tests/golden/ref_field.json and tests/golden/ref_rows_*.npz were produced by oracle/_ref/libref_backend.so = the reference's
src/bn254.cpp, include/zkp/finite_field_gmp.hpp, include/util/{csprng,mpz_vector}.hpp and include/zkp/backend/{witness_manager,core}.hpp
compiled in the build container (oracle/ref_backend.cpp, tests/golden/make_ref_backend.py).  The oracle's restatement must reproduce them:
the AES-CTR field sampler incl. the 16 KiB refill boundaries and the draws that take the subtraction, generate_omegas, the field
operations, the limb export, and -- on the row streams the reference's witness_manager emitted for tests/i32_add.wat and a multiply-add
guest -- the pads, the masks and the constant sum; the oracle's prover over those rows must still give the recorded envelope.
Where libref_backend.so is present the same comparisons also run LIVE (fresh keys, regenerated fixtures).  The HIP path is checked
against the same files in tests/test_gpu_ref_backend.py.
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
import ref_backend_lib as rb

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
P = ol.P
ROW_SETS = ["i32_add_320", "i32_add_8000", "mul_add_320"]
live = pytest.mark.skipif(rb.load() is None, reason="oracle/_ref/libref_backend.so (reference-built) is not present here")


def field_gold():
    with open(os.path.join(GOLD, "ref_field.json")) as f:
        return json.load(f)


def load_rows(name):
    z = np.load(os.path.join(GOLD, "ref_rows_%s.npz" % name))
    meta = json.loads(str(z["meta"]))
    return dict(kinds=z["kinds"], vals=z["vals"], rands=z["rands"], constsum=z["constsum"].tobytes(), meta=meta)


def fr(v):
    return ol.to_limbs([v])


def oracle_op(name, a, b=0):
    L = ol.lib()
    x, y, out = fr(a), fr(b), np.zeros((1, 8), dtype=np.uint32)
    p = ol.ptr
    if name == "mulmod":
        L.lo_fr_mul(p(out), p(x), p(y))
    elif name == "mont_mulmod":
        L.lo_fr_montmul(p(out), p(x), p(y))
    elif name in ("addmod", "submod"):
        f = L.lo_fr_add if name == "addmod" else L.lo_fr_sub
        f.argtypes = [C.c_void_p] * 3
        f(p(out), p(x), p(y))
    elif name == "negate":
        L.lo_fr_neg.argtypes = [C.c_void_p] * 2
        L.lo_fr_neg(p(out), p(x))
    elif name == "invmod":
        L.lo_fr_inv.argtypes = [C.c_void_p] * 2
        L.lo_fr_inv(p(out), p(x))
    elif name == "powmod":
        L.lo_fr_pow.argtypes = [C.c_void_p] * 3
        L.lo_fr_pow(p(out), p(x), p(y))
    elif name == "powmod_ui":
        L.lo_fr_pow_u64.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.lo_fr_pow_u64(p(out), p(x), b)
    elif name == "divmod":
        ol.eltwise(11, x, y, out)            # LO_OP_DIV: the executor's EltwiseDivMod
    else:
        raise KeyError(name)
    return ol.from_limbs(out)[0]


def test_sampler_equals_reference_generate_random():
    """bn254_gmp::generate_random over mpz_random_engine (finite_field_gmp.hpp:66-78, csprng.hpp:28-110): 1100 elements per key --
    the refills at elements 512 and 1024, and every draw that took `- p` after `>> 2`"""
    for s in field_gold()["sampler"]:
        key, n = bytes.fromhex(s["key"]), s["count"]
        mine = ol.rng_fill(key, 0, n)
        assert hashlib.sha256(mine.tobytes()).hexdigest() == s["sha256_of_all"]
        for i, hx in s["elements"].items():
            assert mine[int(i)].tobytes().hex() == hx, "element %s" % i
        # the draws whose raw value >> 2 was >= p: the oracle's keystream says the same, and there are enough of them to matter
        ks = ol.keystream(key, 0, 2 * n)
        took = [i for i in range(n) if (int.from_bytes(ks[32 * i:32 * i + 32], "little") >> 2) >= P]
        assert took[:40] == s["took_subtraction"] and len(took) == s["n_took_subtraction"] and len(took) > 100
        # positions are independent of how the stream is cut (counter mode): the oracle may start anywhere
        assert np.array_equal(ol.rng_fill(key, 510, 6), mine[510:516])


def test_omegas_equal_reference_generate_omegas():
    """src/bn254.cpp:51-64: w_k, w_2k from root1, w_4k from root2"""
    for k, want in field_gold()["omegas"].items():
        w = [np.zeros(8, dtype=np.uint32) for _ in range(3)]
        ol.lib().lo_omegas(int(k), *[ol.ptr(x) for x in w])
        assert [x.tobytes().hex() for x in w] == want, "k = %s" % k


def test_constants_equal_the_reference():
    c = field_gold()["constants"]
    val = {k: int.from_bytes(bytes.fromhex(v), "little") for k, v in c.items()}
    assert val["modulus"] == P and val["modulus_2x"] == 2 * P and val["modulus_4x"] == 4 * P
    assert val["root1"] == pow(7, (P - 1) >> 28, P) and val["root2"] == pow(val["root1"], (1 << 61) - 1, P)
    assert val["barrett_factor"] == (1 << 508) // P
    assert np.array_equal(np.ctypeslib.as_array((C.c_uint64 * 4).in_dll(ol.lib(), "LO_P")), np.frombuffer(P.to_bytes(32, "little"), dtype=np.uint64))
    assert np.array_equal(np.ctypeslib.as_array((C.c_uint64 * 4).in_dll(ol.lib(), "LO_MU")), np.frombuffer(val["barrett_factor"].to_bytes(32, "little"), dtype=np.uint64))
    # the host constant montgomery_factor (src/bn254.cpp:46) is NOT p^-1 mod 2^256 -- the shader's J is (bn254fr.wgsl.in:30-34);
    # recorded so that nobody "fixes" the oracle towards it
    J = pow(P, -1, 1 << 256)
    assert val["montgomery_factor"] != J
    assert np.array_equal(np.ctypeslib.as_array((C.c_uint64 * 4).in_dll(ol.lib(), "LO_J")), np.frombuffer(J.to_bytes(32, "little"), dtype=np.uint64))


@pytest.mark.parametrize("name", ["mulmod", "addmod", "submod", "negate", "invmod", "divmod", "powmod", "powmod_ui"])
def test_field_operations_equal_the_reference(name):
    """bn254_gmp::mulmod (Barrett, src/bn254.cpp:110-121), invmod, powmod, divmod, addmod, submod, negate against oracle/field.c"""
    for row in field_gold()["ops"][name]:
        *args, want = [int(x, 16) for x in row]
        assert oracle_op(name, *args) == want, (name, row)
        if name == "mulmod":
            assert want == args[0] * args[1] % P


def test_reference_host_mont_mulmod_is_recorded_not_followed():
    """bn254_gmp::mont_mulmod uses the defective host constant (see test_constants...): its outputs are in the fixture, and they are NOT
    a * b * R^-1 -- the device path (shader montgomery_mul, which the oracle restates and the powmod KATs pin) is"""
    rows = [[int(x, 16) for x in r] for r in field_gold()["ops"]["mont_mulmod"]]
    Rinv = pow(1 << 256, -1, P)
    proper = [a * b * Rinv % P for a, b, _ in rows]
    assert [oracle_op("mont_mulmod", a, b) for a, b, _ in rows] == proper
    assert any(w != pr for (_, _, w), pr in zip(rows, proper))


def test_limb_layout_equals_mpz_vector_export_import():
    """mpz_vector::export_limbs / import_limbs (mpz_vector.hpp:108-160): 4 x u64 and 8 x u32 little-endian limbs are the same 32 bytes,
    short values keep zero high limbs -- the layout of every buffer of oracle/ and of the HIP path"""
    g = field_gold()["limbs"]
    vals = [int(v, 16) for v in g["values"]]
    blob = b"".join(v.to_bytes(32, "little") for v in vals)
    assert bytes.fromhex(g["u64x4_to_u32x8"]) == blob and bytes.fromhex(g["u32x8_to_u64x4"]) == blob
    assert ol.to_limbs(vals).tobytes() == blob and ol.from_limbs(np.frombuffer(blob, dtype=np.uint32)) == vals


@pytest.mark.parametrize("name", ROW_SETS)
def test_row_stream_of_the_reference_backend_through_the_oracle(name):
    """the stream recorded from witness_manager (process_reset_linear_row / _quadratic_rows / process_masks / finalize,
    witness_manager.hpp:200-321,497-507) for a guest: the oracle's sampler reproduces every pad and the three masks at the positions the
    commit order implies, its constant sum equals constsum(), and its prover over these rows gives the recorded envelope"""
    f = load_rows(name)
    m = f["meta"]
    l, k, n, t = m["l"], m["k"], m["n"], m["t"]
    key = bytes.fromhex(m["encoding_seed"])
    kinds, vals, rands = f["kinds"], f["vals"], f["rands"]
    assert len(kinds) == m["rows"] and set(kinds.tolist()) <= {0, 1, 2, 3}
    pos = 0
    for r in range(len(kinds)):                       # pad_encoding_random: k - l draws per row, in callback order
        assert np.array_equal(vals[r, l:], ol.rng_fill(key, pos, k - l)), "pads of row %d" % r
        assert not rands[r, l:].any()                 # push_back_zeros(pad_zero + pad_random) on the randomness row
        pos += k - l
    masks = ol.form_masks(key, pos, l, k)
    for arr, nm in zip(masks, ("mask_code", "mask_lin", "mask_quad")):
        assert hashlib.sha256(arr.tobytes()).hexdigest() == m["masks_sha256"][nm], nm
    for r in range(len(kinds)):                       # quadratic triples hold z = x * y slot by slot
        if kinds[r] == 3:
            x, y, z = (ol.from_limbs(vals[r - 2 + i, :l]) for i in range(3))
            assert all(a * b % P == c for a, b, c in zip(x, y, z))
    p = ol.prove_rows(l, k, n, t, kinds, vals, *masks, rands, None, generated_at=m["generated_at"])
    assert p["const_sum"] == f["constsum"], "minus the sum of <row, randomness row> must equal witness_manager::constsum()"
    assert p["valid"] == [1, 1, 1]
    assert p["root"].hex() == m["oracle_root"] and p["stage1_seed"].hex() == m["oracle_stage1_seed"]
    assert hashlib.sha256(p["proof"]).hexdigest() == m["oracle_proof_sha256"] and len(p["proof"]) == m["oracle_proof_len"]


def test_i32_add_stream_shape():
    """tests/i32_add.wat on the reference backend: 8 x (3 x 32-bit range checks + one 33-bit decomposition) = 1032 quadratic constraints"""
    f = load_rows("i32_add_320")
    assert f["kinds"].tolist() == [1, 2, 3, 1, 2, 3, 1, 2, 3, 0, 1, 2, 3]          # three full triples are flushed before the linear row
    quad_slots = sum(int(np.any(f["vals"][r, :320] != 0, axis=1).sum()) for r in range(13) if f["kinds"][r] == 1)
    assert load_rows("i32_add_8000")["kinds"].tolist() == [0, 1, 2, 3]
    assert quad_slots <= 8 * (3 * 32 + 33)
    x_rows = [r for r in range(13) if f["kinds"][r] == 1]
    bits = np.concatenate([f["vals"][r, :320] for r in x_rows])
    assert set(ol.from_limbs(bits)) <= {0, 1}                                      # every x of a triple is a bit (constrain_bit: b * b = b)


# ---------------------------------------------------------------------------------------------------------------- live
@live
def test_live_fixtures_are_what_the_reference_code_emits():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_ref_backend", os.path.join(GOLD, "make_ref_backend.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fresh = json.loads(json.dumps(mod.field_vectors()))
    assert fresh == field_gold()
    for name, guest, l, k, reps, gen in mod.ROW_SETS:
        f = load_rows(name)
        g1 = rb.guest(guest, l, k, mod.KEYS[0], reps=reps)
        g2 = rb.guest(guest, l, k, mod.KEYS[0], wit_key=bytes.fromhex(f["meta"]["oracle_stage1_seed"]), reps=reps)
        assert np.array_equal(g1["kinds"], f["kinds"]) and np.array_equal(g1["vals"], f["vals"]) and np.array_equal(g2["vals"], f["vals"])
        assert np.array_equal(g2["rands"], f["rands"]) and g2["constsum"] == f["constsum"]
        assert not g1["rands"].any()
        for nm in ("mask_code", "mask_lin", "mask_quad"):
            assert hashlib.sha256(g2[nm].tobytes()).hexdigest() == f["meta"]["masks_sha256"][nm]


@live
def test_live_sampler_on_fresh_keys_and_long_streams():
    rng = np.random.default_rng(20260929)
    for _ in range(3):
        key = rng.integers(0, 256, 32, dtype=np.uint8).tobytes()
        n = int(rng.integers(3000, 9000))
        assert np.array_equal(rb.field_random(key, n), ol.rng_fill(key, 0, n))


@live
def test_live_field_operations_on_random_operands():
    rng = np.random.default_rng(7)
    for _ in range(200):
        a, b = (int.from_bytes(rng.integers(0, 256, 32, dtype=np.uint8).tobytes(), "little") % P for _ in range(2))
        assert rb.field_op("mulmod", a, b) == oracle_op("mulmod", a, b) == a * b % P
        if b:
            assert rb.field_op("divmod", a, b) == oracle_op("divmod", a, b)
            assert rb.field_op("invmod", b) == oracle_op("invmod", b)
    for k in (512, 4096, 1 << 22):
        w = [np.zeros(8, dtype=np.uint32) for _ in range(3)]
        ol.lib().lo_omegas(k, *[ol.ptr(x) for x in w])
        assert np.array_equal(rb.omegas(k), np.stack(w))


@live
def test_live_verifier_policy_derives_the_provers_public_data():
    """the reference's verifier runs the guest itself (verifier_random_policy: no pads, all checks on) to obtain the per-row randomness rows and
    the constant sum it checks the proof against; they must be the ones the prover's stage-2 run produced -- the inputs of lig_rows_verify_*"""
    key, seed1 = bytes(range(32)), hashlib.sha256(b"some stage-1 seed").digest()
    for which, l, k, reps in (("i32_add", 320, 512, 0), ("mul_add", 320, 512, 700)):
        p = rb.guest(which, l, k, key, wit_key=seed1, reps=reps)
        v = rb.guest(which, l, k, key, wit_key=seed1, reps=reps, verifier=True)
        assert np.array_equal(p["kinds"], v["kinds"]) and np.array_equal(p["rands"], v["rands"]) and p["constsum"] == v["constsum"]
        assert np.array_equal(p["vals"][:, :l], v["vals"][:, :l]) and not v["vals"][:, l:].any() and p["vals"][:, l:].any()      # no pads on the verifier's side
