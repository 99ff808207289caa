"""ctypes binding of oracle/_ref/libref_backend.so = the REFERENCE's own GMP-side code (src/bn254.cpp, finite_field_gmp.hpp,
util/csprng.hpp, util/mpz_vector.hpp, zkp/backend/{witness_manager,core}.hpp) behind oracle/ref_backend.cpp.  Built by `make -C oracle`
where /root/reference and GMP headers exist (the build container); load() returns None elsewhere and the tests fall back on the
committed vectors (tests/golden/ref_field.json, ref_rows_*.npz, made with it by tests/golden/make_ref_backend.py)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_ref", "libref_backend.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            return None
        try:
            L = C.CDLL(SO)
        except OSError:
            return None
        vp, sz, u64 = C.c_void_p, C.c_size_t, C.c_uint64
        L.ref_field_random.argtypes = [C.c_char_p, sz, vp]
        L.ref_engine_raw.argtypes = [C.c_char_p, sz, vp]
        L.ref_omegas.argtypes = [u64, vp]
        L.ref_constants.argtypes = [vp]
        L.ref_field_op.argtypes = [C.c_int, vp, vp, vp]
        L.ref_limbs_roundtrip.restype = sz
        L.ref_limbs_roundtrip.argtypes = [vp, sz, sz, sz, vp, sz, sz]
        L.ref_guest_run.restype = vp
        L.ref_guest_run.argtypes = [C.c_int, u64, u64, C.c_char_p, C.c_char_p, C.c_int, u64]
        L.ref_guest_rows.restype = sz
        L.ref_guest_rows.argtypes = [vp]
        L.ref_guest_read.argtypes = [vp] * 6
        L.ref_guest_free.argtypes = [vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def field_random(key, count):
    out = np.zeros((count, 8), dtype=np.uint32)
    load().ref_field_random(bytes(key), count, _p(out))
    return out


def engine_raw(key, count):
    out = np.zeros((count, 8), dtype=np.uint32)
    load().ref_engine_raw(bytes(key), count, _p(out))
    return out


def omegas(k):
    out = np.zeros((3, 8), dtype=np.uint32)
    load().ref_omegas(k, _p(out))
    return out


def constants():
    out = np.zeros((8, 8), dtype=np.uint32)
    load().ref_constants(_p(out))
    return out


OPS = dict(mulmod=0, invmod=1, powmod=2, divmod=3, mont_mulmod=4, addmod=5, submod=6, negate=7, reduce=8, powmod_ui=9, reduce_u256=10)


def field_op(op, a, b=0):
    """a, b: Python ints < 2^256 -> int"""
    x = np.frombuffer(int(a).to_bytes(32, "little"), dtype=np.uint8).copy()
    y = np.frombuffer(int(b).to_bytes(32, "little"), dtype=np.uint8).copy()
    out = np.zeros(32, dtype=np.uint8)
    assert load().ref_field_op(OPS[op], _p(x), _p(y), _p(out)) == 0
    return int.from_bytes(out.tobytes(), "little")


def limbs_roundtrip(data, count, limb_size, limb_count, out_limb_size, out_limb_count):
    src = np.frombuffer(bytes(data), dtype=np.uint8).copy()
    out = np.zeros(count * out_limb_size * out_limb_count, dtype=np.uint8)
    words = load().ref_limbs_roundtrip(_p(src), count, limb_size, limb_count, _p(out), out_limb_size, out_limb_count)
    return out.tobytes(), words


GUESTS = dict(i32_add=0, mul_add=1)


def guest(which, l, k, enc_key, wit_key=None, reps=0, verifier=False):
    """the row stream the reference's witness_manager emits for a guest: stage-1 policy (wit_key None) or stage-2 policy.
    -> dict(kinds (R,), vals (R,k,8), rands (R,k,8), mask_code (k,8), mask_lin (2k,8), mask_quad (2k,8), constsum bytes)"""
    L = load()
    h = L.ref_guest_run(GUESTS[which], l, k, bytes(enc_key), bytes(wit_key) if wit_key is not None else None, (2 if verifier else 1) if wit_key is not None else 0, reps)
    if not h:
        raise RuntimeError("the reference guest failed")
    try:
        R = L.ref_guest_rows(h)
        kinds = np.zeros(R, dtype=np.uint8)
        vals, rands = np.zeros((R, k, 8), dtype=np.uint32), np.zeros((R, k, 8), dtype=np.uint32)
        masks = np.zeros((5 * k, 8), dtype=np.uint32)
        cs = np.zeros(32, dtype=np.uint8)
        L.ref_guest_read(h, _p(kinds), _p(vals), _p(rands), _p(masks), _p(cs))
    finally:
        L.ref_guest_free(h)
    return dict(kinds=kinds, vals=vals, rands=rands, mask_code=masks[:k].copy(), mask_lin=masks[k:3 * k].copy(), mask_quad=masks[3 * k:].copy(),
                constsum=cs.tobytes())
