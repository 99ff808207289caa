"""ctypes binding of oracle/liblig_oracle.so -- the CPU restatement used ONLY as the checker.

Builds the library with `make -C oracle` when it is missing (gcc only, no GPU needed).
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "liblig_oracle.so")

P = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
R = (1 << 256) % P


def build():
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("field.c", "ntt.c", "hash.c", "prover.c", "lig_oracle.h")]
    if (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return _SO


class Job(C.Structure):
    _fields_ = [("l", C.c_uint32), ("k", C.c_uint32), ("n", C.c_uint32), ("t", C.c_uint32),
                ("n_linear", C.c_uint64), ("n_quad", C.c_uint64),
                ("encoding_seed", C.c_uint8 * 32), ("witness_key", C.c_uint8 * 32),
                ("generated_at", C.c_int64), ("threads", C.c_int),
                ("batch_ops", C.c_void_p), ("n_batch_ops", C.c_uint64), ("batch_data", C.c_void_p), ("batch_data_bytes", C.c_uint64),
                ("public_args", C.c_void_p), ("public_arg_lens", C.c_void_p), ("n_public_args", C.c_uint64),
                ("witness_bits", C.c_uint32)]

    def set_public_args(self, args):
        args = [bytes(a) for a in (args or [])]
        blob = np.frombuffer(b"".join(args) or b"\0", dtype=np.uint8).copy()
        lens = np.array([len(a) for a in args] or [0], dtype=np.uint64)
        self._pub_keep = (blob, lens)
        self.public_args = blob.ctypes.data if args else None
        self.public_arg_lens = lens.ctypes.data if args else None
        self.n_public_args = len(args)


class Proof(C.Structure):
    _fields_ = [("root", C.c_uint8 * 32), ("stage1_seed", C.c_uint8 * 32), ("stage2_seed", C.c_uint8 * 32),
                ("sample_idx", C.POINTER(C.c_uint32)),
                ("code", C.c_void_p), ("lin", C.c_void_p), ("quad", C.c_void_p), ("samples", C.c_void_p),
                ("rows", C.c_size_t), ("proof", C.POINTER(C.c_uint8)), ("proof_len", C.c_size_t),
                ("const_sum", C.c_uint64 * 4),
                ("valid_code", C.c_int), ("valid_linear", C.c_int), ("valid_quad", C.c_int),
                ("t_stage1", C.c_double), ("t_stage2", C.c_double), ("t_stage3", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        L = _lib
        L.lo_ctx_new.restype = C.c_void_p
        L.lo_ctx_new.argtypes = [C.c_uint32] * 3
        L.lo_ctx_free.argtypes = [C.c_void_p]
        for f in ("lo_ntt_forward", "lo_ntt_inverse"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        for f in ("lo_encode", "lo_encode_2k", "lo_decode"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p]
        L.lo_encode_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.lo_eltwise.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32]
        L.lo_powmod.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.lo_sizeof_sha256.restype = C.c_size_t
        L.lo_colsha_init.argtypes = [C.c_void_p, C.c_size_t]
        L.lo_colsha_update.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.lo_colsha_final.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.lo_sha256_buf.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.lo_merkle_nodes.restype = C.c_size_t
        L.lo_merkle_nodes.argtypes = [C.c_size_t]
        L.lo_merkle_build.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.lo_merkle_decommit.restype = C.c_size_t
        L.lo_merkle_decommit.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.lo_merkle_recommit.argtypes = [C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.lo_aes256_expand.argtypes = [C.c_void_p, C.c_void_p]
        L.lo_aes256_encrypt_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.lo_rng_init.argtypes = [C.c_void_p, C.c_void_p]
        L.lo_rng_keystream.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t]
        L.lo_rng_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.lo_stage1_seed.argtypes = [C.c_void_p] * 3
        L.lo_stage2_seed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.lo_instance_hash_default.argtypes = [C.c_void_p]
        L.lo_instance_hash.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.lo_hash_engine_bytes.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.lo_rand_rows.argtypes = [C.POINTER(Job), C.c_void_p, C.c_void_p, C.c_void_p]
        L.lo_sample_indices.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.lo_synth_key.argtypes = [C.c_uint64, C.c_void_p]
        L.lo_job_rows.restype = C.c_size_t
        L.lo_job_rows.argtypes = [C.POINTER(Job)]
        L.lo_prove.argtypes = [C.POINTER(Job), C.POINTER(Proof)]
        L.lo_verify.argtypes = [C.POINTER(Job), C.c_void_p, C.c_void_p, C.c_size_t]
        L.lo_proof_free.argtypes = [C.POINTER(Proof)]
        L.lo_form_rows.argtypes = [C.POINTER(Job)] + [C.c_void_p] * 4
        L.lo_omegas.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.lo_fr_mul.argtypes = [C.c_void_p] * 3
        L.lo_fr_montmul.argtypes = [C.c_void_p] * 3
    return _lib


# ---- helpers: field elements <-> numpy (count, 8) uint32 little-endian limbs ----
def to_limbs(vals):
    out = np.zeros((len(vals), 8), dtype=np.uint32)
    for i, v in enumerate(vals):
        v = int(v)
        for j in range(8):
            out[i, j] = (v >> (32 * j)) & 0xFFFFFFFF
    return out


def from_limbs(arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint32).reshape(-1, 8)
    b = arr.tobytes()
    return [int.from_bytes(b[32 * i:32 * i + 32], "little") for i in range(arr.shape[0])]


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def rand_field(rng, count):
    """uniform-ish canonical field elements as (count, 8) uint32"""
    raw = rng.integers(0, 1 << 32, size=(count, 8), dtype=np.uint64).astype(np.uint32)
    raw[:, 7] &= 0x3FFFFFFF
    vals = from_limbs(raw)
    return to_limbs([v % P for v in vals])


class Ctx:
    def __init__(self, l, k, n):
        self.l, self.k, self.n = l, k, n
        self.h = lib().lo_ctx_new(l, k, n)
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            lib().lo_ctx_free(self.h)
            self.h = None

    def encode(self, msg):
        buf = np.zeros((self.n, 8), dtype=np.uint32)
        buf[:self.k] = msg
        lib().lo_encode(self.h, ptr(buf))
        return buf

    def encode_2k(self, msg2k):
        buf = np.zeros((self.n, 8), dtype=np.uint32)
        buf[:2 * self.k] = msg2k
        lib().lo_encode_2k(self.h, ptr(buf))
        return buf

    def decode(self, cw):
        buf = np.ascontiguousarray(cw, dtype=np.uint32).copy()
        lib().lo_decode(self.h, ptr(buf))
        return buf

    def ntt(self, which, inverse, data):
        buf = np.ascontiguousarray(data, dtype=np.uint32).copy()
        (lib().lo_ntt_inverse if inverse else lib().lo_ntt_forward)(self.h, which, ptr(buf))
        return buf

    def encode_rows(self, msgs, threads=1):
        rows = msgs.shape[0]
        out = np.zeros((rows, self.n, 8), dtype=np.uint32)
        lib().lo_encode_rows(self.h, ptr(np.ascontiguousarray(msgs)), ptr(out), rows, threads)
        return out


def eltwise(op, x, y, out, scalar=None, bit=0):
    count = x.shape[0]
    sc = to_limbs([scalar]) if scalar is not None else None
    lib().lo_eltwise(op, ptr(x), ptr(y) if y is not None else None, ptr(out), count,
                     ptr(sc) if sc is not None else None, bit)
    return out


def colsha(rows_cw):
    """rows_cw: (rows, ncols, 8) -> leaves (ncols, 32) uint8"""
    rows, ncols = rows_cw.shape[0], rows_cw.shape[1]
    st = np.zeros(lib().lo_sizeof_sha256() * ncols, dtype=np.uint8)
    lib().lo_colsha_init(ptr(st), ncols)
    for r in range(rows):
        lib().lo_colsha_update(ptr(st), ptr(np.ascontiguousarray(rows_cw[r])), ncols)
    leaves = np.zeros((ncols, 32), dtype=np.uint8)
    lib().lo_colsha_final(ptr(st), ptr(leaves), ncols)
    return leaves


def merkle_build(leaves):
    n = leaves.shape[0]
    nodes = np.zeros((lib().lo_merkle_nodes(n), 32), dtype=np.uint8)
    lib().lo_merkle_build(ptr(np.ascontiguousarray(leaves)), n, ptr(nodes))
    return nodes


def merkle_decommit(nodes, nleaves, idx):
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    cap = 64 * len(idx) + 64
    sib = np.zeros((cap, 32), dtype=np.uint8)
    cnt = lib().lo_merkle_decommit(ptr(nodes), nleaves, ptr(idx), len(idx), ptr(sib), cap)
    return sib[:cnt].copy()


def sample_indices(seed, n, t):
    out = np.zeros(t, dtype=np.uint32)
    s = np.frombuffer(bytes(seed), dtype=np.uint8).copy()
    lib().lo_sample_indices(ptr(s), n, t, ptr(out))
    return out


def rng_fill(key, first_elem, count):
    st = np.zeros(60 * 4 + 8, dtype=np.uint8)
    k = np.frombuffer(bytes(key), dtype=np.uint8).copy()
    lib().lo_rng_init(ptr(st), ptr(k))
    st[240:248] = np.frombuffer(int(first_elem).to_bytes(8, "little"), dtype=np.uint8)
    out = np.zeros((count, 8), dtype=np.uint32)
    lib().lo_rng_fill(ptr(st), ptr(out), count)
    return out


def keystream(key, first_block, nblocks):
    st = np.zeros(60 * 4 + 8, dtype=np.uint8)
    k = np.frombuffer(bytes(key), dtype=np.uint8).copy()
    lib().lo_rng_init(ptr(st), ptr(k))
    out = np.zeros(16 * nblocks, dtype=np.uint8)
    lib().lo_rng_keystream(ptr(st), first_block, ptr(out), nblocks)
    return out.tobytes()


def instance_hash(args):
    args = [bytes(a) for a in args]
    blob = np.frombuffer(b"".join(args) or b"\0", dtype=np.uint8).copy()
    lens = np.array([len(a) for a in args] or [0], dtype=np.uint64)
    out = np.zeros(32, dtype=np.uint8)
    lib().lo_instance_hash(ptr(blob), ptr(lens), len(args), ptr(out))
    return out.tobytes()


def hash_engine_bytes(seed, count):
    s = np.frombuffer(bytes(seed), dtype=np.uint8).copy()
    out = np.zeros(count, dtype=np.uint8)
    lib().lo_hash_engine_bytes(ptr(s), count, ptr(out))
    return out.tobytes()


def rand_rows(job, stage1_seed):
    """-> ((R, k, 8) dense stage-2 randomness rows, const_sum bytes)"""
    R = lib().lo_job_rows(C.byref(job)) - 3
    out = np.zeros((max(R, 1), job.k, 8), dtype=np.uint32)
    cs = np.zeros(8, dtype=np.uint32)
    s = np.frombuffer(bytes(stage1_seed), dtype=np.uint8).copy()
    lib().lo_rand_rows(C.byref(job), ptr(s), ptr(out), ptr(cs))
    return out[:R], cs.tobytes()


def row_kinds(job):
    R = lib().lo_job_rows(C.byref(job)) - 3
    out = np.zeros(max(R, 1), dtype=np.uint8)
    lib().lo_row_kinds.argtypes = [C.POINTER(Job), C.c_void_p]
    lib().lo_row_kinds(C.byref(job), ptr(out))
    return out[:R]


def form_rows(job):
    """-> ((R, k, 8) message rows in commit order incl. pads, mask_code (k,8), mask_lin (2k,8), mask_quad (2k,8))"""
    R = lib().lo_job_rows(C.byref(job)) - 3
    rows = np.zeros((max(R, 1), job.k, 8), dtype=np.uint32)
    mc, ml, mq = (np.zeros((m * job.k, 8), dtype=np.uint32) for m in (1, 2, 2))
    lib().lo_form_rows(C.byref(job), ptr(rows), ptr(mc), ptr(ml), ptr(mq))
    return rows[:R], mc, ml, mq


def form_masks(encoding_seed, pos, l, k):
    """the three masks from the encoding stream at element position pos -> (k,8), (2k,8), (2k,8)"""
    mc, ml, mq = (np.zeros((m * k, 8), dtype=np.uint32) for m in (1, 2, 2))
    s = np.frombuffer(bytes(encoding_seed), dtype=np.uint8).copy()
    lib().lo_form_masks.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib().lo_form_masks(ptr(s), pos, l, k, ptr(mc), ptr(ml), ptr(mq))
    return mc, ml, mq


def prove_rows(l, k, n, t, kinds, rows, mask_code, mask_lin, mask_quad, rands, const_sum, generated_at=0, threads=4, public_args=None):
    """the oracle's three stages over a row stream formed elsewhere (lo_prove_rows) -> dict(proof, root, stage1_seed, const_sum, valid)"""
    j = Job()
    j.set_public_args(public_args)
    j.l, j.k, j.n, j.t, j.generated_at, j.threads = l, k, n, t, generated_at, threads
    kinds = np.ascontiguousarray(kinds, dtype=np.uint8)
    arrs = [np.ascontiguousarray(a, dtype=np.uint32) for a in (rows, mask_code, mask_lin, mask_quad)]
    rands = np.ascontiguousarray(rands, dtype=np.uint32) if rands is not None else np.zeros((max(1, len(kinds)), k, 8), dtype=np.uint32)
    cs = np.frombuffer(bytes(const_sum), dtype=np.uint8).copy() if const_sum is not None else None
    pr = Proof()
    L = lib()
    L.lo_prove_rows.argtypes = [C.POINTER(Job), C.c_void_p, C.c_size_t] + [C.c_void_p] * 6 + [C.POINTER(Proof)]
    rc = L.lo_prove_rows(C.byref(j), ptr(kinds), len(kinds), *[ptr(a) for a in arrs], ptr(rands), ptr(cs) if cs is not None else None, C.byref(pr))
    if rc != 0:
        raise ValueError("lo_prove_rows rejected the row stream")
    out = dict(proof=bytes(pr.proof[:pr.proof_len]), root=bytes(pr.root), stage1_seed=bytes(pr.stage1_seed), stage2_seed=bytes(pr.stage2_seed),
               const_sum=bytes(pr.const_sum), valid=[pr.valid_code, pr.valid_linear, pr.valid_quad], rows=pr.rows)
    L.lo_proof_free(C.byref(pr))
    return out


def make_job(l, k, n, t, n_linear, n_quad=0, synth_seed=1, generated_at=0, threads=1, public_args=None):
    j = Job()
    j.set_public_args(public_args)
    j.l, j.k, j.n, j.t = l, k, n, t
    j.n_linear, j.n_quad = n_linear, n_quad
    for i in range(32):
        j.encoding_seed[i] = i
    key = (C.c_uint8 * 32)()
    lib().lo_synth_key(synth_seed, key)
    for i in range(32):
        j.witness_key[i] = key[i]
    j.generated_at = generated_at
    j.threads = threads
    return j
