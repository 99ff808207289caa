"""ligero-prover_amd -- ctypes binding of liblig_hip.so (the MI355X-native Ligero prover backend).

This is a thin loader for tests and bench.py; the product is the C-ABI library (include/lig_hip.h) and the
C++ `hip_context` mirror of the reference's `webgpu_context` (include/lig_hip_context.hpp).  There is NO CPU
fallback: if the HIP library is missing or no GPU is visible, construction raises.

The directory name contains a hyphen, so import it by path:
    spec = importlib.util.spec_from_file_location("ligero_prover_amd", ".../ligero-prover_amd/__init__.py")
(tests/hip_lib.py and bench.py do exactly that).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LIG_HIP_LIB") or os.path.join(_HERE, "liblig_hip.so")      # LIG_HIP_LIB: A/B builds of experiments

OPS = dict(ADD=0, SUB=1, ADD_ASSIGN=2, ADD_CONST=3, SUB_CONST=4, CONST_SUB=5, MUL=6, MUL_CONST=7,
           MONTMUL_CONST=8, FMA=9, FMA_CONST=10, DIV=11, BIT_DECOMPOSE=12)
SIZE_K, SIZE_2K, SIZE_N = 0, 1, 2

EXPORTS = [
    "lig_ctx_create", "lig_ctx_destroy", "lig_sync", "lig_last_error", "lig_version", "lig_message_size",
    "lig_padding_size", "lig_encoding_size", "lig_stream", "lig_malloc", "lig_free", "lig_write", "lig_write_clear",
    "lig_clear", "lig_copy", "lig_read", "lig_encode", "lig_encode_2k", "lig_decode", "lig_ntt", "lig_eltwise",
    "lig_powmod", "lig_sha_state_bytes", "lig_sha_init", "lig_sha_update", "lig_sha_final", "lig_sample_init",
    "lig_sample_gather", "lig_encode_rows", "lig_sha_update_rows", "lig_merkle_nodes", "lig_merkle_build",
    "lig_rlc_rows", "lig_gather_rows", "lig_rng_fill", "lig_profile_enable", "lig_profile_read",
    "lig_synth_prepare", "lig_synth_prove", "lig_trace_rows", "lig_trace_destroy",
    "lig_shard_prepare", "lig_shard_prove", "lig_shard_destroy", "lig_synth_verify", "lig_verify_release",
    "lig_proof_gzip_bound", "lig_proof_gzip", "lig_proof_gunzip_size", "lig_proof_gunzip",
    "lig_rows_begin", "lig_rows_commit", "lig_rows_prove", "lig_rows_restart", "lig_rng_fill_rows",
    "lig_rows_verify_begin", "lig_rows_verify_finish", "lig_vtrace_destroy",
    "lig_public_arg_bytes", "lig_instance_hash", "lig_sample_columns",
    "lig_rccl_unique_id", "lig_rccl_comm_create", "lig_rccl_comm_destroy", "lig_rccl_available", "lig_rccl_comm_count",
    "lig_shard_plan", "lig_ipc_comm_create", "lig_ipc_comm_destroy",
    "lig_device_pci_bus_id", "lig_device_peer_access", "lig_host_alloc", "lig_host_free", "lig_write_async", "lig_fence_record", "lig_fence_wait", "lig_fence_destroy", "lig_rows_push_rands", "lig_rows_push_rands_sparse",
    "lig_abi_sizes", "lig_shard_rows_plan", "lig_shard_rows_begin", "lig_shard_rows_restart", "lig_shard_rows_commit", "lig_shard_rows_prove",
    "lig_upload_health", "lig_profile_read_launches",
]

ROW_KINDS = dict(LINEAR=0, QX=1, QY=2, QZ=3, INIT=4, BIT=5, EQX=6, EQY=7, BQX=8, BQY=9, BQZ=10)
ROW_DRAW_PAD = 0x80
ARG_I64, ARG_STR, ARG_HEX = 0, 1, 2


class VerifyInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("parsed", "indices_match", "valid_merkle", "valid_code", "valid_linear", "valid_quad",
                                          "code_equal", "linear_equal", "quad_equal", "accept", "reserved")] + [("ms_total", C.c_double)]

A2A_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)


A2A_ON_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class Comm(C.Structure):
    """lig_comm: host-synchronous callbacks (tests over gloo) and, when made by lig_rccl_comm_create, the stream-ordered RCCL forms"""
    _fields_ = [("user", C.c_void_p), ("all_to_all", A2A_FN), ("all_gather", A2A_FN),
                ("all_to_all_on", A2A_ON_FN), ("all_gather_on", A2A_ON_FN), ("forget", C.CFUNCTYPE(None, C.c_void_p)),
                ("failed", C.CFUNCTYPE(C.c_int, C.c_void_p)), ("abort", C.CFUNCTYPE(None, C.c_void_p))]


class SynthJob(C.Structure):
    _fields_ = [("n_linear", C.c_uint64), ("n_quad", C.c_uint64), ("encoding_seed", C.c_uint8 * 32),
                ("witness_key", C.c_uint8 * 32), ("program_hash", C.c_uint8 * 32), ("generated_at", C.c_int64),
                ("version", C.c_char * 16),
                ("batch_ops", C.c_void_p), ("n_batch_ops", C.c_uint64), ("batch_data", C.c_void_p), ("batch_data_bytes", C.c_uint64),
                ("public_args", C.c_void_p), ("public_arg_lens", C.c_void_p), ("n_public_args", C.c_uint64)]

    def set_public_args(self, args):
        """args: list of byte strings in input_args form (see public_arg_bytes); kept alive on the job object"""
        _attach_public_args(self, args)


class RowsJob(C.Structure):
    _fields_ = [("rows", C.c_uint64), ("kinds", C.c_void_p), ("msgs", C.c_void_p), ("msgs_on_device", C.c_int32),
                ("reserved", C.c_int32), ("encoding_seed", C.c_uint8 * 32), ("program_hash", C.c_uint8 * 32),
                ("generated_at", C.c_int64), ("version", C.c_char * 16),
                ("public_args", C.c_void_p), ("public_arg_lens", C.c_void_p), ("n_public_args", C.c_uint64),
                ("dense_rands_per_row", C.c_void_p), ("elem_bytes", C.c_void_p)]

    def set_public_args(self, args):
        _attach_public_args(self, args)


def _attach_public_args(job, args):
    args = [bytes(a) for a in (args or [])]
    blob = np.frombuffer(b"".join(args) or b"\0", dtype=np.uint8).copy()
    lens = np.array([len(a) for a in args] or [0], dtype=np.uint64)
    job._pub_keep = (blob, lens)
    job.public_args = blob.ctypes.data if args else None
    job.public_arg_lens = lens.ctypes.data if args else None
    job.n_public_args = len(args)


class ProofInfo(C.Structure):
    _fields_ = [("root", C.c_uint8 * 32), ("stage1_seed", C.c_uint8 * 32), ("stage2_seed", C.c_uint8 * 32),
                ("const_sum", C.c_uint8 * 32), ("rows", C.c_uint64), ("valid_code", C.c_int32),
                ("valid_linear", C.c_int32), ("valid_quad", C.c_int32), ("reserved", C.c_int32),
                ("ms_stage1", C.c_double), ("ms_stage2", C.c_double), ("ms_stage3", C.c_double), ("ms_total", C.c_double)]


def build(force=False):
    """compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)"""
    import subprocess
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", csrc, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", csrc, "-j8"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def load_library():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("liblig_hip.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "-- there is no CPU fallback for the HIP path" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, sz, u32, u64 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64
    # struct layouts of this binding against the library's own sizeof (include/lig_hip.h: lig_abi_sizes): a stale binding would hand the
    # library structs it reads past the end of
    sizes = (u32 * 6)()
    L.lig_abi_sizes.argtypes = [C.POINTER(u32)]
    L.lig_abi_sizes.restype = None
    L.lig_abi_sizes(sizes)
    for idx, (name, cls) in {1: ("lig_synth_job", SynthJob), 2: ("lig_proof_info", ProofInfo), 3: ("lig_verify_info", VerifyInfo),
                             4: ("lig_rows_job", RowsJob), 5: ("lig_comm", Comm)}.items():
        if C.sizeof(cls) != sizes[idx]:
            raise RuntimeError("binding out of date: sizeof(%s) is %d in %s, %d in this module" % (name, sizes[idx], LIB_PATH, C.sizeof(cls)))
    L.lig_ctx_create.argtypes = [C.POINTER(vp), C.c_int, u32, u32, u32]
    L.lig_ctx_destroy.argtypes = [vp]
    L.lig_ctx_destroy.restype = None
    L.lig_sync.argtypes = [vp]
    L.lig_last_error.argtypes = [vp]
    L.lig_last_error.restype = C.c_char_p
    L.lig_version.restype = C.c_char_p
    for f in ("lig_message_size", "lig_padding_size", "lig_encoding_size"):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = u32
    L.lig_stream.argtypes = [vp]
    L.lig_stream.restype = vp
    L.lig_malloc.argtypes = [vp, sz, C.POINTER(vp)]
    L.lig_free.argtypes = [vp, vp]
    L.lig_write.argtypes = [vp, vp, vp, sz]
    L.lig_write_clear.argtypes = [vp, vp, sz, vp, sz]
    L.lig_clear.argtypes = [vp, vp, sz]
    L.lig_copy.argtypes = [vp, vp, vp, sz]
    L.lig_read.argtypes = [vp, vp, vp, sz]
    for f in ("lig_encode", "lig_encode_2k", "lig_decode"):
        getattr(L, f).argtypes = [vp, vp]
    L.lig_ntt.argtypes = [vp, vp, C.c_int, C.c_int]
    L.lig_eltwise.argtypes = [vp, C.c_int, vp, vp, vp, sz, vp, u32]
    L.lig_powmod.argtypes = [vp, vp, vp, vp, vp, sz, C.c_int]
    L.lig_sha_state_bytes.argtypes = [sz]
    L.lig_sha_state_bytes.restype = sz
    L.lig_sha_init.argtypes = [vp, vp, sz]
    L.lig_sha_update.argtypes = [vp, vp, vp]
    L.lig_sha_final.argtypes = [vp, vp, vp]
    L.lig_sample_init.argtypes = [vp, vp, sz]
    L.lig_sample_gather.argtypes = [vp, vp, vp, sz]
    L.lig_encode_rows.argtypes = [vp, vp, vp, sz]
    L.lig_sha_update_rows.argtypes = [vp, vp, vp, sz]
    L.lig_merkle_nodes.argtypes = [sz]
    L.lig_merkle_nodes.restype = sz
    L.lig_merkle_build.argtypes = [vp, vp, sz, vp]
    L.lig_rlc_rows.argtypes = [vp, vp, vp, sz, vp, vp, vp, vp, vp, sz, vp]
    L.lig_gather_rows.argtypes = [vp, vp, sz, vp]
    L.lig_rng_fill.argtypes = [vp, vp, u64, vp, sz]
    L.lig_synth_prepare.argtypes = [vp, C.POINTER(SynthJob), C.POINTER(vp)]
    L.lig_synth_prove.argtypes = [vp, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(sz), C.POINTER(ProofInfo)]
    L.lig_trace_rows.argtypes = [vp]
    L.lig_trace_rows.restype = u64
    L.lig_trace_destroy.argtypes = [vp]
    L.lig_trace_destroy.restype = None
    L.lig_proof_gzip_bound.restype = sz
    L.lig_proof_gzip_bound.argtypes = [sz]
    L.lig_proof_gzip.argtypes = [vp, sz, vp, sz, C.POINTER(sz)]
    L.lig_proof_gunzip_size.restype = sz
    L.lig_proof_gunzip_size.argtypes = [vp, sz]
    L.lig_proof_gunzip.argtypes = [vp, sz, vp, sz, C.POINTER(sz)]
    L.lig_synth_verify.argtypes = [vp, C.POINTER(SynthJob), vp, vp, sz, C.POINTER(VerifyInfo)]
    L.lig_verify_release.argtypes = [vp]
    L.lig_shard_prepare.argtypes = [vp, C.POINTER(SynthJob), u32, u32, C.POINTER(Comm), C.POINTER(vp)]
    L.lig_shard_prove.argtypes = [vp, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(sz), C.POINTER(ProofInfo)]
    L.lig_shard_destroy.argtypes = [vp]
    L.lig_shard_destroy.restype = None
    L.lig_rows_begin.argtypes = [vp, C.POINTER(RowsJob), C.POINTER(vp)]
    L.lig_rows_commit.argtypes = [vp, vp, vp]
    L.lig_rows_restart.argtypes = [vp, vp, C.c_int]
    L.lig_rows_verify_begin.argtypes = [vp, C.POINTER(RowsJob), vp, sz, C.POINTER(vp), vp, C.POINTER(VerifyInfo)]
    L.lig_rows_verify_finish.argtypes = [vp, vp, C.c_int, vp, C.POINTER(VerifyInfo)]
    L.lig_vtrace_destroy.argtypes = [vp]
    L.lig_vtrace_destroy.restype = None
    L.lig_rows_prove.argtypes = [vp, vp, C.c_int, vp, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(sz), C.POINTER(ProofInfo)]
    L.lig_rng_fill_rows.argtypes = [vp, vp, u64, vp, sz, vp]
    L.lig_public_arg_bytes.argtypes = [C.c_int, C.c_char_p, vp, sz, C.POINTER(sz)]
    L.lig_instance_hash.argtypes = [vp, vp, sz, vp]
    L.lig_sample_columns.argtypes = [vp, u32, u32, vp]
    L.lig_rccl_unique_id.argtypes = [vp]
    L.lig_rccl_comm_create.argtypes = [vp, vp, u32, u32, C.POINTER(Comm)]
    L.lig_rccl_comm_destroy.argtypes = [C.POINTER(Comm)]
    L.lig_rccl_comm_destroy.restype = None
    L.lig_shard_rows_plan.argtypes = [vp, sz, u32, C.POINTER(u64), vp, sz]
    L.lig_shard_rows_begin.argtypes = [vp, C.POINTER(RowsJob), u32, u32, C.POINTER(Comm), C.POINTER(vp)]
    L.lig_shard_rows_restart.argtypes = [vp, vp, C.c_int]
    L.lig_shard_rows_commit.argtypes = [vp, vp, vp]
    L.lig_shard_rows_prove.argtypes = [vp, vp, C.c_int, vp, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(sz), C.POINTER(ProofInfo)]
    L.lig_ipc_comm_create.argtypes = [vp, C.c_char_p, u32, u32, C.POINTER(Comm)]
    L.lig_ipc_comm_destroy.argtypes = [C.POINTER(Comm)]
    L.lig_ipc_comm_destroy.restype = None
    L.lig_rccl_available.argtypes = [C.c_char_p, sz, C.POINTER(C.c_int)]
    L.lig_rccl_comm_count.argtypes = [C.POINTER(Comm), C.POINTER(u32)]
    L.lig_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, sz]
    L.lig_device_peer_access.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.lig_host_alloc.argtypes = [vp, sz, C.POINTER(vp)]
    L.lig_host_free.argtypes = [vp, vp]
    L.lig_write_async.argtypes = [vp, vp, vp, sz]
    L.lig_fence_record.argtypes = [vp, C.POINTER(vp)]
    L.lig_fence_wait.argtypes = [vp, vp]
    L.lig_fence_destroy.argtypes = [vp, vp]
    L.lig_fence_destroy.restype = None
    L.lig_rows_push_rands.argtypes = [vp, u64, u64, vp]
    L.lig_upload_health.argtypes = [vp, C.POINTER(u32), C.POINTER(u32)]
    L.lig_rows_push_rands_sparse.argtypes = [vp, u64, u64, vp, vp]
    L.lig_profile_enable.argtypes = [vp, C.c_int]
    L.lig_profile_read.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_double)]
    L.lig_profile_read_launches.argtypes = [vp, u32, C.POINTER(u64), C.POINTER(C.c_double)]
    return L


class LigError(RuntimeError):
    pass


# ---- host-only transcript helpers (no GPU needed)
def public_arg_bytes(kind, text):
    """one JSON "args" entry ({"i64": ..} / {"str": ..} / {"hex": ..}) -> the bytes the reference hashes"""
    L = load_library()
    kind = {"i64": ARG_I64, "str": ARG_STR, "hex": ARG_HEX}.get(kind, kind)
    ln = C.c_size_t()
    txt = str(text).encode()
    L.lig_public_arg_bytes(kind, txt, None, 0, C.byref(ln))
    buf = (C.c_uint8 * max(1, ln.value))()
    if L.lig_public_arg_bytes(kind, txt, buf, ln.value, C.byref(ln)) != 0:
        raise LigError("malformed public argument %r" % (text,))
    return bytes(buf[:ln.value])


def instance_hash(args):
    L = load_library()
    args = [bytes(a) for a in args]
    blob = np.frombuffer(b"".join(args) or b"\0", dtype=np.uint8).copy()
    lens = np.array([len(a) for a in args] or [0], dtype=np.uint64)
    out = np.zeros(32, dtype=np.uint8)
    if L.lig_instance_hash(_hptr(blob), _hptr(lens), len(args), _hptr(out)) != 0:
        raise LigError("lig_instance_hash failed")
    return out.tobytes()


def rccl_available():
    """-> (ok, path of the librccl the library resolved or the reason it could not, ncclGetVersion code)"""
    L = load_library()
    buf, ver = C.create_string_buffer(512), C.c_int()
    rc = L.lig_rccl_available(buf, 512, C.byref(ver))
    return rc == 0, buf.value.decode(), ver.value


def device_pci_bus_id(device):
    """-> "0000:c1:00.0" of a HIP device ordinal (None if the runtime cannot say)"""
    L = load_library()
    buf = C.create_string_buffer(64)
    return buf.value.decode() if L.lig_device_pci_bus_id(device, buf, 64) == 0 else None


def device_peer_access(device, peer):
    L = load_library()
    can = C.c_int()
    return bool(can.value) if L.lig_device_peer_access(device, peer, C.byref(can)) == 0 else None


def shard_plan(job, l, world):
    """-> (rounds, boundaries): the block-cyclic deal of the job's rows over `world` ranks (host only)"""
    L = load_library()
    L.lig_shard_plan.argtypes = [C.POINTER(SynthJob), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.c_void_p, C.c_size_t]
    rounds = C.c_uint64()
    cap = 1 << 16
    b = np.zeros(cap, dtype=np.uint64)
    if L.lig_shard_plan(C.byref(job), l, world, C.byref(rounds), _hptr(b), cap) != 0:
        raise LigError("lig_shard_plan failed")
    return int(rounds.value), [int(x) for x in b[:rounds.value * world + 1]]


def pack_rows(rows, widths, l):
    """rows (R, k, 8) uint32 + per-row width (4 / 8 / 32) -> the packed byte array of the narrow row format
    (lig_rows_job.elem_bytes): a narrow row contributes its l data slots as little-endian integers of that width"""
    parts = []
    for r, w in enumerate(widths):
        if w in (0, 32):
            parts.append(np.ascontiguousarray(rows[r], dtype=np.uint32).tobytes())
        else:
            assert not rows[r, :l, w // 4:].any(), "row %d does not fit %d-byte elements" % (r, w)
            parts.append(np.ascontiguousarray(rows[r, :l, :w // 4], dtype=np.uint32).tobytes())
    return np.frombuffer(b"".join(parts), dtype=np.uint8).copy()


def shard_rows_plan(kinds, world):
    """-> (rounds, boundaries): the block-cyclic deal of a rows job's committed rows over `world` ranks (host only):
    global chunk g = rows [b[g], b[g+1]) belongs to rank g mod world"""
    L = load_library()
    kinds = np.ascontiguousarray(kinds, dtype=np.uint8)
    rounds = C.c_uint64()
    cap = 1 << 16
    b = np.zeros(cap, dtype=np.uint64)
    if L.lig_shard_rows_plan(_hptr(kinds) if len(kinds) else None, len(kinds), world, C.byref(rounds), _hptr(b), cap) != 0:
        raise LigError("lig_shard_rows_plan failed")
    return int(rounds.value), [int(x) for x in b[:rounds.value * world + 1]]


def local_rows_of(boundaries, rank, world):
    """global row indices of `rank`'s chunks, in commit order"""
    out = []
    for g in range(rank, len(boundaries) - 1, world):
        out.extend(range(boundaries[g], boundaries[g + 1]))
    return out


def sample_columns(seed, n, t=192):
    L = load_library()
    s = np.frombuffer(bytes(seed), dtype=np.uint8).copy()
    out = np.zeros(min(t, n), dtype=np.uint32)
    if L.lig_sample_columns(_hptr(s), n, t, _hptr(out)) != 0:
        raise LigError("lig_sample_columns failed")
    return out


def _hptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Context:
    """owns a lig_ctx; device buffers are plain integer device pointers"""

    def __init__(self, l, k, n, device=0):
        self.L = load_library()
        self.l, self.k, self.n = l, k, n
        h = C.c_void_p()
        rc = self.L.lig_ctx_create(C.byref(h), device, l, k, n)
        self.h = h
        if rc != 0:
            msg = self.L.lig_last_error(h).decode() if h else "invalid arguments"
            if h:
                self.L.lig_ctx_destroy(h)
            self.h = None
            raise LigError("lig_ctx_create failed (%d): %s" % (rc, msg))
        self._bufs = []

    def close(self):
        if getattr(self, "h", None):
            for p in self._bufs:
                self.L.lig_free(self.h, p)
            self._bufs = []
            self.L.lig_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != 0:
            raise LigError("lig call failed (%d): %s" % (rc, self.L.lig_last_error(self.h).decode()))

    # ---- buffers
    def malloc(self, nbytes):
        p = C.c_void_p()
        self.check(self.L.lig_malloc(self.h, nbytes, C.byref(p)))
        self._bufs.append(p)
        return p

    def free(self, p):
        self._bufs = [q for q in self._bufs if q.value != p.value]
        self.check(self.L.lig_free(self.h, p))

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.malloc(arr.nbytes)
        self.check(self.L.lig_write(self.h, p, _hptr(arr), arr.nbytes))
        return p

    def write(self, p, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        self.check(self.L.lig_write(self.h, C.c_void_p(p.value + offset), _hptr(arr), arr.nbytes))

    def download(self, p, shape, dtype=np.uint32, offset=0):
        out = np.zeros(shape, dtype=dtype)
        self.check(self.L.lig_read(self.h, _hptr(out), C.c_void_p(p.value + offset), out.nbytes))
        return out

    def sync(self):
        self.check(self.L.lig_sync(self.h))

    # ---- ops (thin)
    def encode(self, p): self.check(self.L.lig_encode(self.h, p))
    def encode_2k(self, p): self.check(self.L.lig_encode_2k(self.h, p))
    def decode(self, p): self.check(self.L.lig_decode(self.h, p))
    def ntt(self, p, which, inverse): self.check(self.L.lig_ntt(self.h, p, which, int(inverse)))
    def encode_rows(self, msgs, cws, rows): self.check(self.L.lig_encode_rows(self.h, msgs, cws, rows))

    def eltwise(self, op, x, y, out, count, scalar=None, bit=0):
        sc = None
        if scalar is not None:
            sc = np.frombuffer(int(scalar).to_bytes(32, "little"), dtype=np.uint8).copy()
        self.check(self.L.lig_eltwise(self.h, OPS[op] if isinstance(op, str) else op, x, y, out, count, _hptr(sc), bit))

    def powmod(self, base, exp_p, coeff_p, out_p, count, add=False):
        b = np.frombuffer(int(base).to_bytes(32, "little"), dtype=np.uint8).copy()
        self.check(self.L.lig_powmod(self.h, _hptr(b), exp_p, coeff_p, out_p, count, int(add)))

    def sha_state(self, n_inst):
        p = self.malloc(self.L.lig_sha_state_bytes(n_inst))
        self.check(self.L.lig_sha_init(self.h, p, n_inst))
        return p

    def sha_update_rows(self, st, cws, rows): self.check(self.L.lig_sha_update_rows(self.h, st, cws, rows))
    def sha_final(self, st, digests): self.check(self.L.lig_sha_final(self.h, st, digests))

    def merkle_build(self, leaves, n_leaves):
        nodes = self.malloc(32 * self.L.lig_merkle_nodes(n_leaves))
        self.check(self.L.lig_merkle_build(self.h, leaves, n_leaves, nodes))
        return nodes

    def sample_init(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        self.check(self.L.lig_sample_init(self.h, _hptr(idx), len(idx)))

    def gather_rows(self, cws, rows, out): self.check(self.L.lig_gather_rows(self.h, cws, rows, out))

    def rlc_rows(self, U, Rn, rows, rc, code, lin, triples=None, rq=None, quad=None):
        def sc(vals):
            return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), dtype=np.uint8).copy()
        rcb = sc(rc) if rc is not None else None
        nt = 0 if triples is None else len(triples)
        tri = np.ascontiguousarray(np.array(triples, dtype=np.uint32).reshape(-1)) if nt else None
        rqb = sc(rq) if nt else None
        self.check(self.L.lig_rlc_rows(self.h, U, Rn, rows, _hptr(rcb), code, lin, _hptr(tri), _hptr(rqb), nt, quad))

    # ---- batched prover over a synthetic trace
    @staticmethod
    def make_job(n_linear, n_quad=0, synth_seed=1, generated_at=0, encoding_seed=None, public_args=None):
        import hashlib
        job = SynthJob()
        job.set_public_args(public_args)
        job.n_linear, job.n_quad, job.generated_at = n_linear, n_quad, generated_at
        es = bytes(range(32)) if encoding_seed is None else bytes(encoding_seed)
        wk = hashlib.sha256(b"lig-synth" + int(synth_seed).to_bytes(8, "little")).digest()
        for i in range(32):
            job.encoding_seed[i] = es[i]
            job.witness_key[i] = wk[i]
            job.program_hash[i] = 0
        job.version = b"1.5.0"
        return job

    # ---- RCCL communicator of the sharded prover (csrc/comm_rccl.hip)
    def rccl_unique_id(self):
        out = np.zeros(128, dtype=np.uint8)
        self.check(self.L.lig_rccl_unique_id(_hptr(out)))
        return out.tobytes()

    def rccl_comm(self, unique_id, rank, world):
        comm = Comm()
        uid = np.frombuffer(bytes(unique_id), dtype=np.uint8).copy()
        self.check(self.L.lig_rccl_comm_create(self.h, _hptr(uid), rank, world, C.byref(comm)))
        return comm

    def rccl_comm_destroy(self, comm):
        self.L.lig_rccl_comm_destroy(C.byref(comm))

    def ipc_comm(self, shm_name, rank, world):
        """the process-to-process communicator (csrc/comm_ipc.hip): same fresh "/name" on every rank"""
        comm = Comm()
        self.check(self.L.lig_ipc_comm_create(self.h, shm_name.encode(), rank, world, C.byref(comm)))
        return comm

    def ipc_comm_destroy(self, comm):
        self.L.lig_ipc_comm_destroy(C.byref(comm))

    def rccl_comm_count(self, comm):
        n = C.c_uint32()
        self.check(self.L.lig_rccl_comm_count(C.byref(comm), C.byref(n)))
        return n.value

    # ---- one trace sharded over ranks (comm: a Comm built by dist.Group.make_comm)
    def shard_prepare(self, job, rank, world, comm):
        t = C.c_void_p()
        self.check(self.L.lig_shard_prepare(self.h, C.byref(job), rank, world, C.byref(comm), C.byref(t)))
        return t

    def shard_prove(self, shard, copy=True):
        proof, ln, info = C.POINTER(C.c_uint8)(), C.c_size_t(), ProofInfo()
        self.check(self.L.lig_shard_prove(shard, C.byref(proof), C.byref(ln), C.byref(info)))
        if not copy:
            return (C.addressof(proof.contents), ln.value), info
        return C.string_at(proof, ln.value), info

    def shard_destroy(self, shard):
        self.L.lig_shard_destroy(shard)
        getattr(self, "_shard_keep", {}).pop(shard.value, None)

    # ---- one trace sharded over ranks, rows supplied by the caller (lig_shard_rows_*)
    def shard_rows_begin(self, kinds_all, local_msgs, rank, world, comm, on_device=False, encoding_seed=None, generated_at=0,
                         public_args=None, dense_rands_per_row=None):
        """kinds_all: the kinds of ALL committed rows; local_msgs: this rank's rows (lig_shard_rows_plan), (rows_local, k, 8) uint32"""
        kinds = np.ascontiguousarray(kinds_all, dtype=np.uint8)
        job = RowsJob()
        job.rows = len(kinds)
        job.kinds = kinds.ctypes.data if len(kinds) else None
        keep = (kinds,)
        if on_device:
            job.msgs = local_msgs.value if hasattr(local_msgs, "value") else int(local_msgs)
        else:
            local_msgs = np.ascontiguousarray(local_msgs, dtype=np.uint32)
            job.msgs = local_msgs.ctypes.data if local_msgs.size else None
            keep += (local_msgs,)
        job.msgs_on_device = int(bool(on_device))
        es = bytes(range(32)) if encoding_seed is None else bytes(encoding_seed)
        for i in range(32):
            job.encoding_seed[i] = es[i]
            job.program_hash[i] = 0
        job.generated_at = generated_at
        job.version = b"1.5.0"
        job.set_public_args(public_args)
        if dense_rands_per_row is not None:
            dr = np.ascontiguousarray(dense_rands_per_row, dtype=np.uint32)
            job.dense_rands_per_row = dr.ctypes.data if len(dr) else None
            keep += (dr,)
        t = C.c_void_p()
        self.check(self.L.lig_shard_rows_begin(self.h, C.byref(job), rank, world, C.byref(comm), C.byref(t)))
        # host rows are read asynchronously (uploader thread) until lig_shard_rows_commit returns: the arrays live as long as the shard
        if not hasattr(self, "_shard_keep"):
            self._shard_keep = {}
        self._shard_keep[t.value] = keep
        return t

    def shard_rows_restart(self, shard, local_msgs, on_device=False):
        """the next trace of the same shape (after shard_rows_prove of the previous one)"""
        if on_device:
            ptr, keep = (local_msgs if hasattr(local_msgs, "value") else C.c_void_p(int(local_msgs))), ()
        else:
            local_msgs = np.ascontiguousarray(local_msgs, dtype=np.uint32)
            ptr, keep = C.c_void_p(local_msgs.ctypes.data if local_msgs.size else None), (local_msgs,)
        if hasattr(self, "_shard_keep"):
            self._shard_keep[shard.value] = keep
        self.check(self.L.lig_shard_rows_restart(shard, ptr, int(bool(on_device))))

    def shard_rows_commit(self, shard):
        root, seed = np.zeros(32, dtype=np.uint8), np.zeros(32, dtype=np.uint8)
        self.check(self.L.lig_shard_rows_commit(shard, _hptr(root), _hptr(seed)))
        return root.tobytes(), seed.tobytes()

    def shard_rows_prove(self, shard, local_rands, const_sum, on_device=False):
        proof, ln, info = C.POINTER(C.c_uint8)(), C.c_size_t(), ProofInfo()
        cs = np.frombuffer(bytes(const_sum), dtype=np.uint8).copy() if const_sum is not None else None
        if local_rands is None:
            rp = None
        elif on_device:
            rp = local_rands
        else:
            local_rands = np.ascontiguousarray(local_rands, dtype=np.uint32)
            rp = C.c_void_p(local_rands.ctypes.data if local_rands.size else None)
        self.check(self.L.lig_shard_rows_prove(shard, rp, int(bool(on_device)), _hptr(cs), C.byref(proof), C.byref(ln), C.byref(info)))
        return C.string_at(proof, ln.value), info

    def synth_prepare(self, n_linear, n_quad=0, synth_seed=1, generated_at=0, encoding_seed=None):
        import hashlib
        job = SynthJob()
        job.n_linear, job.n_quad, job.generated_at = n_linear, n_quad, generated_at
        es = bytes(range(32)) if encoding_seed is None else bytes(encoding_seed)
        wk = hashlib.sha256(b"lig-synth" + int(synth_seed).to_bytes(8, "little")).digest()
        for i in range(32):
            job.encoding_seed[i] = es[i]
            job.witness_key[i] = wk[i]
            job.program_hash[i] = 0
        job.version = b"1.5.0"
        return self.synth_prepare_job(job)

    def synth_prepare_job(self, job):
        """prepare from a SynthJob built by make_job (e.g. with a batch program attached, tests/batch_prog.py)"""
        t = C.c_void_p()
        self.check(self.L.lig_synth_prepare(self.h, C.byref(job), C.byref(t)))
        return t

    def synth_prove(self, trace, copy=True):
        """-> (proof bytes, ProofInfo); copy=False returns (address, length) of the trace-owned pinned buffer"""
        proof, ln, info = C.POINTER(C.c_uint8)(), C.c_size_t(), ProofInfo()
        self.check(self.L.lig_synth_prove(trace, C.byref(proof), C.byref(ln), C.byref(info)))
        if not copy:
            return (C.addressof(proof.contents), ln.value), info
        return C.string_at(proof, ln.value), info

    def synth_verify(self, job, const_sum, proof):
        """-> VerifyInfo (accept = 1 iff the reference's seven verifier predicates hold).  const_sum=None: the verifier
        derives the constant of the linear test from the public statement (the sound mode)"""
        info = VerifyInfo()
        cs = np.frombuffer(bytes(const_sum), dtype=np.uint8).copy() if const_sum is not None else None
        pb = np.frombuffer(bytes(proof), dtype=np.uint8).copy()
        self.check(self.L.lig_synth_verify(self.h, C.byref(job), _hptr(cs), _hptr(pb), len(pb), C.byref(info)))
        return info

    def verify_release(self):
        self.check(self.L.lig_verify_release(self.h))

    # ---- the same prover over rows supplied by the caller (lig_rows_*)
    def rows_begin(self, kinds, msgs, on_device=False, encoding_seed=None, generated_at=0, public_args=None, program_hash=None,
                   dense_rands_per_row=None, elem_bytes=None):
        """kinds: uint8 array (ROW_KINDS | ROW_DRAW_PAD); msgs: device pointer (on_device) or a (rows, k, 8) uint32 host array.
        -> (trace, keepalive); keep `keepalive` referenced until rows_commit has returned"""
        kinds = np.ascontiguousarray(kinds, dtype=np.uint8)
        job = RowsJob()
        job.rows = len(kinds)
        job.kinds = kinds.ctypes.data if len(kinds) else None
        if on_device:
            job.msgs, keep = (msgs.value if hasattr(msgs, "value") else int(msgs)), (kinds,)
        elif elem_bytes is not None:              # narrow format: msgs is the packed byte string (pack_rows)
            msgs = np.frombuffer(bytes(msgs), dtype=np.uint8).copy() if not isinstance(msgs, np.ndarray) else np.ascontiguousarray(msgs, dtype=np.uint8)
            job.msgs, keep = (msgs.ctypes.data if msgs.size else None), (kinds, msgs)
        else:
            msgs = np.ascontiguousarray(msgs, dtype=np.uint32)
            job.msgs, keep = (msgs.ctypes.data if msgs.size else None), (kinds, msgs)
        if elem_bytes is not None:
            eb = np.ascontiguousarray(elem_bytes, dtype=np.uint8)
            job.elem_bytes = eb.ctypes.data if len(eb) else None
            keep = keep + (eb,)
        job.msgs_on_device = int(bool(on_device))
        es = bytes(range(32)) if encoding_seed is None else bytes(encoding_seed)
        ph = bytes(32) if program_hash is None else bytes(program_hash)
        for i in range(32):
            job.encoding_seed[i] = es[i]
            job.program_hash[i] = ph[i]
        job.generated_at = generated_at
        job.version = b"1.5.0"
        job.set_public_args(public_args)
        if dense_rands_per_row is not None:
            dr = np.ascontiguousarray(dense_rands_per_row, dtype=np.uint32)
            job.dense_rands_per_row = dr.ctypes.data if len(dr) else None
            keep = keep + (dr,)
        t = C.c_void_p()
        self.check(self.L.lig_rows_begin(self.h, C.byref(job), C.byref(t)))
        return t, keep + (job,)

    def rows_restart(self, trace, msgs, on_device=False):
        p = msgs if on_device else C.c_void_p(msgs if isinstance(msgs, int) else msgs.ctypes.data)
        self.check(self.L.lig_rows_restart(trace, p, int(bool(on_device))))

    def rows_commit(self, trace):
        root, seed = np.zeros(32, dtype=np.uint8), np.zeros(32, dtype=np.uint8)
        self.check(self.L.lig_rows_commit(trace, _hptr(root), _hptr(seed)))
        return root.tobytes(), seed.tobytes()

    def rows_prove(self, trace, rands, const_sum, on_device=False, copy=True):
        proof, ln, info = C.POINTER(C.c_uint8)(), C.c_size_t(), ProofInfo()
        cs = np.frombuffer(bytes(const_sum), dtype=np.uint8).copy() if const_sum is not None else None
        if rands is None:
            rp = None
        elif on_device:
            rp = rands
        elif isinstance(rands, int):              # a raw host address (e.g. a pinned torch tensor's data_ptr()): no copy, no conversion
            rp = C.c_void_p(rands)
        else:
            rands = np.ascontiguousarray(rands, dtype=np.uint32)
            rp = C.c_void_p(rands.ctypes.data if rands.size else None)
        self.check(self.L.lig_rows_prove(trace, rp, int(bool(on_device)), _hptr(cs), C.byref(proof), C.byref(ln), C.byref(info)))
        if not copy:
            return (C.addressof(proof.contents), ln.value), info
        return C.string_at(proof, ln.value), info

    def host_alloc(self, nbytes):
        """page-locked host memory (lig_host_alloc) as a uint8 numpy view; keep the returned (array, ptr) until host_free(ptr)"""
        p = C.c_void_p()
        self.check(self.L.lig_host_alloc(self.h, nbytes, C.byref(p)))
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(1, nbytes),))[:nbytes]
        return arr, p

    def host_free(self, p):
        self.check(self.L.lig_host_free(self.h, p))

    def upload_health(self):
        """lig_upload_health -> (retries, unsettled): calls that re-made a timed-out upload / abandoned transfers still pending"""
        r, u = C.c_uint32(0), C.c_uint32(0)
        self.check(self.L.lig_upload_health(self.h, C.byref(r), C.byref(u)))
        return r.value, u.value

    def rows_push_rands(self, trace, first_row, n_rows, host_ptr):
        """lig_rows_push_rands: randomness rows [first_row, first_row + n_rows), host memory valid until rows_prove returns"""
        self.check(self.L.lig_rows_push_rands(trace, first_row, n_rows, C.c_void_p(host_ptr)))

    def rows_push_rands_sparse(self, trace, first_row, present, host_ptr):
        """lig_rows_push_rands_sparse: `present` (uint8 per row) says which of the rows have a randomness row at host_ptr (packed)"""
        present = np.ascontiguousarray(present, dtype=np.uint8)
        self.check(self.L.lig_rows_push_rands_sparse(trace, first_row, len(present), _hptr(present), C.c_void_p(host_ptr)))

    def rows_verify_begin(self, kinds, proof, public_args=None):
        """-> (vtrace or None, stage1_seed bytes, VerifyInfo): the verifier's first half for a rows job (kinds + public data)"""
        kinds = np.ascontiguousarray(kinds, dtype=np.uint8)
        job = RowsJob()
        job.rows = len(kinds)
        job.kinds = kinds.ctypes.data if len(kinds) else None
        job.set_public_args(public_args)
        pb = np.frombuffer(bytes(proof), dtype=np.uint8).copy()
        vt, seed, info = C.c_void_p(), np.zeros(32, dtype=np.uint8), VerifyInfo()
        self.check(self.L.lig_rows_verify_begin(self.h, C.byref(job), _hptr(pb), len(pb), C.byref(vt), _hptr(seed), C.byref(info)))
        return (vt if vt.value else None), seed.tobytes(), info

    def rows_verify_finish(self, vtrace, rands, const_sum, on_device=False):
        info = VerifyInfo()
        cs = np.frombuffer(bytes(const_sum), dtype=np.uint8).copy()
        if on_device:
            rp = rands
        else:
            rands = np.ascontiguousarray(rands, dtype=np.uint32)
            rp = C.c_void_p(rands.ctypes.data if rands.size else None)
        self.check(self.L.lig_rows_verify_finish(vtrace, rp, int(bool(on_device)), _hptr(cs), C.byref(info)))
        return info

    def vtrace_destroy(self, vtrace):
        """give up a verification between begin and finish (finish frees the trace itself)"""
        self.L.lig_vtrace_destroy(vtrace)

    def rng_fill_rows(self, key, first_elem, per_row, out):
        k = np.frombuffer(bytes(key), dtype=np.uint8).copy()
        pr = np.ascontiguousarray(per_row, dtype=np.uint32)
        self.check(self.L.lig_rng_fill_rows(self.h, _hptr(k), first_elem, _hptr(pr), len(pr), out))

    def trace_destroy(self, trace):
        self.L.lig_trace_destroy(trace)

    def profile_enable(self, on=True):
        self.check(self.L.lig_profile_enable(self.h, int(on)))

    def profile_read(self):
        a, b, ms = C.c_uint64(), C.c_uint64(), C.c_double()
        self.check(self.L.lig_profile_read(self.h, C.byref(a), C.byref(b), C.byref(ms)))
        return a.value, b.value, ms.value

    def profile_read_launches(self, rows_in_launch):
        """lig_profile_read_launches -> (launches of exactly that many rows, their summed ms)"""
        a, ms = C.c_uint64(), C.c_double()
        self.check(self.L.lig_profile_read_launches(self.h, rows_in_launch, C.byref(a), C.byref(ms)))
        return a.value, ms.value

    def rng_fill(self, key, first_elem, out, count):
        k = np.frombuffer(bytes(key), dtype=np.uint8).copy()
        self.check(self.L.lig_rng_fill(self.h, _hptr(k), first_elem, out, count))
