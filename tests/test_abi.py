"""CPU-only: the C-ABI library loads and exports every symbol include/lig_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import hip_lib

ROOT = hip_lib.ROOT


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "lig_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(lig_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_exported():
    mod = hip_lib.load()
    if not os.path.exists(mod.LIB_PATH):
        mod.build()
    lib = ctypes.CDLL(mod.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), "missing export %s" % s
    assert sorted(mod.EXPORTS) == syms, "python binding list out of sync with the header"


def test_pure_host_entry_points():
    mod = hip_lib.load()
    if not os.path.exists(mod.LIB_PATH):
        mod.build()
    L = mod.load_library()
    assert L.lig_version().startswith(b"lig_hip")
    assert L.lig_sha_state_bytes(32768) == 32768 * 64
    assert L.lig_merkle_nodes(32768) == 65535 and L.lig_merkle_nodes(5) == 15 and L.lig_merkle_nodes(1) == 1
    # argument validation happens before any device call
    h = ctypes.c_void_p()
    assert L.lig_ctx_create(ctypes.byref(h), 0, 320, 500, 2000) == -1      # k not a power of two
    assert L.lig_ctx_create(ctypes.byref(h), 0, 320, 256, 1024) == -1      # k < 512 (reference: log2N >= 9)
    assert L.lig_ctx_create(ctypes.byref(h), 0, 600, 512, 2048) == -1      # l > k
    assert L.lig_sync(None) == -1


def test_struct_layouts_of_the_bindings_match_the_library():
    """lig_abi_sizes: the ctypes structs (and tests/batch_prog.py's BatchOp) have the library's sizeof; load_library refuses a mismatch"""
    import batch_prog
    mod = hip_lib.load()
    L = mod.load_library()                       # raises on a mismatch of the five structs it binds
    sizes = (ctypes.c_uint32 * 6)()
    L.lig_abi_sizes(sizes)
    assert sizes[0] == ctypes.sizeof(batch_prog.BatchOp) == 32
    assert sizes[5] == ctypes.sizeof(mod.Comm) == 8 * ctypes.sizeof(ctypes.c_void_p)
    orig = mod.Comm
    class Short(ctypes.Structure):
        _fields_ = orig._fields_[:-2]
    mod.Comm = Short
    try:
        mod.load_library()
    except RuntimeError as e:
        assert "lig_comm" in str(e)
    else:
        raise AssertionError("a short lig_comm must be refused")
    finally:
        mod.Comm = orig


def test_binding_fails_loudly_without_library(tmp_path, monkeypatch):
    mod = hip_lib.load()
    monkeypatch.setattr(mod, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        mod.load_library()
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("missing library must raise")


def test_proof_file_gzip_framing_roundtrip_with_python_gzip():
    """host-only helpers (no GPU): gzip(level 6) member of an envelope, as webgpu_prover.cpp:437-457 writes the proof file;
    interoperable with any gzip implementation in both directions"""
    import ctypes as C
    import gzip
    mod = hip_lib.load()
    if not os.path.exists(mod.LIB_PATH):
        mod.build()
    lib = mod.load_library()
    env = bytes(range(256)) * 300 + b"\x0a\x10" + bytes(5000)
    cap = lib.lig_proof_gzip_bound(len(env))
    out = (C.c_uint8 * cap)()
    n = C.c_size_t()
    src = (C.c_uint8 * len(env)).from_buffer_copy(env)
    assert lib.lig_proof_gzip(src, len(env), out, cap, C.byref(n)) == 0
    gz = bytes(out[:n.value])
    assert gz[:2] == b"\x1f\x8b" and gzip.decompress(gz) == env
    assert lib.lig_proof_gunzip_size(out, n.value) == len(env)
    # the other direction: a member written by another gzip
    theirs = gzip.compress(env, compresslevel=6)
    tsrc = (C.c_uint8 * len(theirs)).from_buffer_copy(theirs)
    back = (C.c_uint8 * len(env))()
    assert lib.lig_proof_gunzip(tsrc, len(theirs), back, len(env), C.byref(n)) == 0 and bytes(back[:n.value]) == env
    # too small an output buffer / not gzip
    small = (C.c_uint8 * 10)()
    assert lib.lig_proof_gunzip(tsrc, len(theirs), small, 10, C.byref(n)) != 0
    assert lib.lig_proof_gunzip(src, len(env), back, len(env), C.byref(n)) != 0
    assert lib.lig_proof_gunzip_size(src, len(env)) == 0


def test_rccl_is_resolved_lazily_and_only_once():
    """liblig_hip.so has no link-time dependency on librccl: the host-only helpers work in a process that never touches
    RCCL, and lig_rccl_available reports which copy the first lig_rccl_* call would use -- the one already mapped into the
    process if there is one (torch brings its own), so that one process never holds two different RCCL builds"""
    import subprocess, sys
    lib = os.path.join(ROOT, "ligero-prover_amd", "liblig_hip.so")
    needed = subprocess.check_output(["readelf", "-d", lib]).decode()
    assert "librccl" not in needed, "librccl must not be a DT_NEEDED entry"
    code = ("import sys; sys.path.insert(0, %r); import hip_lib; m = hip_lib.load(); ok, path, ver = m.rccl_available(); print(ok, path, ver)"
            % os.path.join(ROOT, "tests"))
    plain = subprocess.check_output([sys.executable, "-c", code]).decode().split()
    with_torch = subprocess.check_output([sys.executable, "-c", "import torch; " + code]).decode().split()
    assert plain[0] == "True" and with_torch[0] == "True"
    assert "torch" in with_torch[1], "inside a torch process the copy torch mapped must win: %r" % (with_torch,)
