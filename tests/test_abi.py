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


def test_binding_fails_loudly_without_library(tmp_path, monkeypatch):
    mod = hip_lib.load()
    monkeypatch.setattr(mod, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        mod.load_library()
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("missing library must raise")
