"""The C++ mirror of the reference executor (include/lig_hip_context.hpp) compiles against a driver written
like nonbatch_stage1_context (CPU test: syntax + link), and on the GPU produces the oracle's leaves."""
import json
import os
import subprocess

import numpy as np
import pytest

import hip_lib
import oracle_lib as ol

ROOT = hip_lib.ROOT
SRC = os.path.join(ROOT, "tests", "cpp", "stage1_rows.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "stage1_rows")


def build_exe():
    mod = hip_lib.load()
    if not os.path.exists(mod.LIB_PATH):
        mod.build()
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), SRC,
                           "-L" + os.path.dirname(mod.LIB_PATH), "-llig_hip",
                           "-Wl,-rpath," + os.path.dirname(mod.LIB_PATH), "-o", EXE])
    return EXE


def test_hip_context_header_compiles_and_links():
    subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-fsyntax-only", SRC])
    assert os.path.exists(build_exe())


@pytest.mark.gpu
def test_hip_context_stage1_rows_match_oracle():
    exe = build_exe()
    rows, k, n = 3, 512, 2048
    out = subprocess.check_output([exe, str(rows)]).decode().strip()
    msgs = np.zeros((rows, k, 8), dtype=np.uint32)
    for r in range(rows):
        vals = 1000003 * (r + 1) + np.arange(k, dtype=np.uint64)
        msgs[r, :, 0] = (vals & 0xFFFFFFFF).astype(np.uint32)
        msgs[r, :, 1] = (vals >> 32).astype(np.uint32)
    cws = ol.Ctx(320, k, n).encode_rows(msgs, threads=2)
    assert out == ol.colsha(cws).tobytes().hex()


PSRC = os.path.join(ROOT, "tests", "cpp", "powmod_kat.cpp")
PEXE = os.path.join(ROOT, "tests", "cpp", "powmod_kat")


def build_powmod_exe():
    mod = hip_lib.load()
    if not os.path.exists(mod.LIB_PATH):
        mod.build()
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), PSRC,
                           "-L" + os.path.dirname(mod.LIB_PATH), "-llig_hip",
                           "-Wl,-rpath," + os.path.dirname(mod.LIB_PATH), "-o", PEXE])
    return PEXE


def test_hip_context_powmod_surface_compiles():
    assert os.path.exists(build_powmod_exe())


@pytest.mark.gpu
def test_hip_context_powmod_like_reference_fixture():
    """powmod_init / powmod_set_base / bind_powmod / EltwisePowMod / EltwisePowAddMod (wgpu.hpp:84-107) driven like
    tests/webgpu/test_powmod.cpp's generator case"""
    lines = subprocess.check_output([build_powmod_exe()]).decode().strip().splitlines()
    assert lines[0] == "misuse_throws 1"
    for ln in lines[1:]:
        parts = ln.split()
        i = int(parts[0])
        got = int("".join(parts[1:]), 16)
        assert got == 2 * pow(7, i, ol.P) % ol.P


BSRC = os.path.join(ROOT, "tests", "cpp", "row_batcher_prog.cpp")
BEXE = os.path.join(ROOT, "tests", "cpp", "row_batcher_prog")


def build_batcher_exe():
    mod = hip_lib.load()
    if not os.path.exists(mod.LIB_PATH):
        mod.build()
    odir = os.path.join(ROOT, "oracle")
    ol.build()
    subprocess.check_call(["g++", "-std=c++17", "-O1", BSRC, "-L" + os.path.dirname(mod.LIB_PATH), "-llig_hip", "-L" + odir, "-llig_oracle",
                           "-Wl,-rpath," + os.path.dirname(mod.LIB_PATH), "-Wl,-rpath," + odir, "-o", BEXE])
    return BEXE


def test_row_batcher_shim_compiles_and_links():
    assert os.path.exists(build_batcher_exe())


@pytest.mark.gpu
def test_row_batcher_shim_ships_narrow_rows():
    """hip_proof_meta::narrow_rows: a small-witness trace (lo_job.witness_bits = 64) goes through the shim's callbacks as full rows;
    the shim uploads every row that fits as l x 8 bytes and lets the library draw its pads -- same envelope as the oracle's"""
    out = subprocess.check_output([build_batcher_exe(), "320", "512", "2048", str(320 * 40 + 7), "700", "64"]).decode()
    assert out.startswith("equal 1 "), out


@pytest.mark.gpu
@pytest.mark.parametrize("l,k,n,n_lin,n_quad", [(320, 512, 2048, 2000, 700), (320, 512, 2048, 320 * 530 + 1, 0), (8000, 8192, 32768, 20000, 8001),
                                                (8000, 8192, 32768, 1 << 22, 0)])       # 525 rows: the staging grows, pass 2 pushes randomness rows twice before prove
def test_row_batcher_shim_gives_the_oracle_envelope(l, k, n, n_lin, n_quad):
    """include/lig_hip_row_batcher.hpp driven like the reference's stage contexts (two passes of per-row callbacks) produces
    the envelope of the oracle's reference-structured prover, public arguments included"""
    out = subprocess.check_output([build_batcher_exe(), str(l), str(k), str(n), str(n_lin), str(n_quad)]).decode()
    assert out.startswith("equal 1 "), out


@pytest.mark.gpu
def test_row_batcher_shim_without_the_uploader_thread():
    """with LIG_UPLOAD_MODE=1 (or on a device without stream memory operations) lig_rows_push_rands has no uploader thread to hand
    the rows to -- it copies them on the copy stream and waits, the shim's two passes still end in the oracle's envelope"""
    out = subprocess.check_output([build_batcher_exe(), "320", "512", "2048", str(320 * 530 + 1), "330"], env=dict(os.environ, LIG_UPLOAD_MODE="1")).decode()
    assert out.startswith("equal 1 "), out


BBSRC = os.path.join(ROOT, "tests", "cpp", "row_batcher_batch_prog.cpp")
BBEXE = os.path.join(ROOT, "tests", "cpp", "row_batcher_batch_prog")


def build_batch_batcher_exe():
    mod = hip_lib.load()
    if not os.path.exists(mod.LIB_PATH):
        mod.build()
    odir = os.path.join(ROOT, "oracle")
    ol.build()
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-I" + os.path.join(ROOT, "include"), BBSRC, "-L" + os.path.dirname(mod.LIB_PATH), "-llig_hip",
                           "-L" + odir, "-llig_oracle", "-Wl,-rpath," + os.path.dirname(mod.LIB_PATH), "-Wl,-rpath," + odir, "-o", BBEXE])
    return BBEXE


def test_row_batcher_batch_hooks_compile_and_link():
    assert os.path.exists(build_batch_batcher_exe())


@pytest.mark.gpu
@pytest.mark.parametrize("n_lin,n_quad,with_bits,k", [(0, 0, 0, 512), (900, 330, 1, 512), (320 * 3, 0, 1, 512), (20000, 8001, 0, 8192)])
def test_row_batcher_batch_hooks_give_the_oracle_envelope(n_lin, n_quad, with_bits, k):
    """on_batch_init / _bit / _equal / _quadratic of the shim, raised by the vbn254fr layer itself over two runs of a guest:
    the init pads are drawn at the row's encoding-stream position and written INTO the variable (they flow into the product,
    quotient, copy and bit rows derived from it), and the envelope equals the oracle's for batch program + synthetic stream
    (the shim used to commit init rows without their pads)"""
    out = subprocess.check_output([build_batch_batcher_exe(), str(n_lin), str(n_quad), str(with_bits), str(k)]).decode()
    assert out.startswith("equal 1 "), out


RBSRC = os.path.join(ROOT, "tests", "cpp", "row_batcher_bench.cpp")
RBEXE = os.path.join(ROOT, "tests", "cpp", "row_batcher_bench")


def build_batcher_bench_exe():
    mod = hip_lib.load()
    if not os.path.exists(mod.LIB_PATH):
        mod.build()
    odir = os.path.join(ROOT, "oracle")
    ol.build()
    subprocess.check_call(["g++", "-std=c++17", "-O2", RBSRC, "-L" + os.path.dirname(mod.LIB_PATH), "-llig_hip", "-L" + odir, "-llig_oracle", "-lpthread",
                           "-Wl,-rpath," + os.path.dirname(mod.LIB_PATH), "-Wl,-rpath," + odir, "-o", RBEXE])
    return RBEXE


def test_row_batcher_bench_compiles_and_links():
    assert os.path.exists(build_batcher_bench_exe())


@pytest.mark.gpu
def test_row_batcher_bench_at_2p20_matches_the_oracle_on_every_path():
    """tests/cpp/row_batcher_bench.cpp at 2^20 constraints with the oracle check on: copying callbacks, the zero-copy slots
    (next_slot / commit_slot), re-use of one batcher for several proofs (reset) and two batchers on two contexts all self-check,
    and the first envelope equals the oracle's"""
    out = json.loads(subprocess.check_output([build_batcher_bench_exe(), "20", "2", "1"]).decode().strip().splitlines()[-1])
    assert out["ok"] == 1 and out["checked_against_oracle"] == 1 and out["rows"] == 135


@pytest.mark.gpu
@pytest.mark.parametrize("slicing", ["declared", "upstream"])
def test_vbn254fr_layer_under_both_slicing_semantics_gives_the_oracle_envelope(slicing):
    """hip_context::set_upstream_slice_compat: the vbn254fr layer + the shim's on_batch_init run a guest under buffer_view's declared
    slicing and under upstream's definition of slice_bytes (src/webgpu/buffer_view.cpp:91-95: parameters swapped -- pads land in
    variable 0, write_buffer_clear wipes the slab); the program the layer RECORDS says which (LIG_BOP_UPSTREAM_COMPAT,
    LIG_BOP_F_WRITE_LIMBS) and the oracle's interpreter of that program gives the same envelope"""
    out = subprocess.check_output([build_batch_batcher_exe(), "900", "330", "0", "512", slicing]).decode()
    assert out.startswith("equal 1 "), out


SBSRC = os.path.join(ROOT, "tests", "cpp", "sharded_batcher_prog.cpp")
SBEXE = os.path.join(ROOT, "tests", "cpp", "sharded_batcher_prog")


def build_sharded_batcher_exe():
    mod = hip_lib.load()
    if not os.path.exists(mod.LIB_PATH):
        mod.build()
    odir = os.path.join(ROOT, "oracle")
    ol.build()
    subprocess.check_call(["g++", "-std=c++17", "-O1", SBSRC, "-L" + os.path.dirname(mod.LIB_PATH), "-llig_hip", "-L" + odir, "-llig_oracle",
                           "-Wl,-rpath," + os.path.dirname(mod.LIB_PATH), "-Wl,-rpath," + odir, "-o", SBEXE])
    return SBEXE


def test_sharded_row_batcher_compiles_and_links():
    assert os.path.exists(build_sharded_batcher_exe())


@pytest.mark.gpu
@pytest.mark.parametrize("world,n_lin,n_quad", [(2, 2000, 900), (4, 320 * 1100 + 3, 330)])
def test_sharded_row_batcher_gives_the_oracle_envelope_on_every_rank(world, n_lin, n_quad):
    """hip_row_batcher::shard_over: the same guest on `world` ranks (processes on the one GPU, comm_ipc), each keeping the rows of its
    chunks; every rank's envelope is the oracle's -- the backend side of configs[4] (one guest trace on the GPUs of a node)"""
    exe = build_sharded_batcher_exe()
    import multirank as mr
    name = "/lig_sb_" + mr.fresh_tag()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = mr.run_ranks(lambda r: [exe, str(r), str(world), name, str(n_lin), str(n_quad)], world, env, timeout=300)
    for r, (o, e) in enumerate(outs):
        assert ("rank %d: equal 1 " % r) in o, (o, e[-2000:])


S3SRC = os.path.join(ROOT, "tests", "cpp", "stage123_rows.cpp")
S3EXE = os.path.join(ROOT, "tests", "cpp", "stage123_rows")


def build_stage123_exe():
    mod = hip_lib.load()
    if not os.path.exists(mod.LIB_PATH):
        mod.build()
    odir = os.path.join(ROOT, "oracle")
    ol.build()
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), S3SRC, "-L" + os.path.dirname(mod.LIB_PATH), "-llig_hip",
                           "-L" + odir, "-llig_oracle", "-Wl,-rpath," + os.path.dirname(mod.LIB_PATH), "-Wl,-rpath," + odir, "-o", S3EXE])
    return S3EXE


def test_three_stage_per_row_driver_compiles_and_links():
    assert os.path.exists(build_stage123_exe())


@pytest.mark.gpu
@pytest.mark.parametrize("constraints,n_quad,k,deferred", [
    (-700, 330, 512, 0),            # eager: every executor call is a launch (the literal three-line drop-in)
    (-700, 330, 512, 512),          # deferred rows: linear rows batched, the triples' check_quadratic falls back (flush + eager)
    (-(320 * 40 + 7), 0, 512, 5),   # a 5-row ring: flushes in the middle of every stage, staging halves alternate
    (-(320 * 300), 0, 512, 128),    # stage 3's 256-slot sample ring wraps (sync_sample_to_host in the middle of the rows)
    (-1, 0, 512, 512),              # one constraint: one row + the masks
    (17, 0, 8192, 512),             # production geometry, 17 rows + masks
])
def test_three_stage_per_row_driver_gives_the_oracle_envelope(constraints, n_quad, k, deferred):
    """tests/cpp/stage123_rows.cpp: miniatures of nonbatch_stage{1,2,3}_context (nonbatch_context.hpp:445-471,654-780,924-970) written
    against the Executor template parameter drive ligero::hip_context row by row through a complete proof -- eagerly and in the
    deferred row mode (set_deferred_rows: the same calls recorded and flushed through the batched entry points); the envelope
    assembled from what the executor returned equals the oracle's reference-structured prover's, the self-check predicates hold"""
    out = json.loads(subprocess.check_output([build_stage123_exe(), str(constraints), str(deferred), str(n_quad), str(k)]).decode().strip().splitlines()[-1])
    assert out["equal"] == 1, out
