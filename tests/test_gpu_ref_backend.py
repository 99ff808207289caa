"""GPU: the HIP path against vectors produced by the REFERENCE's own GMP-side code (tests/golden/ref_field.json, ref_rows_*.npz; made by
tests/golden/make_ref_backend.py from oracle/_ref/libref_backend.so -- the reference's bn254.cpp, finite_field_gmp.hpp, csprng.hpp,
mpz_vector.hpp, witness_manager.hpp and core.hpp compiled in the build container):

* the device AES-256-CTR field sampler (csrc/aes.hip) == bn254_gmp::generate_random over mpz_random_engine;
* the executor's eltwise products / quotients / sums == bn254_gmp::mulmod / divmod / addmod / submod;
* the rows entry (lig_rows_begin / _commit / _prove) on the row stream the reference's witness_manager + ligetron_backend emitted for
  tests/i32_add.wat (configs[0]) and for a multiply-add guest: root, stage-1 seed and envelope equal what the oracle computes from the
  same stream -- with the rows' own pads, and with the pads wiped and drawn by the library (LIG_ROW_DRAW_PAD: the library's sampler at
  the positions the commit order implies must then reproduce pad_encoding_random).
"""
import hashlib
import json
import os

import numpy as np
import pytest

import hip_lib
import oracle_lib as ol

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
P = ol.P


@pytest.fixture(scope="module")
def amd():
    return hip_lib.load()


def field_gold():
    with open(os.path.join(GOLD, "ref_field.json")) as f:
        return json.load(f)


def load_rows(name):
    z = np.load(os.path.join(GOLD, "ref_rows_%s.npz" % name))
    return dict(kinds=z["kinds"], vals=z["vals"], rands=z["rands"], constsum=z["constsum"].tobytes(), meta=json.loads(str(z["meta"])))


def test_device_sampler_equals_reference_generate_random(amd):
    c = amd.Context(320, 512, 2048)
    try:
        for s in field_gold()["sampler"]:
            key, n = bytes.fromhex(s["key"]), s["count"]
            d = c.malloc(n * 32)
            c.rng_fill(key, 0, d, n)
            got = c.download(d, (n, 8))
            assert hashlib.sha256(got.tobytes()).hexdigest() == s["sha256_of_all"]
            for i, hx in s["elements"].items():
                assert got[int(i)].tobytes().hex() == hx
            c.rng_fill(key, 1020, d, 8)                                  # any start position (the refill boundary at element 1024 inside)
            assert np.array_equal(c.download(d, (8, 8)), got[1020:1028])
            c.free(d)
    finally:
        c.close()


def test_device_field_operations_equal_the_reference(amd):
    ops = field_gold()["ops"]
    c = amd.Context(320, 512, 2048)
    try:
        for name, op in (("mulmod", "MUL"), ("divmod", "DIV"), ("addmod", "ADD"), ("submod", "SUB")):
            rows = [[int(x, 16) for x in r] for r in ops[name]]
            a, b = ol.to_limbs([r[0] for r in rows]), ol.to_limbs([r[1] for r in rows])
            da, db, do = c.upload(a), c.upload(b), c.malloc(len(rows) * 32)
            c.eltwise(op, da, db, do, len(rows))
            assert ol.from_limbs(c.download(do, (len(rows), 8))) == [r[2] for r in rows], name
            for p in (da, db, do):
                c.free(p)
    finally:
        c.close()


@pytest.mark.parametrize("name", ["i32_add_320", "i32_add_8000", "mul_add_320"])
@pytest.mark.parametrize("pads", ["own", "library"])
def test_rows_entry_on_the_reference_backends_row_stream(amd, name, pads):
    """configs[0]: the row stream of tests/i32_add.wat as the reference's constraint backend emits it (13 rows at l = 320, 4 at l = 8000)
    through lig_rows_*: same root / seed / envelope as the oracle over the same rows, randomness rows and constant sum from the
    reference's stage-2 replay"""
    f = load_rows(name)
    m = f["meta"]
    l, k, n, t = m["l"], m["k"], m["n"], m["t"]
    key = bytes.fromhex(m["encoding_seed"])
    kinds, vals = f["kinds"].copy(), f["vals"].copy()
    if pads == "library":
        kinds |= amd.ROW_DRAW_PAD
        vals[:, l:] = 0xDEADBEEF
    masks = ol.form_masks(key, len(kinds) * (k - l), l, k)
    want = ol.prove_rows(l, k, n, t, f["kinds"], f["vals"], *masks, f["rands"], f["constsum"], generated_at=m["generated_at"])
    assert hashlib.sha256(want["proof"]).hexdigest() == m["oracle_proof_sha256"]          # (the fixture's own record of the same run)
    c = amd.Context(l, k, n)
    try:
        tr, keep = c.rows_begin(kinds, vals, encoding_seed=key, generated_at=m["generated_at"])
        root, seed1 = c.rows_commit(tr)
        assert root.hex() == m["oracle_root"] and seed1.hex() == m["oracle_stage1_seed"]
        proof, info = c.rows_prove(tr, f["rands"], f["constsum"])
        assert [info.valid_code, info.valid_linear, info.valid_quad] == [1, 1, 1]
        assert proof == want["proof"]
        c.trace_destroy(tr)
        # the constant the library derives when none is given (minus the sum of the inner products) is witness_manager::constsum()
        tr, keep = c.rows_begin(kinds, vals, encoding_seed=key, generated_at=m["generated_at"])
        c.rows_commit(tr)
        proof3, info3 = c.rows_prove(tr, f["rands"], None)
        assert bytes(info3.const_sum) == f["constsum"] and proof3 == want["proof"]
        c.trace_destroy(tr)
        # the HIP verifier on the same public data: kinds + proof -> the stage-1 seed -> the reference's randomness rows and constant
        vt, vseed, vinfo = c.rows_verify_begin(f["kinds"], proof)
        assert vt is not None and vseed == seed1 and vinfo.parsed == 1 and vinfo.indices_match == 1
        v = c.rows_verify_finish(vt, f["rands"], f["constsum"])
        assert [v.valid_merkle, v.valid_code, v.valid_linear, v.valid_quad, v.code_equal, v.linear_equal, v.quad_equal, v.accept] == [1] * 8
    finally:
        c.close()


SHARDED_REF_WORKER = '''
import hashlib, importlib.util, json, os, sys
import numpy as np
root, name = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.path.join(root, "tests"))
def load(mod, rel):
    spec = importlib.util.spec_from_file_location(mod, os.path.join(root, "ligero-prover_amd", rel))
    m = importlib.util.module_from_spec(spec); sys.modules[mod] = m; spec.loader.exec_module(m); return m
pkg = load("ligero_prover_amd", "__init__.py")
dist = load("lig_dist", "dist.py")
z = np.load(os.path.join(root, "tests", "golden", "ref_rows_%s.npz" % name))
m = json.loads(str(z["meta"]))
g = dist.Group("gloo")
ctx = pkg.Context(m["l"], m["k"], m["n"], device=0)
kinds, vals, rands = z["kinds"], z["vals"], z["rands"]
rounds, b = pkg.shard_rows_plan(kinds, g.world)
mine = pkg.local_rows_of(b, g.rank, g.world)
comm = g.make_comm(pkg, ctx)
sh = ctx.shard_rows_begin(kinds, vals[mine] if len(mine) else np.zeros((0, m["k"], 8), np.uint32), g.rank, g.world, comm,
                          encoding_seed=bytes.fromhex(m["encoding_seed"]), generated_at=m["generated_at"])
root_, seed1 = ctx.shard_rows_commit(sh)
proof, info = ctx.shard_rows_prove(sh, rands[mine] if len(mine) else np.zeros((0, m["k"], 8), np.uint32), z["constsum"].tobytes())
ctx.shard_destroy(sh)
print(json.dumps({"rank": g.rank, "local": len(mine), "root": root_.hex(), "seed1": seed1.hex(), "sha": hashlib.sha256(proof).hexdigest(),
                  "valid": [info.valid_code, info.valid_linear, info.valid_quad]}))
g.close(); ctx.close()
'''


@pytest.mark.parametrize("name,world", [("i32_add_320", 2), ("mul_add_320", 4)])
def test_reference_row_stream_sharded_over_ranks(tmp_path, name, world):
    """the reference backend's row stream (configs[0]; real callback order: triples and linear rows interleaved) as ONE trace over W ranks
    (processes on the one GPU, comm_ipc): every rank's envelope is the one the oracle computes from the unsharded stream"""
    import multirank as mr
    m = load_rows(name)["meta"]
    script = tmp_path / "sharded_ref_worker.py"
    script.write_text(SHARDED_REF_WORKER)
    outs = [mr.last_json(o) for o, _ in mr.run_ranks(mr.python_argv(script, os.path.dirname(os.path.dirname(GOLD)), name), world, mr.rendezvous_env(world, "ipc"))]
    assert sorted(o["rank"] for o in outs) == list(range(world)) and sum(o["local"] for o in outs) == m["rows"]
    for o in outs:
        assert o["valid"] == [1, 1, 1] and o["root"] == m["oracle_root"] and o["seed1"] == m["oracle_stage1_seed"] and o["sha"] == m["oracle_proof_sha256"], o


def test_live_reference_row_former_feeds_the_hip_prover_at_configs1_scale(amd):
    """oracle/_ref/libref_backend.so travels with the snapshot (built in the build container from the reference's own sources): the reference's
    witness_manager + ligetron_backend form a 200-row stream at k = 8192 (400 000 multiply-add constraints with constants: 50 linear rows and
    50 quadratic triples in the real emission order) ON THIS BOX, the HIP rows entry commits to it, the reference's stage-2 replay -- keyed by
    the seed the HIP path produced -- delivers the randomness rows and constsum(), and the envelope must be the oracle's over the same stream.
    No fixture in between: reference row former -> HIP prover -> HIP verifier."""
    import ref_backend_lib as rb
    if rb.load() is None:
        pytest.skip("oracle/_ref/libref_backend.so is not present on this box")
    l, k, n, t, reps, gen = 8000, 8192, 32768, 192, 400000, 21
    key = bytes(range(32))
    g1 = rb.guest("mul_add", l, k, key, reps=reps)
    kinds, vals = g1["kinds"], g1["vals"]
    assert len(kinds) == 200 and np.bincount(kinds).tolist() == [50, 50, 50, 50] and not g1["rands"].any()
    c = amd.Context(l, k, n)
    try:
        tr, keep = c.rows_begin(kinds, vals, encoding_seed=key, generated_at=gen)
        root, seed1 = c.rows_commit(tr)
        g2 = rb.guest("mul_add", l, k, key, wit_key=seed1, reps=reps)            # the reference's own stage-2 run of the guest
        assert np.array_equal(g2["vals"], vals) and np.array_equal(g2["kinds"], kinds)
        proof, info = c.rows_prove(tr, g2["rands"], g2["constsum"])
        assert [info.valid_code, info.valid_linear, info.valid_quad] == [1, 1, 1]
        c.trace_destroy(tr)
        want = ol.prove_rows(l, k, n, t, kinds, vals, g1["mask_code"], g1["mask_lin"], g1["mask_quad"], g2["rands"], g2["constsum"], generated_at=gen, threads=8)
        assert want["root"] == root and want["stage1_seed"] == seed1 and want["valid"] == [1, 1, 1]
        assert proof == want["proof"]
        vt, vseed, vinfo = c.rows_verify_begin(kinds, proof)
        assert vt is not None and vseed == seed1
        v = c.rows_verify_finish(vt, g2["rands"], g2["constsum"])
        assert v.accept == 1
        # the same rows with the pads wiped and drawn by the library: the device sampler at the positions the emission order implies
        wiped = vals.copy()
        wiped[:, l:] = 0xA5A5A5A5
        tr, keep = c.rows_begin(kinds | amd.ROW_DRAW_PAD, wiped, encoding_seed=key, generated_at=gen)
        root2, seed2 = c.rows_commit(tr)
        assert root2 == root and seed2 == seed1
        c.trace_destroy(tr)
    finally:
        c.close()


@pytest.mark.parametrize("name", ["i32_add_320", "i32_add_8000", "mul_add_320"])
def test_reference_constraint_backend_drives_the_hip_backend_through_the_shim(name):
    """the binding a maintainer adds, compiled and run: oracle/_ref/libref_hip_guest.so = the reference's witness_manager + ligetron_backend
    (from /root/reference, built in the build container) whose linear / quadratic / mask callbacks export their rows straight into
    ligero::hip_row_batcher's slots (include/lig_hip_row_batcher.hpp), two runs of the guest around commit(), constsum() into prove() --
    all in one process with liblig_hip.so.  configs[0] (tests/i32_add.wat) end to end below the interpreter: root, seed, constant and
    envelope equal the fixture's (the oracle over the recorded stream)."""
    import ctypes as C
    so = os.path.join(os.path.dirname(os.path.dirname(GOLD)), "oracle", "_ref", "libref_hip_guest.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libref_hip_guest.so is not present on this box")
    f = load_rows(name)
    m = f["meta"]
    L = C.CDLL(so)
    L.ref_guest_prove_hip.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_char_p, C.c_int64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.POINTER(C.c_int * 3), C.c_char_p, C.c_size_t]
    root, seed1, cs = (np.zeros(32, dtype=np.uint8) for _ in range(3))
    cap = 64 << 20
    proof = np.zeros(cap, dtype=np.uint8)
    ln, rows, valid, err = C.c_size_t(), C.c_uint64(), (C.c_int * 3)(), C.create_string_buffer(512)
    rc = L.ref_guest_prove_hip({"i32_add": 0, "mul_add": 1}[m["guest"]], m["l"], m["k"], bytes.fromhex(m["encoding_seed"]), m["generated_at"], m["reps"],
                               root.ctypes.data, seed1.ctypes.data, cs.ctypes.data, proof.ctypes.data, cap, C.byref(ln), C.byref(rows), C.byref(valid), err, 512)
    assert rc == 0, err.value.decode()
    assert rows.value == m["rows"] and list(valid) == [1, 1, 1]
    assert root.tobytes().hex() == m["oracle_root"] and seed1.tobytes().hex() == m["oracle_stage1_seed"] and cs.tobytes() == f["constsum"]
    assert hashlib.sha256(proof[:ln.value].tobytes()).hexdigest() == m["oracle_proof_sha256"] and ln.value == m["oracle_proof_len"]
