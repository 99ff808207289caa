#!/usr/bin/env python3
"""Golden vectors produced by the REFERENCE's own GMP-side code: oracle/_ref/libref_backend.so is built by `make -C oracle` from
/root/reference/src/bn254.cpp, include/zkp/finite_field_gmp.hpp, include/util/{csprng,mpz_vector}.hpp and
include/zkp/backend/{witness_manager,core,lazy_witness}.hpp where they lie, behind oracle/ref_backend.cpp.  Run in the BUILD container
(the upstream tree does not exist on the GPU box); what is written here travels instead:

  tests/golden/ref_field.json          sampler (generate_random over mpz_random_engine: refill boundaries, draws that take the
                                       subtraction), generate_omegas, mulmod / invmod / powmod / divmod / mont_mulmod / ..., the constants,
                                       mpz_vector::export_limbs / import_limbs
  tests/golden/ref_rows_<guest>.npz    the row stream the reference's witness_manager + ligetron_backend emit for a guest (kinds, rows
                                       with pads, SHA-256 of the masks; stage-2 replay: randomness rows, constant sum), and what the ORACLE computes
                                       from it (root, seeds, envelope hash) -- the reference has no CPU encoder / hasher to ask.
                                       Guests: tests/i32_add.wat replayed by hand on the reference backend at (l, k) = (320, 512) and
                                       (8000, 8192); a multiply-add stream with constants at (320, 512).

  make -C oracle && python tests/golden/make_ref_backend.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol          # noqa: E402  (the oracle computes the expected commitment / envelope of a recorded row stream)
import ref_backend_lib as rb     # noqa: E402

P = ol.P
KEYS = [bytes(range(32)), hashlib.sha256(b"ref-field-key-1").digest()]
T = 192
ROW_SETS = [("i32_add_320", "i32_add", 320, 512, 0, 5), ("i32_add_8000", "i32_add", 8000, 8192, 0, 5), ("mul_add_320", "mul_add", 320, 512, 700, 9)]


def xof_ints(tag, count, bits=256):
    out, c = [], 0
    while len(out) < count:
        h = b"".join(hashlib.sha256(tag + c.to_bytes(4, "little") + bytes([i])).digest() for i in range((bits + 255) // 256))
        out.append(int.from_bytes(h, "little") >> (len(h) * 8 - bits))
        c += 1
    return out


def hexes(arr):
    return [bytes(x).hex() for x in np.ascontiguousarray(arr, dtype=np.uint32).reshape(-1, 8).view(np.uint8).reshape(-1, 32)]


def field_vectors():
    out = {"note": "made by tests/golden/make_ref_backend.py from oracle/_ref/libref_backend.so (reference code); 32-byte values are "
                   "little-endian hex as mpz_vector::export_limbs(4 x u64) writes them"}
    samp = []
    for key in KEYS:
        n = 1100                                  # two 16 KiB refills inside (elements 512 and 1024 start a fresh buffer)
        vals, raw = rb.field_random(key, n), rb.engine_raw(key, n)
        took = [i for i, v in enumerate(ol.from_limbs(raw)) if (v >> 2) >= P]
        pick = sorted(set(list(range(8)) + list(range(508, 516)) + list(range(1020, 1028)) + took[:6] + [n - 1]))
        samp.append({"key": key.hex(), "count": n, "sha256_of_all": hashlib.sha256(vals.tobytes()).hexdigest(),
                     "took_subtraction": took[:40], "n_took_subtraction": len(took),
                     "elements": {str(i): hexes(vals[i])[0] for i in pick}})
    out["sampler"] = samp
    out["omegas"] = {str(k): hexes(rb.omegas(k)) for k in (512, 1024, 2048, 4096, 8192, 16384, 32768, 1 << 20, 1 << 26)}
    names = ["modulus", "modulus_2x", "modulus_4x", "modulus_middle", "root1", "root2", "montgomery_factor", "barrett_factor"]
    out["constants"] = dict(zip(names, hexes(rb.constants())))
    ops = {}
    a_s = [v % P for v in xof_ints(b"ref-op-a", 12)] + [0, 1, P - 1, P - 2, 2, (P - 1) // 2]
    b_s = [v % P for v in xof_ints(b"ref-op-b", 12)] + [1, P - 1, P - 1, 3, 0, 7]
    for name in ("mulmod", "addmod", "submod", "mont_mulmod"):
        ops[name] = [[hex(a), hex(b), hex(rb.field_op(name, a, b))] for a, b in zip(a_s, b_s)]
    ops["divmod"] = [[hex(a), hex(b), hex(rb.field_op("divmod", a, b))] for a, b in zip(a_s, b_s) if b]
    ops["invmod"] = [[hex(a), hex(rb.field_op("invmod", a))] for a in a_s if a]
    ops["negate"] = [[hex(a), hex(rb.field_op("negate", a))] for a in a_s]
    ops["powmod"] = [[hex(a), hex(e), hex(rb.field_op("powmod", a, e))] for a, e in zip(a_s, xof_ints(b"ref-op-e", len(a_s)))]
    ops["powmod_ui"] = [[hex(a), hex(e & 0xffffffff), hex(rb.field_op("powmod_ui", a, e & 0xffffffff))] for a, e in zip(a_s, xof_ints(b"ref-op-u", len(a_s)))]
    wide = xof_ints(b"ref-op-w", 10) + [P, 2 * P, 4 * P, 4 * P - 1, (1 << 256) - 1]
    ops["reduce"] = [[hex(a), hex(rb.field_op("reduce", a))] for a in wide]
    ops["reduce_u256"] = [[hex(a), hex(rb.field_op("reduce_u256", a))] for a in wide]
    out["ops"] = ops
    # export_limbs / import_limbs: 4 x u64 <-> 8 x u32 are the same bytes; a short value leaves its high limbs zero
    vals = [0, 1, (1 << 64) - 1, 1 << 64, P - 1, (1 << 200) + 12345] + [v % P for v in xof_ints(b"ref-limbs", 4)]
    blob = b"".join(v.to_bytes(32, "little") for v in vals)
    as32, w32 = rb.limbs_roundtrip(blob, len(vals), 8, 4, 4, 8)
    as64, w64 = rb.limbs_roundtrip(blob, len(vals), 4, 8, 8, 4)
    out["limbs"] = {"values": [hex(v) for v in vals], "u64x4_to_u32x8": as32.hex(), "words32": w32, "u32x8_to_u64x4": as64.hex(), "words64": w64}
    return out


def row_set(name, guest, l, k, reps, generated_at):
    n = 4 * k
    key = KEYS[0]
    g1 = rb.guest(guest, l, k, key, reps=reps)
    # the reference has no CPU encoder: the ORACLE commits to the recorded rows; its stage-1 seed keys the stage-2 replay
    p0 = ol.prove_rows(l, k, n, T, g1["kinds"], g1["vals"], g1["mask_code"], g1["mask_lin"], g1["mask_quad"], None, None, generated_at=generated_at)
    g2 = rb.guest(guest, l, k, key, wit_key=p0["stage1_seed"], reps=reps)
    assert np.array_equal(g1["vals"], g2["vals"]) and np.array_equal(g1["kinds"], g2["kinds"])
    for m in ("mask_code", "mask_lin", "mask_quad"):
        assert np.array_equal(g1[m], g2[m])
    p1 = ol.prove_rows(l, k, n, T, g2["kinds"], g2["vals"], g2["mask_code"], g2["mask_lin"], g2["mask_quad"], g2["rands"], g2["constsum"], generated_at=generated_at)
    assert p1["valid"] == [1, 1, 1] and p1["root"] == p0["root"]
    meta = dict(guest=guest, l=l, k=k, n=n, t=T, reps=reps, generated_at=generated_at, encoding_seed=key.hex(), rows=int(len(g2["kinds"])),
                oracle_root=p1["root"].hex(), oracle_stage1_seed=p1["stage1_seed"].hex(), oracle_stage2_seed=p1["stage2_seed"].hex(),
                oracle_proof_sha256=hashlib.sha256(p1["proof"]).hexdigest(), oracle_proof_len=len(p1["proof"]),
                masks_sha256={m: hashlib.sha256(g2[m].tobytes()).hexdigest() for m in ("mask_code", "mask_lin", "mask_quad")},
                note="kinds / vals / masks / rands / constsum: recorded from the reference's witness_manager + ligetron_backend (oracle/ref_backend.cpp); "
                     "the guest is a hand replay of the interpreter's calls; oracle_*: computed by oracle/ from the recorded stream")
    path = os.path.join(HERE, "ref_rows_%s.npz" % name)
    # the masks are 5k incompressible elements (770 KB at k = 8192): the fixture carries their SHA-256; a test rebuilds them with the
    # sampler under test and must hit the hashes before it uses them
    np.savez_compressed(path, kinds=g2["kinds"], vals=g2["vals"], rands=g2["rands"], constsum=np.frombuffer(g2["constsum"], dtype=np.uint8),
                        meta=np.array(json.dumps(meta)))
    print("%s: %d rows %r, constsum %s, %d bytes" % (name, len(g2["kinds"]), g2["kinds"].tolist(), g2["constsum"].hex()[:16], os.path.getsize(path)))


def main():
    if rb.load() is None:
        sys.exit("oracle/_ref/libref_backend.so is missing: run `make -C oracle` where /root/reference and GMP headers exist")
    with open(os.path.join(HERE, "ref_field.json"), "w") as f:
        json.dump(field_vectors(), f, indent=1)
    for rs in ROW_SETS:
        row_set(*rs)


if __name__ == "__main__":
    main()
