#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.

Run in the BUILD container (needs the `openssl` CLI for the AES-256-CTR vectors and the
pure-Python definitions of tests/pydef.py).  The reference itself cannot be executed (no Dawn /
wabt / Boost / protobuf here) and is C++ only, so no fixture is produced by importing it; the
powmod known answers restate tests/webgpu/test_powmod.cpp (the only device KATs upstream holds).

  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import pydef  # noqa: E402

P = pydef.P


def openssl_ctr(key: bytes, nbytes: int) -> bytes:
    """zero plaintext, IV = 0 -> raw keystream (what include/util/csprng.hpp:64-83 reads)"""
    out = subprocess.run(["openssl", "enc", "-aes-256-ctr", "-K", key.hex(), "-iv", "00" * 16, "-nosalt"],
                         input=b"\0" * nbytes, stdout=subprocess.PIPE, check=True).stdout
    assert len(out) == nbytes
    return out


def main():
    # ---- AES-256-CTR keystream + field sampler
    vec = []
    for name, key in (("bytes0_31", bytes(range(32))),
                      ("synth1", hashlib.sha256(b"lig-synth" + (1).to_bytes(8, "little")).digest()),
                      ("zeros", bytes(32))):
        ks = openssl_ctr(key, 16384 + 64)          # spans one 16 KiB refill boundary
        elems = [pydef.field_from_keystream(ks[32 * i:32 * i + 32]) for i in range(8)]
        last = [pydef.field_from_keystream(ks[32 * i:32 * i + 32]) for i in (511, 512, 513)]
        vec.append({"name": name, "key": key.hex(), "keystream_first64": ks[:64].hex(),
                    "keystream_sha256": hashlib.sha256(ks).hexdigest(),
                    "field_first8": [hex(e) for e in elems], "field_511_512_513": [hex(e) for e in last]})
    json.dump({"generator": "openssl enc -aes-256-ctr (OpenSSL CLI of the build image)", "vectors": vec},
              open(os.path.join(HERE, "aes_ctr.json"), "w"), indent=1)

    # ---- encode / decode / leaves / merkle by definition at the smallest legal packing k=512
    k, n, l, t = 512, 2048, 320, 192
    key = hashlib.sha256(b"golden-encode").digest()
    ks = openssl_ctr(key, 32 * (3 * k))
    rows = [[pydef.field_from_keystream(ks[32 * (r * k + i):32 * (r * k + i) + 32]) for i in range(k)] for r in range(3)]
    cws = [pydef.encode(r, k, n) for r in rows]
    leaves = [pydef.leaf([cw[j] for cw in cws]) for j in range(n)]
    nodes = pydef.merkle_nodes(leaves)
    msg2k = rows[0] + rows[1]
    cw2k = pydef.encode(msg2k, k, n, two_k=True)
    dec = pydef.decode(cws[0], k, n)
    h = lambda vals: hashlib.sha256(b"".join(v.to_bytes(32, "little") for v in vals)).hexdigest()
    json.dump({"k": k, "n": n, "key": key.hex(),
               "note": "rows = AES-CTR field samples; codeword by Lagrange/evaluate definition (pydef.encode)",
               "row0_first4": [hex(v) for v in rows[0][:4]],
               "cw0_first4": [hex(v) for v in cws[0][:4]], "cw0_last2": [hex(v) for v in cws[0][-2:]],
               "cw_sha256": [h(c) for c in cws], "cw2k_sha256": h(cw2k), "decode0_sha256": h(dec),
               "leaf0": leaves[0].hex(), "leaf_last": leaves[-1].hex(),
               "leaves_sha256": hashlib.sha256(b"".join(leaves)).hexdigest(), "root": nodes[0].hex()},
              open(os.path.join(HERE, "encode_k512.json"), "w"), indent=1)

    # ---- column sampling (restated Boost algorithm -- parity unpinned upstream)
    smp = []
    for seed in (bytes(32), bytes(range(32)), hashlib.sha256(b"seed").digest()):
        for (nn, tt) in ((32768, 192), (2048, 192), (200, 192)):
            smp.append({"seed": seed.hex(), "n": nn, "t": tt, "indices": pydef.sample_indices(seed, nn, tt)})
    json.dump({"note": "hash_random_engine + restated boost uniform_int + partial Fisher-Yates + sort", "cases": smp},
              open(os.path.join(HERE, "sampling.json"), "w"))

    # ---- powmod KATs restating tests/webgpu/test_powmod.cpp:89-197 (N reduced to 64 per case for file size;
    #      the test suite recomputes all 8192 with Python pow())
    N = 64
    kat = {"zero_coeff": [0] * N,
           "base_one": [1] * N,
           "generator": [hex(pow(7, i, P)) for i in range(N)],
           "minus": [hex((P - 1) * pow(P - 1, (1 << 16) + i, P) % P) for i in range(N)],
           "powmod_add": [hex(10 * pow(7, i, P) % P) for i in range(N)]}
    json.dump(kat, open(os.path.join(HERE, "powmod_kat.json"), "w"), indent=0)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
