#!/usr/bin/env python3
"""Golden transcript vectors produced by the REFERENCE's own code: oracle/_ref/libref_transcript.so is built by
`make -C oracle` from /root/reference/include/zkp/{hash,random,merkle_tree}.hpp + params.hpp (where they lie; only
OpenSSL is needed) behind the thin driver oracle/ref_transcript.cpp.  Run in the BUILD container (the upstream tree does
not exist on the GPU box); the vectors written to tests/golden/ref_transcript.json travel instead.

  make -C oracle && python tests/golden/make_ref_transcript.py
"""
import ctypes as C
import hashlib
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_transcript.so")


def load():
    L = C.CDLL(REF_SO)
    vp, sz = C.c_void_p, C.c_size_t
    L.ref_hash_engine_bytes.argtypes = [C.c_char_p, sz, vp]
    L.ref_instance_hash.argtypes = [C.c_char_p, vp, sz, vp]
    L.ref_stage1_seed.argtypes = [C.c_char_p, C.c_char_p, vp]
    L.ref_stage2_seed.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, sz, vp]
    L.ref_aes_engine_words.argtypes = [C.c_char_p, C.c_char_p, sz, vp]
    L.ref_merkle_build.restype = sz
    L.ref_merkle_build.argtypes = [C.c_char_p, sz, vp]
    L.ref_merkle_decommit.restype = sz
    L.ref_merkle_decommit.argtypes = [C.c_char_p, sz, vp, sz, vp, vp, sz]
    L.ref_merkle_recommit.argtypes = [sz, vp, sz, C.c_char_p, vp, C.c_char_p, sz, vp]
    return L


def engine_bytes(L, seed, count):
    out = (C.c_uint8 * count)()
    L.ref_hash_engine_bytes(seed, count, out)
    return bytes(out)


def instance_hash(L, args):
    """args: list of byte strings INCLUDING arg0"""
    lens = (C.c_uint64 * max(1, len(args)))(*[len(a) for a in args])
    out = (C.c_uint8 * 32)()
    L.ref_instance_hash(b"".join(args), lens, len(args), out)
    return bytes(out)


def stage1(L, root, ih):
    out = (C.c_uint8 * 32)()
    L.ref_stage1_seed(root, ih, out)
    return bytes(out)


def stage2(L, root, code, lin, quad):
    out = (C.c_uint8 * 32)()
    L.ref_stage2_seed(root, code, lin, quad, len(code) // 4, out)
    return bytes(out)


def aes_words(L, key, iv, count):
    out = (C.c_uint64 * count)()
    L.ref_aes_engine_words(key, iv, count, out)
    return list(out)


def merkle(L, leaves, idx):
    n = len(leaves)
    size = L.ref_merkle_build(b"".join(leaves), n, None)
    nodes = (C.c_uint8 * (32 * size))()
    L.ref_merkle_build(b"".join(leaves), n, nodes)
    nodes = bytes(nodes)
    cidx = (C.c_uint64 * len(idx))(*idx)
    cap = 64 * len(idx) + 64
    pos, dig = (C.c_uint64 * cap)(), (C.c_uint8 * (32 * cap))()
    cnt = L.ref_merkle_decommit(b"".join(leaves), n, cidx, len(idx), pos, dig, cap)
    leafd = b"".join(nodes[32 * (size // 2 + i):32 * (size // 2 + i) + 32] for i in idx)
    root = (C.c_uint8 * 32)()
    L.ref_merkle_recommit(size, cidx, len(idx), leafd, pos, bytes(dig), cnt, root)
    return dict(n_leaves=n, nodes=size, root=nodes[:32].hex(), nodes_sha256=hashlib.sha256(nodes).hexdigest(),
                idx=list(idx), decommit=[[int(pos[i]), bytes(dig[32 * i:32 * i + 32]).hex()] for i in range(cnt)],
                recommit_root=bytes(root).hex())


def xof(tag, n):
    out, c = b"", 0
    while len(out) < n:
        out += hashlib.sha256(tag + c.to_bytes(4, "little")).digest()
        c += 1
    return out[:n]


def main():
    L = load()
    seeds = [bytes(32), bytes(range(32)), hashlib.sha256(b"seed").digest()]
    out = {"generator": "oracle/_ref/libref_transcript.so = the reference's include/zkp/{hash,random,merkle_tree}.hpp + "
                        "params.hpp (ligero-prover v1.5.0) compiled here; see oracle/ref_transcript.cpp",
           "hash_engine": [{"seed": s.hex(), "bytes": engine_bytes(L, s, 200).hex()} for s in seeds]}
    arg0 = b"Ligero\0"
    i64 = lambda v: int(v).to_bytes(8, "little", signed=True)
    cases = {"none": [], "i64": [i64(42)], "i64_neg": [i64(-1)], "str": [b"hello\0"], "str_empty": [b"\0"],
             "hex": [bytes.fromhex("0abc")], "mixed": [i64(7), b"abc\0", bytes.fromhex("deadbeef"), i64(1 << 40)]}
    out["instance_hash"] = [{"name": k, "args": [a.hex() for a in v], "hash": instance_hash(L, [arg0] + v).hex()}
                            for k, v in cases.items()]
    s1 = []
    for i in range(3):
        root, ih = xof(b"root%d" % i, 32), instance_hash(L, [arg0] + list(cases.values())[i])
        s1.append({"root": root.hex(), "instance_hash": ih.hex(), "seed": stage1(L, root, ih).hex()})
    out["stage1_seed"] = s1
    s2 = []
    for i, nel in enumerate((1, 16, 2048)):
        root = xof(b"root2%d" % i, 32)
        code, lin, quad = (xof(b"%s%d" % (t, i), 32 * nel) for t in (b"code", b"lin", b"quad"))
        s2.append({"root": root.hex(), "n_elems": nel, "xof_tags": ["code%d" % i, "lin%d" % i, "quad%d" % i],
                   "seed": stage2(L, root, code, lin, quad).hex()})
    out["stage2_seed"] = s2
    out["stage2_seed_note"] = "vectors = xof(tag, 32*n_elems): SHA256(tag || le32(counter)) blocks concatenated"
    aes = []
    for key in (bytes(range(32)), hashlib.sha256(b"lig-synth" + (1).to_bytes(8, "little")).digest()):
        w = aes_words(L, key, bytes(16), 2048 * 2 + 4)          # 2048 words per 16 KiB refill: two refill boundaries
        aes.append({"key": key.hex(), "first4": w[:4], "around_refill_1": w[2046:2050], "around_refill_2": w[4094:4098],
                    "sha256_le64": hashlib.sha256(b"".join(x.to_bytes(8, "little") for x in w)).hexdigest()})
    out["aes_engine_u64"] = aes
    mk = []
    for n, idx in ((8, [1, 6]), (8, [0, 1, 2, 3, 4, 5, 6, 7]), (37, [0, 5, 36]), (2048, sorted(set(int.from_bytes(xof(b"idx", 4 * 192)[4 * i:4 * i + 4], "little") % 2048 for i in range(192))))):
        leaves = [xof(b"leaf%d_%d" % (n, i), 32) for i in range(n)]
        mk.append(merkle(L, leaves, idx))
    out["merkle"] = mk
    out["merkle_note"] = "leaf i of a case with n leaves = xof('leaf<n>_<i>', 32); decommit = (heap position, digest) pairs sorted by position"
    with open(os.path.join(HERE, "ref_transcript.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote ref_transcript.json")


if __name__ == "__main__":
    main()
