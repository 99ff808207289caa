#!/usr/bin/env python3
"""Envelope known answers made with the REAL protobuf runtime from the REFERENCE's schema files.

The image has the Python `google.protobuf` runtime but no protoc, so the two schema files are read where they lie
(/root/reference/proto/common.proto, ligero_proof.proto) by the ~60-line proto3 reader below and turned into dynamic
descriptors; messages are then filled the way include/zkp/proof_serializer.hpp:119-191 and src/webgpu_prover.cpp:410-427
fill them and serialised by the runtime.  The resulting bytes pin the hand-written encoders (oracle lo_serialize_proof,
csrc/prover_common.hpp write_envelope) -- field order, packed encodings, the empty-Timestamp and empty-vector cases.
Run in the BUILD container; the digests travel as tests/golden/ref_envelope.json.

  python tests/golden/make_ref_envelope.py
"""
import hashlib
import json
import os
import re

from google.protobuf import descriptor_pb2, descriptor_pool, message_factory, timestamp_pb2

HERE = os.path.dirname(os.path.abspath(__file__))
PROTO_DIR = "/root/reference/proto"
SCALARS = {"string": 9, "bytes": 12, "uint32": 13, "int32": 5, "int64": 3, "uint64": 4, "fixed32": 7, "bool": 8}


def parse_proto(path, name):
    """minimal proto3 reader: package, imports, enums, messages with scalar / message / enum fields, repeated, oneof"""
    src = re.sub(r"//[^\n]*", "", open(path).read())
    fd = descriptor_pb2.FileDescriptorProto(name=name, syntax="proto3")
    fd.package = re.search(r"package\s+([\w.]+)\s*;", src).group(1)
    for m in re.finditer(r'import\s+"([^"]+)"\s*;', src):
        fd.dependency.append(m.group(1))
    for m in re.finditer(r"enum\s+(\w+)\s*\{([^}]*)\}", src):
        e = fd.enum_type.add(name=m.group(1))
        for v in re.finditer(r"(\w+)\s*=\s*(\d+)\s*;", m.group(2)):
            e.value.add(name=v.group(1), number=int(v.group(2)))
    pos = 0
    while True:
        m = re.compile(r"message\s+(\w+)\s*\{").search(src, pos)
        if not m:
            break
        depth, i = 1, m.end()
        while depth:
            depth += {"{": 1, "}": -1}.get(src[i], 0)
            i += 1
        body, pos = src[m.end():i - 1], i
        msg = fd.message_type.add(name=m.group(1))
        oneofs = {}
        for om in re.finditer(r"oneof\s+(\w+)\s*\{([^}]*)\}", body):
            oneofs[om.group(2)] = len(msg.oneof_decl)
            msg.oneof_decl.add(name=om.group(1))
        flat = re.sub(r"oneof\s+\w+\s*\{([^}]*)\}", r"\1", body)
        for f in re.finditer(r"(repeated\s+)?([\w.]+)\s+(\w+)\s*=\s*(\d+)\s*;", flat):
            fld = msg.field.add(name=f.group(3), number=int(f.group(4)))
            fld.label = 3 if f.group(1) else 1
            t = f.group(2)
            if t in SCALARS:
                fld.type = SCALARS[t]
            else:
                fld.type_name = "." + (t if "." in t else fd.package + "." + t)     # resolved as message or enum below
            for text, idx in oneofs.items():
                if re.search(r"\b%s\s*=\s*%s\s*;" % (f.group(3), f.group(4)), text):
                    fld.oneof_index = idx
    return fd


def build_pool():
    pool = descriptor_pool.DescriptorPool()
    pool.AddSerializedFile(timestamp_pb2.DESCRIPTOR.serialized_pb)
    files = [parse_proto(os.path.join(PROTO_DIR, "common.proto"), "common.proto"),
             parse_proto(os.path.join(PROTO_DIR, "ligero_proof.proto"), "ligero_proof.proto")]
    enums = {"." + f.package + "." + e.name for f in files for e in f.enum_type}
    for f in files:
        for m in f.message_type:
            for fld in m.field:
                if fld.type_name:
                    fld.type = 14 if fld.type_name in enums else 11
        pool.AddSerializedFile(f.SerializeToString())
    return pool


def xof(tag, n):
    out, c = b"", 0
    while len(out) < n:
        out += hashlib.sha256(tag + c.to_bytes(4, "little")).digest()
        c += 1
    return out[:n]


def u32s(b):
    return [int.from_bytes(b[i:i + 4], "little") for i in range(0, len(b), 4)]


CASES = [
    dict(name="typical", version="1.5.0", generated_at=1700000000, k=512, n=2048, t=192, n_sib=300, n_idx=192, rows=9),
    dict(name="zero_timestamp", version="1.5.0", generated_at=0, k=8192, n=32768, t=192, n_sib=5, n_idx=192, rows=4),
    dict(name="empty_samples_and_siblings", version="1.5.0", generated_at=7, k=512, n=2048, t=192, n_sib=0, n_idx=3, rows=0),
    dict(name="no_version_no_indices", version="", generated_at=-1, k=1024, n=4096, t=192, n_sib=2, n_idx=0, rows=1),
]


def inputs(c):
    """deterministic inputs of a case (shared with tests/test_oracle.py through the same xof construction)"""
    tag = c["name"].encode()
    idx = sorted(set(int.from_bytes(xof(tag + b"idx", 4 * c["n_idx"])[4 * i:4 * i + 4], "little") % c["n"] for i in range(c["n_idx"])))
    return dict(program_hash=xof(tag + b"ph", 32), root=xof(tag + b"root", 32),
                siblings=[xof(tag + b"sib%d" % i, 32) for i in range(c["n_sib"])], idx=idx,
                code=xof(tag + b"code", 32 * c["n"]), lin=xof(tag + b"lin", 32 * c["n"]), quad=xof(tag + b"quad", 32 * c["n"]),
                samples=xof(tag + b"smp", 32 * c["rows"] * c["t"]))


def serialize(pool, c):
    cls = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName(n))
    env = cls("ligero.v1.LigeroProofEnvelope")()
    x = inputs(c)
    md = env.metadata                                      # src/webgpu_prover.cpp:410-427
    md.prover_version = c["version"]
    md.proof_schema_version = 1
    md.proof_type = 1                                      # PROOF_TYPE_CLASSIC
    md.program_hash.value = x["program_hash"]
    md.packing_size, md.codeword_size, md.sample_size, md.security_level = c["k"], c["n"], c["t"], 128
    md.generated_at.seconds = c["generated_at"]
    md.generated_at.SetInParent()                          # mutable_generated_at(): present even when seconds == 0
    proof = env.ligero_proof                               # serialize_proof, proof_serializer.hpp:166-191
    mt = proof.merkle_tree
    mt.algorithm = 1                                       # HASH_ALGORITHM_SHA256
    mt.root.value = x["root"]
    mt.leaf_indices.extend(x["idx"])
    for s in x["siblings"]:
        mt.sibling_hashes.add().value = s
    for fld, key in (("encoded_code", "code"), ("encoded_linear", "lin"), ("encoded_quadratic", "quad"), ("sampled_data", "samples")):
        v = getattr(proof, fld)
        v.SetInParent()                                    # mutable_*(): the submessage is present even when empty
        v.values.extend(u32s(x[key]))
    return env.SerializeToString(deterministic=True)


def main():
    pool = build_pool()
    out = {"generator": "google.protobuf %s runtime, descriptors parsed from %s/{common,ligero_proof}.proto" %
           (__import__("google.protobuf").protobuf.__version__, PROTO_DIR),
           "inputs": "see inputs() in tests/golden/make_ref_envelope.py: xof(tag, n) = SHA256(tag || le32(counter)) blocks",
           "cases": []}
    for c in CASES:
        b = serialize(pool, c)
        out["cases"].append(dict(c, length=len(b), sha256=hashlib.sha256(b).hexdigest(), head=b[:96].hex()))
    with open(os.path.join(HERE, "ref_envelope.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote ref_envelope.json:", [(c["name"], c["length"]) for c in out["cases"]])


if __name__ == "__main__":
    main()
