"""CPU-only: a COMPLETE proof of the oracle's reference-structured prover (k = 512, linear + quadratic + batch rows) is
PARSED by the protobuf runtime with descriptors read from the reference's own proto/*.proto, every field is compared with
what the prover put in, the Merkle decommitment inside it is recommitted to the root from the opened columns alone, and the
re-serialised message is byte-identical -- i.e. an unmodified reader of proto/ligero_proof.proto (the reference's verifier,
src/webgpu_verifier.cpp:249-262, include/zkp/proof_serializer.hpp:193-212) reads this file as the prover wrote it.
(The HIP prover's envelope equals the oracle's byte for byte in every -m gpu prover test.)

Needs /root/reference/proto (build container); skipped where the reference is absent (the GPU box)."""
import ctypes as C
import hashlib
import importlib.util
import os

import numpy as np
import pytest

import oracle_lib as ol
import test_batch_rows as tb

HERE = os.path.dirname(os.path.abspath(__file__))
PROTO_DIR = "/root/reference/proto"
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(PROTO_DIR, "ligero_proof.proto")),
                                reason="the reference's .proto files are not on this machine")


def envelope_class():
    spec = importlib.util.spec_from_file_location("make_ref_envelope", os.path.join(HERE, "golden", "make_ref_envelope.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    from google.protobuf import message_factory
    pool = m.build_pool()
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("ligero.v1.LigeroProofEnvelope"))


def u32_bytes(values):
    return np.asarray(values, dtype=np.uint32).tobytes()


@pytest.mark.parametrize("with_batch", [False, True])
def test_full_oracle_proof_parses_field_by_field_and_reserialises(with_batch):
    l, k, n, t = 320, 512, 2048, 192
    job = ol.make_job(l, k, n, t, 2000, 900, generated_at=1712345678, threads=2)
    if with_batch:
        tb.demo_program().attach(job)
    pr = ol.Proof()
    assert ol.lib().lo_prove(C.byref(job), C.byref(pr)) == 0
    try:
        blob = bytes(pr.proof[:pr.proof_len])
        rows = pr.rows
        idx = np.ctypeslib.as_array(pr.sample_idx, shape=(t,)).copy()
        vec = lambda p, cnt: np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(cnt * 8,)).copy()
        code, lin, quad = vec(pr.code, n), vec(pr.lin, n), vec(pr.quad, n)
        samples = vec(pr.samples, rows * t)
        root = bytes(pr.root)
    finally:
        ol.lib().lo_proof_free(C.byref(pr))
    env = envelope_class()()
    assert env.ParseFromString(blob) == len(blob)
    # metadata, src/webgpu_prover.cpp:410-427
    md = env.metadata
    assert (md.prover_version, md.proof_schema_version, md.proof_type) == ("1.5.0", 1, 1)
    assert md.program_hash.value == bytes(32)                       # the synthetic job has no program file: 32 zero bytes
    assert (md.generated_at.seconds, md.generated_at.nanos) == (1712345678, 0)
    assert (md.packing_size, md.codeword_size, md.sample_size, md.security_level) == (k, n, t, 128)
    assert env.WhichOneof("payload") == "ligero_proof"
    # the proof body, include/zkp/proof_serializer.hpp:166-191
    p = env.ligero_proof
    mt = p.merkle_tree
    assert mt.algorithm == 1 and mt.root.value == root
    assert list(mt.leaf_indices) == [int(i) for i in idx] == sorted(set(int(i) for i in idx))
    assert u32_bytes(p.encoded_code.values) == code.tobytes()
    assert u32_bytes(p.encoded_linear.values) == lin.tobytes()
    assert u32_bytes(p.encoded_quadratic.values) == quad.tobytes()
    assert u32_bytes(p.sampled_data.values) == samples.tobytes() and len(p.sampled_data.values) == rows * t * 8
    # what the verifier does with it (webgpu_verifier.cpp:412-419): hash the opened columns (shader/sha256.wgsl byte order:
    # every limb big-endian, digest words stored little-endian), recommit with the sibling hashes, compare with the root
    smp = samples.reshape(rows, t, 8)
    leaves = np.zeros((t, 32), dtype=np.uint8)
    for j in range(t):
        d = hashlib.sha256(smp[:, j, :].astype(">u4").tobytes()).digest()
        leaves[j] = np.frombuffer(np.frombuffer(d, dtype=">u4").astype("<u4").tobytes(), dtype=np.uint8)
    sib = np.frombuffer(b"".join(h.value for h in mt.sibling_hashes), dtype=np.uint8).copy()
    assert all(len(h.value) == 32 for h in mt.sibling_hashes)
    got_root = np.zeros(32, dtype=np.uint8)
    lo = ol.lib()
    lo.lo_merkle_recommit.argtypes = [C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    ok = lo.lo_merkle_recommit(n, ol.ptr(np.ascontiguousarray(idx)), t, ol.ptr(leaves), ol.ptr(sib) if sib.size else None,
                               len(mt.sibling_hashes), ol.ptr(got_root))
    assert ok == 1 and got_root.tobytes() == root, "the decommitment inside the envelope does not lead to the root"
    # and back: the runtime's serialisation of the parsed message is the file
    assert env.SerializeToString(deterministic=True) == blob
