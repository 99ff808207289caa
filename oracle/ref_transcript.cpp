// ref_transcript.cpp -- C entry points around the REFERENCE's own transcript headers, compiled from where they lie
// under /root/reference (oracle/Makefile target `_ref`; output oracle/_ref/libref_transcript.so, git-ignored).
//
// TEST INFRASTRUCTURE ONLY.  This file contains no algorithm: every function below instantiates the reference's
// templates and forwards to them, so that tests can check the oracle's restatement (oracle/hash.c) and the HIP
// backend's host transcript (ligero-prover_amd/csrc/prover_common.hpp) against the code the reference prover runs:
//
//   include/zkp/hash.hpp:44-118,153-214,341-346   overload_hash byte streams, openssl_hash, hash<>()
//   include/zkp/random.hpp:29-84                  aes256ctr_engine (16 KiB refills of one AES-256-CTR stream)
//   include/zkp/random.hpp:87-146                 hash_random_engine
//   include/zkp/merkle_tree.hpp:155-375           build_tree / decommit / recommit
//   include/params.hpp:34                         params::hasher = sha256
//
// The GMP side of the reference (util/csprng.hpp, finite_field_gmp.hpp, src/bn254.cpp, util/mpz_vector.hpp, the constraint backend and
// witness_manager) is built by the sibling driver ref_backend.cpp (round 5: the image has GMP headers under /opt/conda/include).
// What is NOT reachable in this image: util/portable_sample.hpp (Boost uniform_int_distribution; Boost is absent) and
// zkp/proof_serializer.hpp (protobuf-generated code).  Those stay pinned as DESIGN.md section 5 says.
//
// Build flags (oracle/Makefile): -std=c++20, -D__EMSCRIPTEN__ selects the reference's own no-Boost.Log branch of
// util/log.hpp:19-28 (reached from merkle_tree.hpp via util/timer.hpp); `-include <std header>` supplies standard headers
// the reference relies on transitively.  No stand-in headers are written.
#include <params.hpp>
#include <zkp/hash.hpp>
#include <zkp/merkle_tree.hpp>
#include <zkp/random.hpp>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace zkp = ligero::vm::zkp;
namespace params = ligero::vm::params;
using hasher = params::hasher;
using digest = hasher::digest;
using tree_t = zkp::merkle_tree<hasher>;

static digest to_digest(const uint8_t* p) { digest d; std::memcpy(d.data, p, 32); return d; }

extern "C" {

// zkp::hash_random_engine<params::hasher> engine(seed); out[i] = engine()      (src/webgpu_prover.cpp:343)
void ref_hash_engine_bytes(const uint8_t seed[32], size_t count, uint8_t* out) {
    zkp::hash_random_engine<hasher> engine(to_digest(seed));
    for (size_t i = 0; i < count; i++) out[i] = engine();
}

// instance_hash = hash<hasher>(instance_hash, input_args[i]) over the public arguments, arg0 = "Ligero\0" first
// (src/webgpu_prover.cpp:110-168).  args: nargs byte strings back to back, lens[i] bytes each, already in the form the
// prover holds them in input_args (i64 = 8 LE bytes, str with its NUL, hex decoded).
void ref_instance_hash(const uint8_t* args, const uint64_t* lens, size_t nargs, uint8_t out[32]) {
    digest instance_hash;                           // zero-initialised (hash.hpp:158)
    for (size_t i = 0; i < nargs; i++) {
        std::vector<uint8_t> a(args, args + lens[i]);
        args += lens[i];
        instance_hash = zkp::hash<hasher>(instance_hash, a);
    }
    std::memcpy(out, instance_hash.data, 32);
}

// zkp::hash<params::hasher>("LigetronStage1", stage1_root, instance_hash)     (src/webgpu_prover.cpp:281-282)
void ref_stage1_seed(const uint8_t root[32], const uint8_t instance_hash[32], uint8_t out[32]) {
    const digest r = to_digest(root), ih = to_digest(instance_hash);
    const digest s = zkp::hash<hasher>("LigetronStage1", r, ih);
    std::memcpy(out, s.data, 32);
}

// zkp::hash<params::hasher>("LigetronStage2", stage1_root, code_limbs, linear_limbs, quad_limbs)   (:337-341)
void ref_stage2_seed(const uint8_t root[32], const uint32_t* code, const uint32_t* lin, const uint32_t* quad, size_t nlimbs,
                     uint8_t out[32]) {
    const digest r = to_digest(root);
    const std::vector<uint32_t> c(code, code + nlimbs), l(lin, lin + nlimbs), q(quad, quad + nlimbs);
    const digest s = zkp::hash<hasher>("LigetronStage2", r, c, l, q);
    std::memcpy(out, s.data, 32);
}

// aes256ctr_engine<uint64_t> (random.hpp:29-84): `count` successive 64-bit words of the keystream
void ref_aes_engine_words(const uint8_t key[32], const uint8_t iv[16], size_t count, uint64_t* out) {
    unsigned char k[32], v[16];
    std::memcpy(k, key, 32);
    std::memcpy(v, iv, 16);
    zkp::aes256ctr_engine<uint64_t> e(k, v);
    for (size_t i = 0; i < count; i++) out[i] = e();
}

// merkle_tree = vector<digest> (initialize_from_digest + build_tree, merkle_tree.hpp:344-375); nodes_out: size() x 32 B
size_t ref_merkle_build(const uint8_t* leaves, size_t nleaves, uint8_t* nodes_out) {
    std::vector<digest> d(nleaves);
    for (size_t i = 0; i < nleaves; i++) d[i] = to_digest(leaves + 32 * i);
    tree_t t;
    t = d;
    if (nodes_out) for (size_t i = 0; i < t.size(); i++) std::memcpy(nodes_out + 32 * i, t[i].data, 32);
    return t.size();
}

// tree.decommit(known_index) (merkle_tree.hpp:155-215): the (position, digest) pairs it holds, sorted by position
// descending level / ascending position is the serializer's business (proof_serializer.hpp:82-117, not buildable here);
// here they are returned sorted by heap position so that the comparison is order-independent.
size_t ref_merkle_decommit(const uint8_t* leaves, size_t nleaves, const uint64_t* idx, size_t nidx, uint64_t* pos_out,
                           uint8_t* dig_out, size_t cap) {
    std::vector<digest> d(nleaves);
    for (size_t i = 0; i < nleaves; i++) d[i] = to_digest(leaves + 32 * i);
    tree_t t;
    t = d;
    const std::vector<size_t> known(idx, idx + nidx);
    const auto dec = t.decommit(known);
    std::vector<size_t> pos;
    for (const auto& kv : dec.nodes()) pos.push_back(kv.first);
    std::sort(pos.begin(), pos.end());
    for (size_t i = 0; i < pos.size() && i < cap; i++) {
        pos_out[i] = pos[i];
        std::memcpy(dig_out + 32 * i, dec[pos[i]].data, 32);
    }
    return pos.size();
}

// merkle_tree::recommit(leaf digests of the opened columns, decommitment) (merkle_tree.hpp:232-318)
void ref_merkle_recommit(size_t n_nodes, const uint64_t* idx, size_t nidx, const uint8_t* leaf_digests, const uint64_t* pos,
                         const uint8_t* digs, size_t npos, uint8_t root[32]) {
    const std::vector<size_t> known(idx, idx + nidx);
    tree_t::decommitment dec(n_nodes, known);
    for (size_t i = 0; i < npos; i++) dec.insert(pos[i], to_digest(digs + 32 * i));
    std::vector<digest> b(nidx);
    for (size_t i = 0; i < nidx; i++) b[i] = to_digest(leaf_digests + 32 * i);
    const digest r = tree_t::recommit(b, dec);
    std::memcpy(root, r.data, 32);
}

}  // extern "C"
