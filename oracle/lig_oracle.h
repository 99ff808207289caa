/*
 * lig_oracle.h -- CPU restatement of the ligero-prover v1.5.0 hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / CPU baseline.  The HIP product
 * (ligero-prover_amd/csrc) never links or calls it.
 *
 * Parity status.  The reference as a whole cannot be compiled here (Dawn/wabt/
 * Boost/protobuf-generated code/GMP headers are absent) and
 * ships no golden vectors for NTT outputs, SHA leaves, Merkle roots, sample
 * indices or proof bytes.  What IS pinned to reference code or vectors:
 *   - montgomery_mul + limb layout: the five powmod cases of
 *     tests/webgpu/test_powmod.cpp (tests/test_oracle.py);
 *   - the transcript: the reference's own headers include/zkp/hash.hpp,
 *     random.hpp, merkle_tree.hpp and params.hpp compile with OpenSSL alone and
 *     are built into oracle/_ref/libref_transcript.so (ref_transcript.cpp,
 *     Makefile target _ref): hash_random_engine byte streams, both seed byte
 *     streams, instance_hash with i64 / str / hex arguments, the AES-256-CTR
 *     engine's refills, Merkle build / decommit set / recommit.  Vectors
 *     generated from it are committed (tests/golden/ref_transcript.json,
 *     tests/golden/make_ref_transcript.py) and checked against this oracle;
 *   - the proof envelope: serialised with the protobuf runtime from descriptors
 *     equal to the files under proto/ (tests/golden/make_ref_envelope.py) and compared
 *     byte for byte with lo_serialize_proof.
 * Still "parity unpinned": Boost's uniform_int_distribution (not vendored
 * upstream, absent here; restated from boost/random/uniform_int_distribution.hpp
 * generate_uniform_int) and therefore the 192 sample indices; NTT outputs,
 * SHA leaves and the AES field sampler are pinned against independent
 * definitions (Python big-int Lagrange interpolation, hashlib, OpenSSL CLI,
 * FIPS-197 / FIPS-180 vectors) in tests/.
 *
 * Every function cites the reference file:line (relative to the upstream
 * tree) whose behaviour it restates.  All field elements are 8 x u32 little
 * endian limbs, canonical in [0,p), exactly as the reference keeps them in
 * device buffers (include/ligetron/webgpu/device_bignum.hpp:76-86).
 */
#ifndef LIG_ORACLE_H
#define LIG_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t v[4]; } lo_fr;          /* little-endian 4 x u64 == 8 x u32 LE */
typedef struct { uint64_t v[8]; } lo_wide;        /* 512-bit product */

/* ---- field (shader/bigint.wgsl.in, shader/bn254fr.wgsl.in, src/bn254.cpp) ---- */
extern const lo_fr LO_P, LO_2P, LO_J, LO_R, LO_MU;
void lo_fr_from_u64(lo_fr *o, uint64_t x);
int  lo_fr_cmp(const lo_fr *a, const lo_fr *b);
int  lo_fr_is_zero(const lo_fr *a);
void lo_fr_add(lo_fr *o, const lo_fr *a, const lo_fr *b);      /* mod p */
void lo_fr_sub(lo_fr *o, const lo_fr *a, const lo_fr *b);      /* mod p */
void lo_fr_neg(lo_fr *o, const lo_fr *a);
void lo_fr_mul(lo_fr *o, const lo_fr *a, const lo_fr *b);      /* Barrett, plain x plain -> plain */
void lo_fr_montmul(lo_fr *o, const lo_fr *a, const lo_fr *b);  /* a*b*R^-1, result [0,p) */
void lo_fr_to_mont(lo_fr *o, const lo_fr *a);                  /* a*R mod p */
void lo_fr_pow(lo_fr *o, const lo_fr *a, const lo_fr *e);      /* 256-bit exponent */
void lo_fr_pow_u64(lo_fr *o, const lo_fr *a, uint64_t e);
void lo_fr_inv(lo_fr *o, const lo_fr *a);                      /* 0 -> 0 */
void lo_mul_wide(lo_wide *o, const lo_fr *a, const lo_fr *b);
void lo_barrett_reduce(lo_fr *o, const lo_wide *w);
void lo_omegas(uint32_t k, lo_fr *wk, lo_fr *w2k, lo_fr *w4k); /* src/bn254.cpp:51-64 */

/* ---- NTT context (src/webgpu/engine.cpp:196-, 1382-1503) ---- */
typedef struct lo_ctx lo_ctx;
lo_ctx *lo_ctx_new(uint32_t l, uint32_t k, uint32_t n);
void    lo_ctx_free(lo_ctx *c);
uint32_t lo_ctx_k(const lo_ctx *c);
uint32_t lo_ctx_l(const lo_ctx *c);
uint32_t lo_ctx_n(const lo_ctx *c);
enum { LO_SIZE_K = 0, LO_SIZE_2K = 1, LO_SIZE_N = 2 };
void lo_ntt_forward(const lo_ctx *c, int which, lo_fr *buf);   /* ntt_forward_{k,2k,n} */
void lo_ntt_inverse(const lo_ctx *c, int which, lo_fr *buf);   /* ntt_inverse_{k,2k,n} */
void lo_encode(const lo_ctx *c, lo_fr *buf);                   /* encode_ntt_device, buf = n elems */
void lo_encode_2k(const lo_ctx *c, lo_fr *buf);                /* ntt_inverse_2k + ntt_forward_n */
void lo_decode(const lo_ctx *c, lo_fr *buf);                   /* decode_ntt_device */
void lo_encode_rows(const lo_ctx *c, const lo_fr *msgs, lo_fr *codewords, size_t rows, int threads);

/* ---- eltwise kernels (shader/kernels.wgsl.in:326-538) ---- */
enum {
    LO_OP_ADD = 0, LO_OP_SUB, LO_OP_ADD_ASSIGN, LO_OP_ADD_CONST, LO_OP_SUB_CONST, LO_OP_CONST_SUB,
    LO_OP_MUL, LO_OP_MUL_CONST, LO_OP_MONTMUL_CONST, LO_OP_FMA, LO_OP_FMA_CONST, LO_OP_DIV,
    LO_OP_BIT_DECOMPOSE
};
void lo_eltwise(int op, const lo_fr *x, const lo_fr *y, lo_fr *out, size_t count,
                const lo_fr *scalar, uint32_t bit);
void lo_powmod(const lo_fr *base, const uint32_t *exp, const lo_fr *coeff, lo_fr *out,
               size_t count, int add);                         /* src/webgpu/powmod_context.cpp */

/* ---- SHA-256 ---- */
typedef struct { uint32_t h[8]; uint8_t buf[64]; uint64_t len; uint32_t fill; } lo_sha256;
void lo_sha256_init(lo_sha256 *s);
void lo_sha256_update(lo_sha256 *s, const void *data, size_t n);
void lo_sha256_final(lo_sha256 *s, uint8_t out[32]);
void lo_sha256_buf(const void *data, size_t n, uint8_t out[32]);
/* column hashing (shader/sha256.wgsl:128-230): one lo_sha256 per column */
void lo_colsha_init(lo_sha256 *st, size_t ncols);
void lo_colsha_update(lo_sha256 *st, const lo_fr *row, size_t ncols);
void lo_colsha_final(lo_sha256 *st, uint8_t *leaves, size_t ncols);   /* LE-word digests */
size_t lo_sizeof_sha256(void);

/* ---- Merkle (include/zkp/merkle_tree.hpp, include/zkp/proof_serializer.hpp:82-117) ---- */
size_t lo_merkle_nodes(size_t nleaves);                       /* 2*bit_ceil(n)-1 */
void   lo_merkle_build(const uint8_t *leaves, size_t nleaves, uint8_t *nodes);
size_t lo_merkle_decommit(const uint8_t *nodes, size_t nleaves, const uint32_t *idx, size_t nidx,
                          uint8_t *siblings, size_t cap);     /* canonical sibling order */
int    lo_merkle_recommit(size_t nleaves, const uint32_t *idx, size_t nidx, const uint8_t *leaf_digests,
                          const uint8_t *siblings, size_t nsib, uint8_t root[32]);

/* ---- AES-256-CTR field sampler (include/util/csprng.hpp, finite_field_gmp.hpp:66-78) ---- */
typedef struct { uint32_t rk[60]; uint64_t pos; } lo_rng;     /* pos = element index (32 B each) */
void lo_aes256_expand(const uint8_t key[32], uint32_t rk[60]);
void lo_aes256_encrypt_block(const uint32_t rk[60], const uint8_t in[16], uint8_t out[16]);
void lo_rng_init(lo_rng *r, const uint8_t key[32]);
void lo_rng_keystream(const lo_rng *r, uint64_t first_block, uint8_t *out, size_t nblocks);
void lo_rng_next(lo_rng *r, lo_fr *out);
void lo_rng_fill(lo_rng *r, lo_fr *out, size_t count);

/* ---- Fiat-Shamir + sampling (webgpu_prover.cpp:162-168,281-282,337-351; random.hpp; portable_sample.hpp) ---- */
void lo_stage1_seed(const uint8_t root[32], const uint8_t instance_hash[32], uint8_t out[32]);
void lo_stage2_seed(const uint8_t root[32], const lo_fr *code, const lo_fr *lin, const lo_fr *quad,
                    size_t n, uint8_t out[32]);
void lo_instance_hash_default(uint8_t out[32]);               /* only arg0 = "Ligero\0" */
/* instance_hash chained over arg0 = "Ligero\0" and n_args further public arguments (webgpu_prover.cpp:162-168) */
void lo_instance_hash(const uint8_t *args, const uint64_t *lens, size_t n_args, uint8_t out[32]);
/* `count` bytes of hash_random_engine<sha256>(seed) (include/zkp/random.hpp:87-146) */
void lo_hash_engine_bytes(const uint8_t seed[32], size_t count, uint8_t *out);
void lo_sample_indices(const uint8_t seed[32], uint32_t n, uint32_t t, uint32_t *out_sorted);

/* ---- proof envelope (proto/{common,ligero_proof}.proto, proof_serializer.hpp:166-191) ---- */
size_t lo_serialize_proof(uint8_t *out, size_t cap,
                          const char *version, const uint8_t program_hash[32], int64_t generated_at,
                          uint32_t k, uint32_t n, uint32_t t,
                          const uint8_t root[32], const uint8_t *siblings, size_t nsib,
                          const uint32_t *leaf_idx, size_t nidx,
                          const lo_fr *code, const lo_fr *lin, const lo_fr *quad,
                          const lo_fr *samples, size_t nsample_elems);

/* ---- synthetic row stream + reference-structured prover / verifier ---- */
/* Batch ("vbn254fr") program: the guest-visible batch operations of include/host_modules/vbn254fr.hpp:138-565 as a
 * list, executed before the synthetic stream.  Variables are slots 0..511 of k elements (l data + k-l padding); every
 * operation works on all k elements (the reference binds k-element windows, vbn254fr.hpp:64-69) and raises the
 * constraint hook the reference raises (nonbatch_context.hpp:497-553): the rows it names are committed in program order.
 *   SET x <- data[data_off .. +32*len), rest 0                      -> on_batch_init(x)   (k-l pads drawn, 1 row)
 *   SET_SCALAR x[0..l) <- data[data_off .. +32)                      -> on_batch_init(x)
 *   COPY out <- x                                                    -> on_batch_equal(out, x)        (2 rows)
 *   ADD/SUB out <- x op y;  *_CONST with the 32-byte constant at data_off (CONST_SUB: c - x)          (no row)
 *   MUL out <- x*y                                                   -> on_batch_quadratic(x, y, x*y) (3 rows)
 *   DIV out <- x/y                                                   -> on_batch_quadratic(x/y, y, x) (3 rows)
 *   ASSERT_EQUAL                                                     -> on_batch_equal(x, y)
 *   BIT_DECOMPOSE: slot table of `len` (= 254) u32 at data_off; out_i <- bit i of x -> on_batch_bit(out_i) (1 row each)
 *   FREE x <- 0
 * SET / SET_SCALAR with reserved bit 0: written by the write_limbs family (vbn254fr.hpp:200,222,251,271), nothing cleared.
 * LO_BOP_UPSTREAM_COMPAT switches to buffer_view::slice_bytes as src/webgpu/buffer_view.cpp:91-95 DEFINES it (parameters
 * swapped against the declaration buffer_view.hpp:52): x.slice(B) = {offset B, size X + size(x) - B} of the slab, hence
 * on_batch_init's pad (nonbatch_context.hpp:502-505) lands in variable 0's pad slots and write_buffer_clear
 * (device_context.hpp:95-98) clears slab bytes [len*32, X + k*32).  Default: the declared semantics. */
enum {
    LO_BOP_SET = 0, LO_BOP_SET_SCALAR, LO_BOP_COPY, LO_BOP_ADD, LO_BOP_SUB, LO_BOP_MUL, LO_BOP_DIV, LO_BOP_ADD_CONST,
    LO_BOP_SUB_CONST, LO_BOP_CONST_SUB, LO_BOP_MUL_CONST, LO_BOP_MONTMUL_CONST, LO_BOP_ASSERT_EQUAL, LO_BOP_BIT_DECOMPOSE,
    LO_BOP_FREE, LO_BOP_UPSTREAM_COMPAT, LO_BOP_COUNT
};
typedef struct { uint32_t op, out, x, y, len, reserved; uint64_t data_off; } lo_batch_op;

typedef struct {
    uint32_t l, k, n, t;
    uint64_t n_linear;            /* number of linear constraints (witness slots) */
    uint64_t n_quad;              /* number of quadratic constraints (slots of x*y=z) */
    uint8_t  encoding_seed[32];
    uint8_t  witness_key[32];     /* AES key of the synthetic witness stream */
    int64_t  generated_at;
    int      threads;
    const lo_batch_op *batch_ops; uint64_t n_batch_ops;        /* optional batch program (NULL / 0: none) */
    const uint8_t *batch_data;    uint64_t batch_data_bytes;
    /* public arguments after arg0 = "Ligero\0" (src/webgpu_prover.cpp:110-168), byte strings back to back in the form the
     * reference holds them in input_args (i64 = 8 LE bytes, str with its NUL, hex decoded); NULL / 0: none */
    const uint8_t *public_args; const uint64_t *public_arg_lens; uint64_t n_public_args;
    /* 0: witnesses are full field elements.  32 / 64: every witness DRAW (linear, x, y slots) keeps only its low 32 / 64 bits --
     * the synthetic counterpart of a real trace, whose witnesses are mostly bits and machine words; used to check the narrow
     * row format of the HIP rows entry (lig_rows_job.elem_bytes) against this prover */
    uint32_t witness_bits;
} lo_job;

typedef struct {
    uint8_t  root[32], stage1_seed[32], stage2_seed[32];
    uint32_t *sample_idx;         /* t */
    lo_fr   *code, *lin, *quad;   /* n each (encoded accumulators) */
    lo_fr   *samples;             /* rows * t */
    size_t   rows;                /* committed rows incl. 3 masks */
    uint8_t *proof; size_t proof_len;
    lo_fr    const_sum;           /* linear constant sum */
    int      valid_code, valid_linear, valid_quad;
    double   t_stage1, t_stage2, t_stage3;
} lo_proof;

void lo_synth_key(uint64_t seed, uint8_t key[32]);            /* SHA256("lig-synth" || le64(seed)) */
size_t lo_job_rows(const lo_job *j);                          /* committed rows incl. masks */
void lo_row_kinds(const lo_job *j, uint8_t *kinds);           /* kind of every non-mask row in commit order (0 linear, 1-3 x/y/z, 4.. batch rows) */
int  lo_prove(const lo_job *j, lo_proof *out);                /* reference-structured 3-stage prover */
/* const_sum == NULL: the verifier derives the constant of the linear test from the public statement of the synthetic
 * stream (the witness_key stream b and the coefficients rho of the stage-1 seed: -sum rho_i b_i), as the reference's
 * verifier accumulates linear_sums from the public constraint stream (src/webgpu_verifier.cpp:318) */
int  lo_verify(const lo_job *j, const lo_fr *const_sum, const uint8_t *proof, size_t len);   /* 1 = accept */
/* the dense stage-2 randomness rows of the synthetic stream ((rows - 3) x k: one linear-stream draw per data slot in
 * commit order, zeros elsewhere and for batch rows) and the constant -sum <witness row, randomness row> */
void lo_rand_rows(const lo_job *j, const uint8_t stage1_seed[32], lo_fr *rands, lo_fr *const_sum);
void lo_proof_free(lo_proof *p);
/* row former exposed for tests: fills rows[(rows) * k] message rows in commit order (masks: code row k elems,
 * linear/quad masks 2k elems are returned separately) */
void lo_form_rows(const lo_job *j, lo_fr *rows /* (R-3)*k */, lo_fr *mask_code /* k */,
                  lo_fr *mask_lin /* 2k */, lo_fr *mask_quad /* 2k */);

/* the three masks from the encoding stream at element position `pos` (witness_manager.hpp:271-321) */
void lo_form_masks(const uint8_t encoding_seed[32], uint64_t pos, uint32_t l, uint32_t k, lo_fr *mask_code, lo_fr *mask_lin, lo_fr *mask_quad);
/* the three-stage prover over a row stream formed elsewhere (kinds 0..3; rows with pads; masks; randomness rows and constant sum of the
 * constraint generator's stage-2 replay): what lo_prove does after lo_form_rows.  Uses j->l,k,n,t, generated_at, threads, public args. */
int  lo_prove_rows(const lo_job *j, const uint8_t *kinds, size_t rows_count, const lo_fr *rows, const lo_fr *mask_code,
                   const lo_fr *mask_lin, const lo_fr *mask_quad, const lo_fr *rands, const lo_fr *const_sum, lo_proof *out);

#ifdef __cplusplus
}
#endif
#endif
