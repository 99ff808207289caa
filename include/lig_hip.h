/*
 * lig_hip.h -- C ABI of the MI355X-native Ligero prover backend (liblig_hip.so).
 *
 * This is the drop-in boundary for the reference's GPU executor `ligero::webgpu_context`
 * (include/wgpu.hpp:50-183, include/ligetron/webgpu/device_context.hpp:29-98).  In the reference the
 * executor is a C++ template parameter (`using executor_t = webgpu_context`, src/webgpu_prover.cpp:54)
 * passed into the stage contexts (include/zkp/nonbatch_context.hpp:68-80); `include/lig_hip_context.hpp`
 * wraps this ABI in a class with the same member names so those drivers compile against it.
 *
 * Conventions
 *   - plain C: opaque context, raw DEVICE pointers (hipMalloc'ed or any HIP-visible allocation, e.g. a
 *     torch tensor's data_ptr()), sizes in elements unless a name says bytes; no C++/torch types.
 *   - a field element is 32 bytes: 8 little-endian u32 limbs, canonical in [0,p) (BN254 Fr), the
 *     reference's device/host/proof format (include/ligetron/webgpu/device_bignum.hpp:76-86).
 *   - every op is asynchronous and ordered on the context's single HIP stream (the reference's single
 *     in-order WebGPU queue, src/webgpu/device_context.cpp:344-354); lig_sync() blocks.
 *   - functions return 0 on success, a negative LIG_E_* code otherwise; lig_last_error() gives text.
 *     (The reference has no error returns: device errors abort, device_context.cpp:121-127.)
 *   - thread-compatible, not thread-safe (the reference is single-threaded).
 */
#ifndef LIG_HIP_H
#define LIG_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lig_ctx lig_ctx;

enum { LIG_OK = 0, LIG_E_ARG = -1, LIG_E_HIP = -2, LIG_E_STATE = -3, LIG_E_NOMEM = -4 };
enum { LIG_ELEM_BYTES = 32, LIG_DIGEST_BYTES = 32 };

/* ---- lifecycle: webgpu_init + ntt_init (include/wgpu.hpp:71-82; src/webgpu/engine.cpp:196-260).
 * p, mu and the three roots are derived from the fixed BN254 constants (the reference's shader hard-codes
 * them too, shader/bn254fr.wgsl.in:19-45); k must be a power of two >= 512 (engine.cpp:849-850), n = 4k. */
int  lig_ctx_create(lig_ctx **out, int device, uint32_t l, uint32_t k, uint32_t n);
void lig_ctx_destroy(lig_ctx *ctx);                       /* syncs, then frees (wgpu.hpp dtor) */
int  lig_sync(lig_ctx *ctx);                              /* device_synchronize() */
const char *lig_last_error(const lig_ctx *ctx);
const char *lig_version(void);
uint32_t lig_message_size(const lig_ctx *ctx);            /* l  (wgpu.hpp:153) */
uint32_t lig_padding_size(const lig_ctx *ctx);            /* k  (wgpu.hpp:154) */
uint32_t lig_encoding_size(const lig_ctx *ctx);           /* n  (wgpu.hpp:155) */
void *lig_stream(lig_ctx *ctx);                           /* hipStream_t, for callers that enqueue their own work */
/* which physical device an ordinal is ("0000:c1:00.0") and whether it can map a peer's memory: what a multi-GPU launcher
 * records before it builds a communicator (bench.py's preflight).  No context needed. */
int lig_device_pci_bus_id(int device, char *out, size_t cap);
int lig_device_peer_access(int device, int peer, int *can);

/* ---- buffers: make_device_buffer (zero-initialised like WebGPU), write_buffer, write_buffer_clear,
 * clear_buffer, copy_buffer_to_buffer, copy_to_host (device_context.hpp:79-98, device_context.cpp:364-449) */
int lig_malloc(lig_ctx *ctx, size_t bytes, void **dptr);
int lig_free(lig_ctx *ctx, void *dptr);
int lig_write(lig_ctx *ctx, void *dst, const void *host_src, size_t bytes);
int lig_write_clear(lig_ctx *ctx, void *dst, size_t dst_bytes, const void *host_src, size_t bytes);
int lig_clear(lig_ctx *ctx, void *dst, size_t bytes);
int lig_copy(lig_ctx *ctx, void *dst, const void *src, size_t bytes);
int lig_read(lig_ctx *ctx, void *host_dst, const void *src, size_t bytes);   /* blocking */
/* page-locked host memory for rows a driver hands to lig_rows_* (uploads from it are true DMA transfers; from pageable memory
 * the runtime stages every copy).  Replaces the std::vector `limbs_` staging of include/zkp/nonbatch_context.hpp:447. */
int lig_host_alloc(lig_ctx *ctx, size_t bytes, void **host_ptr);
int lig_host_free(lig_ctx *ctx, void *host_ptr);
/* Asynchronous form of lig_write for page-locked sources (lig_host_alloc): enqueued on the context stream and NOT waited for.
 * The source must stay untouched until a fence recorded after the call has been waited for.  Fences: lig_fence_record
 * (re)records *fence at the current end of the context stream (allocating it when *fence == NULL), lig_fence_wait blocks the
 * host until that point has been reached.  Used by hip_context's deferred row mode (include/lig_hip_context.hpp): the
 * reference's per-row wgpuQueueWriteBuffer (src/webgpu/device_context.cpp:364-380) batched into one transfer per <= 512 rows
 * that runs under the caller's next rows. */
int  lig_write_async(lig_ctx *ctx, void *dst, const void *pinned_src, size_t bytes);
int  lig_fence_record(lig_ctx *ctx, void **fence);
int  lig_fence_wait(lig_ctx *ctx, void *fence);
void lig_fence_destroy(lig_ctx *ctx, void *fence);

/* ---- Reed-Solomon transforms on ONE n-element buffer, in place
 * (encode_ntt_device / decode_ntt_device / ntt_{forward,inverse}_{k,2k,n}: engine.cpp:755-968) */
enum { LIG_SIZE_K = 0, LIG_SIZE_2K = 1, LIG_SIZE_N = 2 };
int lig_encode(lig_ctx *ctx, void *buf);                  /* buf[0..k) message, buf[k..n) must be 0 */
int lig_encode_2k(lig_ctx *ctx, void *buf);               /* ntt_inverse_2k + ntt_forward_n (mask rows) */
int lig_decode(lig_ctx *ctx, void *buf);                  /* INTT_n, fold k..2k onto 0..k, NTT_k */
int lig_ntt(lig_ctx *ctx, void *buf, int which, int inverse);

/* ---- eltwise (shader/kernels.wgsl.in:326-538; wgpu.hpp:98-139).  x, y, out may alias; `scalar` is a host
 * pointer to one 32-byte element (the reference passes an mpz through a uniform buffer). */
enum {
    LIG_OP_ADD = 0,        /* out = x + y          EltwiseAddMod            */
    LIG_OP_SUB,            /* out = x - y          EltwiseSubMod            */
    LIG_OP_ADD_ASSIGN,     /* out += x             EltwiseAddAssignMod      */
    LIG_OP_ADD_CONST,      /* out = x + c          EltwiseAddMod(scalar)    */
    LIG_OP_SUB_CONST,      /* out = x - c          EltwiseSubConstMod       */
    LIG_OP_CONST_SUB,      /* out = c - x          EltwiseConstSubMod       */
    LIG_OP_MUL,            /* out = x * y          EltwiseMultMod           */
    LIG_OP_MUL_CONST,      /* out = x * c          EltwiseMultMod(scalar)   */
    LIG_OP_MONTMUL_CONST,  /* out = x * c / R      EltwiseMontMultMod       */
    LIG_OP_FMA,            /* out += x * y         EltwiseFMAMod            */
    LIG_OP_FMA_CONST,      /* out += x * c         EltwiseFMAMod(scalar)    */
    LIG_OP_DIV,            /* out = x / y (y=0->0) EltwiseDivMod            */
    LIG_OP_BIT_DECOMPOSE   /* out = bit `bit` of x EltwiseBitDecompose      */
};
int lig_eltwise(lig_ctx *ctx, int op, const void *x, const void *y, void *out, size_t count,
                const uint8_t *scalar32, uint32_t bit);
/* EltwisePowMod / EltwisePowAddMod (src/webgpu/powmod_context.cpp:178-268): out (=|+=) coeff * base^exp */
int lig_powmod(lig_ctx *ctx, const uint8_t *base32, const void *exp_u32, const void *coeff, void *out,
               size_t count, int add);

/* ---- column SHA-256 (shader/sha256.wgsl; engine.cpp:1514-1686).  `state` is a device buffer of
 * lig_sha_state_bytes(n_inst) bytes owned by the caller (the reference's sha256_context, wgpu.hpp:63-68). */
size_t lig_sha_state_bytes(size_t n_inst);
int lig_sha_init(lig_ctx *ctx, void *state, size_t n_inst);                   /* sha256_digest_init  */
int lig_sha_update(lig_ctx *ctx, void *state, const void *row);               /* sha256_digest_update: n_inst elems */
int lig_sha_final(lig_ctx *ctx, void *state, void *digests);                  /* sha256_digest_final: n_inst x 32 B */

/* ---- sampling (engine.cpp:1689-1809; kernels.wgsl.in:541-549) */
int lig_sample_init(lig_ctx *ctx, const uint32_t *host_idx, size_t count);    /* sampling_init */
int lig_sample_gather(lig_ctx *ctx, const void *from, void *to, size_t slot); /* to[slot*count + i] = from[idx[i]] */

/* ==== batched entry points (no reference counterpart: the reference pushes one row per call; these are what
 * a row-batching driver and bench.py call so the GPU sees hundreds of rows per launch) ==== */
/* msgs: rows x k elements (row-major, contiguous); codewords: rows x n elements */
int lig_encode_rows(lig_ctx *ctx, const void *msgs, void *codewords, size_t rows);
/* absorb `rows` codeword rows (rows x n_inst elements, row-major) in order */
int lig_sha_update_rows(lig_ctx *ctx, void *state, const void *codewords, size_t rows);
/* Merkle tree over n_leaves digests (merkle_tree::initialize_from_digest/build_tree, merkle_tree.hpp:344-375):
 * nodes = (2*bit_ceil(n_leaves)-1) x 32 B device buffer in heap order, root = node 0 */
size_t lig_merkle_nodes(size_t n_leaves);
int lig_merkle_build(lig_ctx *ctx, const void *leaves, size_t n_leaves, void *nodes);
/* stage-2 accumulators over a batch (check_code / check_linear / check_quadratic,
 * nonbatch_context.hpp:756-780): for r < rows
 *    code[j]  += rc[r] * U[r][j]                      (rc: rows host scalars, 32 B each)
 *    lin[j]   += U[r][j] * Rn[r][j]                   (Rn may be NULL: skipped)
 *    quad[j]  += rq[t] * (U[x_t][j]*U[y_t][j] - U[z_t][j])   for each triple t (triples: 3 row indices each)
 */
int lig_rlc_rows(lig_ctx *ctx, const void *U, const void *Rn, size_t rows,
                 const uint8_t *rc_host, void *code, void *lin,
                 const uint32_t *triples_host, const uint8_t *rq_host, size_t n_triples, void *quad);
/* out[r*count + i] = codewords[r][idx[i]]  for r < rows (stage 3, nonbatch_context.hpp:924-942) */
int lig_gather_rows(lig_ctx *ctx, const void *codewords, size_t rows, void *out);
/* AES-256-CTR field sampler on the GPU (include/util/csprng.hpp:54-107 + finite_field_gmp.hpp:66-78):
 * out[i] = element number first_elem + i of the stream keyed by key32 (IV = 0) */
int lig_rng_fill(lig_ctx *ctx, const uint8_t *key32, uint64_t first_elem, void *out, size_t count);

/* ==== batched three-stage prover over a resident witness matrix (src/webgpu_prover.cpp:226-494 restructured:
 * rows are encoded once, codewords stay in HBM for stages 2 and 3).  The guest interpreter is replaced by the
 * synthetic constraint stream of BASELINE.md: n_linear witness slots + n_quad slots of x*y=z, witnesses from the
 * AES-256-CTR field stream keyed by witness_key, one dense linear-test coefficient per witness.  The envelope
 * bytes equal the reference's LigeroProofEnvelope (proto/ligero_proof.proto) for the same rows and seeds. ==== */
typedef struct lig_trace lig_trace;
/* Optional batch ("vbn254fr") program executed before the synthetic stream: the guest-visible batch operations of
 * include/host_modules/vbn254fr.hpp:138-565 as a list.  Variables are slots 0..511 of k elements (l data + k-l padding);
 * every operation works on all k elements (the reference binds k-element windows, :64-69) and raises the constraint hook
 * the reference raises (include/zkp/nonbatch_context.hpp:497-553), whose rows are committed in program order:
 *   SET            x <- data[data_off .. +32*len), rest 0            on_batch_init(x): k-l pads drawn, 1 row
 *   SET_SCALAR     x[0..l) <- the element at data_off                on_batch_init(x)
 *   COPY           out <- x                                          on_batch_equal(out, x): 2 rows
 *   ADD, SUB       out <- x op y                                     no row
 *   MUL            out <- x*y                                        on_batch_quadratic(x, y, x*y): 3 rows
 *   DIV            out <- x/y  (x/0 = 0)                             on_batch_quadratic(x/y, y, x): 3 rows
 *   ADD_CONST, SUB_CONST, CONST_SUB (c - x), MUL_CONST, MONTMUL_CONST with the element at data_off: no row
 *   ASSERT_EQUAL                                                     on_batch_equal(x, y)
 *   BIT_DECOMPOSE  slot table of len (= 254) u32 at data_off; out_i <- bit i of x; on_batch_bit(out_i): 1 row each
 *   FREE           x <- 0
 * Stage 2 treats the rows as the reference does: check_code for init / bit / quadratic rows (not for equal rows),
 * check_quadratic for bit (x*x - x) and quadratic triples, quad += r*(x - y) for equal rows; no linear-test randomness.
 * SET / SET_SCALAR with `reserved` bit 0 set: the variable was written by the write_limbs family (vbn254fr_set_str / _set_bytes
 * and their _scalar forms, vbn254fr.hpp:200,222,251,271: write_buffer, nothing cleared): slots beyond the written ones keep
 * their content.  Bit 0 clear: the write_buffer_clear family (vbn254fr_set_ui / _set_ui_scalar, :154,:169).
 *
 * Two slicing semantics.  DECLARED (default): buffer_view::slice_bytes(from, n_bytes) as include/ligetron/webgpu/
 * buffer_view.hpp:52 declares it -- the pad of on_batch_init goes to the variable's own slots [l, k), write_buffer_clear zeroes
 * the rest of the variable.  UPSTREAM (after a LIG_BOP_UPSTREAM_COMPAT op): the definition src/webgpu/buffer_view.cpp:91-95
 * swaps the two parameters, so x.slice(B) of a variable x at slab byte offset X is the view {offset = B, size = X + size(x) - B}
 * of the SLAB: (1) on_batch_init (nonbatch_context.hpp:502-505) writes its 192 pad elements at slab byte l*32 = the pad slots
 * of VARIABLE 0, whatever x is; (2) write_buffer_clear(x, data, len) (device_context.hpp:95-98) clears slab bytes
 * [len*32, X + k*32) after the write -- for a variable other than 0 that is variable 0 from slot `len` on, every variable in
 * between and x itself, the data just written included.  An unmodified v1.5.0 build proves THAT; a proof of a vbn254fr program
 * matches / cross-verifies with it only in this mode. */
enum {
    LIG_BOP_SET = 0, LIG_BOP_SET_SCALAR, LIG_BOP_COPY, LIG_BOP_ADD, LIG_BOP_SUB, LIG_BOP_MUL, LIG_BOP_DIV, LIG_BOP_ADD_CONST,
    LIG_BOP_SUB_CONST, LIG_BOP_CONST_SUB, LIG_BOP_MUL_CONST, LIG_BOP_MONTMUL_CONST, LIG_BOP_ASSERT_EQUAL,
    LIG_BOP_BIT_DECOMPOSE, LIG_BOP_FREE,
    LIG_BOP_UPSTREAM_COMPAT,         /* no operands, no row: from here on slices behave as upstream DEFINES them (see above) */
    LIG_BOP_COUNT
};
enum { LIG_BOP_F_WRITE_LIMBS = 1 };  /* lig_batch_op.reserved, SET / SET_SCALAR */
typedef struct { uint32_t op, out, x, y, len, reserved; uint64_t data_off; } lig_batch_op;
typedef struct {
    uint64_t n_linear, n_quad;
    uint8_t  encoding_seed[32];      /* src/webgpu_prover.cpp:239-245 (there: std::random_device) */
    uint8_t  witness_key[32];
    uint8_t  program_hash[32];
    int64_t  generated_at;           /* metadata timestamp seconds (there: wall clock) */
    char     version[16];            /* "1.5.0" */
    const lig_batch_op *batch_ops; uint64_t n_batch_ops;      /* NULL / 0: no batch rows; read by prepare and verify only */
    const uint8_t *batch_data;     uint64_t batch_data_bytes;
    /* public arguments of the instance after arg0 = "Ligero\0" (src/webgpu_prover.cpp:110-168): n_public_args byte strings
     * back to back in the form the reference holds them in input_args (lig_public_arg_bytes converts the JSON forms);
     * NULL / 0: none.  They enter the proof through instance_hash -> stage1_seed.  Copied by prepare. */
    const uint8_t *public_args; const uint64_t *public_arg_lens; uint64_t n_public_args;
} lig_synth_job;
typedef struct {
    uint8_t  root[32], stage1_seed[32], stage2_seed[32], const_sum[32];
    uint64_t rows;                   /* committed rows including the 3 mask rows */
    int32_t  valid_code, valid_linear, valid_quad;   /* prover self-check, webgpu_prover.cpp:465-469 */
    int32_t  reserved;
    double   ms_stage1, ms_stage2, ms_stage3, ms_total;
} lig_proof_info;
/* untimed: plans the rows, allocates the resident buffers, generates the witness matrix on the GPU */
int  lig_synth_prepare(lig_ctx *ctx, const lig_synth_job *job, lig_trace **out);
/* the hot path: stage 1 (pads, masks, encode, column hash, Merkle), stage 2 (randomness rows, accumulators,
 * seeds, sampling, self-check), stage 3 (column gather, envelope).  *proof points into pinned host memory owned
 * by the trace: valid until the next lig_synth_prove on it or lig_trace_destroy. */
int  lig_synth_prove(lig_trace *trace, const uint8_t **proof, size_t *proof_len, lig_proof_info *info);
uint64_t lig_trace_rows(const lig_trace *trace);
void lig_trace_destroy(lig_trace *trace);

/* ==== verifier for the same synthetic stream (src/webgpu_verifier.cpp:263-452, nonbatch_verifier_context): returns
 * LIG_OK with out->accept = 1 iff the reference's seven predicates hold; a malformed envelope gives accept = 0
 * (parsed = 0), not an error.
 * The constant of the linear test is PUBLIC data upstream: the verifier re-runs the constraint stream and accumulates
 * linear_sums itself (src/webgpu_verifier.cpp:318).  The synthetic stream's public statement is "the committed witness is
 * the AES-256-CTR field stream of job->witness_key" (one linear constraint w_i = b_i per slot), so the verifier derives
 * the constant -sum_i rho_i b_i from the job alone: it regenerates b from witness_key and the coefficients rho from the
 * stage-1 seed.  const_sum == NULL selects that (the sound mode).  A non-NULL const_sum is used as given instead: the
 * hook for a caller whose own constraint generator produced the rows (lig_rows_*) and therefore the constant. ==== */
typedef struct {
    int32_t parsed, indices_match;
    int32_t valid_merkle, valid_code, valid_linear, valid_quad, code_equal, linear_equal, quad_equal;   /* webgpu_verifier.cpp:412-442 */
    int32_t accept;
    int32_t reserved;
    double  ms_total;                /* wall time of the call */
} lig_verify_info;
int lig_synth_verify(lig_ctx *ctx, const lig_synth_job *job, const uint8_t *const_sum /* NULL: derived */, const uint8_t *proof, size_t proof_len,
                     lig_verify_info *out);
/* the verifiers keep their device workspace on the context between calls (~2.5 GB after a 2^24-constraint verification: allocating it
 * per call costs milliseconds); this gives it back (it is also freed by lig_ctx_destroy) */
int lig_verify_release(lig_ctx *ctx);

/* ==== transcript helpers (host only, no GPU work): what a driver needs to stay byte-compatible with the reference ==== */
enum { LIG_ARG_I64 = 0, LIG_ARG_STR = 1, LIG_ARG_HEX = 2 };
/* one JSON "args" entry -> the bytes the reference appends to input_args (src/webgpu_prover.cpp:116-146): i64 (decimal
 * text) = 8 little-endian bytes, str = the characters plus the terminating NUL, hex (optional 0x, odd length gets a leading
 * 0) = the decoded bytes.  Returns LIG_E_ARG for malformed text or a too small buffer (*len = needed size). */
int lig_public_arg_bytes(int kind, const char *text, uint8_t *out, size_t cap, size_t *len);
/* instance_hash over arg0 = "Ligero\0" followed by the given public arguments (src/webgpu_prover.cpp:162-168) */
int lig_instance_hash(const uint8_t *args, const uint64_t *lens, size_t n_args, uint8_t out[32]);
/* hash_random_engine(seed) + portable_sample + sort (src/webgpu_prover.cpp:343-351): the t opened columns */
int lig_sample_columns(const uint8_t seed[32], uint32_t n, uint32_t t, uint32_t *out_sorted);

/* ==== the same three-stage prover over rows SUPPLIED BY THE CALLER: the entry a row-batching driver of the reference's
 * constraint generator uses.  It replaces the per-row callbacks of
 * include/zkp/nonbatch_context.hpp:445-471 (stage 1), :654-780 (stage 2) and :924-970 (stage 3); the rows are what
 * witness_manager hands to those callbacks (include/zkp/backend/witness_manager.hpp:200-269), in commit order.
 *   lig_rows_begin   takes the job: row kinds + the rows x k message matrix (host or device memory).  A host matrix is
 *                    uploaded asynchronously on a copy stream, chunk by chunk, and each chunk is encoded as soon as it has
 *                    arrived (pinned host memory makes the copy truly asynchronous; the memory must stay valid until
 *                    lig_rows_commit returns).
 *   lig_rows_commit  stage 1: pads of the rows flagged LIG_ROW_DRAW_PAD from the encoding stream (pad_encoding_random, in
 *                    commit order), the three mask rows, encode, column hash, Merkle root, stage-1 seed.  The caller now
 *                    derives its linear-test randomness from stage1_seed (the reference re-runs the guest for that).
 *   lig_rows_prove   stage 2 + 3 with the caller's randomness rows (rows x k; the row of a batch-kind row must be zero) and
 *                    the public constant of the linear test (linear_sums, src/webgpu_prover.cpp:307).
 * Encoding-stream positions follow the reference whoever draws: every LINEAR / QX / QY / QZ / INIT row owns the next k-l
 * elements of the stream keyed by encoding_seed (witness_manager pads each of them when it is formed), the three mask rows
 * take what follows; a row flagged LIG_ROW_DRAW_PAD gets exactly its own k-l elements, an unflagged one must carry them.
 * Row kinds: LINEAR rows stand alone; QX,QY,QZ / BQX,BQY,BQZ are consecutive triples (z = x*y); EQX,EQY a consecutive
 * pair; INIT and BIT stand alone.  Code-test coefficients are drawn per row (not for EQ rows), quadratic-test
 * coefficients per triple / pair / bit row, from the streams keyed by stage1_seed, exactly as lig_synth_prove does. ==== */
enum {
    LIG_ROW_LINEAR = 0, LIG_ROW_QX = 1, LIG_ROW_QY = 2, LIG_ROW_QZ = 3, LIG_ROW_INIT = 4, LIG_ROW_BIT = 5, LIG_ROW_EQX = 6,
    LIG_ROW_EQY = 7, LIG_ROW_BQX = 8, LIG_ROW_BQY = 9, LIG_ROW_BQZ = 10,
    LIG_ROW_DRAW_PAD = 0x80      /* or-ed in: slots [l, k) of this row are drawn from the encoding stream by the library */
};
typedef struct {
    uint64_t rows;                   /* committed rows, the 3 mask rows excluded */
    const uint8_t *kinds;            /* one byte per row (host memory) */
    const void *msgs;                /* rows x k elements, row-major */
    int32_t  msgs_on_device;         /* 0: host memory (uploaded inside the timed path), 1: device memory */
    int32_t  reserved;
    uint8_t  encoding_seed[32];
    uint8_t  program_hash[32];
    int64_t  generated_at;
    char     version[16];
    const uint8_t *public_args; const uint64_t *public_arg_lens; uint64_t n_public_args;    /* as in lig_synth_job */
    /* optional (NULL: none): the linear-test randomness rows are the DENSE rows of the synthetic stream -- row r = this many
     * successive elements of the stream keyed by stage1_seed, zeros up to k -- so lig_rows_prove(rands = NULL) may
     * generate them on the device (sampled under the encodes, as lig_synth_prove does) instead of reading them */
    const uint32_t *dense_rands_per_row;
    /* optional (NULL: every row is k full 32-byte elements): the NARROW row format.  One byte per row: 0 or 32 = the row as
     * above; 4 or 8 = only the row's l data slots are supplied, each as a little-endian unsigned integer of that many bytes
     * (real traces are mostly bits and machine words: 8x / 4x less to move over PCIe; the reference ships 32 bytes per slot,
     * include/util/mpz_vector.hpp:108-127).  `msgs` then holds the rows back to back, a narrow row taking l * 4 (or 8)
     * bytes; the library expands it on the device.  A narrow row must be LINEAR / QX / QY / QZ and flagged
     * LIG_ROW_DRAW_PAD (its k - l pad slots are drawn by the library).  lig_rows_restart takes the same packed layout. */
    const uint8_t *elem_bytes;
} lig_rows_job;
int lig_rows_begin(lig_ctx *ctx, const lig_rows_job *job, lig_trace **out);
int lig_rows_commit(lig_trace *trace, uint8_t root[32], uint8_t stage1_seed[32]);
/* const_sum == NULL: the constant is taken to be -sum_r <row_r, rand_r> (returned in info->const_sum) -- for statements that
 * are the rows themselves, like the synthetic stream; the prover's own linear self-check is then vacuous. */
int lig_rows_prove(lig_trace *trace, const void *rands, int rands_on_device, const uint8_t *const_sum,
                   const uint8_t **proof, size_t *proof_len, lig_proof_info *info);
/* Randomness rows handed over WHILE the constraint generator is still producing them (the reference's stage-2 callbacks deliver
 * one row at a time, include/zkp/nonbatch_context.hpp:654-712): after lig_rows_commit, rows [first_row, first_row + n_rows) of the
 * rows x k randomness matrix, in order and without gaps from row 0; host memory (page-locked: lig_host_alloc), valid until
 * lig_rows_prove returns.  The upload starts at once (the library's uploader thread) into a device-resident matrix.  When all
 * rows have been pushed, lig_rows_prove(rands = NULL) uses them: stage 2 waits chunk by chunk for what is still on the bus. */
int lig_rows_push_rands(lig_trace *trace, uint64_t first_row, uint64_t n_rows, const void *host_rows);
/* The same for a generator that puts no linear constraint on many rows (the narrow format of randomness rows): present[i] != 0
 * says that row first_row + i has a randomness row; the present rows follow each other in host_rows (k x 32 bytes each), the
 * others are zero rows and are not shipped -- batch-kind rows always are (nonbatch_context.hpp:782-850 hands no randomness to the
 * vbn254fr hooks), quadratic rows often.  present == NULL: every row is present. */
int lig_rows_push_rands_sparse(lig_trace *trace, uint64_t first_row, uint64_t n_rows, const uint8_t *present, const void *host_rows);
/* the next trace of the same shape (same kinds, seeds, metadata) with new message rows, reusing every buffer of `trace`
 * (no allocation on the proving path of a service); same upload semantics as lig_rows_begin.  It may be called right
 * after lig_rows_commit, BEFORE lig_rows_prove of the committed trace: the new rows then go to a second message matrix
 * and arrive while the current trace is being proved (commit(i) -> restart(i+1) -> prove(i) -> commit(i+1) -> ...). */
int lig_rows_restart(lig_trace *trace, const void *msgs, int msgs_on_device);
/* Host rows travel through one uploader thread per device that waits for every transfer on the host.  A
 * transfer that has not completed after LIG_UPLOAD_TIMEOUT_S (default 5 s; never seen with one process per GPU) does not fail the call:
 * lig_rows_commit / lig_rows_prove discard the stage that ran over the missing rows, wait (bounded by the same time) for the abandoned
 * copy to leave the bus, bring the same rows again with stream-ordered copies and run the stage again -- the proof is the same proof.
 * Only if the abandoned copy is STILL pending after that second wait (it cannot be cancelled) do the calls return LIG_E_HIP; the
 * caller's rows and the trace's device buffers are then still referenced by a DMA transfer: keep the rows alive until
 * *unsettled == 0 here (lig_trace_destroy keeps the device side alive by itself, lig_rows_restart / _commit refuse that trace).
 * *retries = calls of this process that needed the second attempt on ctx's device.  No reference counterpart
 * (src/webgpu/device_context.cpp:364-449 blocks in wgpuQueueWriteBuffer).  Either pointer may be NULL. */
int lig_upload_health(lig_ctx *ctx, uint32_t *retries, uint32_t *unsettled);
/* The verifier's side of a rows job (src/webgpu_verifier.cpp:263-452 with the rows of nonbatch_verifier_context,
 * include/zkp/nonbatch_context.hpp:1219-1287): begin parses the envelope, re-derives both seeds and the sample indices from the
 * job's public data (kinds, public arguments; msgs is ignored) and returns the stage-1 seed; the caller's constraint generator
 * produces the randomness rows and the linear constant from it; finish recommits the 192 opened columns, encodes the
 * randomness rows, evaluates the seven predicates and frees the trace.  A malformed envelope is not an error: begin returns
 * LIG_OK with out->accept = 0 (out->parsed / out->indices_match say why) and *trace = NULL. */
typedef struct lig_vtrace lig_vtrace;
int lig_rows_verify_begin(lig_ctx *ctx, const lig_rows_job *job, const uint8_t *proof, size_t proof_len, lig_vtrace **trace,
                          uint8_t stage1_seed[32], lig_verify_info *out);
int lig_rows_verify_finish(lig_vtrace *trace, const void *rands, int rands_on_device, const uint8_t const_sum[32], lig_verify_info *out);
/* a verifier that gives up between begin and finish (its constraint generator threw) frees the trace here; NULL is a no-op */
void lig_vtrace_destroy(lig_vtrace *trace);
/* rows x k dense randomness rows on the device: row r = per_row[r] successive elements of the AES-256-CTR field stream
 * keyed by key32 (row r starts where row r-1 ended, the first at first_elem), zeros up to k -- the linear-test
 * coefficient rows of the synthetic stream, for callers that feed lig_rows_prove from the device. */
int lig_rng_fill_rows(lig_ctx *ctx, const uint8_t *key32, uint64_t first_elem, const uint32_t *per_row_host, size_t rows, void *out);

/* ==== proof file framing (src/webgpu_prover.cpp:437-457 writes gzip(level 6) of the serialized envelope with
 * Boost.iostreams; src/webgpu_verifier.cpp:249-253 reads it back).  Host-only helpers on zlib: the output is a standard
 * gzip member that any gzip reader (the reference's gzip_decompressor included) accepts; compressed BYTES depend on the
 * zlib version and on Boost's header fields, so bit-exactness is defined on the uncompressed envelope. ==== */
size_t lig_proof_gzip_bound(size_t envelope_len);
int    lig_proof_gzip(const uint8_t *envelope, size_t len, uint8_t *out, size_t cap, size_t *out_len);
size_t lig_proof_gunzip_size(const uint8_t *gz, size_t len);           /* ISIZE trailer; 0 if not a gzip member */
int    lig_proof_gunzip(const uint8_t *gz, size_t len, uint8_t *out, size_t cap, size_t *out_len);

/* ==== one trace sharded over the GPUs of a node (configs[3]).  Rows are dealt to ranks block-cyclically
 * (global chunk g of <= 512 rows belongs to rank g mod world, chunks never split an x,y,z triple), so that after exchange
 * round c every rank holds the W consecutive global chunks cW .. cW+W-1 restricted to ITS columns: the column hash is
 * column-partitioned (rank h owns columns [h n/W, (h+1) n/W)) and absorbs the rows in global order while the next round
 * is still being encoded and exchanged.  One all-to-all of codeword column slices per round; leaves, partial stage-2 sums
 * (k + 2k + 2k values per rank, added mod p locally: RCCL has no modular reduction) and opened columns are all-gathered.
 * Every rank obtains the same envelope, byte-identical to lig_synth_prove of the same job.
 *
 * Collectives: lig_rccl_comm_create gives the RCCL communicator of the product (librccl over xGMI: grouped
 * ncclSend/ncclRecv and ncclAllGather enqueued on HIP streams of the context, nothing blocks the host).  The plain
 * callbacks exist so that tests can run the same sharded logic over gloo on CPU-staged buffers: they are called after
 * the context stream has been drained and must return 0 once the data is in place. ==== */
typedef struct lig_shard lig_shard;
/* A caller that fills in a lig_comm itself MUST zero the whole struct first (memset / = {0}): members may be appended (as
 * `forget` was), a non-NULL optional member is called.  lig_rccl_comm_create / lig_ipc_comm_create zero it themselves. */
typedef struct {
    void *user;
    /* rank g's `send` holds world blocks of block_bytes; block h goes to rank h; `recv` block g comes from rank g */
    int (*all_to_all)(void *user, const void *send_dev, void *recv_dev, size_t block_bytes);
    /* `recv` = world blocks of `bytes` in rank order */
    int (*all_gather)(void *user, const void *send_dev, void *recv_dev, size_t bytes);
    /* optional stream-ordered forms (NULL: not available): the collective is enqueued on `hip_stream`; the data is in
     * place for work enqueued on that stream afterwards */
    int (*all_to_all_on)(void *user, const void *send_dev, void *recv_dev, size_t block_bytes, void *hip_stream);
    int (*all_gather_on)(void *user, const void *send_dev, void *recv_dev, size_t bytes, void *hip_stream);
    /* optional (NULL: nothing to do): the caller is about to free device buffers it has passed as send buffers -- a
     * communicator that exports them to its peers (lig_ipc_comm_create) must not keep handles / mappings of a recycled
     * address.  Contract: call it before freeing any buffer that has been a `send_dev` (lig_shard_destroy does). */
    void (*forget)(void *user);
    /* optional (NULL: the communicator cannot tell): non-zero once the communicator has FAILED -- a peer died, left, or stopped
     * responding -- and the reason is in lig_last_error of its context.  A stream-ordered collective cannot return an error from
     * inside a queue: lig_shard_* asks this after it has drained its streams and returns LIG_E_STATE instead of an envelope made
     * of garbage.  lig_ipc_comm_create: watchdog thread (peer pids, an abort word, a stall timer; waits already queued are
     * released); lig_rccl_comm_create: ncclCommGetAsyncError. */
    int (*failed)(void *user);
    /* optional: make every collective already queued on this rank's streams give up NOW (ncclCommAbort; comm_ipc: poison).  lig_shard_*
     * calls it when queued work containing collectives has not completed LIG_COMM_TIMEOUT_S seconds (default 300) after the host
     * started to wait for it; the communicator is unusable afterwards (failed() != 0). */
    void (*abort)(void *user);
} lig_comm;
/* RCCL communicator: rank 0 calls lig_rccl_unique_id and hands the 128 bytes to every rank through the launcher's
 * rendezvous (torch.distributed / MPI / a file); every rank then calls lig_rccl_comm_create with its context. */
enum { LIG_RCCL_ID_BYTES = 128 };
int  lig_rccl_unique_id(uint8_t out[LIG_RCCL_ID_BYTES]);
int  lig_rccl_comm_create(lig_ctx *ctx, const uint8_t id[LIG_RCCL_ID_BYTES], uint32_t rank, uint32_t world, lig_comm *out);
void lig_rccl_comm_destroy(lig_comm *comm);               /* safe before or after lig_ctx_destroy of its context */
/* librccl is resolved on first use (the copy already mapped into the process wins, then the loader path, then /opt/rocm/lib;
 * LIG_RCCL_LIB overrides): LIG_OK and the library's path / ncclGetVersion, or LIG_E_STATE and the reason in path_out */
int  lig_rccl_available(char *path_out, size_t cap, int *version);
int  lig_rccl_comm_count(const lig_comm *comm, uint32_t *ranks);      /* ncclCommCount of a communicator made here */
/* A second communicator with stream-ordered forms, between processes that can map each other's device memory (same GPU, or
 * GPUs of one node with peer access): every rank pulls its blocks out of the peers' send buffers (hipIpcMemHandle), ordering is
 * done by the GPU on flags in the POSIX shared-memory segment `shm_name` ("/name"; rank 0 creates it, all ranks must pass the
 * same fresh name), no host thread waits for GPU work.  This is what the tests use to run the exchange pipeline of
 * lig_shard_prove with real peers on a one-GPU box (W processes on the device); world <= 16. */
int  lig_ipc_comm_create(lig_ctx *ctx, const char *shm_name, uint32_t rank, uint32_t world, lig_comm *out);
void lig_ipc_comm_destroy(lig_comm *comm);                /* collective: returns when every rank has left (or after 15 s) */
/* host only: the block-cyclic deal of a job's committed rows (masks excluded) for packing size l: *rounds exchange rounds,
 * world * rounds + 1 chunk boundaries (chunk g = rows [b[g], b[g+1]) belongs to rank g mod world).  LIG_E_NOMEM: cap too small. */
int  lig_shard_plan(const lig_synth_job *job, uint32_t l, uint32_t world, uint64_t *rounds, uint64_t *boundaries, size_t cap);
int  lig_shard_prepare(lig_ctx *ctx, const lig_synth_job *job, uint32_t rank, uint32_t world, const lig_comm *comm, lig_shard **out);
int  lig_shard_prove(lig_shard *shard, const uint8_t **proof, size_t *proof_len, lig_proof_info *info);
void lig_shard_destroy(lig_shard *shard);

/* ==== one trace sharded over the GPUs, rows SUPPLIED BY THE CALLER (configs[4]: a real constraint generator on N GPUs): what
 * lig_rows_* is to lig_synth_*.  Every rank passes the kinds of ALL committed rows -- the deal (lig_shard_rows_plan: global
 * chunk g belongs to rank g mod world) and the encoding-stream positions are global -- and the message rows of ITS OWN chunks
 * only, in commit order (job->rows = all rows, job->msgs = the local rows; dense_rands_per_row, if given, covers all rows).
 * lig_shard_rows_commit = stage 1 on all ranks (same root and seed everywhere); each rank's constraint generator then derives
 * the randomness rows of its own rows; lig_shard_rows_prove takes those (local rows x k, zero rows for batch kinds) and the
 * public linear constant (NULL: minus the sum of all inner products, as lig_rows_prove).  Every rank obtains the envelope of
 * lig_rows_prove on the whole trace.  Replaces the per-row callbacks of include/zkp/nonbatch_context.hpp:445-471, :654-780,
 * :924-970 when the rows of one trace live on several GPUs.
 * The sharded entry takes full-width rows only: job->elem_bytes must be NULL (LIG_E_ARG otherwise).
 * LIFETIME: the local rows passed to lig_shard_rows_begin / lig_shard_rows_restart (host or device memory) are copied before the call
 * returns; randomness rows passed to lig_shard_rows_prove are consumed before it returns.  (Only with LIG_SHARD_UPLOADER=1 -- the round-4
 * path through the library's uploader thread, off by default: profiles/r05_rows_entry_hang.md -- host rows are read until
 * lig_shard_rows_commit has returned.)
 * FAILURE: a peer that dies, leaves or stops responding makes these calls return LIG_E_STATE (lig_comm.failed / .abort), they do not hang:
 * every wait on the error paths and in lig_shard_destroy is a bounded poll.  If kernels queued behind the failed collective still have
 * not drained 20 s after the communicator was aborted, the error text says "poisoned": the shard refuses further calls, lig_shard_destroy
 * returns at once and keeps the device buffers (the queued kernels may still touch them), the rows / randomness rows handed to the failed
 * call must stay alive, and the context must not be reused -- tear the process down. ==== */
int lig_shard_rows_plan(const uint8_t *kinds, size_t n_rows, uint32_t world, uint64_t *rounds, uint64_t *boundaries, size_t cap);
int lig_shard_rows_begin(lig_ctx *ctx, const lig_rows_job *job, uint32_t rank, uint32_t world, const lig_comm *comm, lig_shard **out);
int lig_shard_rows_restart(lig_shard *shard, const void *local_msgs, int msgs_on_device);   /* next trace, same shape; after lig_shard_rows_prove of the previous one (LIG_E_STATE otherwise) */
int lig_shard_rows_commit(lig_shard *shard, uint8_t root[32], uint8_t stage1_seed[32]);
int lig_shard_rows_prove(lig_shard *shard, const void *local_rands, int rands_on_device, const uint8_t *const_sum,
                         const uint8_t **proof, size_t *proof_len, lig_proof_info *info);

/* sizeof of the public structs as this build of the library sees them, in the order {lig_batch_op, lig_synth_job, lig_proof_info,
 * lig_verify_info, lig_rows_job, lig_comm}: a binding in another language (ctypes, cgo, JNI) checks its own layouts against these at
 * load time -- members are appended to these structs over time and a short struct on the caller's side is read past its end. */
enum { LIG_ABI_STRUCTS = 6 };
void lig_abi_sizes(uint32_t out[LIG_ABI_STRUCTS]);

/* Measurement hook (no reference counterpart): while enabled, every lig_encode_rows launch group records HIP
 * events on the context stream immediately around the dominant kernel (k_encode_tiles).  lig_profile_read syncs
 * and returns the number of bracketed launches, the rows they covered and the summed kernel time. */
int lig_profile_enable(lig_ctx *ctx, int on);
int lig_profile_read(lig_ctx *ctx, uint64_t *launches, uint64_t *rows, double *total_ms);
/* the same restricted to the bracketed launches of exactly rows_in_launch rows (512 = the full chunks: what a rocprofv3 kernel table
 * lists per launch size) */
int lig_profile_read_launches(lig_ctx *ctx, uint32_t rows_in_launch, uint64_t *launches, double *total_ms);

#ifdef __cplusplus
}
#endif
#endif /* LIG_HIP_H */
