// kernels.hpp -- internal interface between the C ABI (lig_capi.hip) and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <string>
#include "fr.hpp"

namespace lig {
// Every run-time knob of the library, read ONCE from the environment by the first lig_ctx_create of the process
// (lig_capi.hip: lig::knobs()).  None is needed on the product path: the defaults are what was measured best; the
// table in DESIGN.md (appendix) names the experiment each one belongs to.  No other getenv exists in the library
// (tests/test_generated_code.py greps for that).
struct Knobs {
    int      encode_kmask = 15;        // LIG_ENCODE_KMASK   run only K1 (1) / K2 (6) / K3 (8) of the encoder
    uint32_t k13_block = 256;          // LIG_K13_BLOCK      workgroup size of K1 / K3 (64 | 128 | 256)
    uint32_t k2_dyn_lds = 0;           // LIG_K2_DYN_LDS     extra dynamic LDS bytes of the tile kernel (occupancy experiments)
    size_t   encode_chunk = 512;       // LIG_ENCODE_CHUNK   rows per encode launch group
    bool     encode_generic = false;   // LIG_ENCODE_GENERIC radix-2 row path even where the tiled encoder exists (tests)
    uint32_t sha_block = 256;          // LIG_SHA_BLOCK      workgroup size of the column hash
    int      sha_ws = 2;               // LIG_SHA_WS         column hash: 0 one wave per 64 columns; 1 / 2 / 4 wave-specialised (producer + consumer waves), groups per workgroup
    uint32_t aes_blocks = 0;           // LIG_AES_BLOCKS     persistent workgroups of the big sampler launches (0: two per CU)
    bool     shared_side = true;       // LIG_SHARED_SIDE    1: one side stream (column hash, samplers) per device for all contexts of the process; 0: one per context (rounds 1-5)
    int      aes_layout = 1;           // LIG_AES_LAYOUT     1 (default since round 6): AES tables of the big sampler launches entry-major, a lookup address is one v_perm_b32; 0: table-major (rounds 2-5)
    int      sha_gate = 1;             // LIG_SHA_GATE       1: place every chunk's hash before the encode stream goes on; 2: the next chunk's K1 runs first, ALONE (the hash
                                       //                    waits for it), then the hash is placed, then the tile kernel goes on
    size_t   sha_gate_rows = 2;        // LIG_SHA_GATE_ROWS  rows hashed before the encode stream is released
    int      sha_prio = 0;             // LIG_SHA_PRIO       1: the side stream (column hash, samplers) is a high-priority stream
    int      ctx_low_prio_every = 0;   // LIG_CTX_LOW_PRIO_EVERY  n > 0: every n-th context of the process gets lowest-priority streams (a "filler" proof)
    int      sha_cumask = 0;           // LIG_SHA_CUMASK     1: the stage-1 hash of even / odd contexts runs on disjoint halves of the CUs (CU-masked stream)
    size_t   s1_head = 128, s1_tail = 96, s2_head = 192;   // LIG_S1_HEAD / LIG_S1_TAIL / LIG_S2_HEAD  chunk schedule
    bool     fused_rlc = true;         // LIG_NO_FUSED_RLC   (set: the two-kernel sampler / accumulate path of round 2)
    bool     early_code = true;        // LIG_EARLY_CODE=0   accumulate the code test inside the row loop
    int      upload_mode = 2;          // LIG_UPLOAD_MODE    2: uploader thread, 1: per-context copy stream + events
    bool     upload_prio = true;       // LIG_UPLOAD_PRIO=0  the uploader thread's stream at normal priority (shares the proof streams' hardware-queue pool: A/B only)
    int      fault_upload = 0;         // LIG_FAULT_UPLOAD   tests: a transfer of the uploader thread never completes -- 1: the first of witness rows, 2: the first of randomness rows (exercises LIG_UPLOAD_TIMEOUT_S)
    int      upload_timeout_s = 5;     // LIG_UPLOAD_TIMEOUT_S  uploader thread: seconds after which a transfer that has not completed is given up on (the call makes the upload again with stream copies)
    bool     shard_uploader = false;   // LIG_SHARD_UPLOADER=1  lig_shard_rows_*: host rows / randomness rows through the uploader thread + stream waits (round 4; hangs with several processes per GPU)
    int      rands_upload_mode = 2;    // LIG_RANDS_UPLOAD_MODE  caller randomness rows from host: 2 uploader thread, 1: event-chained copies on the side stream (round 3)
    bool     spin_wait = true;         // LIG_SPIN_WAIT=0    the proof's host waits block in the runtime instead of polling
    int      spin_wait_ms = 50;        // LIG_SPIN_WAIT_MS   polling gives way to the blocking wait after this long
    bool     d2h_kernel = true;        // LIG_D2H_KERNEL=0   proof downloads by hipMemcpyAsync instead of copy kernels
    bool     shard_force_exchange = false;   // LIG_SHARD_FORCE_EXCHANGE  pack + all-to-all with one rank too (tests)
    bool     zres = false;             // LIG_ZRES=1         stage 1 keeps the encoder's Z tiles instead of codeword planes: K3 inside the column hash (round 6 A/B)
    bool     trace = false;            // LIG_TRACE          synchronised phase timeline on stderr
    int      fault_comm = 0;           // LIG_FAULT_COMM     tests: 1 = the stream-ordered all-to-all of the library's communicators
                                       //                    fails on first use, 2 = the host-synchronous one fails as well, 3 = the stream-ordered one hangs on the
                                       //                    host, 4 = (comm_ipc) rank 1 never raises its ready flag: a stall inside the GPU queues
    int      comm_timeout_s = 300;     // LIG_COMM_TIMEOUT_S  lig_shard_*: seconds the host waits for queued work with collectives before it calls lig_comm.abort
    int      ipc_host_s = 120;         // LIG_IPC_HOST_S     comm_ipc: seconds a rank waits on the host for a peer to reach (publish) the same collective
    int      ipc_stall_s = 120;        // LIG_IPC_STALL_S    comm_ipc watchdog: seconds without any flag changing, once every rank has published the collective, before it is declared dead
    std::string stream_map;            // LIG_STREAM_MAP     experiment: six digits, the physical stream of [main, side, copy] of even | odd contexts (lig_ctx_create)
    std::string stream_pad;            // LIG_STREAM_PAD     experiment: "even,odd[,order]" dummy streams before a context's streams / their creation order
    std::string rccl_lib;              // LIG_RCCL_LIB       the librccl to load instead of the one already mapped / found
};
const Knobs& knobs();
}  // namespace lig

namespace lig {

// Twiddle tables of one transform size (the reference's ntt_config_t + omega buffers,
// include/wgpu.hpp:54-61, src/webgpu/engine.cpp:1382-1503); all entries Montgomery form.
struct NttPlan {
    uint32_t N = 0, log2N = 0;
    fr* w = nullptr;      // w^i * R,  i < N/2   (device)
    fr* winv = nullptr;   // w^-i * R, i < N/2   (device)
    fr ninv;              // N^-1 * R
};

// Tables of the batched row-encode path (ntt_encode.hip) in 9 x 29-bit limbs: the per-element tables (stage twiddles, seams,
// twists) in windowed form (f29w: three shifted copies of every constant, fr29.hpp: f29_mulw), the radix-8 constants in
// Montgomery form (f29s, radix 2^261).
struct f29s;
struct EncodePlan {
    uint32_t k = 0, n = 0, log2k = 0;
    uint32_t A = 0, B = 0, log2B = 0;      // k = A * B, A = 8 outer radix, B = tile length
    f29wt tw_b{nullptr, 0, 0};       // DIT stage twiddles of the size-B forward transform (root psi^8, psi = w_n^4): span M' at M'/2-1
    f29wt tw_b_inv{nullptr, 0, 0};   // same for the inverse size-B transform (root w_k^-8)
    f29wt seam_inv{nullptr, 0, 0};   // w_k^(-i2*j1), [8][B]
    f29wt twist{nullptr, 0, 0};      // w_n^(r*(j1 + 8*i2)) / k, [3][8][B] for r = 1..3, inside a tile in THREAD order: entry q*B/4 + t belongs to
                                  // position i2 = brev(4t + q), the element thread t loads as its q-th (adjacent lanes, adjacent entries)
    f29wt seam_fwd{nullptr, 0, 0};   // psi^(i1*q2), [8][B]
    f29wt w8_inv{nullptr, 0, 0};     // powers of w_k^-(k/8): radix-8 constants (inverse), 8 entries
    f29wt w8_fwd{nullptr, 0, 0};     // powers of psi^(k/8): radix-8 constants (forward), 8 entries
    f29s* kinv = nullptr;       // k^-1, 1 entry
};

// Tables of the two-launch single-row transforms (ntt_tiled.hip): N = A * B, Montgomery radix 2^261, 48-byte entries.
struct TiledPlan {
    uint32_t N = 0, log2N = 0, log2A = 0, log2B = 0;
    f29s* tw_a = nullptr;       // DIT stage twiddles of the size-A transform (root w^B): span M' at M'/2-1
    f29s* tw_b = nullptr;       // same for the size-B transform (root w^A)
    f29s* mid = nullptr;        // [A][B]: w^(k1*n2), times N^-1 for an inverse transform
};
bool tiled_supported(uint32_t log2N);
void ntt_tiled(hipStream_t s, const TiledPlan& tp, const fr* in, size_t in_stride, fr* out, size_t out_stride, size_t rows, fr* scratch,
               uint32_t fold = 0);

// ---- ntt_generic.hip
void ntt_generic_forward(hipStream_t s, const NttPlan& pl, fr* buf, size_t rows, size_t row_stride);
void ntt_generic_inverse(hipStream_t s, const NttPlan& pl, fr* buf, size_t rows, size_t row_stride);
void ntt_generic_fold(hipStream_t s, fr* buf, uint32_t half, size_t rows, size_t row_stride);

// ---- ntt_encode.hip
bool encode_fast_supported(uint32_t k);
// ev0/ev1 (optional): HIP events recorded on `s` immediately before/after the dominant kernel (k_encode_tiles)
// mode ENC_FULL: out = rows x n codewords in the reference layout; msgs must not overlap out (K3 copies coset 0 of the
//   codeword from the message row).
// mode ENC_HALF: out = rows x k, only the values on the coset w_n^2 <w_n^4> (out[q] = P(w_n^(4q + 2))).
// mode ENC_PLANAR: out = rows x 3k, out[(r-1)*k + q] = P(w_n^(4q + r)) for r = 1, 2, 3; coset 0 of a codeword is its message
//   row reversed (codeword[4q] = msg[(k - q) mod k]) and is not stored.  CwView below addresses such a matrix by column.
// mode ENC_DOT: like ENC_HALF, but the k coset values of row r are not stored: their products with cw2[r] (rows of another
//   matrix on the same coset, cw2_stride elements apart) are added, per group of group_rows rows, to part[group][k]
//   (group partials in the format of k_rlc_partial's lin_part: plain values < 2p; `out` is unused).
// mode ENC_ZRES (round 6, LIG_ZRES): out = rows x 3k like ENC_PLANAR, but holding the Z tiles of the three computed cosets as the
//   tile kernel leaves them (element ((ci*8 + j1) << log2B) + q2 of a row, ci = coset - 1): the last radix-8 pass is NOT run; its
//   consumers (the column hash, the coset-2 dot, the column gather) take the radix-8 outputs they need from the tiles themselves.
// mode ENC_DOT with dot->cw2_z: cw2 rows are such Z tiles (of coset 2) instead of coset values.
enum { ENC_FULL = 0, ENC_HALF = 1, ENC_PLANAR = 2, ENC_DOT = 3, ENC_ZRES = 4 };
struct EncodeDot { const fr* cw2; size_t cw2_stride; uint32_t group_rows; fr* part; bool cw2_z = false; };
void encode_rows_fast(hipStream_t s, const EncodePlan& ep, const fr* msgs, fr* out, fr* scratch_y,
                      fr* scratch_z, size_t rows, hipEvent_t ev0, hipEvent_t ev1, int mode = ENC_FULL, const EncodeDot* dot = nullptr,
                      int phases = 15);      // phases: 1 = K1 only (into the Y scratch), 14 = the rest (from the Y scratch): a caller that runs K1 of the next chunk ahead

// The batched prover's resident codeword matrix: message rows (coset 0, reversed) + the three computed cosets as planes.
struct CwView {
    const fr* msgs;       // rows x k
    const fr* planes;     // rows x 3k
    uint32_t k;
    // codeword element `col` (0 <= col < 4k) of row `row`
    __host__ __device__ const fr* at(size_t row, uint32_t col) const {
        const uint32_t r = col & 3u, q = col >> 2;
        return r == 0 ? msgs + row * k + ((k - q) & (k - 1)) : planes + row * 3 * (size_t)k + (size_t)(r - 1) * k + q;
    }
};

// ---- eltwise.hip
void launch_eltwise(hipStream_t s, int op, const fr* x, const fr* y, fr* out, size_t count, fr scalar, uint32_t bit);
void launch_powmod(hipStream_t s, const fr* table32, const uint32_t* exp, const fr* coeff, fr* out, size_t count, int add);
void launch_gather_rows(hipStream_t s, const fr* cw, size_t row_stride, size_t rows, const uint32_t* idx, uint32_t count, fr* out);
void launch_gather_rows_planar(hipStream_t s, CwView cw, size_t rows, const uint32_t* idx, uint32_t count, fr* out);
// the same over a CwView whose `planes` are Z tiles (ENC_ZRES): every opened element of cosets 1..3 is one radix-8 output (ntt_encode.hip)
bool launch_gather_rows_z(hipStream_t s, const EncodePlan& ep, CwView cw, size_t rows, const uint32_t* idx, uint32_t count, fr* out);
void launch_rlc_rows(hipStream_t s, const fr* U, const fr* Rn, size_t rows, uint32_t n, const fr* rc_dev,
                     fr* code, fr* lin, const uint32_t* triples_dev, const fr* rq_dev, size_t n_triples, fr* quad);

// ---- sha.hip
struct ShaState {          // layout of the caller-owned device state buffer
    uint32_t* h;           // [8][n_inst]
    uint32_t* pend;        // [8][n_inst]  buffered element when an odd number of rows has been absorbed
};
void launch_sha_init(hipStream_t s, uint32_t* state, size_t n_inst);
// plane_k = 0: instance j = column j of `rows`.  plane_k = k: plane-major instances (j = r*k + q is column 4q + r), reading
// interleaved rows (msgs == nullptr) or a CwView {msgs, planes = rows} (sha.hip); launch_sha_final then needs the same plane_k
void launch_sha_update_rows(hipStream_t s, uint32_t* state, size_t n_inst, const fr* rows, size_t row_stride,
                            size_t nrows, uint64_t rows_before, uint32_t plane_k = 0, const fr* msgs = nullptr);
// rows whose cosets 1..3 are Z tiles (ENC_ZRES): the encoder's last radix-8 pass runs inside the hash's producer waves (sha.hip)
bool launch_sha_update_rows_z(hipStream_t s, uint32_t* state, const EncodePlan& ep, const fr* z, size_t nrows, uint64_t rows_before, const fr* msgs);
void launch_sha_final(hipStream_t s, const uint32_t* state, size_t n_inst, uint64_t rows_total, uint32_t* digests, uint32_t plane_k = 0);
void launch_merkle_build(hipStream_t s, const uint32_t* leaves, size_t n_leaves, uint32_t* nodes);

// ---- aes.hip
void launch_rng_fill(hipStream_t s, const uint32_t* rk60_dev, uint64_t first_elem, fr* out, size_t count);
void aes256_expand_host(const uint8_t key[32], uint32_t rk[60]);

}  // namespace lig
