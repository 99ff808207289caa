// prover.hip -- batched three-stage Ligero prover over a resident witness matrix, host side.
//
// MI355X-native counterpart of the orchestration in src/webgpu_prover.cpp:226-494 and the stage contexts of
// include/zkp/nonbatch_context.hpp (stage 1 :445-558, stage 2 :654-780, stage 3 :924-1000).  The reference
// streams one row at a time through the executor and re-runs the guest program (and therefore re-encodes every
// row) three times because a WebGPU device cannot hold the witness matrix.  With 288 GB of HBM the matrix is at
// rest: every message row is encoded ONCE in stage 1, the codewords (1 MiB per row) stay resident and are re-used
// by the stage-2 accumulators and the stage-3 column gather; only the dense stage-2 randomness rows need a second
// encode.  Transcript bytes (seeds, sample indices, Merkle decommitment, protobuf envelope) follow the reference
// byte for byte, so an unmodified verifier accepts the proof.
//
// The guest interpreter / constraint generator is out of scope; rows come from the synthetic
// constraint stream of BASELINE.md 3: n_linear witness slots + n_quad slots of x*y=z, one dense linear-test
// coefficient per witness.
#include "prover_common.hpp"
#include <cerrno>
#include <functional>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

namespace lig {
// Narrow rows -> message rows: element i < l of row r is the little-endian integer of widths[r] (4 / 8) bytes at
// packed + off[r] + i * widths[r]; slots l..k-1 are zeroed (their pads are drawn right after); a row of width 32 is copied.
__global__ void __launch_bounds__(256) k_expand_rows(const uint8_t* __restrict__ packed, const uint64_t* __restrict__ off, const uint8_t* __restrict__ widths,
                                                     size_t first_row, size_t rows, uint32_t l, uint32_t k, fr* __restrict__ out) {
    const size_t total = rows * k;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = first_row + e / k;
        const uint32_t i = (uint32_t)(e % k), w = widths[r];
        const uint8_t* src = packed + off[r];
        fr v = fr_zero();
        if (w == 32) v = fr_load(reinterpret_cast<const fr*>(src) + i);
        else if (i < l) {
            const uint32_t* p = reinterpret_cast<const uint32_t*>(src + (size_t)i * w);
            v.v[0] = p[0];
            if (w == 8) v.v[1] = p[1];
        }
        fr_store(out + r * k + i, v);
    }
}
}  // namespace lig

struct lig_trace {
    lig_ctx* c = nullptr;
    lig_synth_job job;                  // synthetic jobs only (n_linear / n_quad / witness_key); pointers cleared
    bool from_rows = false;             // rows supplied by the caller (lig_rows_*): no witness generation here
    bool loaded = false, committed = false;   // lig_rows_*: rows for the next commit are loaded / stage 1 done, proof pending
    bool dense_rands = false;           // lig_rows_job.dense_rands_per_row given: randomness rows may be generated here
    fr* msgs_alt = nullptr;             // second message matrix: the next trace is uploaded while the current one is proved
    bool alt_pending = false;           // the rows for the next commit are (arriving) in msgs_alt
    uint8_t encoding_seed[32] = {0}, program_hash[32] = {0}, ih[32] = {0};
    int64_t generated_at = 0;
    char version[17] = {0};
    std::vector<RowDesc> rows;          // committed non-mask rows in commit order
    std::vector<PadRun> pad_runs;       // pads drawn at commit time (pad_encoding_random), runs of consecutive rows
    uint64_t mask_pos = 0;              // encoding-stream position of the first mask element
    std::vector<std::pair<size_t, size_t>> sched1;     // stage-1 chunk schedule
    const uint8_t* host_msgs = nullptr; // lig_rows_begin with host memory: uploaded chunk by chunk under the encodes
    std::vector<hipEvent_t> ev_up;      // one per stage-1 chunk: its rows have arrived (per-context copy stream, LIG_UPLOAD_MODE=1)
    volatile uint32_t* up_flag = nullptr; uint32_t* up_flag_dev = nullptr;   // pinned: word ci = sequence number of the last upload whose chunk ci has arrived
    uint32_t up_seq = 0;                // sequence number of the last witness-rows upload (what stage 1 waits for)
    uint32_t rand_seq = 0;              // ... of the last randomness-rows upload (its own flag words, its own counter: a restart
                                        // between commit and prove must not move what the next commit waits for)
    bool up_by_thread = false;
    fr* rands_full = nullptr;           // lig_rows_push_rands: R x k, device resident
    uint64_t rands_pushed = 0;          // rows handed to the uploader so far (word up_words - 1 counts the rows that have ARRIVED)
    size_t up_words = 0;                // words in up_flag: [stage-1 chunks | stage-2 chunks: randomness rows arrived | ... consumed]
    bool push_sync = false;             // lig_rows_push_rands without an uploader thread: the pushed rows were copied synchronously
    std::vector<UploadJob> push_log;    // the pushes of the committed trace (their host rows stay valid until lig_rows_prove returns): what a retry copies again
    bool up_retry = false;              // prove_stage1 failed because a witness-rows transfer timed out (LIG_UPLOAD_TIMEOUT_S): lig_rows_commit makes the upload again
    bool leak = false;                  // an abandoned transfer of this trace is still pending and cannot be cancelled: its buffers are never freed or reused
    std::atomic<int> up_abort{0};       // a failed lig_rows_prove: the uploader drops the randomness-row copies it still holds
    std::atomic<int> up_pending{0};     // chunk copies of this trace the uploader thread still has to make
    std::atomic<int> up_failed{0}, rand_pending{0}, rand_failed{0};      // hipError_t of a chunk copy that failed (the chunk is published all the same: no stream may hang)   -- rand_*: the same for randomness-row uploads, kept apart: lig_rows_prove must not wait for (or swallow the error of) the NEXT trace's witness prefetch
    // narrow row format (lig_rows_job.elem_bytes): packed byte offset of every row (+1 entry), the widths, the device staging
    // area the packed rows are uploaded to (expanded into `msgs` chunk by chunk in stage 1)
    bool narrow = false;
    std::vector<uint64_t> src_off;
    std::vector<uint8_t> widths;
    uint64_t* src_off_dev = nullptr; uint8_t* widths_dev = nullptr; uint8_t* packed_dev = nullptr;
    lig_proof_info info1;               // stage-1 results kept between lig_rows_commit and lig_rows_prove
    size_t R = 0, RB = 0, n_init = 0;   // all rows, leading rows committed by the batch program, of those: init rows
    fr* msgs = nullptr;                 // R x k witness matrix (pads are re-drawn by every prove)
    bool zres = false;                  // LIG_ZRES: `cw` holds the encoder's Z tiles (lig::ENC_ZRES) instead of planes -- K3 runs inside the column hash,
                                        // stage 2 / 3 take single radix-8 outputs from the tiles (linear rows only: a trace with quadratic triples keeps planes)
    fr* cw = nullptr;                   // R x 3k: cosets 1..3 of every codeword as planes (lig::ENC_PLANAR), resident across the stages;
                                        // coset 0 of a codeword is its message row reversed and is read from `msgs` (lig::CwView)
    fr* maskcw = nullptr;               // 3 x n: the mask rows' codewords, reference layout
    fr* randb = nullptr;                // chunk x k randomness rows
    fr* rcw = nullptr;                  // chunk x n their codewords
    fr* acc = nullptr;                  // code | lin | quad | tmp   (4 x n)
    fr* parts = nullptr;                // 2 x groups x n partial accumulators
    fr* dots = nullptr;                 // one element: a device-side sum (mask closing slot, linear-test constant)
    fr* samples = nullptr;              // (R+3) x t
    uint32_t* sha_state = nullptr; uint32_t* leaves = nullptr; uint32_t* nodes = nullptr;
    uint32_t* tri_dev = nullptr;
    lig::f29s* coef_dev = nullptr;      // rc (R) | rq2 (T) | rq1 (T)
    std::vector<uint32_t> triples;
    uint8_t* h_proof = nullptr; size_t h_proof_cap = 0;   // pinned: the envelope is assembled here (owned by the trace)
    uint8_t* h_enc = nullptr;                              // pinned: 3 x n accumulators
    uint8_t* h_nodes = nullptr;                            // pinned: Merkle nodes
    hipEvent_t ev_ready[2] = {nullptr, nullptr}, ev_used[2] = {nullptr, nullptr};   // double-buffered randomness rows
    hipEvent_t ev_gate = nullptr, ev_k1 = nullptr, ev_acc[3] = {nullptr, nullptr, nullptr};
    uint8_t* h_small = nullptr;                            // pinned: the device-side sum (32 B) | 3 decoded accumulators (3 x n x 32)
};

// Host waits of a proof (root of stage 1, the three accumulators, the decodes, the envelope).  The runtime's blocking waits sleep on an
// interrupt: every wake-up costs tens of microseconds during which the GPU has nothing of this proof to run.  LIG_SPIN_WAIT=1 (default)
// polls instead (the calling thread yields between polls for LIG_SPIN_WAIT_MS, then falls back on the blocking wait): A/B in
// profiles/r05_spin_wait_ab.md.
static hipError_t wait_stream(hipStream_t st) {
    if (!lig::knobs().spin_wait) return hipStreamSynchronize(st);
    const auto t0 = clk::now();
    for (unsigned spins = 0;; spins++) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
        std::this_thread::yield();
        if ((spins & 255) == 255 && ms_since(t0) > (double)lig::knobs().spin_wait_ms) return hipStreamSynchronize(st);
    }
}
static hipError_t wait_event(hipEvent_t ev) {
    if (!lig::knobs().spin_wait) return hipEventSynchronize(ev);
    const auto t0 = clk::now();
    for (unsigned spins = 0;; spins++) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        std::this_thread::yield();
        if ((spins & 255) == 255 && ms_since(t0) > (double)lig::knobs().spin_wait_ms) return hipEventSynchronize(ev);
    }
}
static void uploader_drain(lig_trace* T);      // (below, with the uploader thread)
static void rand_drain(lig_trace* T);
static int ensure_up_flags(lig_ctx* c, lig_trace* T);

// The batch program on the device (lig_hip.h, lig_batch_op): k-element variables in a slab, every operation one eltwise
// kernel into a temporary + a copy (as vbn254fr_module does), every hook a device-to-device copy of the rows it names
// into the witness matrix.  The padding of an initialised variable comes from the encoding stream, 192 draws per init
// in program order (pad_encoding_random, nonbatch_context.hpp:497-510).
int lig_run_batch_program(lig_ctx* c, const lig_synth_job& job, fr* rows_out) {
    const uint32_t l = c->l, k = c->k, pad = k - l;
    hipStream_t s = c->stream;
    uint32_t nvars = 1;
    for (uint64_t i = 0; i < job.n_batch_ops; i++) {
        const lig_batch_op& o = job.batch_ops[i];
        nvars = std::max(nvars, std::max(o.out, std::max(o.x, o.y)) + 1);
        if (o.op == LIG_BOP_BIT_DECOMPOSE)
            for (uint32_t b = 0; b < o.len; b++) { uint32_t slot; std::memcpy(&slot, job.batch_data + o.data_off + 4ull * b, 4); nvars = std::max(nvars, (slot & 511u) + 1); }
    }
    fr* vars = nullptr; fr* tmp = nullptr;
    HIP_TRY(c, hipMalloc((void**)&vars, (size_t)nvars * k * sizeof(fr)));
    struct Free { fr*& a; fr*& b; lig_ctx* c; ~Free() { (void)hipStreamSynchronize(c->stream); (void)hipFree(a); (void)hipFree(b); } } guard{vars, tmp, c};
    HIP_TRY(c, hipMalloc((void**)&tmp, (size_t)k * sizeof(fr)));
    HIP_TRY(c, hipMemsetAsync(vars, 0, (size_t)nvars * k * sizeof(fr), s));
    uint32_t rk[60];
    lig::aes256_expand_host(job.encoding_seed, rk);
    HIP_TRY(c, hipMemcpyAsync(c->rk_dev, rk, sizeof rk, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    const size_t vb = (size_t)k * sizeof(fr);
    size_t r = 0, inits = 0;
    auto var = [&](uint32_t i) { return vars + (size_t)i * k; };
    auto commit = [&](const fr* src) -> int { HIP_TRY(c, hipMemcpyAsync(rows_out + (r++) * (size_t)k, src, vb, hipMemcpyDeviceToDevice, s)); return LIG_OK; };
    auto to_out = [&](uint32_t out) -> int { HIP_TRY(c, hipMemcpyAsync(var(out), tmp, vb, hipMemcpyDeviceToDevice, s)); return LIG_OK; };
    std::vector<uint8_t> stage;
    bool compat = false;                 // LIG_BOP_UPSTREAM_COMPAT seen: slices as upstream defines them (lig_hip.h)
    for (uint64_t i = 0; i < job.n_batch_ops; i++) {
        const lig_batch_op& o = job.batch_ops[i];
        const uint8_t* data = job.batch_data ? job.batch_data + o.data_off : nullptr;
        switch (o.op) {
            case LIG_BOP_UPSTREAM_COMPAT: compat = true; break;
            case LIG_BOP_SET: case LIG_BOP_SET_SCALAR: {
                const bool limbs = (o.reserved & LIG_BOP_F_WRITE_LIMBS) != 0;           // write_limbs family: write_buffer only
                const uint32_t len = o.op == LIG_BOP_SET ? o.len : l;
                stage.assign(32ull * len, 0);
                if (o.op == LIG_BOP_SET) std::memcpy(stage.data(), data, 32ull * o.len);
                else for (uint32_t e = 0; e < l; e++) std::memcpy(stage.data() + 32ull * e, data, 32);
                if (!limbs && !compat) HIP_TRY(c, hipMemsetAsync(var(o.x), 0, vb, s));  // write_buffer_clear, declared slice: the rest of x
                if (len) HIP_TRY(c, hipMemcpyAsync(var(o.x), stage.data(), 32ull * len, hipMemcpyHostToDevice, s));
                HIP_TRY(c, hipStreamSynchronize(s));                                    // `stage` is reused
                // write_buffer_clear as upstream runs it: clear_buffer(x.slice(len * 32)) after the write, the slice being
                // {offset len*32, size X + k*32 - len*32} of the slab: elements [len, x*k + k)
                if (!limbs && compat) HIP_TRY(c, hipMemsetAsync(vars + len, 0, ((size_t)o.x * k + k - len) * sizeof(fr), s));
                // on_batch_init's pad: x's own pad slots (declared) / slab element l = variable 0's pad slots (upstream)
                lig::launch_rng_fill_rows(s, c->rk_dev, (uint64_t)(inits++) * pad, compat ? vars : var(o.x), 1, pad, k, l, 1, pad);
                TRY(commit(var(o.x)));
                break;
            }
            case LIG_BOP_COPY:
                if (o.out != o.x) HIP_TRY(c, hipMemcpyAsync(var(o.out), var(o.x), vb, hipMemcpyDeviceToDevice, s));
                TRY(commit(var(o.out))); TRY(commit(var(o.x)));
                break;
            case LIG_BOP_ADD: TRY(lig_eltwise(c, LIG_OP_ADD, var(o.x), var(o.y), tmp, k, nullptr, 0)); TRY(to_out(o.out)); break;
            case LIG_BOP_SUB: TRY(lig_eltwise(c, LIG_OP_SUB, var(o.x), var(o.y), tmp, k, nullptr, 0)); TRY(to_out(o.out)); break;
            case LIG_BOP_MUL:
                TRY(lig_eltwise(c, LIG_OP_MUL, var(o.x), var(o.y), tmp, k, nullptr, 0));
                TRY(commit(var(o.x))); TRY(commit(var(o.y))); TRY(commit(tmp));
                TRY(to_out(o.out));
                break;
            case LIG_BOP_DIV:
                TRY(lig_eltwise(c, LIG_OP_DIV, var(o.x), var(o.y), tmp, k, nullptr, 0));
                TRY(commit(tmp)); TRY(commit(var(o.y))); TRY(commit(var(o.x)));
                TRY(to_out(o.out));
                break;
            case LIG_BOP_ADD_CONST: case LIG_BOP_SUB_CONST: case LIG_BOP_CONST_SUB: case LIG_BOP_MUL_CONST: case LIG_BOP_MONTMUL_CONST: {
                static const int map[5] = {LIG_OP_ADD_CONST, LIG_OP_SUB_CONST, LIG_OP_CONST_SUB, LIG_OP_MUL_CONST, LIG_OP_MONTMUL_CONST};
                TRY(lig_eltwise(c, map[o.op - LIG_BOP_ADD_CONST], var(o.x), nullptr, tmp, k, data, 0));
                TRY(to_out(o.out));
                break;
            }
            case LIG_BOP_ASSERT_EQUAL: TRY(commit(var(o.x))); TRY(commit(var(o.y))); break;
            case LIG_BOP_BIT_DECOMPOSE:
                for (uint32_t b = 0; b < o.len; b++) {
                    uint32_t slot;
                    std::memcpy(&slot, data + 4ull * b, 4);
                    slot &= 511u;
                    TRY(lig_eltwise(c, LIG_OP_BIT_DECOMPOSE, var(o.x), nullptr, tmp, k, nullptr, b));
                    TRY(to_out(slot));
                    TRY(commit(var(slot)));
                }
                break;
            case LIG_BOP_FREE: HIP_TRY(c, hipMemsetAsync(var(o.x), 0, vb, s)); break;
            default: break;
        }
    }
    HIP_TRY(c, hipStreamSynchronize(s));
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}

int lig_internal_synth_witness(lig_ctx* c, const uint8_t witness_key[32], const std::vector<RowDesc>& rows, size_t first, fr* msgs) {
    const uint32_t k = c->k;
    uint32_t rk[60];
    lig::aes256_expand_host(witness_key, rk);
    TRY(lig_internal_upload_small(c, c->rk_dev, rk, sizeof rk, c->stream));
    uint64_t pos = 0;
    const size_t R = rows.size();
    for (size_t r = first; r < R;) {
        const RowDesc d = rows[r];
        if (d.kind == 0) {                      // run of linear rows with the same fill
            size_t run = 1;
            while (r + run < R && rows[r + run].kind == 0 && rows[r + run].data == d.data) run++;
            lig::launch_rng_fill_rows(c->stream, c->rk_dev, pos, msgs + r * k, run, d.data, k, 0, 1, d.data);
            pos += (uint64_t)run * d.data; r += run;
        } else {                                 // x, y, z triple
            lig::launch_rng_fill_rows(c->stream, c->rk_dev, pos, msgs + r * k, 2, d.data, k, 0, 1, d.data);
            pos += 2ull * d.data;
            lig::launch_eltwise(c->stream, LIG_OP_MUL, msgs + r * k, msgs + (r + 1) * k, msgs + (r + 2) * k, d.data, fr{}, 0);
            r += 3;
        }
    }
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// shared by lig_synth_* and lig_rows_*: buffers of a trace whose row plan (T->rows) is known
static int trace_alloc(lig_ctx* c, lig_trace* T) {
    const uint32_t k = c->k, n = c->n, t = 192;
    const size_t R = T->R = T->rows.size();
    T->triples = quad_terms(T->rows);
    T->zres = lig::knobs().zres && c->fast && T->triples.empty();
    const size_t chunk = lig_tune::CHUNK, groups = (chunk + lig_tune::GROUP - 1) / lig_tune::GROUP;
    auto dm = [&](void** p, size_t bytes) -> int { HIP_TRY(c, hipMalloc(p, bytes ? bytes : 16)); HIP_TRY(c, hipMemsetAsync(*p, 0, bytes, c->stream)); return LIG_OK; };
    TRY(dm((void**)&T->msgs, (R ? R : 1) * (size_t)k * 32));
    TRY(dm((void**)&T->cw, (R ? R : 1) * 3 * (size_t)k * 32));
    TRY(dm((void**)&T->maskcw, 3 * (size_t)n * 32));
    TRY(dm((void**)&T->randb, 2 * chunk * (size_t)k * 32));          // double-buffered
    TRY(dm((void**)&T->rcw, chunk * (size_t)n * 32));
    TRY(dm((void**)&T->acc, 4 * (size_t)n * 32));
    TRY(dm((void**)&T->parts, (2 * groups * (size_t)n + (chunk + lig_tune::DOT_GROUP - 1) / lig_tune::DOT_GROUP * k) * 32));
    TRY(dm((void**)&T->dots, 32));
    TRY(dm((void**)&T->samples, (R + 3) * (size_t)t * 32));
    TRY(dm((void**)&T->sha_state, lig_sha_state_bytes(n)));
    TRY(dm((void**)&T->leaves, (size_t)n * 32));
    TRY(dm((void**)&T->nodes, lig_merkle_nodes(n) * 32));
    TRY(dm((void**)&T->tri_dev, (T->triples.size() ? T->triples.size() : 1) * sizeof(uint32_t)));
    TRY(dm((void**)&T->coef_dev, (R + 2 * T->triples.size() / 3 + 1) * sizeof(lig::f29s)));
    T->h_proof_cap = (size_t)1 << 19;
    T->h_proof_cap += 3 * (size_t)n * 32 + (R + 3) * (size_t)t * 32;
    HIP_TRY(c, hipHostMalloc((void**)&T->h_proof, T->h_proof_cap, hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc((void**)&T->h_enc, 3 * (size_t)n * 32, hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc((void**)&T->h_nodes, lig_merkle_nodes(n) * 32, hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc((void**)&T->h_small, (1 + 3 * (size_t)n) * 32, hipHostMallocDefault));
    for (int i = 0; i < 2; i++) {
        HIP_TRY(c, hipEventCreateWithFlags(&T->ev_ready[i], hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&T->ev_used[i], hipEventDisableTiming));
    }
    HIP_TRY(c, hipEventCreateWithFlags(&T->ev_gate, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&T->ev_k1, hipEventDisableTiming));
    for (int a3 = 0; a3 < 3; a3++) HIP_TRY(c, hipEventCreateWithFlags(&T->ev_acc[a3], hipEventDisableTiming));
    if (!T->triples.empty()) HIP_TRY(c, hipMemcpyAsync(T->tri_dev, T->triples.data(), T->triples.size() * 4, hipMemcpyHostToDevice, c->stream));
    {   // a short first chunk (the column hash -- the longest chain of stage 1 -- starts after 128 rows instead of 512) and a short
        // last one (the hash tail after the last encode); LIG_S1_HEAD / LIG_S1_TAIL override for experiments
        // (profiles/r02_stage1_schedule_ab.md: head 128 is +3 % proofs/s with two proofs in flight; one stage 1 at a time -- a
        // process-wide lock around this stage -- measured 3-5 % slower)
        T->sched1 = chunk_schedule(R, lig_tune::CHUNK, lig::knobs().s1_head, lig::knobs().s1_tail);
    }
    TRY(lig_internal_reserve_scratch(c, R < lig_tune::CHUNK ? (R ? R : 1) : lig_tune::CHUNK));     // sized once: never re-allocated under a running stream
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return LIG_OK;
}

// instance_hash over arg0 = "Ligero\0" and the public arguments (src/webgpu_prover.cpp:110-168)
static bool instance_hash_of(const uint8_t* args, const uint64_t* lens, uint64_t n_args, uint8_t out[32]) {
    if (n_args && (!args || !lens)) return false;
    std::memset(out, 0, 32);
    Sha256().add(out, 32).add("Ligero", 7).finish(out);
    for (uint64_t i = 0; i < n_args; i++) {
        uint8_t prev[32];
        std::memcpy(prev, out, 32);
        Sha256().add(prev, 32).add(args, lens[i]).finish(out);
        args += lens[i];
    }
    return true;
}

// ================= stage 1: row forming (pads + masks from the encoding stream), encode, column hash, Merkle root
static int prove_stage1(lig_trace* T, lig_proof_info* info, const std::function<void(const char*)>& mark) {
    lig_ctx* c = T->c;
    const uint32_t l = c->l, k = c->k, n = c->n, pad = k - l;
    const size_t k3 = 3 * (size_t)k;
    hipStream_t s = c->stream;
    uint32_t rk[60];
    lig::aes256_expand_host(T->encoding_seed, rk);
    TRY(lig_internal_upload_small(c, c->rk_dev, rk, sizeof rk, s));
    const bool streamed = T->host_msgs != nullptr;       // rows arrive chunk by chunk: their pads are drawn per chunk below
    if (!streamed)
        for (const PadRun& pr : T->pad_runs)             // pad_encoding_random of every row that draws at commit time
            lig::launch_rng_fill_rows(s, c->rk_dev, pr.pos, T->msgs + pr.first * (size_t)k, pr.count, pad, k, l, 1, pad);
    mark("  pads");
    uint64_t epos = T->mask_pos;
    fr* mask = T->maskcw;                                                                           // the 3 mask rows are formed in place
    HIP_TRY(c, hipMemsetAsync(mask, 0, 3 * (size_t)n * 32, s));
    lig::launch_rng_fill_rows(s, c->rk_dev, epos, mask, 1, l, 0, 0, 1, 0); epos += l;               // code mask: l randoms, zeros to k
    fr* mlin = mask + n; fr* mquad = mask + 2 * (size_t)n;
    lig::launch_rng_fill_rows(s, c->rk_dev, epos, mlin, 1, l - 1, 0, 1, 2, 0); epos += l - 1;       // (0, r) x (l-1)
    // last odd slot = -(sum of the others) (witness_manager.hpp:283-297), on the device: no host round trip
    lig::launch_sum_elems(s, mlin + 1, l - 1, 2, T->dots, mlin + 2 * (size_t)(l - 1) + 1);
    lig::launch_rng_fill_rows(s, c->rk_dev, epos, mlin, 1, 2 * pad, 0, 2 * l, 1, 0); epos += 2 * pad;
    lig::launch_rng_fill_rows(s, c->rk_dev, epos, mquad, 1, l, 0, 1, 2, 0); epos += l;              // (0, r) x l
    lig::launch_rng_fill_rows(s, c->rk_dev, epos, mquad, 1, 2 * pad, 0, 2 * l, 1, 0); epos += 2 * pad;

    mark("row forming (pads, masks)");
    // Encode chunk by chunk on the main stream; the column hash of chunk b runs on the side stream while chunk
    // b+1 is being encoded (the hash has only n = 32768 lanes of parallelism -- 512 waves -- and would otherwise
    // leave most of the chip idle for its whole duration).  Row order = hash order is preserved by stream order.
    hipStream_t s2 = c->stream2;
    // (dedicated CUs for the hash via CU-masked streams were measured with one proof: no gain, profiles/r01_overlap_experiments.md;
    // LIG_SHA_CUMASK: disjoint halves for the hashes of two proofs in flight, profiles/r04_sha_cumask_ab.md)
    hipStream_t s_sha = c->stream_sha ? c->stream_sha : s2, s_enc = s;
    TRY(lig_sha_init(c, T->sha_state, n));
    HIP_TRY(c, hipEventRecord(c->ev_fork, s));
    HIP_TRY(c, hipStreamWaitEvent(s_sha, c->ev_fork, 0));
    // The three mask rows do not depend on the witness: their transforms run on the side stream under the
    // first chunk's encode (the hash of chunk 0 is queued behind them and has to wait for that encode anyway).
    TRY(lig_internal_encode_generic(c, mask, s_sha));
    TRY(lig_internal_encode_2k_rows(c, mlin, 2, s_sha));      // mlin and mquad are adjacent rows: one pass
    uint64_t absorbed = 0;
    size_t pr_i = 0;
    // what has to be on the encode stream in front of chunk ci's first kernel: its rows have arrived, are expanded, have their pads
    auto pre = [&](size_t ci, hipStream_t s_enc) -> int {
        const size_t b = T->sched1[ci].first, nb = T->sched1[ci].second - b;
        if (streamed) {
            if (T->up_by_thread) HIP_TRY(c, hipStreamWaitValue32(s_enc, T->up_flag_dev + ci, T->up_seq, hipStreamWaitValueGte, 0xffffffffu));
            else HIP_TRY(c, hipStreamWaitEvent(s_enc, T->ev_up[ci], 0));                   // this chunk's rows have arrived
            if (T->narrow) hipLaunchKernelGGL(lig::k_expand_rows, dim3((uint32_t)std::min<size_t>((nb * k + 255) / 256, 4096)), dim3(256), 0, s_enc, T->packed_dev,
                                              T->src_off_dev, T->widths_dev, b, nb, l, k, T->msgs);
            for (; pr_i < T->pad_runs.size() && T->pad_runs[pr_i].first < b + nb; pr_i++) {      // runs never straddle chunks (split in begin)
                const PadRun& pr = T->pad_runs[pr_i];
                lig::launch_rng_fill_rows(s_enc, c->rk_dev, pr.pos, T->msgs + pr.first * (size_t)k, pr.count, pad, k, l, 1, pad);
            }
        }
        return LIG_OK;
    };
    const int enc_mode = T->zres ? lig::ENC_ZRES : lig::ENC_PLANAR;
    const int gate = lig::knobs().sha_gate;
    // LIG_SHA_GATE=2 (round 6): K1 of chunk ci+1 is queued BEHIND the last kernel of chunk ci and runs alone -- the hash of chunk ci
    // waits for it (K1 is latency-bound: next to the hash waves it takes 2.2x as long, on the critical path of a lone proof) --
    // then the hash is placed, then the tile kernel of chunk ci+1 goes on.  (One Y scratch: K1 of ci+1 starts after chunk ci is done with it.)
    const bool k1_ahead = gate == 2 && c->fast && lig_tune::CHUNK <= lig::knobs().encode_chunk;
    if (k1_ahead && !T->sched1.empty()) {
        TRY(pre(0, s_enc));
        TRY(lig_internal_encode_rows(c, T->msgs, T->cw, T->sched1[0].second - T->sched1[0].first, enc_mode, s_enc, 1));
    }
    for (size_t ci = 0; ci < T->sched1.size(); ci++) {
        const size_t b = T->sched1[ci].first, nb = T->sched1[ci].second - b;
        if (!k1_ahead) TRY(pre(ci, s_enc));
        TRY(lig_internal_encode_rows(c, T->msgs + b * k, T->cw + b * k3, nb, enc_mode, s_enc, k1_ahead ? 14 : 15));
        HIP_TRY(c, hipEventRecord(c->ev_fork, s_enc));
        HIP_TRY(c, hipStreamWaitEvent(s_sha, c->ev_fork, 0));
        if (k1_ahead && ci + 1 < T->sched1.size()) {
            const size_t b1 = T->sched1[ci + 1].first, nb1 = T->sched1[ci + 1].second - b1;
            TRY(pre(ci + 1, s_enc));
            TRY(lig_internal_encode_rows(c, T->msgs + b1 * k, T->cw + b1 * k3, nb1, enc_mode, s_enc, 1));
            HIP_TRY(c, hipEventRecord(T->ev_k1, s_enc));
            HIP_TRY(c, hipStreamWaitEvent(s_sha, T->ev_k1, 0));
        }
        const size_t g = lig::knobs().sha_gate_rows;
        if (T->zres) {
            // (same placement rule as below; the gate is a whole batch of the in-hash butterflies: 8 rows)
            const size_t gz = g < 8 ? 8 : g;
            if (gate && nb > 2 * gz) {
                lig::launch_sha_update_rows_z(s_sha, T->sha_state, c->ep, T->cw + b * k3, gz, absorbed, T->msgs + b * k);
                HIP_TRY(c, hipEventRecord(T->ev_gate, s_sha));
                lig::launch_sha_update_rows_z(s_sha, T->sha_state, c->ep, T->cw + (b + gz) * k3, nb - gz, absorbed + gz, T->msgs + (b + gz) * k);
                HIP_TRY(c, hipStreamWaitEvent(s_enc, T->ev_gate, 0));
            } else {
                lig::launch_sha_update_rows_z(s_sha, T->sha_state, c->ep, T->cw + b * k3, nb, absorbed, T->msgs + b * k);
            }
        } else if (gate && nb > 2 * g) {
            // the hash waves must be placed while the chip is idle (one per SIMD, evenly): hash the first two rows, let the
            // encode stream wait for that, and queue the rest of the chunk right behind it on the hash stream
            lig::launch_sha_update_rows(s_sha, T->sha_state, n, T->cw + b * k3, 0, g, absorbed, k, T->msgs + b * k);
            HIP_TRY(c, hipEventRecord(T->ev_gate, s_sha));
            lig::launch_sha_update_rows(s_sha, T->sha_state, n, T->cw + (b + g) * k3, 0, nb - g, absorbed + g, k, T->msgs + (b + g) * k);
            HIP_TRY(c, hipStreamWaitEvent(s_enc, T->ev_gate, 0));
        } else {
            lig::launch_sha_update_rows(s_sha, T->sha_state, n, T->cw + b * k3, 0, nb, absorbed, k, T->msgs + b * k);
        }
        absorbed += nb;
    }
    mark("encode message rows");
    HIP_TRY(c, hipEventRecord(c->ev_fork, s));
    HIP_TRY(c, hipStreamWaitEvent(s_sha, c->ev_fork, 0));
    lig::launch_sha_update_rows(s_sha, T->sha_state, n, mask, n, 3, absorbed, k, nullptr);     // same plane-major instances, interleaved rows
    absorbed += 3;
    c->sha[T->sha_state].second = absorbed;
    HIP_TRY(c, hipEventRecord(c->ev_join, s_sha));
    HIP_TRY(c, hipStreamWaitEvent(s, c->ev_join, 0));
    mark("encode mask rows + column sha tail");
    uint32_t* leaf_level = T->nodes + 8 * ((size_t)n - 1);                    // n = 4k is a power of two: the leaves ARE the last level of the heap
    lig::launch_sha_final(s, T->sha_state, n, absorbed, leaf_level, k);      // plane-major instances -> leaves in column order, written in place
    TRY(lig_merkle_build(c, leaf_level, n, T->nodes));
    TRY(lig_internal_download(c, T->h_nodes, T->nodes, 32, s));              // the root now ...
    HIP_TRY(c, wait_stream(s));
    if (streamed && T->up_by_thread) {
        // every chunk has been waited for by now; a copy the uploader thread could not make leaves garbage rows behind
        if (const int e = T->up_failed.exchange(0)) {
            T->up_retry = e == (int)hipErrorLaunchTimeOut;
            FAIL(c, LIG_E_HIP, std::string("rows upload failed: ") + hipGetErrorString((hipError_t)e));
        }
    }
    std::memcpy(info->root, T->h_nodes, 32);
    TRY(lig_internal_download(c, T->h_nodes, T->nodes, lig_merkle_nodes(n) * 32, s));      // ... the tree for the decommitment (stage 3) under stage 2
    Sha256().add("LigetronStage1", 15).add(info->root, 32).add(T->ih, 32).finish(info->stage1_seed);
    mark("merkle + seed");
    return LIG_OK;
}

// where the stage-2 randomness rows come from: generated (dense rows of the synthetic stream, from the linear stream
// keyed by the stage-1 seed) or supplied by the caller (device pointer used in place / host rows uploaded chunk-wise)
struct RandSource { const fr* dev = nullptr; const uint8_t* host = nullptr; bool pushed = false; };

// ================= stage 2 + 3
static int prove_stage23(lig_trace* T, const RandSource& rs, const uint8_t* const_sum_given, const uint8_t** proof, size_t* proof_len,
                         lig_proof_info* info, const std::function<void(const char*)>& mark) {
    lig_ctx* c = T->c;
    const uint32_t l = c->l, k = c->k, n = c->n, t = 192;
    const size_t R = T->R;
    hipStream_t s = c->stream, s2 = c->stream2;
    auto t0 = clk::now();
    uint32_t rk[60];
    // ================= stage 2: code / linear / quadratic accumulators over the resident codewords
    const size_t NT = T->triples.size() / 3;
    lig::aes256_expand_host(info->stage1_seed, rk);           // key of the code / linear / quadratic streams (three engines, same key)
    TRY(lig_internal_upload_small(c, c->rk_dev, rk, sizeof rk, s));
    // Accumulators.  All three tests are sums of low-degree polynomials, so they are accumulated where they are
    // cheapest and extended to the n evaluation points once per proof (exact field arithmetic => same values as the
    // reference's per-row n-point updates, nonbatch_context.hpp:756-780):
    //   code  = sum_r rc_r * U_r          degree < k : combine the MESSAGE rows (k values per row), encode once;
    //   lin   = sum_r U_r o R_r           degree < 2k: accumulate on the order-2k subgroup <w_n^2> (the even codeword
    //   quad  = sum_t rq_t (X o Y - Z)                 positions), then INTT_2k + NTT_n once.
    // Half of <w_n^2> is the message domain itself (w_n^4 = w_k^-1), where U_r and R_r are the message and randomness rows
    // as given; so a randomness row is only ever evaluated on ONE extra coset (w_n^2 <w_n^4>, k points): linH accumulates
    // msg_r o rand_r in message order (fused with the code test, same pass over the message rows), linC accumulates
    // codeword coset 2 against the coset-2 values of the randomness rows, and the two are interleaved once per proof.
    fr* code = T->acc; fr* lin = T->acc + n; fr* quad = T->acc + 2 * (size_t)n; fr* tmp = T->acc + 3 * (size_t)n;
    fr* linH = lin + 2 * (size_t)k; fr* linC = lin + 3 * (size_t)k;
    // group partials of the three k-column accumulators: group g of every chunk adds into slot g, combined once at the end
    const size_t groups = (lig_tune::CHUNK + lig_tune::GROUP - 1) / lig_tune::GROUP;
    fr* p_code = T->parts; fr* p_linH = T->parts + groups * (size_t)n; fr* p_linC = T->parts + 2 * groups * (size_t)n;
    const size_t dot_groups = (lig_tune::CHUNK + lig_tune::DOT_GROUP - 1) / lig_tune::DOT_GROUP;      // p_linC: dot_groups x k
    HIP_TRY(c, hipMemsetAsync(T->parts, 0, (2 * groups * (size_t)n + dot_groups * k) * 32, s));
    HIP_TRY(c, hipMemsetAsync(T->acc, 0, 4 * (size_t)n * 32, s));
    fr* rhalf = T->rcw;                                   // chunk x 2k
    // The randomness rows of chunk b+1 (AES sampling: LDS-bound; or the upload of the caller's rows) are formed on the side
    // stream, double-buffered, while the main stream encodes / accumulates chunk b (VALU-bound).
    // first chunk 192 rows (the encode stream waits for the first randomness rows); LIG_S2_HEAD overrides for experiments
    // (profiles/r02_stage1_schedule_ab.md: 192-256 rows +3 % proofs/s over 96 with two proofs in flight)
    const size_t s2_head = lig::knobs().s2_head;
    const std::vector<std::pair<size_t, size_t>> sched2 = chunk_schedule(R, lig_tune::CHUNK, s2_head, 0);
    const size_t n_chunks = sched2.size();
    std::vector<uint64_t> chunk_pos(n_chunks + 1, 0);
    for (size_t ci = 0; ci < n_chunks; ci++) {
        uint64_t cnt = 0;
        for (size_t r = sched2[ci].first; r < sched2[ci].second; r++) cnt += T->rows[r].data;
        chunk_pos[ci + 1] = chunk_pos[ci] + cnt;
    }
    auto rand_buf = [&](size_t ci) -> fr* { return rs.dev ? const_cast<fr*>(rs.dev) + sched2[ci].first * (size_t)k : T->randb + (ci & 1) * lig_tune::CHUNK * (size_t)k; };
    // Generated (dense) randomness rows: the sampler also accumulates the message-domain halves of the code and linear tests
    // while the elements are in registers (aes.hip: k_rand_rlc) -- all of it on the side stream; the main stream only encodes.
    const bool fused_rlc = !rs.dev && !rs.host && (k % 256 == 0) && lig::knobs().fused_rlc;
    const bool early_code = lig::knobs().early_code;      // (see below)
    const lig::f29s* rc_loop = early_code ? nullptr : T->coef_dev;      // code coefficients of the row loop (null: accumulated up front)
    // Caller rows in HOST memory (what a constraint generator delivers, nonbatch_context.hpp:654-780 / mpz_vector.hpp:108-127) take
    // the road of the witness rows (rows_load): the uploader thread copies chunk after chunk into the double buffer and
    // publishes each arrival in pinned host memory, the main stream waits for that word in its own queue and writes a second word
    // when it has consumed a chunk -- which is what the uploader waits for before it overwrites that half of the buffer.  No copy,
    // event or barrier packet of this transfer ever sits in a queue of the proof (the event-chained copy
    // on the side stream that this replaces is LIG_UPLOAD_MODE=1, profiles/r04_caller_rands_ab.md).
    const bool rands_by_thread = rs.host && lig::knobs().upload_mode == 2 && lig::knobs().rands_upload_mode == 2 && lig_internal_uploader_available(c) && n_chunks;
    uint32_t rseq = 0;
    size_t rflag0 = 0, uflag0 = 0;
    if (rands_by_thread) {
        TRY(ensure_up_flags(c, T));
        rflag0 = T->sched1.size(); uflag0 = rflag0 + n_chunks;
        if (uflag0 + n_chunks > T->up_words) FAIL(c, LIG_E_STATE, "rows job: flag page too small for the stage-2 schedule");
        rseq = ++T->rand_seq;
        T->up_abort.store(0, std::memory_order_release);
        std::vector<UploadJob> jobs;
        for (size_t ci = 0; ci < n_chunks; ci++) {
            const size_t b = sched2[ci].first, nb = sched2[ci].second - b;
            UploadJob j{(uint8_t*)rand_buf(ci), rs.host + b * (size_t)k * 32, nb * (size_t)k * 32, T->up_flag + rflag0 + ci, rseq, &T->rand_failed};
            if (ci >= 2) { j.wait = T->up_flag + uflag0 + ci - 2; j.wait_val = rseq; }
            j.abort = &T->up_abort; j.prio = 1;
            jobs.push_back(j);
        }
        lig_internal_uploader_submit(c->device, jobs, &T->rand_pending);
    }
    auto form_rand_chunk = [&](size_t ci) -> int {        // enqueued on the side stream
        const size_t b = sched2[ci].first, nb = sched2[ci].second - sched2[ci].first;
        fr* rb = rand_buf(ci);
        if (rs.pushed) return LIG_OK;                                                           // arriving in T->rands_full (lig_rows_push_rands)
        if (rs.dev) { HIP_TRY(c, hipEventRecord(T->ev_ready[ci & 1], s2)); return LIG_OK; }     // used in place
        if (rands_by_thread) return LIG_OK;                                                     // the uploader thread brings them
        if (ci >= 2) HIP_TRY(c, hipStreamWaitEvent(s2, T->ev_used[ci & 1], 0));      // buffer free again
        if (rs.host) {
            HIP_TRY(c, hipMemcpyAsync(rb, rs.host + b * (size_t)k * 32, nb * (size_t)k * 32, hipMemcpyHostToDevice, s2));
        } else {
            uint64_t lpos = chunk_pos[ci];
            for (size_t r = 0; r < nb;) {          // dense linear-test coefficients: one draw per witness slot, commit order; zeros after
                size_t run = 1;
                const uint32_t d = T->rows[b + r].data;
                while (r + run < nb && T->rows[b + r + run].data == d) run++;
                if (fused_rlc) lig::launch_rand_rlc(s2, c->rk_dev, lpos, rb + r * k, T->msgs + (b + r) * k, run, d, k, rc_loop ? rc_loop + b + r : nullptr, lig_tune::GROUP / 4, p_code, p_linH);
                else lig::launch_rng_fill_rows_dense(s2, c->rk_dev, lpos, rb + r * k, run, d, k);
                lpos += (uint64_t)run * d; r += run;
            }
        }
        HIP_TRY(c, hipEventRecord(T->ev_ready[ci & 1], s2));
        return LIG_OK;
    };
    {   // coefficients: one code-stream draw per row, one quadratic-stream draw per triple (the sampler's fused pass reads them)
        std::vector<H::Fr> rc, rq;
        FieldStream code_s(info->stage1_seed), quad_s(info->stage1_seed);
        size_t n_code = 0;
        for (size_t r = 0; r < R; r++) n_code += has_code_check(T->rows[r].kind);
        code_s.next(n_code, rc);
        quad_s.next(NT, rq);
        std::vector<lig::f29s> coef(R + 2 * NT + 1);
        std::memset(coef.data(), 0, coef.size() * sizeof(lig::f29s));
        const H::Fr R261sq = H::mul(R261, R261);
        for (size_t r = 0, ci = 0; r < R; r++) if (has_code_check(T->rows[r].kind)) coef[r] = to_f29s_host(rc[ci++], R261);
        for (size_t i = 0; i < NT; i++) { coef[R + i] = to_f29s_host(rq[i], R261sq); coef[R + NT + i] = to_f29s_host(rq[i], R261); }
        TRY(lig_internal_upload_small(c, T->coef_dev, coef.data(), coef.size() * sizeof(lig::f29s), s));
    }
    HIP_TRY(c, hipEventRecord(c->ev_fork, s));            // the side stream starts after the key / coefficient uploads and the memsets above
    HIP_TRY(c, hipStreamWaitEvent(s2, c->ev_fork, 0));
    if (n_chunks) TRY(form_rand_chunk(0));
    // (the side stream is already sampling the first randomness rows while the main stream does this)
    // The code-test accumulator does not depend on the randomness rows: it is formed FIRST (one pass over the message rows, one
    // encode, mask, download), so that the host can absorb it into the stage-2 seed hash -- 1 MiB of the 3 MiB sequential
    // SHA-256 that is the longest host step of a proof -- while the GPU is still busy with the row loop below.
    // LIG_EARLY_CODE=0: accumulate it inside the row loop as before (profiles/r03_early_code_ab.md).
    fr* mask = T->maskcw; fr* mlin = mask + n; fr* mquad = mask + 2 * (size_t)n;
    uint8_t* enc = T->h_enc;
    const size_t vec_bytes = (size_t)n * 32;
    if (early_code) {
        for (size_t b = 0; b < R; b += lig_tune::CHUNK) {
            const size_t nb = std::min<size_t>(lig_tune::CHUNK, R - b);
            lig::launch_rlc_accumulate29(s, T->msgs + b * k, k, 1, nullptr, 0, nb, k, T->coef_dev + b, p_code, nullptr, lig_tune::GROUP / 4);
        }
        lig::launch_rlc_combine(s, tmp, p_code, (uint32_t)((lig_tune::CHUNK + lig_tune::GROUP / 4 - 1) / (lig_tune::GROUP / 4)), k);
        TRY(lig_internal_encode_rows(c, tmp, code, 1, false));      // out of place: no device-to-device staging copy
        lig::launch_eltwise(s, LIG_OP_ADD_ASSIGN, mask, nullptr, code, n, fr{}, 0);
        TRY(lig_internal_download(c, enc, code, vec_bytes, s));
        HIP_TRY(c, hipEventRecord(T->ev_acc[0], s));
    }
    for (size_t ci = 0; ci < n_chunks; ci++) {
        const size_t b = sched2[ci].first, nb = sched2[ci].second - sched2[ci].first;
        fr* rb = rand_buf(ci);
        if (ci + 1 < n_chunks) TRY(form_rand_chunk(ci + 1));
        if (rs.pushed && !T->push_sync) HIP_TRY(c, hipStreamWaitValue32(s, T->up_flag_dev + T->up_words - 1, (uint32_t)sched2[ci].second, hipStreamWaitValueGte, 0xffffffffu));
        else if (rands_by_thread) HIP_TRY(c, hipStreamWaitValue32(s, T->up_flag_dev + rflag0 + ci, rseq, hipStreamWaitValueGte, 0xffffffffu));
        else HIP_TRY(c, hipStreamWaitEvent(s, T->ev_ready[ci & 1], 0));
        if (c->fast) {
            // coset-2 values of the randomness rows times the coset-2 plane of the codewords, summed per group of rows inside the
            // encoder's output kernel: the values themselves are never written
            TRY(lig_internal_encode_dot(c, rb, nb, T->cw + b * 3 * (size_t)k + k, 3 * (size_t)k, lig_tune::DOT_GROUP, p_linC, nullptr, T->zres));
        } else {
            TRY(lig_internal_encode_rows(c, rb, rhalf, nb, lig::ENC_HALF));
            lig::launch_rlc_accumulate29(s, T->cw + b * 3 * (size_t)k + k, 3 * (size_t)k, 1, rhalf, k, nb, k, nullptr, nullptr, p_linC, lig_tune::DOT_GROUP);
        }
        // k columns per pass: groups of 16 rows (4x more workgroups than the n-column grouping; same partial-sum space)
        if (!fused_rlc) lig::launch_rlc_accumulate29(s, T->msgs + b * k, k, 1, rb, k, nb, k, rc_loop ? rc_loop + b : nullptr, p_code, p_linH, lig_tune::GROUP / 4);
        if (rands_by_thread) HIP_TRY(c, hipStreamWriteValue32(s, T->up_flag_dev + uflag0 + ci, rseq, 0));      // chunk consumed: its half of the buffer is free
        else HIP_TRY(c, hipEventRecord(T->ev_used[ci & 1], s));
    }
    {   // one combine per accumulator and proof
        const uint32_t pg = (uint32_t)((lig_tune::CHUNK + lig_tune::GROUP / 4 - 1) / (lig_tune::GROUP / 4));
        if (!early_code) lig::launch_rlc_combine(s, tmp, p_code, pg, k);         // the code test's k message values (tmp was zeroed with acc)
        lig::launch_rlc_combine(s, linH, p_linH, pg, k);
        lig::launch_rlc_combine(s, linC, p_linC, (uint32_t)dot_groups, k);
    }
    mark("stage2 rows (rng+dot+encode+rlc)");
    lig::launch_sum_elems(s, linH, k, 1, T->dots, nullptr);       // sum of all <witness row, randomness row> (prover_kernels.hip)
    lig::launch_lin_interleave(s, lin, linH, linC, k);
    HIP_TRY(c, hipMemsetAsync(lin + 2 * (size_t)k, 0, (size_t)(n - 2 * k) * 32, s));
    const lig::CwView view{T->msgs, T->cw, k};
    // (the group partials of the row loop have been combined above: their space holds the quadratic test's group sums now)
    lig::launch_quad_rows29_view(s, view, 2 * k, T->tri_dev, T->coef_dev + R, T->coef_dev + R + NT, NT, quad, T->parts, 2 * groups * (size_t)n);
    // Each accumulator is extended to the n evaluation points, masked (nonbatch_context.hpp:739-753) and sent to the host
    // as soon as it is final; the host absorbs it into the stage-2 seed hash (a sequential SHA-256 over 3 MiB, the longest
    // host step of the proof) while the GPU extends the next one.
    const H::Fr* dots = reinterpret_cast<const H::Fr*>(T->h_small);
    TRY(lig_internal_download(c, T->h_small, T->dots, 32, s));
    if (!early_code) {
        TRY(lig_internal_encode_rows(c, tmp, code, 1, false));      // out of place: no device-to-device staging copy
        lig::launch_eltwise(s, LIG_OP_ADD_ASSIGN, mask, nullptr, code, n, fr{}, 0);
        TRY(lig_internal_download(c, enc, code, vec_bytes, s));
        HIP_TRY(c, hipEventRecord(T->ev_acc[0], s));
    }
    TRY(lig_internal_extend_2k(c, lin));
    lig::launch_eltwise(s, LIG_OP_ADD_ASSIGN, mlin, nullptr, lin, n, fr{}, 0);
    TRY(lig_internal_download(c, enc + vec_bytes, lin, vec_bytes, s));
    HIP_TRY(c, hipEventRecord(T->ev_acc[1], s));
    TRY(lig_internal_extend_2k(c, quad));
    lig::launch_eltwise(s, LIG_OP_ADD_ASSIGN, mquad, nullptr, quad, n, fr{}, 0);
    TRY(lig_internal_download(c, enc + 2 * vec_bytes, quad, vec_bytes, s));
    HIP_TRY(c, hipEventRecord(T->ev_acc[2], s));
    // prover self-check (src/webgpu_prover.cpp:355-386,465-469): the three decodes run on the GPU while the host hashes
    H::Fr* dec = reinterpret_cast<H::Fr*>(T->h_small + 32);    // 3 x n
    const fr* accs[3] = {code, lin, quad};
    for (int a3 = 0; a3 < 3; a3++) {
        TRY(lig_internal_decode_to(c, accs[a3], tmp));
        TRY(lig_internal_download(c, dec + (size_t)a3 * n, tmp, vec_bytes, s));
    }
    HIP_TRY(c, hipEventRecord(c->ev_join, s));
    {
        Sha256 h2;
        h2.add("LigetronStage2", 15).add(info->root, 32);
        for (int a3 = 0; a3 < 3; a3++) {
            HIP_TRY(c, wait_event(T->ev_acc[a3]));
            h2.add(enc + (size_t)a3 * vec_bytes, vec_bytes);
        }
        h2.finish(info->stage2_seed);
    }
    if (const_sum_given) std::memcpy(info->const_sum, const_sum_given, 32);     // the caller's public constant (linear_sums)
    else {
        // synthetic stream: every witness slot carries the constraint w_i = b_i with b public (derived from witness_key), so
        // the constant is minus the sum of all inner products <witness row, randomness row> = minus the sum of linH
        const H::Fr sum = H::neg(dots[0]);
        std::memcpy(info->const_sum, sum.v, 32);
    }
    const std::vector<uint32_t> idx = sample_columns(info->stage2_seed, n, t);
    mark("accumulators to host, seed hash, sampling, decodes");
    info->ms_stage2 = ms_since(t0);
    t0 = clk::now();

    // ================= stage 3: open the sampled columns of every committed row, assemble the envelope.  The gather
    // runs while the host derives the decommitment (the Merkle nodes were downloaded in stage 1) and lays out the envelope;
    // the opened columns then land in place while the host evaluates the self-check predicates.
    TRY(lig_sample_init(c, idx.data(), idx.size()));
    if (T->zres) lig::launch_gather_rows_z(s, c->ep, view, R, c->sample_idx, t, T->samples);
    else lig::launch_gather_rows_planar(s, view, R, c->sample_idx, t, T->samples);
    lig::launch_gather_rows(s, T->maskcw, n, 3, c->sample_idx, t, T->samples + R * (size_t)t);
    const size_t n_nodes = lig_merkle_nodes(n);
    const std::vector<uint8_t> sib = decommit(T->h_nodes, (n_nodes + 1) / 2, idx);
    const size_t smp_bytes = (R + 3) * (size_t)t * 32;
    const EnvelopeLayout lay = write_envelope(T->h_proof, T->h_proof_cap, T->version, T->program_hash, T->generated_at, k, n, t,
                                              info->root, sib, idx, nullptr, smp_bytes);       // framing only: the layout is known now
    if (lay.total > T->h_proof_cap) FAIL(c, LIG_E_NOMEM, "proof buffer too small");
    TRY(lig_internal_download(c, T->h_proof + lay.samples_off, T->samples, smp_bytes, s));   // opened columns land in place (any byte offset)
    for (int a3 = 0; a3 < 3; a3++) std::memcpy(T->h_proof + lay.vec_off[a3], enc + (size_t)a3 * vec_bytes, vec_bytes);     // 3 MiB, under the 13 MB download
    HIP_TRY(c, wait_event(c->ev_join));          // decoded accumulators are on the host
    auto is_zero = [](const H::Fr& v) { return !(v.v[0] | v.v[1] | v.v[2] | v.v[3]); };
    info->valid_code = 1;
    for (uint32_t i = k; i < n; i++) if (!is_zero(dec[i])) info->valid_code = 0;
    {
        H::Fr a;
        std::memcpy(a.v, info->const_sum, 32);
        for (uint32_t i = 0; i < l; i++) a = H::add(a, dec[(size_t)n + i]);
        info->valid_linear = is_zero(a);
    }
    info->valid_quad = 1;
    for (uint32_t i = 0; i < l; i++) if (!is_zero(dec[2 * (size_t)n + i])) info->valid_quad = 0;
    HIP_TRY(c, wait_stream(s));
    *proof = T->h_proof;
    *proof_len = lay.total;
    mark("serialize");
    info->ms_stage3 = ms_since(t0);
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}

// LIG_TRACE=1: print a synchronised timeline of the prove call to stderr (debug aid, off by default)
static std::function<void(const char*)> make_mark(lig_ctx* c) {
    const bool trace_on = lig::knobs().trace;
    auto t_mark = std::make_shared<clk::time_point>(clk::now());
    return [c, trace_on, t_mark](const char* what) {
        if (!trace_on) return;
        (void)hipStreamSynchronize(c->stream);
        std::fprintf(stderr, "[lig_trace] %-28s %8.3f ms\n", what, ms_since(*t_mark));
        *t_mark = clk::now();
    };
}

extern "C" {

// plan_rows: commit order of witness_manager (witness_manager.hpp:497-503): full linear rows, full quadratic
// triples, partial linear row, partial quadratic triple
static int synth_prepare_impl(lig_ctx* c, const lig_synth_job* job, lig_trace* T);
int lig_synth_prepare(lig_ctx* c, const lig_synth_job* job, lig_trace** out) {
    CHECK_CTX(c);
    if (!job || !out) return LIG_E_ARG;
    *out = nullptr;
    lig_trace* T = new lig_trace();
    T->c = c; T->job = *job;
    T->job.batch_ops = nullptr; T->job.batch_data = nullptr;       // the program is consumed here; the caller's memory is not kept
    T->job.public_args = nullptr; T->job.public_arg_lens = nullptr;
    const int rc = synth_prepare_impl(c, job, T);
    if (rc != LIG_OK) { lig_trace_destroy(T); return rc; }         // nothing is handed out on failure
    *out = T;
    return LIG_OK;
}
static int synth_prepare_impl(lig_ctx* c, const lig_synth_job* job, lig_trace* T) {
    const uint32_t l = c->l, k = c->k, n = c->n, t = 192;
    // k - l = sample_size upstream (include/params.hpp:24-32); with fewer random pads than opened columns the proof
    // would silently stop being zero-knowledge
    if (l >= k || l < 2 || t > n || k - l < t) FAIL(c, LIG_E_ARG, "synthetic trace: need 2 <= l <= k - 192 (k - l random pads cover the 192 opened columns)");
    if (!plan_rows(*job, l, T->rows, T->n_init)) FAIL(c, LIG_E_ARG, "malformed batch program");
    if (T->n_init && k - l != 192) FAIL(c, LIG_E_ARG, "batch program: on_batch_init draws params::sample_size = 192 pads, k - l must be 192");
    if (!instance_hash_of(job->public_args, job->public_arg_lens, job->n_public_args, T->ih)) FAIL(c, LIG_E_ARG, "public arguments: null pointer");
    std::memcpy(T->encoding_seed, job->encoding_seed, 32);
    std::memcpy(T->program_hash, job->program_hash, 32);
    std::memcpy(T->version, job->version, 16);
    T->generated_at = job->generated_at;
    const size_t R = T->rows.size();
    for (T->RB = 0; T->RB < R && T->rows[T->RB].kind >= RK_INIT; T->RB++) {}
    TRY(trace_alloc(c, T));
    // pads: init rows of the batch program drew theirs while the program ran (prepare); every stream row draws at commit time
    if (R > T->RB) T->pad_runs.push_back({T->RB, R - T->RB, (uint64_t)T->n_init * (k - l)});
    T->mask_pos = (uint64_t)(T->n_init + (R - T->RB)) * (k - l);
    // witness values: one draw per data slot of every linear / x / y row, in commit order; z = x*y
    if (T->RB) TRY(lig_run_batch_program(c, *job, T->msgs));
    TRY(lig_internal_synth_witness(c, job->witness_key, T->rows, T->RB, T->msgs));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}

void lig_trace_destroy(lig_trace* T) {
    if (!T) return;
    (void)hipSetDevice(T->c->device);
    (void)hipStreamSynchronize(T->c->stream);
    (void)hipStreamSynchronize(T->c->stream2);
    if (T->c->stream3) (void)hipStreamSynchronize(T->c->stream3);
    T->up_abort.store(1, std::memory_order_release);      // (copies that wait for a buffer of a proof that never ran)
    uploader_drain(T);                                    // an upload still in flight
    rand_drain(T);
    if (T->c->stream_sha) (void)hipStreamSynchronize(T->c->stream_sha);      // (experiment knob LIG_SHA_CUMASK: the stage-1 hash stream)
    T->c->sha.erase(T->sha_state);
    if (T->leak) {      // destinations of a transfer that was given up on and has never completed (upload_settled): they outlive the trace
        T->rands_full = nullptr; T->msgs_alt = nullptr; T->msgs = nullptr; T->randb = nullptr; T->packed_dev = nullptr;
    }
    for (void* p : {(void*)T->rands_full, (void*)T->msgs_alt, (void*)T->msgs, (void*)T->cw, (void*)T->maskcw, (void*)T->randb, (void*)T->rcw, (void*)T->acc, (void*)T->parts, (void*)T->dots,
                    (void*)T->samples, (void*)T->sha_state, (void*)T->leaves, (void*)T->nodes, (void*)T->tri_dev,
                    (void*)T->coef_dev})
        (void)hipFree(p);
    for (int i = 0; i < 2; i++) { if (T->ev_ready[i]) (void)hipEventDestroy(T->ev_ready[i]); if (T->ev_used[i]) (void)hipEventDestroy(T->ev_used[i]); }
    if (T->ev_gate) (void)hipEventDestroy(T->ev_gate);
    if (T->ev_k1) (void)hipEventDestroy(T->ev_k1);
    for (int a3 = 0; a3 < 3; a3++) if (T->ev_acc[a3]) (void)hipEventDestroy(T->ev_acc[a3]);
    for (hipEvent_t e : T->ev_up) (void)hipEventDestroy(e);
    if (T->up_flag) (void)hipHostFree((void*)T->up_flag);
    (void)hipFree(T->src_off_dev); (void)hipFree(T->widths_dev); (void)hipFree(T->packed_dev);
    (void)hipHostFree(T->h_proof); (void)hipHostFree(T->h_enc); (void)hipHostFree(T->h_nodes); (void)hipHostFree(T->h_small);
    delete T;
}
uint64_t lig_trace_rows(const lig_trace* T) { return T ? T->R + 3 : 0; }

int lig_synth_prove(lig_trace* T, const uint8_t** proof, size_t* proof_len, lig_proof_info* info) {
    if (!T || !proof || !proof_len || !info) return LIG_E_ARG;
    lig_ctx* c = T->c;
    CHECK_CTX(c);
    if (T->from_rows) FAIL(c, LIG_E_STATE, "lig_synth_prove on a trace made by lig_rows_begin (use lig_rows_commit / lig_rows_prove)");
    std::memset(info, 0, sizeof *info);
    info->rows = T->R + 3;
    const auto t_begin = clk::now();
    const auto mark = make_mark(c);
    TRY(prove_stage1(T, info, mark));
    info->ms_stage1 = ms_since(t_begin);
    TRY(prove_stage23(T, RandSource{}, nullptr, proof, proof_len, info, mark));
    info->ms_total = ms_since(t_begin);
    return LIG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Host rows -> device without a HIP stream of the prover in the path.
//
// What was measured (tools/time_rows2.py, tools/h2d_under_load.py; profiles/r03_h2d_pipeline.md): the PCIe link delivers
// 57 GB/s to this process whatever the GPU is doing (550 MB = one 2^24-constraint trace in 9.7 ms, idle or under two proving
// contexts), and a foreign stream's upload does not slow the proofs down.  The prover's own chunked upload did: HIP maps the
// streams of a process onto GPU_MAX_HW_QUEUES = 4 hardware queues, every event recorded behind a copy (and every wait for one)
// is a barrier packet in such a queue, and while it waits for a 2 ms .. 10 ms transfer the kernels of whichever proof stream
// shares the queue do not start (stage 2 of a proof 4.9 -> 10-16 ms; two alternating contexts: 12.8 ms per proof, slower
// than one).  Priorities (own queue pool) and a shared copy stream move the problem around (A/B table in the profile).
// So: one uploader thread per device copies chunk after chunk on a stream of its own and waits for each copy ON THE HOST
// (no event, no packet behind the copy), then publishes the chunk's arrival in pinned host memory; the encode stream of the
// trace waits for that word with a stream memory operation (hipStreamWaitValue32 -- a wait in ITS OWN queue, where it has to
// wait anyway).  Uploads of all contexts go through the one thread: one at a time, in the order of the calls -- two contexts
// that alternate keep the link busy without ever sharing it.
namespace {
struct Uploader {
    std::atomic<uint64_t> cur_bytes{0}, cur_since_us{0}, done_jobs{0};      // diagnostics
    std::atomic<int> phase{0};                                              // 0 idle, 1 copy call, 2 waiting for the copy, 3 publishing
    std::atomic<bool> broken{false};                                        // a transfer timed out: no further jobs
    bool fault_done = false;                                                // LIG_FAULT_UPLOAD: the injected fault has been spent
    std::atomic<uint32_t> abandoned{0};                                     // transfers given up on that may still be in flight on `st` (cleared by settle())
    std::atomic<uint32_t> retries{0};                                       // calls that re-made a timed-out upload with stream-ordered copies
    // The copy this thread stopped waiting for is still queued on `st` and cannot be cancelled.  settle(): has it finished by now?  (Bounded
    // poll from the calling thread; hipStreamQuery is thread-safe.)  Until it has, its source rows and its destination must stay alive.
    bool settle(double seconds) {
        if (!abandoned.load(std::memory_order_acquire)) return true;
        const auto t0 = clk::now();
        for (;;) {
            const hipError_t e = hipStreamQuery(st);
            if (e != hipErrorNotReady) { (void)hipGetLastError(); abandoned.store(0, std::memory_order_release); return true; }
            if (std::chrono::duration<double>(clk::now() - t0).count() > seconds) return false;
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    }
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::pair<UploadJob, std::atomic<int>*>> q;
    std::thread th;
    hipStream_t st = nullptr;
    int device = 0;
    bool ok = false;
    void run() {
        if (hipSetDevice(device) != hipSuccess) return;
        for (;;) {
            std::pair<UploadJob, std::atomic<int>*> j;
            {
                // the first job whose destination is free: a job that waits for its buffer must not hold up the other contexts'
                // uploads queued behind it (jobs of one trace stay in order: their wait words become true in order)
                std::unique_lock<std::mutex> lk(mu);
                for (;;) {
                    auto it = q.end();
                    for (auto i2 = q.begin(); i2 != q.end(); ++i2) {
                        const UploadJob& u = i2->first;
                        const bool ready = !u.wait || (int32_t)(__atomic_load_n(u.wait, __ATOMIC_ACQUIRE) - u.wait_val) >= 0 || (u.abort && u.abort->load(std::memory_order_acquire));
                        if (!ready) continue;
                        if (it == q.end()) it = i2;                                  // the oldest ready job ...
                        if (u.prio > it->first.prio) { it = i2; break; }             // ... unless a ready one is urgent
                    }
                    if (it != q.end()) { j = *it; q.erase(it); break; }
                    if (q.empty()) cv.wait(lk, [&] { return !q.empty(); });
                    else cv.wait_for(lk, std::chrono::microseconds(20));       // every queued job waits for the GPU: poll
                }
            }
            const bool dead = broken.load(std::memory_order_acquire);             // jobs queued behind a transfer that timed out fail at once
            const bool skip = dead || (j.first.abort && j.first.abort->load(std::memory_order_acquire));
            cur_bytes.store(j.first.bytes, std::memory_order_relaxed);
            cur_since_us.store((uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(clk::now().time_since_epoch()).count(), std::memory_order_relaxed);
            phase.store(1, std::memory_order_release);
            hipError_t e = dead ? hipErrorLaunchTimeOut : hipSuccess;
            if (!skip && j.first.segs) {
                for (const UploadSeg& g : *j.first.segs) {
                    if (!g.bytes || e != hipSuccess) continue;
                    e = g.src ? hipMemcpyAsync(g.dst, g.src, g.bytes, hipMemcpyHostToDevice, st) : hipMemsetAsync(g.dst, 0, g.bytes, st);
                }
            } else if (!skip && j.first.bytes) e = hipMemcpyAsync(j.first.dst, j.first.src, j.first.bytes, hipMemcpyHostToDevice, st);
            phase.store(2, std::memory_order_release);
            // bounded: a transfer that does not complete (seen with several processes on one GPU, profiles/r05_rows_entry_hang.md) must
            // not hang every stream that waits for its word -- after LIG_UPLOAD_TIMEOUT_S the job is reported as failed (the word is
            // published; lig_rows_commit / _prove let their streams drain, wait -- bounded -- for the abandoned copy to leave the bus and
            // make the upload again with stream-ordered copies: rows_retry_*) and this thread takes no more jobs: callers fall back on
            // stream-ordered copies from then on (lig_internal_uploader_available turns false)
            hipError_t e2 = e;
            // tests (LIG_FAULT_UPLOAD): one transfer "never completes" -- 1: the first one of witness rows, 2: the first one of randomness rows
            const int fu = lig::knobs().fault_upload;
            const bool injected = fu && !skip && !fault_done && (fu == 2) == (j.first.prio == 1);
            if (injected) fault_done = true;
            if (e == hipSuccess) {
                const auto t_wait = clk::now();
                const double limit = (double)lig::knobs().upload_timeout_s;
                for (unsigned spins = 0;; spins++) {
                    e2 = injected ? hipErrorNotReady : hipStreamQuery(st);
                    if (e2 != hipErrorNotReady) break;
                    if (spins < 20000) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(20));
                    if ((spins & 1023) == 1023 && std::chrono::duration<double>(clk::now() - t_wait).count() > limit) { e2 = hipErrorLaunchTimeOut; abandoned.fetch_add(1, std::memory_order_acq_rel); broken.store(true, std::memory_order_release); break; }
                }
                if (e2 == hipErrorNotReady) e2 = hipSuccess;
            }
            phase.store(3, std::memory_order_release);
            // (a failed copy publishes too: no stream may hang on the flag; lig_rows_commit reports the error once stage 1 has drained)
            if (e2 != hipSuccess) { (void)hipGetLastError(); j.first.failed->store((int)e2, std::memory_order_release); }
            __atomic_store_n(j.first.flag, j.first.seq, __ATOMIC_RELEASE);
            j.second->fetch_sub(1, std::memory_order_acq_rel);
            done_jobs.fetch_add(1, std::memory_order_relaxed);
            phase.store(0, std::memory_order_release);
        }
    }
};
Uploader* g_uploader[64] = {nullptr};
std::mutex g_uploader_mu;
}  // namespace
bool lig_internal_uploader_available(lig_ctx* c) {
    if (c->device < 0 || c->device >= 64) return false;
    std::lock_guard<std::mutex> lk(g_uploader_mu);
    Uploader*& u = g_uploader[c->device];
    if (!u) {
        int can = 0;
        (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, c->device);
        u = new Uploader();                         // lives for the process: its thread sleeps on the condition variable
        u->device = c->device;
        // The uploader's stream must not share a HARDWARE queue with a stream that may hold a pending hipStreamWaitValue32 for the word this
        // thread publishes: HIP maps the streams of a process onto GPU_MAX_HW_QUEUES (4) hardware queues per priority class, a pending stream
        // wait occupies its queue, and a small host-to-device copy is a blit KERNEL in the copying stream's queue (tools/queue_share_probe.hip,
        // profiles/r05_queue_share_probe.txt) -- behind the wait it would never run.  (A latent deadlock found while hunting the round-4 hang
        // of the sharded rows entry; not its cause, profiles/r05_rows_entry_hang.md.)  Queues are pooled per priority class, the library's
        // proof streams are normal priority: the uploader takes the highest.
        int lo = 0, hi = 0;
        const bool prio = lig::knobs().upload_prio && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo;
        u->ok = can && (prio ? hipStreamCreateWithPriority(&u->st, hipStreamNonBlocking, hi) : hipStreamCreateWithFlags(&u->st, hipStreamNonBlocking)) == hipSuccess;
        if (u->ok) { u->th = std::thread([u] { u->run(); }); u->th.detach(); }
    }
    return u->ok && !u->broken.load(std::memory_order_acquire);
}
}  // extern "C"
std::string lig_internal_uploader_state(int device) {
    if (device < 0 || device >= 64 || !g_uploader[device]) return "uploader: none";
    Uploader* u = g_uploader[device];
    size_t queued = 0;
    { std::lock_guard<std::mutex> lk(u->mu); queued = u->q.size(); }
    const uint64_t now = (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(clk::now().time_since_epoch()).count();
    const int ph = u->phase.load();
    static const char* names[4] = {"idle", "in the copy call", "waiting for the copy", "publishing"};
    return "uploader: " + std::string(names[ph & 3]) + (ph ? " (" + std::to_string(u->cur_bytes.load()) + " bytes, for " + std::to_string((now - u->cur_since_us.load()) / 1000) + " ms)" : "") +
           ", " + std::to_string(queued) + " queued, " + std::to_string(u->done_jobs.load()) + " done";
}
extern "C" {
int lig_upload_health(lig_ctx* c, uint32_t* retries, uint32_t* unsettled) {
    CHECK_CTX(c);
    Uploader* u = (c->device >= 0 && c->device < 64) ? g_uploader[c->device] : nullptr;
    if (u && u->abandoned.load(std::memory_order_acquire)) (void)u->settle(0.0);      // one query: has it finished in the meantime?
    if (retries) *retries = u ? u->retries.load(std::memory_order_relaxed) : 0;
    if (unsettled) *unsettled = u ? u->abandoned.load(std::memory_order_acquire) : 0;
    return LIG_OK;
}
void lig_internal_uploader_submit(int device, const std::vector<UploadJob>& jobs, std::atomic<int>* pending) {
    Uploader* u = g_uploader[device];
    pending->fetch_add((int)jobs.size(), std::memory_order_acq_rel);
    {
        std::lock_guard<std::mutex> lk(u->mu);
        for (const UploadJob& j : jobs) u->q.push_back({j, pending});
    }
    u->cv.notify_one();
}
// every copy of this trace's uploads has been made (its host rows and its device matrix are no longer touched by the thread)
static void uploader_drain(lig_trace* T) {
    while (T->up_pending.load(std::memory_order_acquire) > 0) std::this_thread::yield();
}
static void rand_drain(lig_trace* T) {
    while (T->rand_pending.load(std::memory_order_acquire) > 0) std::this_thread::yield();
}
// pinned flag words of a trace: one per stage-1 chunk (rows arrived), two per stage-2 chunk (caller randomness rows arrived / consumed)
static int ensure_up_flags(lig_ctx* c, lig_trace* T) {
    if (T->up_flag) return LIG_OK;
    T->up_words = T->sched1.size() + 2 * (T->R / lig_tune::CHUNK + 3) + 8;        // (the last word: rows of lig_rows_push_rands that have arrived)
    const size_t bytes = (T->up_words * 4 + 4095) & ~(size_t)4095;
    HIP_TRY(c, hipHostMalloc((void**)&T->up_flag, bytes, hipHostMallocDefault));
    std::memset((void*)T->up_flag, 0, bytes);
    HIP_TRY(c, hipHostGetDevicePointer((void**)&T->up_flag_dev, (void*)T->up_flag, 0));
    return LIG_OK;
}
// witness rows by stream-ordered copies on the context's copy stream, one event per stage-1 chunk (LIG_UPLOAD_MODE=1, devices without
// stream memory operations, and the second attempt after a transfer of the uploader thread has timed out)
static int rows_copy_by_stream(lig_ctx* c, lig_trace* T, uint8_t* up_dst) {
    const uint32_t k = c->k;
    auto chunk_src = [&](size_t b) -> size_t { return T->narrow ? (size_t)T->src_off[b] : b * (size_t)k * 32; };
    T->up_by_thread = false;
    if (T->ev_up.empty()) {
        T->ev_up.resize(T->sched1.size(), nullptr);
        for (auto& e : T->ev_up) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    for (size_t ci = 0; ci < T->sched1.size(); ci++) {
        const size_t b = T->sched1[ci].first, e = T->sched1[ci].second;
        const size_t off = chunk_src(b), bytes = chunk_src(e) - off;
        if (bytes) HIP_TRY(c, hipMemcpyAsync(up_dst + off, T->host_msgs + off, bytes, hipMemcpyHostToDevice, lig_internal_copy_stream(c)));
        HIP_TRY(c, hipEventRecord(T->ev_up[ci], lig_internal_copy_stream(c)));
    }
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}
// A transfer of the uploader thread timed out (Uploader::run): the waiting streams were released and have drained.  The copy itself is
// still queued on the uploader's stream and cannot be cancelled.  Bounded wait for it to leave the bus: true = it has (its bytes are where
// they belong or will be overwritten by the second attempt); false = it is still pending -- what it reads (the caller's rows) and writes
// (this trace's buffers) must stay alive: the trace is marked, its buffers are never freed or reused, lig_upload_health() reports it.
static bool upload_settled(lig_ctx* c, lig_trace* T) {
    Uploader* u = g_uploader[c->device];
    if (!u || u->settle((double)lig::knobs().upload_timeout_s)) return true;
    T->leak = true;
    return false;
}
static const char* const UPLOAD_PENDING_MSG = "; the transfer is still pending and cannot be cancelled: keep the rows alive until lig_upload_health reports no unsettled "
                                              "transfer (the trace's device buffers are kept for the life of the process)";
// message rows of a rows job -> T->msgs.  Device rows: one copy on the main stream.  Host rows: the upload starts now, on
// the copy stream, one event per stage-1 chunk: lig_rows_commit encodes chunk b while chunk b+1 is still on the bus.
static int rows_load(lig_ctx* c, lig_trace* T, const void* msgs, bool on_device) {
    const uint32_t k = c->k;
    const size_t R = T->R;
    T->host_msgs = nullptr;
    T->loaded = true;
    T->alt_pending = false;
    if (!R) return LIG_OK;
    fr* dst = T->msgs;
    // narrow host rows land in the packed staging area and are expanded into `msgs` by lig_rows_commit (after the previous
    // proof has finished with it): no second matrix needed
    if (T->committed && !(T->narrow && !on_device)) {
        // the committed trace still needs its rows for stage 2: the next trace goes to the second matrix and is swapped in
        // by lig_rows_commit -- its upload runs under lig_rows_prove of the current one
        if (!T->msgs_alt) HIP_TRY(c, hipMalloc((void**)&T->msgs_alt, R * (size_t)k * 32));
        dst = T->msgs_alt;
        T->alt_pending = true;
    }
    if (on_device) {
        if (T->narrow) hipLaunchKernelGGL(lig::k_expand_rows, dim3((uint32_t)std::min<size_t>((R * k + 255) / 256, 8192)), dim3(256), 0, c->stream, (const uint8_t*)msgs,
                                          T->src_off_dev, T->widths_dev, (size_t)0, R, c->l, k, dst);
        else HIP_TRY(c, hipMemcpyAsync(dst, msgs, R * (size_t)k * 32, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipGetLastError());
        return LIG_OK;
    }
    T->host_msgs = (const uint8_t*)msgs;
    if (T->narrow && !T->packed_dev) HIP_TRY(c, hipMalloc((void**)&T->packed_dev, T->src_off[R] ? T->src_off[R] : 16));   // (the job began with device rows)
    // byte range of a stage-1 chunk in the caller's matrix / the destination of its copy
    auto chunk_src = [&](size_t b) -> size_t { return T->narrow ? (size_t)T->src_off[b] : b * (size_t)k * 32; };
    uint8_t* up_dst = T->narrow ? T->packed_dev : (uint8_t*)dst;
    // (every earlier reader of `dst` has finished: lig_rows_prove returns only after its stream work is done)
    // (A copy kernel reading the pinned rows over PCIe instead of the DMA engine was measured: 37 GB/s against 56 GB/s, and
    // the long-running kernel serialises with the proof's kernels whenever both streams share a hardware queue:
    // stage 2 5.6 -> 13.8 ms.  The DMA engine it is.)
    const int mode = lig::knobs().upload_mode;   // 2: uploader thread (default), 1: per-context copy stream + events
    if (mode == 2 && lig_internal_uploader_available(c)) {
        TRY(ensure_up_flags(c, T));
        T->up_seq++;
        std::vector<UploadJob> jobs;
        for (size_t ci = 0; ci < T->sched1.size(); ci++) {
            const size_t b = T->sched1[ci].first, e = T->sched1[ci].second;
            const size_t off = chunk_src(b), bytes = chunk_src(e) - off;
            jobs.push_back(UploadJob{up_dst + off, T->host_msgs + off, bytes, T->up_flag + ci, T->up_seq, &T->up_failed});
        }
        lig_internal_uploader_submit(c->device, jobs, &T->up_pending);
        T->up_by_thread = true;
        return LIG_OK;
    }
    return rows_copy_by_stream(c, T, up_dst);
}

// ---------------------------------------------------------------------------------------------------------------------
// rows supplied by the caller (include/lig_hip.h, lig_rows_*)
static int rows_begin_impl(lig_ctx* c, const lig_rows_job* job, lig_trace* T) {
    const uint32_t l = c->l, k = c->k, n = c->n, t = 192, pad = k - l;
    if (l >= k || l < 2 || t > n || k - l < t) FAIL(c, LIG_E_ARG, "rows job: need 2 <= l <= k - 192");
    if (job->rows && (!job->kinds || !job->msgs)) FAIL(c, LIG_E_ARG, "rows job: null kinds / msgs");
    if (!instance_hash_of(job->public_args, job->public_arg_lens, job->n_public_args, T->ih)) FAIL(c, LIG_E_ARG, "public arguments: null pointer");
    std::memcpy(T->encoding_seed, job->encoding_seed, 32);
    std::memcpy(T->program_hash, job->program_hash, 32);
    std::memcpy(T->version, job->version, 16);
    T->generated_at = job->generated_at;
    T->from_rows = true;
    const size_t R = job->rows;
    // row kinds: groups must be complete and consecutive; which kinds draw padding from the encoding stream at the time
    // they are formed: linear rows and the rows of a quadratic triple (witness_manager.hpp:200-269), on_batch_init rows
    // (nonbatch_context.hpp:497-510); bit / equal / batch-quadratic rows are copies of variables and draw nothing
    T->rows.resize(R);
    std::vector<uint8_t> draw(R, 0);
    std::vector<uint64_t> pos(R + 1, 0);
    for (size_t r = 0; r < R; r++) {
        const uint8_t kd = job->kinds[r] & 0x7f;
        if (kd > RK_BQZ) FAIL(c, LIG_E_ARG, "rows job: unknown row kind");
        const bool first_of_3 = kd == 1 || kd == RK_BQX, first_of_2 = kd == RK_EQX;
        if (first_of_3 && !(r + 2 < R && (job->kinds[r + 1] & 0x7f) == kd + 1 && (job->kinds[r + 2] & 0x7f) == kd + 2)) FAIL(c, LIG_E_ARG, "rows job: incomplete x,y,z triple");
        if (first_of_2 && !(r + 1 < R && (job->kinds[r + 1] & 0x7f) == RK_EQY)) FAIL(c, LIG_E_ARG, "rows job: incomplete equality pair");
        const bool follower = kd == 2 || kd == 3 || kd == RK_EQY || kd == RK_BQY || kd == RK_BQZ;
        if (follower && !(r > 0 && (job->kinds[r - 1] & 0x7f) == kd - 1)) FAIL(c, LIG_E_ARG, "rows job: row of a group without its predecessor");
        const bool draws = kd <= 3 || kd == RK_INIT;
        // on_batch_init draws params::sample_size = 192 elements (nonbatch_context.hpp:497-510), the rows of witness_manager
        // k - l; upstream the two are the same number (params.hpp:27-30).  Batch rows are only accepted in that geometry:
        // otherwise the encoding stream would run out of step with the reference's
        if (kd == RK_INIT && pad != 192) FAIL(c, LIG_E_ARG, "rows job: on_batch_init rows need k - l = 192 (params::sample_size)");
        if ((job->kinds[r] & LIG_ROW_DRAW_PAD) && !draws) FAIL(c, LIG_E_ARG, "rows job: LIG_ROW_DRAW_PAD on a row kind that draws no padding upstream");
        draw[r] = (job->kinds[r] & LIG_ROW_DRAW_PAD) ? 1 : 0;
        pos[r + 1] = pos[r] + (draws ? pad : 0);
        const uint32_t dense = job->dense_rands_per_row ? job->dense_rands_per_row[r] : 0;
        if (dense > k || (dense && kd > 3)) FAIL(c, LIG_E_ARG, "rows job: dense_rands_per_row out of range or on a batch row");
        T->rows[r] = RowDesc{kd, dense};
    }
    T->dense_rands = job->dense_rands_per_row != nullptr;
    T->mask_pos = pos[R];
    if (job->elem_bytes) {            // the narrow row format
        T->src_off.assign(R + 1, 0);
        T->widths.assign(R ? R : 1, 32);
        for (size_t r = 0; r < R; r++) {
            const uint8_t w = job->elem_bytes[r] ? job->elem_bytes[r] : 32;
            if (w != 4 && w != 8 && w != 32) FAIL(c, LIG_E_ARG, "rows job: elem_bytes must be 0, 4, 8 or 32");
            if (w != 32 && (T->rows[r].kind > 3 || !draw[r])) FAIL(c, LIG_E_ARG, "rows job: a narrow row must be LINEAR / QX / QY / QZ with LIG_ROW_DRAW_PAD");
            T->narrow = T->narrow || w != 32;
            T->widths[r] = w;
            T->src_off[r + 1] = T->src_off[r] + (w == 32 ? (uint64_t)k * 32 : (uint64_t)l * w);
        }
    }
    TRY(trace_alloc(c, T));
    if (T->narrow) {
        HIP_TRY(c, hipMalloc((void**)&T->src_off_dev, (R + 1) * sizeof(uint64_t)));
        HIP_TRY(c, hipMalloc((void**)&T->widths_dev, R));
        HIP_TRY(c, hipMemcpy(T->src_off_dev, T->src_off.data(), (R + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(T->widths_dev, T->widths.data(), R, hipMemcpyHostToDevice));
        if (!job->msgs_on_device) HIP_TRY(c, hipMalloc((void**)&T->packed_dev, T->src_off[R] ? T->src_off[R] : 16));
    }
    // pad runs: consecutive flagged rows whose stream positions are consecutive, never straddling a stage-1 chunk
    for (const auto& ch : T->sched1)
        for (size_t r = ch.first; r < ch.second;) {
            if (!draw[r]) { r++; continue; }
            size_t e = r + 1;
            while (e < ch.second && draw[e] && pos[e] == pos[e - 1] + pad) e++;
            T->pad_runs.push_back({r, e - r, pos[r]});
            r = e;
        }
    return rows_load(c, T, job->msgs, job->msgs_on_device != 0);
}
int lig_rows_begin(lig_ctx* c, const lig_rows_job* job, lig_trace** out) {
    CHECK_CTX(c);
    if (!job || !out) return LIG_E_ARG;
    *out = nullptr;
    lig_trace* T = new lig_trace();
    T->c = c;
    std::memset(&T->job, 0, sizeof T->job);
    const int rc = rows_begin_impl(c, job, T);
    if (rc != LIG_OK) { lig_trace_destroy(T); return rc; }
    *out = T;
    return LIG_OK;
}
int lig_rows_restart(lig_trace* T, const void* msgs, int msgs_on_device) {
    if (!T) return LIG_E_ARG;
    lig_ctx* c = T->c;
    CHECK_CTX(c);
    if (!T->from_rows) FAIL(c, LIG_E_STATE, "lig_rows_restart: not a rows trace");
    if (T->R && !msgs) FAIL(c, LIG_E_ARG, "lig_rows_restart: null rows");
    if (T->leak) FAIL(c, LIG_E_STATE, "lig_rows_restart: a transfer into this trace's buffers never completed (lig_upload_health): destroy the trace");
    if (T->loaded && T->host_msgs) { uploader_drain(T); if (c->stream3) HIP_TRY(c, hipStreamSynchronize(c->stream3)); }      // an upload nobody committed: let it finish first
    return rows_load(c, T, msgs, msgs_on_device != 0);
}
int lig_rows_commit(lig_trace* T, uint8_t root[32], uint8_t stage1_seed[32]) {
    if (!T) return LIG_E_ARG;
    lig_ctx* c = T->c;
    CHECK_CTX(c);
    if (!T->from_rows || !T->loaded) FAIL(c, LIG_E_STATE, "lig_rows_commit: no rows loaded (lig_rows_begin / lig_rows_restart)");
    if (T->committed) FAIL(c, LIG_E_STATE, "lig_rows_commit: the committed trace has not been proved yet");
    if (T->alt_pending) { std::swap(T->msgs, T->msgs_alt); T->alt_pending = false; }
    std::memset(&T->info1, 0, sizeof T->info1);
    T->info1.rows = T->R + 3;
    const auto t_begin = clk::now();
    {
        if (T->leak) FAIL(c, LIG_E_STATE, "lig_rows_commit: a transfer into this trace's buffers never completed (lig_upload_health): destroy the trace");
        T->up_retry = false;
        int rc = prove_stage1(T, &T->info1, make_mark(c));
        if (rc != LIG_OK && T->up_retry) {
            // a chunk of the rows did not arrive within LIG_UPLOAD_TIMEOUT_S (the uploader thread has released the streams, stage 1 ran over
            // garbage and is discarded): once the abandoned copy has left the bus the same rows are brought by stream-ordered copies and
            // stage 1 runs again -- the caller sees a slow commit, not an error
            T->up_retry = false;
            const std::string why = c->err;
            uploader_drain(T);
            for (hipStream_t st : {c->stream, c->stream2, c->stream3, c->stream_sha}) if (st) (void)hipStreamSynchronize(st);
            if (upload_settled(c, T)) {
                g_uploader[c->device]->retries.fetch_add(1, std::memory_order_relaxed);
                rc = rows_copy_by_stream(c, T, T->narrow ? T->packed_dev : (uint8_t*)T->msgs);
                if (rc == LIG_OK) rc = prove_stage1(T, &T->info1, make_mark(c));
            } else c->err = why + UPLOAD_PENDING_MSG;
        }
        if (rc != LIG_OK) {           // the caller is told it may free its rows: nothing of ours may still read them (but see T->leak)
            const std::string why = c->err;
            uploader_drain(T);
            for (hipStream_t st : {c->stream, c->stream2, c->stream3, c->stream_sha}) if (st) (void)hipStreamSynchronize(st);
            T->loaded = false; T->host_msgs = nullptr;
            c->err = why;
            return rc;
        }
    }
    T->info1.ms_stage1 = ms_since(t_begin);
    T->committed = true;
    T->loaded = false;
    T->host_msgs = nullptr;                               // the caller's memory is no longer referenced
    if (root) std::memcpy(root, T->info1.root, 32);
    if (stage1_seed) std::memcpy(stage1_seed, T->info1.stage1_seed, 32);
    return LIG_OK;
}
int lig_rows_prove(lig_trace* T, const void* rands, int rands_on_device, const uint8_t const_sum[32], const uint8_t** proof,
                   size_t* proof_len, lig_proof_info* info) {
    if (!T || !proof || !proof_len || !info) return LIG_E_ARG;
    lig_ctx* c = T->c;
    CHECK_CTX(c);
    if (!T->from_rows || !T->committed) FAIL(c, LIG_E_STATE, "lig_rows_prove: lig_rows_commit has not run on this trace");
    if (T->R && !rands && !T->dense_rands && !T->rands_pushed) FAIL(c, LIG_E_ARG, "lig_rows_prove: null randomness rows");
    if (const_sum) {
        H::Fr v;
        std::memcpy(v.v, const_sum, 32);
        if (H::geq(v, H::P)) FAIL(c, LIG_E_ARG, "lig_rows_prove: constant not reduced mod p");
    }
    *info = T->info1;
    const auto t_begin = clk::now();
    RandSource rs;                                        // default: generated from the dense counts of the job
    if (rands && T->rands_pushed) {                       // explicit rows win over rows pushed earlier: drop what the uploader still holds of those
        T->up_abort.store(1, std::memory_order_release);
        rand_drain(T);
        T->up_abort.store(0, std::memory_order_release);
        (void)T->rand_failed.exchange(0);
        T->rands_pushed = 0;
    }
    if (rands && rands_on_device) rs.dev = (const fr*)rands; else if (rands) rs.host = (const uint8_t*)rands;
    else if (T->rands_pushed) {
        if (T->rands_pushed != T->R) FAIL(c, LIG_E_STATE, "lig_rows_prove: lig_rows_push_rands has not delivered every row");
        rs.dev = T->rands_full; rs.pushed = true;
    }
    for (int attempt = 0;; attempt++) {
        int rc = prove_stage23(T, rs, const_sum, proof, proof_len, info, make_mark(c));
        const std::string why = c->err;
        if (rc != LIG_OK) {           // randomness-row copies the uploader thread still holds read the caller's memory: drop them, wait
            T->rands_pushed = 0;
            T->push_log.clear();
            T->up_abort.store(1, std::memory_order_release);
            rand_drain(T);
            for (hipStream_t st : {c->stream, c->stream2, c->stream3, c->stream_sha}) if (st) (void)hipStreamSynchronize(st);
            c->err = why;
            return rc;
        }
        if (!(rs.host || rs.pushed)) break;
        rand_drain(T);
        const int e = T->rand_failed.exchange(0);
        if (!e) break;
        if (e == (int)hipErrorLaunchTimeOut && attempt == 0) {
            // a randomness-row transfer timed out: stages 2 / 3 ran over garbage and are discarded.  Same second attempt as in
            // lig_rows_commit: host rows come by stream-ordered copies now (the uploader is out of service), pushed rows are copied again
            // from the pushes' host memory (valid until this call returns)
            for (hipStream_t st : {c->stream, c->stream2, c->stream3, c->stream_sha}) if (st) (void)hipStreamSynchronize(st);
            if (upload_settled(c, T)) {
                g_uploader[c->device]->retries.fetch_add(1, std::memory_order_relaxed);
                if (rs.pushed) {
                    for (const UploadJob& j : T->push_log) {
                        if (j.segs) { for (const UploadSeg& g : *j.segs) { if (!g.bytes) continue; HIP_TRY(c, g.src ? hipMemcpyAsync(g.dst, g.src, g.bytes, hipMemcpyHostToDevice, lig_internal_copy_stream(c)) : hipMemsetAsync(g.dst, 0, g.bytes, lig_internal_copy_stream(c))); } }
                        else if (j.bytes) HIP_TRY(c, hipMemcpyAsync(j.dst, j.src, j.bytes, hipMemcpyHostToDevice, lig_internal_copy_stream(c)));
                    }
                    if (c->stream3) HIP_TRY(c, hipStreamSynchronize(c->stream3));
                    T->push_sync = true;          // every row is there: stage 2 has no arrival word to wait for
                }
                *info = T->info1;
                continue;
            }
            T->rands_pushed = 0; T->push_log.clear();
            FAIL(c, LIG_E_HIP, std::string("randomness rows upload failed: ") + hipGetErrorString((hipError_t)e) + UPLOAD_PENDING_MSG);
        }
        T->rands_pushed = 0; T->push_log.clear();
        FAIL(c, LIG_E_HIP, std::string("randomness rows upload failed: ") + hipGetErrorString((hipError_t)e));
    }
    T->rands_pushed = 0;
    T->push_log.clear();
    info->ms_total = info->ms_stage1 + ms_since(t_begin);
    T->committed = false;
    return LIG_OK;
}

int lig_rows_push_rands(lig_trace* T, uint64_t first_row, uint64_t n_rows, const void* host_rows) {
    return lig_rows_push_rands_sparse(T, first_row, n_rows, nullptr, host_rows);
}
int lig_rows_push_rands_sparse(lig_trace* T, uint64_t first_row, uint64_t n_rows, const uint8_t* present, const void* host_rows) {
    if (!T) return LIG_E_ARG;
    lig_ctx* c = T->c;
    CHECK_CTX(c);
    if (!T->from_rows || !T->committed) FAIL(c, LIG_E_STATE, "lig_rows_push_rands: lig_rows_commit has not run on this trace");
    if (first_row != T->rands_pushed || first_row + n_rows > T->R) FAIL(c, LIG_E_ARG, "lig_rows_push_rands: rows must arrive in order, without gaps, inside the trace");
    if (!n_rows) return LIG_OK;
    size_t n_present = n_rows;
    if (present) { n_present = 0; for (uint64_t i = 0; i < n_rows; i++) n_present += present[i] != 0; }
    if (n_present && !host_rows) FAIL(c, LIG_E_ARG, "lig_rows_push_rands: null rows");
    const size_t row_bytes = (size_t)c->k * 32;
    if (!T->rands_full) HIP_TRY(c, hipMalloc((void**)&T->rands_full, T->R * row_bytes));
    if (lig::knobs().upload_mode != 2 || !lig_internal_uploader_available(c)) {
        // no uploader thread (LIG_UPLOAD_MODE=1, or a device without stream memory operations): the same rows by plain copies on the
        // copy stream, waited for here -- slower (the transfer sits in a HIP queue of the context) but every caller of the push
        // interface (include/lig_hip_row_batcher.hpp) keeps working.  ADVICE r4.
        if (first_row == 0) T->push_sync = true;
        if (!T->push_sync) FAIL(c, LIG_E_STATE, "lig_rows_push_rands: the uploader became unavailable in the middle of a trace");
        uint8_t* dst = (uint8_t*)T->rands_full + first_row * row_bytes;
        const uint8_t* src = (const uint8_t*)host_rows;
        for (uint64_t i = 0; i < n_rows;) {
            uint64_t e = i + 1;
            const bool p = !present || present[i] != 0;
            while (e < n_rows && (!present || (present[e] != 0) == p)) e++;
            if (p) { HIP_TRY(c, hipMemcpyAsync(dst + i * row_bytes, src, (size_t)(e - i) * row_bytes, hipMemcpyHostToDevice, lig_internal_copy_stream(c))); src += (e - i) * row_bytes; }
            else HIP_TRY(c, hipMemsetAsync(dst + i * row_bytes, 0, (size_t)(e - i) * row_bytes, lig_internal_copy_stream(c)));
            i = e;
        }
        if (c->stream3) HIP_TRY(c, hipStreamSynchronize(c->stream3));
        T->rands_pushed = first_row + n_rows;
        return LIG_OK;
    }
    if (first_row == 0) T->push_sync = false;
    TRY(ensure_up_flags(c, T));
    volatile uint32_t* arrived = T->up_flag + T->up_words - 1;
    if (first_row == 0) { __atomic_store_n(arrived, 0u, __ATOMIC_RELEASE); T->up_abort.store(0, std::memory_order_release); T->push_log.clear(); }
    // one job per push: the uploader publishes the number of rows that have arrived (jobs of a trace are taken in order)
    UploadJob j{(uint8_t*)T->rands_full + first_row * row_bytes, (const uint8_t*)host_rows, n_rows * row_bytes, arrived, (uint32_t)(first_row + n_rows), &T->rand_failed};
    if (present && n_present != n_rows) {       // runs of present rows are copied from where they follow each other in host_rows, the others zero-filled on the device
        j.segs = std::make_shared<std::vector<UploadSeg>>();
        const uint8_t* src = (const uint8_t*)host_rows;
        for (uint64_t i = 0; i < n_rows;) {
            uint64_t e = i + 1;
            const bool p = present[i] != 0;
            while (e < n_rows && (present[e] != 0) == p) e++;
            j.segs->push_back(UploadSeg{j.dst + i * row_bytes, p ? src : nullptr, (size_t)(e - i) * row_bytes});
            if (p) src += (e - i) * row_bytes;
            i = e;
        }
    }
    j.abort = &T->up_abort; j.prio = 1;
    T->push_log.push_back(j);
    lig_internal_uploader_submit(c->device, {j}, &T->rand_pending);
    T->rands_pushed = first_row + n_rows;
    return LIG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// host-only transcript helpers
int lig_public_arg_bytes(int kind, const char* text, uint8_t* out, size_t cap, size_t* len) {
    if (!text || !len) return LIG_E_ARG;
    std::vector<uint8_t> b;
    if (kind == LIG_ARG_I64) {
        char* end = nullptr;
        errno = 0;
        const long long v = std::strtoll(text, &end, 10);
        if (errno || end == text || *end) return LIG_E_ARG;
        const int64_t i = (int64_t)v;
        b.resize(8);
        std::memcpy(b.data(), &i, 8);                      // (u8*)&i .. + sizeof(int64_t), little-endian host as upstream
    } else if (kind == LIG_ARG_STR) {
        b.assign(text, text + std::strlen(text) + 1);      // c_str() .. + size() + 1: the NUL is part of the argument
    } else if (kind == LIG_ARG_HEX) {
        std::string h(text);
        if (h.rfind("0x", 0) == 0) h = h.substr(2);
        if (h.size() % 2) h.insert(h.begin(), '0');
        auto nib = [](char ch) -> int { return ch >= '0' && ch <= '9' ? ch - '0' : ch >= 'a' && ch <= 'f' ? ch - 'a' + 10 : ch >= 'A' && ch <= 'F' ? ch - 'A' + 10 : -1; };
        for (size_t i = 0; i < h.size(); i += 2) {
            const int hi = nib(h[i]), lo = nib(h[i + 1]);
            if (hi < 0 || lo < 0) return LIG_E_ARG;        // boost::algorithm::unhex throws on a non-hex character
            b.push_back((uint8_t)(hi * 16 + lo));
        }
    } else return LIG_E_ARG;
    *len = b.size();
    if (!out || cap < b.size()) return LIG_E_ARG;
    if (!b.empty()) std::memcpy(out, b.data(), b.size());
    return LIG_OK;
}
int lig_instance_hash(const uint8_t* args, const uint64_t* lens, size_t n_args, uint8_t out[32]) {
    if (!out || !instance_hash_of(args, lens, n_args, out)) return LIG_E_ARG;
    return LIG_OK;
}
int lig_sample_columns(const uint8_t seed[32], uint32_t n, uint32_t t, uint32_t* out_sorted) {
    if (!seed || !out_sorted || !n) return LIG_E_ARG;
    const std::vector<uint32_t> idx = sample_columns(seed, n, t);
    std::memcpy(out_sorted, idx.data(), idx.size() * sizeof(uint32_t));
    return LIG_OK;
}

}  // extern "C"
