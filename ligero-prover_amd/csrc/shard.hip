// shard.hip -- one trace sharded over the GPUs of a node (lig_shard_*).
#include "prover_common.hpp"
#include <unistd.h>
#include <functional>
#include <thread>

// =====================================================================================================================
// One trace sharded over the GPUs of a node (BASELINE.json configs[3]).
//
// The committed rows are cut into G = W * rounds global chunks (<= 512 rows, equal sizes, never splitting an x,y,z
// triple or an equality pair) and dealt block-cyclically: chunk g belongs to rank g mod W, so the local rows of a rank are
// its chunks r, r+W, r+2W, ...; every rank forms, encodes and keeps only those (codewords as three coset planes + the message
// rows, like the single-GPU prover: lig::CwView).  leaf_j hashes ALL rows in commit order, so
// the hash is column-partitioned (rank h owns columns [h*n/W, (h+1)*n/W), i.e. positions q in [h*k/W, (h+1)*k/W) of all four
// cosets): in round c every rank sends the column slices
// of its c-th chunk to their owners -- ONE all-to-all per round -- and receives the W consecutive global chunks
// cW .. cW+W-1 restricted to its columns, in rank order = commit order.  The column hash therefore runs in commit order
// from the first round on, on the side stream, while the main stream encodes round c+1 and the copy stream exchanges it:
//     main:  encode(c)   pack(c)   encode(c+1)  pack(c+1)  ...
//     comm:                        exchange(c)             exchange(c+1)
//     hash:                                     hash(c)                 hash(c+1)
// n/W leaves per rank are all-gathered, the Merkle tree is built redundantly.  The stage-2 tests are sums over rows: every
// rank accumulates its rows on the low-degree domains (k + 2k + 2k values), the partial sums are all-gathered and added
// mod p locally (RCCL has no modular reduction).  Opened columns are all-gathered and laid out in commit order.
// Collectives: lig_comm (include/lig_hip.h) -- RCCL over xGMI from comm_rccl.hip (stream-ordered), or host-synchronous
// callbacks (tests over gloo).  Every rank ends with the same envelope, byte-identical to lig_synth_prove.
struct lig_shard {
    lig_ctx* c = nullptr;
    lig_synth_job job;
    lig_comm comm;
    uint32_t rank = 0, world = 1;
    std::atomic<const char*> dbg_wait{nullptr}; std::atomic<uint64_t> dbg_wait_since_ms{0};     // what shard_bounded_wait is polling for, since when (shard_debug)
    bool poisoned = false;                     // work queued behind a failed collective never drained (shard_quiesce): no further calls, buffers outlive the shard
    std::vector<RowDesc> rows;                 // global plan
    std::vector<size_t> gb;                    // G + 1 global chunk boundaries
    std::vector<size_t> lrow0;                 // local row offset of my c-th chunk (rounds + 1 entries)
    std::vector<size_t> grow;                  // global row of every local row
    std::vector<uint64_t> wit_pos, lin_pos;    // stream position of every global row (+1 entry)
    std::vector<uint64_t> code_ord;            // code-test draws before every global row (+1 entry)
    std::vector<uint8_t> draw;                 // per global row: its k-l pads are drawn here at commit time (pad_encoding_random)
    std::vector<uint64_t> enc_pos;             // encoding-stream position of every global row's pads (+1 entry = the masks' position)
    bool from_rows = false;                    // lig_shard_rows_*: the local rows and their randomness rows come from the caller
    bool dense_rands = false, committed = false;
    lig_proof_info info1;                      // stage-1 results kept between lig_shard_rows_commit and lig_shard_rows_prove
    uint8_t encoding_seed[32] = {0}, program_hash[32] = {0};
    int64_t generated_at = 0;
    char version[17] = {0};
    size_t RB = 0, n_init = 0;                 // leading rows committed by the batch program, of those: init rows
    size_t R = 0, Rl = 0, rows_max = 0, ncol = 0, rounds = 0, G = 0, ch_cap = 0;
    fr *msgs = nullptr, *cw = nullptr, *maskcw = nullptr, *send = nullptr, *recv = nullptr, *randb = nullptr, *rhalf = nullptr, *acc = nullptr,
       *parts = nullptr, *accp = nullptr, *accg = nullptr, *dots = nullptr, *smp = nullptr, *smpg = nullptr;
    uint32_t *sha_state = nullptr, *leaves_slice = nullptr, *leaves = nullptr, *nodes = nullptr, *tri_dev = nullptr;
    lig::f29s* coef_dev = nullptr;
    std::vector<uint32_t> triples;             // local row indices
    std::vector<size_t> triple_ord;            // global ordinal of each local triple
    uint8_t *h_proof = nullptr, *h_enc = nullptr, *h_nodes = nullptr, *h_small = nullptr;
    size_t h_proof_cap = 0;
    uint8_t ih[32] = {0};
    hipEvent_t ev_enc[2] = {nullptr, nullptr}, ev_comm[2] = {nullptr, nullptr}, ev_hash[2] = {nullptr, nullptr};
    // caller rows in host memory (lig_shard_rows_*): uploaded by the library's uploader thread, as in lig_rows_* (prover.hip).  Flag words
    // in pinned host memory: [round c: local rows of my c-th chunk arrived | round c: randomness rows arrived | ... consumed]
    volatile uint32_t* up_flag = nullptr; uint32_t* up_flag_dev = nullptr;
    uint32_t up_seq = 0, rand_seq = 0;
    bool rows_by_thread = false;               // the local rows of the trace being committed are arriving through the uploader
    std::atomic<int> up_pending{0}, up_failed{0}, up_abort{0};
    bool used_comm = false;                    // a collective has been issued with buffers of this shard as send buffers
    bool exchange_even_alone = false;          // LIG_SHARD_FORCE_EXCHANGE: run pack + all-to-all with world == 1 too (tests)
    size_t chunk_rows(size_t g) const { return g < G ? gb[g + 1] - gb[g] : 0; }
};

static int shard_up_flags(lig_shard* S);
static void shard_up_drain(lig_shard* S);

namespace lig {
// The column slices of `rows` codewords, one block of cap_rows x ncol per destination rank h.  Inside a row of a block the
// ncol = 4*kq columns of rank h are PLANE-MAJOR: element c*kq + ql = codeword column 4*(h*kq + ql) + c, so that both sides
// move contiguous runs (reads: runs of kq elements of one plane / of the message row; the receiver's column hash then reads
// 32-byte neighbours) and instance j of the receiver's hash state is simply element j of every received row.
__global__ void __launch_bounds__(256) k_pack_slices(CwView cw, fr* __restrict__ out, size_t rows, size_t kq, size_t cap_rows) {
    const size_t n = 4 * (size_t)cw.k, ncol = 4 * kq, total = rows * n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / n, j = i - r * n;                  // j: position in the row of blocks = h*ncol + c*kq + ql
        const size_t h = j / ncol, e = j - h * ncol, c = e / kq, ql = e - c * kq;
        fr_store(out + (h * cap_rows + r) * ncol + e, fr_load(cw.at(r, (uint32_t)(4 * (h * kq + ql) + c))));
    }
}
}  // namespace lig

// The block-cyclic deal: rounds = ceil(R / (W * CHUNK)), G = W * rounds global chunks of (nearly) equal size whose boundaries
// never fall inside an x,y,z triple or an equality pair; chunk g belongs to rank g mod W.  gb: G + 1 boundaries.
static void shard_chunks(const std::vector<RowDesc>& rows, uint32_t W, size_t& rounds, std::vector<size_t>& gb) {
    const size_t R = rows.size();
    rounds = std::max<size_t>(1, (R + W * lig_tune::CHUNK - 1) / (W * lig_tune::CHUNK));
    const size_t G = rounds * W;
    const size_t target = std::max<size_t>(1, (R + G - 1) / G);
    gb.assign(G + 1, R);
    gb[0] = 0;
    auto inside_group = [&](uint8_t kd) { return kd == 2 || kd == 3 || kd == RK_EQY || kd == RK_BQY || kd == RK_BQZ; };
    for (size_t g = 1; g < G; g++) {
        size_t b = std::min(R, gb[g - 1] + target);
        while (b < R && inside_group(rows[b].kind)) b++;                       // never split a triple / an equality pair
        gb[g] = b;
    }
}

extern "C" {

// host only (no GPU work): the deal of a job's committed rows over `world` ranks, for drivers and tests
int lig_shard_plan(const lig_synth_job* job, uint32_t l, uint32_t world, uint64_t* rounds_out, uint64_t* boundaries, size_t cap) {
    if (!job || !world || !rounds_out) return LIG_E_ARG;
    std::vector<RowDesc> rows;
    size_t n_init = 0;
    if (!plan_rows(*job, l, rows, n_init)) return LIG_E_ARG;
    size_t rounds = 0;
    std::vector<size_t> gb;
    shard_chunks(rows, world, rounds, gb);
    *rounds_out = rounds;
    if (gb.size() > cap || !boundaries) return gb.size() > cap ? LIG_E_NOMEM : LIG_E_ARG;
    for (size_t i = 0; i < gb.size(); i++) boundaries[i] = gb[i];
    return LIG_OK;
}

static int shard_prepare_impl(lig_ctx* c, const lig_synth_job* job, uint32_t rank, uint32_t world, lig_shard* S);
int lig_shard_prepare(lig_ctx* c, const lig_synth_job* job, uint32_t rank, uint32_t world, const lig_comm* comm, lig_shard** out) {
    CHECK_CTX(c);
    if (!job || !out || !comm || world == 0 || rank >= world) return LIG_E_ARG;
    if (!comm->all_to_all || !comm->all_gather) return LIG_E_ARG;
    *out = nullptr;
    lig_shard* S = new lig_shard();
    S->c = c; S->job = *job; S->comm = *comm; S->rank = rank; S->world = world;
    S->job.batch_ops = nullptr; S->job.batch_data = nullptr;
    std::memcpy(S->encoding_seed, job->encoding_seed, 32);
    std::memcpy(S->program_hash, job->program_hash, 32);
    std::memcpy(S->version, job->version, 16);
    S->generated_at = job->generated_at;
    {   // instance_hash over arg0 = "Ligero\0" and the public arguments (src/webgpu_prover.cpp:110-168)
        if (job->n_public_args && (!job->public_args || !job->public_arg_lens)) { delete S; return LIG_E_ARG; }
        std::memset(S->ih, 0, 32);
        Sha256().add(S->ih, 32).add("Ligero", 7).finish(S->ih);
        const uint8_t* a = job->public_args;
        for (uint64_t i = 0; i < job->n_public_args; i++) {
            uint8_t prev[32];
            std::memcpy(prev, S->ih, 32);
            Sha256().add(prev, 32).add(a, job->public_arg_lens[i]).finish(S->ih);
            a += job->public_arg_lens[i];
        }
        S->job.public_args = nullptr; S->job.public_arg_lens = nullptr;
    }
    const int rc = shard_prepare_impl(c, job, rank, world, S);
    if (rc != LIG_OK) { lig_shard_destroy(S); return rc; }
    *out = S;
    return LIG_OK;
}
// the deal + every buffer of a shard whose global row plan (S->rows) is known
static int shard_alloc(lig_ctx* c, uint32_t rank, uint32_t world, lig_shard* S);
static int shard_prepare_impl(lig_ctx* c, const lig_synth_job* job, uint32_t rank, uint32_t world, lig_shard* S) {
    const uint32_t l = c->l, k = c->k, n = c->n, t = 192;
    if (l >= k || l < 2 || t > n || k - l < t || k % world) FAIL(c, LIG_E_ARG, "sharded trace: need 2 <= l <= k - 192 and world | k");
    if (!plan_rows(*job, l, S->rows, S->n_init)) FAIL(c, LIG_E_ARG, "malformed batch program");
    if (S->n_init && k - l != 192) FAIL(c, LIG_E_ARG, "batch program: on_batch_init draws params::sample_size = 192 pads, k - l must be 192");
    const size_t R = S->R = S->rows.size();
    for (S->RB = 0; S->RB < R && S->rows[S->RB].kind >= RK_INIT; S->RB++) {}
    S->wit_pos.assign(R + 1, 0); S->lin_pos.assign(R + 1, 0); S->code_ord.assign(R + 1, 0);
    S->draw.assign(R, 0); S->enc_pos.assign(R + 1, 0);
    for (size_t r = 0; r < R; r++) {
        const uint8_t kd = S->rows[r].kind;
        S->wit_pos[r + 1] = S->wit_pos[r] + ((kd == 3 || kd >= RK_INIT) ? 0 : S->rows[r].data);   // z rows and batch rows draw nothing
        S->lin_pos[r + 1] = S->lin_pos[r] + S->rows[r].data;
        S->code_ord[r + 1] = S->code_ord[r] + has_code_check(kd);                                    // position in the code-test stream
        S->draw[r] = kd <= 3;                                                                        // stream rows draw at commit time; init rows drew theirs in the program
        S->enc_pos[r + 1] = S->enc_pos[r] + ((kd <= 3 || kd == RK_INIT) ? (k - l) : 0);             // rows that draw k-l pads upstream
    }
    TRY(shard_alloc(c, rank, world, S));
    const size_t Rl = S->Rl;
    // local witness rows: same stream positions as in the single-GPU trace
    uint32_t rk[60];
    if (S->RB) {          // the batch program is small: every rank runs it and keeps the rows it owns
        fr* all = nullptr;
        HIP_TRY(c, hipMalloc((void**)&all, S->RB * (size_t)k * sizeof(fr)));
        int rc = lig_run_batch_program(c, *job, all);
        for (size_t lr = 0; lr < Rl && rc == LIG_OK; lr++)
            if (S->grow[lr] < S->RB && hipMemcpyAsync(S->msgs + lr * (size_t)k, all + S->grow[lr] * (size_t)k, (size_t)k * sizeof(fr), hipMemcpyDeviceToDevice, c->stream) != hipSuccess) rc = LIG_E_HIP;
        (void)hipStreamSynchronize(c->stream);
        (void)hipFree(all);
        if (rc != LIG_OK) return rc;
    }
    lig::aes256_expand_host(job->witness_key, rk);
    TRY(lig_internal_upload_small(c, c->rk_dev, rk, sizeof rk, c->stream));
    for (size_t lr = 0; lr < Rl;) {
        const size_t gr = S->grow[lr];
        const RowDesc d = S->rows[gr];
        if (d.kind >= RK_INIT) { lr++; continue; }
        if (d.kind == 0) {
            lig::launch_rng_fill_rows(c->stream, c->rk_dev, S->wit_pos[gr], S->msgs + lr * k, 1, d.data, k, 0, 1, d.data);
            lr += 1;
        } else {          // x, y, z are consecutive locally as well: chunks never split a triple
            lig::launch_rng_fill_rows(c->stream, c->rk_dev, S->wit_pos[gr], S->msgs + lr * k, 2, d.data, k, 0, 1, d.data);
            lig::launch_eltwise(c->stream, LIG_OP_MUL, S->msgs + lr * k, S->msgs + (lr + 1) * k, S->msgs + (lr + 2) * k, d.data, fr{}, 0);
            lr += 3;
        }
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}

static int shard_alloc(lig_ctx* c, uint32_t rank, uint32_t world, lig_shard* S) {
    const uint32_t k = c->k, n = c->n, t = 192, W = world;
    const size_t R = S->R;
    S->ncol = n / world;
    S->exchange_even_alone = lig::knobs().shard_force_exchange;
    shard_chunks(S->rows, W, S->rounds, S->gb);
    S->G = S->rounds * W;
    for (size_t g = 0; g < S->G; g++) S->ch_cap = std::max(S->ch_cap, S->chunk_rows(g));
    if (!S->ch_cap) S->ch_cap = 1;
    S->lrow0.assign(S->rounds + 1, 0);
    for (size_t cidx = 0; cidx < S->rounds; cidx++) {
        const size_t g = cidx * W + rank;
        for (size_t r = S->gb[g]; r < S->gb[g + 1]; r++) S->grow.push_back(r);
        S->lrow0[cidx + 1] = S->grow.size();
    }
    const size_t Rl = S->Rl = S->grow.size();
    for (uint32_t h = 0; h < W; h++) {
        size_t cnt = 0;
        for (size_t cidx = 0; cidx < S->rounds; cidx++) cnt += S->chunk_rows(cidx * W + h);
        S->rows_max = std::max(S->rows_max, cnt);
    }
    if (!S->rows_max) S->rows_max = 1;
    const size_t RM = S->rows_max;
    std::vector<size_t> local_of(R, (size_t)-1);
    for (size_t lr = 0; lr < Rl; lr++) local_of[S->grow[lr]] = lr;
    {   // quadratic-test terms whose rows are local, with local row indices; triple_ord = position in the quadratic stream
        const std::vector<uint32_t> all = quad_terms(S->rows);
        for (size_t i = 0; i < all.size() / 3; i++) {
            const size_t last = all[3 * i + 2];
            if (local_of[last] == (size_t)-1) continue;
            S->triples.push_back((uint32_t)local_of[all[3 * i]]);
            S->triples.push_back(all[3 * i + 1] == 0xFFFFFFFFu ? 0xFFFFFFFFu : (uint32_t)local_of[all[3 * i + 1]]);
            S->triples.push_back((uint32_t)local_of[last]);
            S->triple_ord.push_back(i);
        }
    }
    const size_t chunk = S->ch_cap;
    auto dm = [&](void** p, size_t bytes) -> int { HIP_TRY(c, hipMalloc(p, bytes ? bytes : 16)); HIP_TRY(c, hipMemsetAsync(*p, 0, bytes, c->stream)); return LIG_OK; };
    TRY(dm((void**)&S->msgs, (Rl ? Rl : 1) * (size_t)k * 32));
    TRY(dm((void**)&S->cw, (Rl ? Rl : 1) * 3 * (size_t)k * 32));      // cosets 1..3 of the local codewords as planes
    TRY(dm((void**)&S->maskcw, 3 * (size_t)n * 32));                  // the mask rows' codewords, reference layout
    TRY(dm((void**)&S->send, 2 * chunk * (size_t)n * 32));          // double-buffered: W blocks of chunk x ncol each
    TRY(dm((void**)&S->recv, 2 * chunk * (size_t)n * 32));
    TRY(dm((void**)&S->randb, 2 * chunk * (size_t)k * 32));         // double-buffered
    TRY(dm((void**)&S->rhalf, chunk * 2 * (size_t)k * 32));
    TRY(dm((void**)&S->acc, 4 * (size_t)n * 32));
    TRY(dm((void**)&S->parts, (2 * ((chunk + lig_tune::GROUP / 4 - 1) / (lig_tune::GROUP / 4)) + (chunk + lig_tune::DOT_GROUP - 1) / lig_tune::DOT_GROUP) * (size_t)k * 32));
    TRY(dm((void**)&S->accp, 5 * (size_t)k * 32));
    TRY(dm((void**)&S->accg, (size_t)world * 5 * k * 32));
    TRY(dm((void**)&S->dots, 32));
    TRY(dm((void**)&S->smp, (RM + 3) * (size_t)t * 32));
    TRY(dm((void**)&S->smpg, (size_t)world * RM * t * 32));
    TRY(dm((void**)&S->sha_state, lig_sha_state_bytes(S->ncol)));
    TRY(dm((void**)&S->leaves_slice, S->ncol * 32));
    TRY(dm((void**)&S->leaves, (size_t)n * 32));
    TRY(dm((void**)&S->nodes, lig_merkle_nodes(n) * 32));
    TRY(dm((void**)&S->tri_dev, (S->triples.size() ? S->triples.size() : 1) * sizeof(uint32_t)));
    TRY(dm((void**)&S->coef_dev, (Rl + 2 * S->triple_ord.size() + 1) * sizeof(lig::f29s)));
    S->h_proof_cap = ((size_t)1 << 19) + 3 * (size_t)n * 32 + (R + 3) * (size_t)t * 32;
    HIP_TRY(c, hipHostMalloc((void**)&S->h_proof, S->h_proof_cap, hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc((void**)&S->h_enc, 3 * (size_t)n * 32, hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc((void**)&S->h_nodes, lig_merkle_nodes(n) * 32, hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc((void**)&S->h_small, (1 + 3 * (size_t)n) * 32, hipHostMallocDefault));
    for (int i = 0; i < 2; i++) {
        HIP_TRY(c, hipEventCreateWithFlags(&S->ev_enc[i], hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&S->ev_comm[i], hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&S->ev_hash[i], hipEventDisableTiming));
    }
    if (!S->triples.empty()) HIP_TRY(c, hipMemcpyAsync(S->tri_dev, S->triples.data(), S->triples.size() * 4, hipMemcpyHostToDevice, c->stream));
    TRY(lig_internal_reserve_scratch(c, chunk));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return LIG_OK;
}

// Bounded replacement for hipStreamSynchronize on the error / teardown paths: when a collective failed and neither the
// communicator's abort nor the watchdog released the kernels queued behind it, a blocking synchronize would hang exactly where
// shard_bounded_wait gave up.  Polls the context's streams for `seconds`; false = something is still queued: the shard is poisoned
// (every later call on it returns LIG_E_STATE; lig_shard_destroy keeps the device buffers the queued kernels may still touch).
static bool shard_quiesce(lig_shard* S, double seconds) {
    lig_ctx* c = S->c;
    const auto t0 = clk::now();
    for (unsigned spins = 0;; spins++) {
        bool busy = false;
        for (hipStream_t st : {c->stream, c->stream2, c->stream3}) {
            if (!st) continue;
            const hipError_t e = hipStreamQuery(st);
            if (e == hipErrorNotReady) busy = true; else if (e != hipSuccess) (void)hipGetLastError();
        }
        if (!busy) return true;
        if (spins < 4000) { std::this_thread::yield(); continue; }
        usleep(50);
        if ((spins & 63) == 0 && std::chrono::duration<double>(clk::now() - t0).count() > seconds) { S->poisoned = true; return false; }
    }
}
static constexpr double SHARD_QUIESCE_S = 20.0;
static const char* const SHARD_POISONED_MSG = "; the streams still hold work queued behind the failed collective: the shard is poisoned (rows and randomness rows of this call "
                                              "must stay alive; destroy the shard and do not reuse the context)";

void lig_shard_destroy(lig_shard* S) {
    if (!S) return;
    (void)hipSetDevice(S->c->device);
    if (!S->poisoned) (void)shard_quiesce(S, SHARD_QUIESCE_S);
    S->up_abort.store(1, std::memory_order_release);
    while (S->up_pending.load(std::memory_order_acquire) > 0) std::this_thread::yield();       // the uploader thread is done with our buffers
    if (S->poisoned) {
        // kernels behind a collective that never completed are still queued: they may run (and touch these buffers) whenever the queue is
        // released.  The device and pinned buffers, the events and the flag page outlive the shard; only the host object goes.  (The
        // context's streams are in the same state: destroy the context's process, or at least never reuse the context.)
        S->c->sha.erase(S->sha_state);
        delete S;
        return;
    }
    if (S->up_flag) (void)hipHostFree((void*)S->up_flag);
    // the send buffers below are about to be freed.  forget() may be a host collective (comm_ipc): a shard that fails before its
    // first collective (a local error in *_begin / *_prepare) has exported nothing and must not wait for peers that are not there
    if (S->comm.forget && S->used_comm) S->comm.forget(S->comm.user);
    S->c->sha.erase(S->sha_state);
    for (void* p : {(void*)S->msgs, (void*)S->cw, (void*)S->maskcw, (void*)S->send, (void*)S->recv, (void*)S->randb, (void*)S->rhalf, (void*)S->acc,
                    (void*)S->parts, (void*)S->accp, (void*)S->accg, (void*)S->dots, (void*)S->smp, (void*)S->smpg, (void*)S->sha_state,
                    (void*)S->leaves_slice, (void*)S->leaves, (void*)S->nodes, (void*)S->tri_dev, (void*)S->coef_dev})
        (void)hipFree(p);
    for (int i = 0; i < 2; i++)
        for (hipEvent_t e : {S->ev_enc[i], S->ev_comm[i], S->ev_hash[i]}) if (e) (void)hipEventDestroy(e);
    (void)hipHostFree(S->h_proof); (void)hipHostFree(S->h_enc); (void)hipHostFree(S->h_nodes); (void)hipHostFree(S->h_small);
    delete S;
}

// Bounded wait for queued work that contains stream-ordered collectives (round 5: a sharded call returns an error, it never hangs).
// A collective inside a queue cannot time out or report a failure; so the host polls instead of blocking, and
//   * asks lig_comm.failed (comm_ipc: watchdog thread -- dead peer, abort word, stall timer; comm_rccl: ncclCommGetAsyncError):
//     a failed communicator releases / aborts its queued waits, the streams drain, the call returns LIG_E_STATE with the reason;
//   * after LIG_COMM_TIMEOUT_S (default 300 s -- fifty times the longest sharded proof measured) calls lig_comm.abort
//     (ncclCommAbort / poison) itself.
// Without stream-ordered forms (host-synchronous callbacks) the wait is a plain synchronize: nothing of a peer is in the queues.
static int shard_bounded_wait(lig_shard* S, const std::function<hipError_t()>& query, const char* what) {
    lig_ctx* c = S->c;
    const lig_comm& cm = S->comm;
    const auto t0 = clk::now();
    const double limit_s = (double)lig::knobs().comm_timeout_s;
    bool aborted = false, failed = false;
    auto t_fail = t0;
    std::string why;
    S->dbg_wait_since_ms.store((uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(t0.time_since_epoch()).count(), std::memory_order_release);
    S->dbg_wait.store(what, std::memory_order_release);
    struct Clear { lig_shard* S; ~Clear() { S->dbg_wait.store(nullptr, std::memory_order_release); } } clear_on_exit{S};
    for (unsigned spins = 0;; spins++) {
        const hipError_t e = query();
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) { (void)hipGetLastError(); c->err = std::string(what) + ": " + hipGetErrorString(e); return LIG_E_HIP; }
        if (spins < 4000) { std::this_thread::yield(); continue; }
        usleep(50);
        if ((spins & 63) != 0) continue;
        const auto now = clk::now();
        if (!failed && cm.failed && cm.failed(cm.user)) { failed = true; t_fail = now; why = c->err; }
        if (!failed && !aborted && std::chrono::duration<double>(now - t0).count() > limit_s) {
            aborted = failed = true; t_fail = now;
            why = std::string(what) + ": no completion after " + std::to_string((int)limit_s) + " s (LIG_COMM_TIMEOUT_S)";
            if (cm.abort) cm.abort(cm.user);
        }
        if (failed && std::chrono::duration<double>(now - t_fail).count() > 20.0) {        // the queues did not drain even so: report, leave the rest to the caller
            c->err = "collective failed and the streams did not drain: " + why;
            return LIG_E_STATE;
        }
    }
    if (failed || (cm.failed && cm.failed(cm.user))) {
        if (failed) c->err = why;
        if (c->err.find("ipc comm") == std::string::npos && c->err.find("nccl") == std::string::npos) c->err = "collective failed: " + c->err;
        return LIG_E_STATE;
    }
    return LIG_OK;
}

// where the stage-2 randomness rows of the LOCAL rows come from: generated (dense rows of the synthetic stream) or the caller's
struct ShardRands { const fr* dev = nullptr; const uint8_t* host = nullptr; };
#define SHARD_COMMON \
    lig_ctx* c = S->c; \
    const uint32_t l = c->l, k = c->k, n = c->n, t = 192, pad = k - l, W = S->world; \
    const size_t R = S->R, Rl = S->Rl, RM = S->rows_max, ncol = S->ncol, CAP = S->ch_cap; \
    hipStream_t s = c->stream, s_hash = c->stream2, s_comm = lig_internal_copy_stream(c); \
    const bool ordered = S->comm.all_to_all_on != nullptr && S->comm.all_gather_on != nullptr; \
    auto comm_fail = [&](const char* what) { if (c->err.find("nccl") == std::string::npos && c->err.find("ipc comm") == std::string::npos && c->err.find("injected fault") == std::string::npos) c->err = std::string("collective failed: ") + what; else c->err = std::string(what) + ": " + c->err; return (int)LIG_E_STATE; }; \
    auto all_gather = [&](const void* src, void* dst, size_t bytes, hipStream_t st, const char* what) -> int { \
        S->used_comm = true; \
        if (ordered) { if (S->comm.all_gather_on(S->comm.user, src, dst, bytes, st)) return comm_fail(what); return LIG_OK; } \
        HIP_TRY(c, hipStreamSynchronize(st)); \
        if (S->comm.all_gather(S->comm.user, src, dst, bytes)) return comm_fail(what); \
        return LIG_OK; \
    }; \
    auto drain = [&](hipStream_t st, const char* what) -> int { \
        if (!ordered) { HIP_TRY(c, hipStreamSynchronize(st)); return LIG_OK; } \
        return shard_bounded_wait(S, [&] { return hipStreamQuery(st); }, what); \
    }; \
    auto drain_event = [&](hipEvent_t ev, const char* what) -> int { \
        if (!ordered) { HIP_TRY(c, hipEventSynchronize(ev)); return LIG_OK; } \
        return shard_bounded_wait(S, [&] { return hipEventQuery(ev); }, what); \
    }; \
    (void)l; (void)pad; (void)RM; (void)t; (void)s_comm; (void)W; (void)Rl; (void)R; (void)CAP; (void)ncol; (void)s_hash; (void)all_gather; (void)drain; (void)drain_event

// what a watchdog prints when it declares the communicator dead while a sharded call is in progress
// (No HIP call in here: it runs on a communicator's watchdog thread under the context's debug mutex, in exactly the situation where the
// runtime may be wedged -- a blocked query would keep the mutex and with it the API call's return.  What the calling thread is waiting
// for is published by shard_bounded_wait in two atomics instead.  ADVICE r5.)
static std::string shard_debug(lig_shard* S, const char* stage, uint32_t rseq) {
    lig_ctx* c = S->c;
    const char* w = S->dbg_wait.load(std::memory_order_acquire);
    const uint64_t now = (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(clk::now().time_since_epoch()).count();
    std::string o = std::string("[lig_shard] rank ") + std::to_string(S->rank) + " in " + stage + ": " +
                    (w ? std::string("the calling thread has been waiting for '") + w + "' for " + std::to_string(now - S->dbg_wait_since_ms.load(std::memory_order_acquire)) + " ms"
                       : std::string("the calling thread is enqueueing"));
    o += "; rows upload seq " + std::to_string(S->up_seq) + ", randomness seq " + std::to_string(rseq) + ", pending " + std::to_string(S->up_pending.load()) + ", flag words [arrived rows | arrived rands | consumed]:";
    if (S->up_flag && S->rounds) for (size_t i = 0; i < 3 * S->rounds && i < 24; i++) o += (i % S->rounds == 0 ? " | " : " ") + std::to_string(S->up_flag[i]);
    else o += " none";
    return o + "; " + lig_internal_uploader_state(c->device);
}
struct ShardDebugScope {
    lig_ctx* c;
    ShardDebugScope(lig_shard* S, const char* stage, const uint32_t* rseq) : c(S->c) { lig_internal_set_debug_state(c, [S, stage, rseq] { return shard_debug(S, stage, rseq ? *rseq : 0); }); }
    ~ShardDebugScope() { lig_internal_set_debug_state(c, nullptr); }
};

static int shard_stage1(lig_shard* S, lig_proof_info* info) {
    SHARD_COMMON;
    ShardDebugScope dbg(S, "stage 1", nullptr);
    // ---------------- stage 1
    uint32_t rk[60];
    lig::aes256_expand_host(S->encoding_seed, rk);
    TRY(lig_internal_upload_small(c, c->rk_dev, rk, sizeof rk, s));
    // pads of the local rows that draw them at commit time (batch init rows carry theirs from the program): the position
    // of a row's pads = number of pad-drawing rows before it in commit order; runs of consecutive rows = one launch
    // (rows that are still arriving from the host -- rows_by_thread -- get theirs round by round below, behind the arrival of the round)
    auto draw_pads = [&](size_t lr0, size_t lr1) {
        for (size_t lr = lr0; lr < lr1;) {
            const size_t gr = S->grow[lr];
            if (!S->draw[gr]) { lr++; continue; }
            size_t run = 1;
            while (lr + run < lr1 && S->grow[lr + run] == gr + run && S->draw[gr + run] && S->enc_pos[gr + run] == S->enc_pos[gr + run - 1] + pad) run++;
            lig::launch_rng_fill_rows(s, c->rk_dev, S->enc_pos[gr], S->msgs + lr * (size_t)k, run, pad, k, l, 1, pad);
            lr += run;
        }
    };
    if (!S->rows_by_thread) draw_pads(0, Rl);
    uint64_t epos = S->enc_pos[R];
    fr* mask = S->maskcw; fr* mlin = mask + n; fr* mquad = mask + 2 * (size_t)n;                        // masks: formed by every rank
    const size_t k3 = 3 * (size_t)k, kq = ncol / 4;                                                      // kq = positions per coset of a rank's columns
    HIP_TRY(c, hipMemsetAsync(mask, 0, 3 * (size_t)n * 32, s));
    lig::launch_rng_fill_rows(s, c->rk_dev, epos, mask, 1, l, 0, 0, 1, 0); epos += l;
    lig::launch_rng_fill_rows(s, c->rk_dev, epos, mlin, 1, l - 1, 0, 1, 2, 0); epos += l - 1;
    lig::launch_sum_elems(s, mlin + 1, l - 1, 2, S->dots, mlin + 2 * (size_t)(l - 1) + 1);      // closing slot = -(sum of the others), on the device
    lig::launch_rng_fill_rows(s, c->rk_dev, epos, mlin, 1, 2 * pad, 0, 2 * l, 1, 0); epos += 2 * pad;
    lig::launch_rng_fill_rows(s, c->rk_dev, epos, mquad, 1, l, 0, 1, 2, 0); epos += l;
    lig::launch_rng_fill_rows(s, c->rk_dev, epos, mquad, 1, 2 * pad, 0, 2 * l, 1, 0); epos += 2 * pad;
    TRY(lig_sha_init(c, S->sha_state, ncol));
    HIP_TRY(c, hipEventRecord(c->ev_fork, s));
    HIP_TRY(c, hipStreamWaitEvent(s_hash, c->ev_fork, 0));
    HIP_TRY(c, hipStreamWaitEvent(s_comm, c->ev_fork, 0));
    // the three mask rows do not depend on the witness: encoded on the hash stream under the first round's encode
    TRY(lig_internal_encode_generic(c, mask, s_hash));
    TRY(lig_internal_encode_2k_rows(c, mlin, 2, s_hash));
    const bool exchange = W > 1 || S->exchange_even_alone;
    const size_t blk_elems = CAP * ncol;                         // one block of the exchange buffers (per rank pair and round)
    uint64_t absorbed = 0;
    for (size_t cidx = 0; cidx < S->rounds; cidx++) {
        const int pb = (int)(cidx & 1);
        const size_t lb = S->lrow0[cidx], nb = S->lrow0[cidx + 1] - lb;
        fr* sendb = S->send + (size_t)pb * CAP * n; fr* recvb = S->recv + (size_t)pb * CAP * n;
        if (S->rows_by_thread) {                              // my c-th chunk has arrived from the host
            HIP_TRY(c, hipStreamWaitValue32(s, S->up_flag_dev + cidx, S->up_seq, hipStreamWaitValueGte, 0xffffffffu));
            draw_pads(lb, lb + nb);
        }
        if (nb) TRY(lig_internal_encode_rows(c, S->msgs + lb * (size_t)k, S->cw + lb * k3, nb, lig::ENC_PLANAR, s));
        if (exchange) {
            if (cidx >= 2) HIP_TRY(c, hipStreamWaitEvent(s, S->ev_comm[pb], 0));          // send buffer free again (exchange c-2 done)
            if (nb) hipLaunchKernelGGL(lig::k_pack_slices, dim3(2048), dim3(256), 0, s, lig::CwView{S->msgs + lb * (size_t)k, S->cw + lb * k3, k}, sendb, nb, kq, CAP);
        }
        HIP_TRY(c, hipEventRecord(S->ev_enc[pb], s));
        if (exchange) {
            S->used_comm = true;
            if (ordered) {
                HIP_TRY(c, hipStreamWaitEvent(s_comm, S->ev_enc[pb], 0));
                if (cidx >= 2) HIP_TRY(c, hipStreamWaitEvent(s_comm, S->ev_hash[pb], 0));   // receive buffer free again (hash c-2 done)
                if (S->comm.all_to_all_on(S->comm.user, sendb, recvb, blk_elems * 32, s_comm)) return comm_fail("all_to_all(codeword column slices)");
                HIP_TRY(c, hipEventRecord(S->ev_comm[pb], s_comm));
            } else {
                HIP_TRY(c, hipStreamSynchronize(s));
                HIP_TRY(c, hipStreamSynchronize(s_hash));                               // receive buffer free again
                if (S->comm.all_to_all(S->comm.user, sendb, recvb, blk_elems * 32)) return comm_fail("all_to_all(codeword column slices)");
                HIP_TRY(c, hipEventRecord(S->ev_comm[pb], s));
            }
            HIP_TRY(c, hipStreamWaitEvent(s_hash, S->ev_comm[pb], 0));
            for (uint32_t h = 0; h < W; h++) {               // global chunks cW .. cW+W-1 in rank order = commit order
                const size_t rg = S->chunk_rows(cidx * W + h);
                if (rg) lig::launch_sha_update_rows(s_hash, S->sha_state, ncol, recvb + (size_t)h * blk_elems, ncol, rg, absorbed);
                absorbed += rg;
            }
        } else {                                             // one rank: its codewords are the rows, all columns are its own
            HIP_TRY(c, hipStreamWaitEvent(s_hash, S->ev_enc[pb], 0));
            if (nb) lig::launch_sha_update_rows(s_hash, S->sha_state, ncol, S->cw + lb * k3, 0, nb, absorbed, k, S->msgs + lb * (size_t)k);
            absorbed += nb;
        }
        HIP_TRY(c, hipEventRecord(S->ev_hash[pb], s_hash));
    }
    lig::launch_sha_update_rows(s_hash, S->sha_state, ncol, mask + (size_t)S->rank * ncol, n, 3, absorbed, (uint32_t)kq, nullptr);   // plane-major instances over interleaved rows
    absorbed += 3;
    c->sha[S->sha_state].second = absorbed;
    HIP_TRY(c, hipEventRecord(c->ev_join, s_hash));
    HIP_TRY(c, hipStreamWaitEvent(s, c->ev_join, 0));
    if (exchange && ordered) {                                // the main stream also joins the exchange stream
        HIP_TRY(c, hipEventRecord(c->ev_fork, s_comm));
        HIP_TRY(c, hipStreamWaitEvent(s, c->ev_fork, 0));
    }
    lig::launch_sha_final(s, S->sha_state, ncol, absorbed, S->leaves_slice, (uint32_t)kq);      // plane-major instances -> the rank's leaves in column order
    uint32_t* leaf_level = S->nodes + 8 * ((size_t)n - 1);         // the gathered leaves ARE the last level of the node heap (n is a power of two)
    TRY(all_gather(S->leaves_slice, leaf_level, ncol * 32, s, "all_gather(leaves)"));
    TRY(lig_merkle_build(c, leaf_level, n, S->nodes));
    HIP_TRY(c, hipMemcpyAsync(info->root, S->nodes, 32, hipMemcpyDeviceToHost, s));
    TRY(drain(s, "stage 1 (exchange, column hash, leaves)"));
    if (S->rows_by_thread) {                                  // every round has been waited for: the caller's rows are no longer read
        S->rows_by_thread = false;
        if (const int e = S->up_failed.exchange(0)) FAIL(c, LIG_E_HIP, std::string("rows upload failed: ") + hipGetErrorString((hipError_t)e));
    }
    Sha256().add("LigetronStage1", 15).add(info->root, 32).add(S->ih, 32).finish(info->stage1_seed);
    return LIG_OK;
}

static int shard_stage23(lig_shard* S, const ShardRands& rs, const uint8_t* const_sum_given, const uint8_t** proof, size_t* proof_len, lig_proof_info* info) {
    SHARD_COMMON;
    auto t0 = clk::now();
    uint32_t rk[60];
    uint32_t rseq = 0;
    ShardDebugScope dbg(S, "stages 2-3", &rseq);

    // ---------------- stage 2
    const size_t NTl = S->triple_ord.size();
    {
        std::vector<H::Fr> rc, rq;
        const size_t NT = quad_terms(S->rows).size() / 3;
        FieldStream code_s(info->stage1_seed), quad_s(info->stage1_seed);
        code_s.next(S->code_ord[R], rc);
        quad_s.next(NT, rq);
        std::vector<lig::f29s> coef(Rl + 2 * NTl + 1);
        const H::Fr R261sq = H::mul(R261, R261);
        std::memset(coef.data(), 0, coef.size() * sizeof(lig::f29s));
        for (size_t lr = 0; lr < Rl; lr++) if (has_code_check(S->rows[S->grow[lr]].kind)) coef[lr] = to_f29s_host(rc[S->code_ord[S->grow[lr]]], R261);
        for (size_t i = 0; i < NTl; i++) { coef[Rl + i] = to_f29s_host(rq[S->triple_ord[i]], R261sq); coef[Rl + NTl + i] = to_f29s_host(rq[S->triple_ord[i]], R261); }
        TRY(lig_internal_upload_small(c, S->coef_dev, coef.data(), coef.size() * sizeof(lig::f29s), s));
        lig::aes256_expand_host(info->stage1_seed, rk);
        TRY(lig_internal_upload_small(c, c->rk_dev, rk, sizeof rk, s));
    }
    fr* code = S->acc; fr* lin = S->acc + n; fr* quad = S->acc + 2 * (size_t)n; fr* tmp = S->acc + 3 * (size_t)n;
    fr* linH = lin + 2 * (size_t)k; fr* linC = lin + 3 * (size_t)k;
    fr* mask = S->maskcw; fr* mlin = mask + n; fr* mquad = mask + 2 * (size_t)n;       // the mask codewords of stage 1
    const size_t kq = ncol / 4;
    (void)kq;
    HIP_TRY(c, hipMemsetAsync(S->acc, 0, 4 * (size_t)n * 32, s));
    // as in lig_synth_prove: the randomness rows of chunk c+1 are sampled on the side stream (dense rows, double-buffered)
    // while the main stream encodes / accumulates chunk c; group partials persist across chunks, one combine per accumulator
    const uint32_t pg = (uint32_t)((CAP + lig_tune::GROUP / 4 - 1) / (lig_tune::GROUP / 4));
    fr* p_code = S->parts; fr* p_linH = S->parts + (size_t)pg * k; fr* p_linC = S->parts + 2 * (size_t)pg * k;
    const uint32_t dot_groups = (uint32_t)((CAP + lig_tune::DOT_GROUP - 1) / lig_tune::DOT_GROUP);      // p_linC: dot_groups x k
    HIP_TRY(c, hipMemsetAsync(S->parts, 0, (2 * (size_t)pg + dot_groups) * k * 32, s));
    HIP_TRY(c, hipEventRecord(c->ev_fork, s));                 // key + memsets above
    HIP_TRY(c, hipStreamWaitEvent(s_hash, c->ev_fork, 0));
    // as in lig_synth_prove: the sampler also accumulates the message-domain halves of the code / linear tests (k_rand_rlc)
    const bool fused_rlc = !rs.dev && !rs.host && (k % 256 == 0) && lig::knobs().fused_rlc;
    // the caller's randomness rows (lig_shard_rows_prove): device rows are used in place, host rows go through the double buffer
    auto rand_buf = [&](size_t cidx) -> fr* { return rs.dev ? const_cast<fr*>(rs.dev) + S->lrow0[cidx] * (size_t)k : S->randb + (cidx & 1) * CAP * (size_t)k; };
    // host rows: the uploader thread fills the double buffer round by round (prover.hip, prove_stage23: same scheme -- arrival and
    // consumption are words in pinned host memory, no copy or event of the transfer in a queue of the proof)
    // (default off in the sharded entry, see shard_rows_load: event-chained copies on the side stream instead)
    const bool rands_by_thread = rs.host && lig::knobs().shard_uploader && lig::knobs().upload_mode == 2 && lig::knobs().rands_upload_mode == 2 && lig_internal_uploader_available(c) && S->rounds;
    const size_t rflag0 = S->rounds, uflag0 = 2 * S->rounds;
    if (rands_by_thread) {
        TRY(shard_up_flags(S));
        rseq = ++S->rand_seq;
        S->up_abort.store(0, std::memory_order_release);
        std::vector<UploadJob> jobs;
        for (size_t cidx = 0; cidx < S->rounds; cidx++) {
            const size_t lb = S->lrow0[cidx], nb = S->lrow0[cidx + 1] - lb;
            UploadJob j{(uint8_t*)rand_buf(cidx), rs.host + lb * (size_t)k * 32, nb * (size_t)k * 32, S->up_flag + rflag0 + cidx, rseq, &S->up_failed};
            if (cidx >= 2) { j.wait = S->up_flag + uflag0 + cidx - 2; j.wait_val = rseq; }
            j.abort = &S->up_abort; j.prio = 1;
            jobs.push_back(j);
        }
        lig_internal_uploader_submit(c->device, jobs, &S->up_pending);
    }
    auto form_rand_chunk = [&](size_t cidx) -> int {           // on the side stream
        const size_t lb = S->lrow0[cidx], nb = S->lrow0[cidx + 1] - lb;
        fr* rb = rand_buf(cidx);
        if (rs.dev) { HIP_TRY(c, hipEventRecord(S->ev_enc[cidx & 1], s_hash)); return LIG_OK; }
        if (rands_by_thread) return LIG_OK;
        if (cidx >= 2) HIP_TRY(c, hipStreamWaitEvent(s_hash, S->ev_comm[cidx & 1], 0));        // buffer consumed (event reused: stage 1 is over)
        if (rs.host) {
            if (nb) HIP_TRY(c, hipMemcpyAsync(rb, rs.host + lb * (size_t)k * 32, nb * (size_t)k * 32, hipMemcpyHostToDevice, s_hash));
            HIP_TRY(c, hipEventRecord(S->ev_enc[cidx & 1], s_hash));
            return LIG_OK;
        }
        for (size_t r = 0; r < nb;) {          // a chunk is a run of consecutive global rows: runs of equal fill are contiguous in the linear stream
            size_t run = 1;
            const size_t gr = S->grow[lb + r];
            const uint32_t d = S->rows[gr].data;
            while (r + run < nb && S->rows[gr + run].data == d) run++;
            if (fused_rlc) lig::launch_rand_rlc(s_hash, c->rk_dev, S->lin_pos[gr], rb + r * k, S->msgs + (lb + r) * (size_t)k, run, d, k, S->coef_dev + lb + r, lig_tune::GROUP / 4, p_code, p_linH);
            else lig::launch_rng_fill_rows_dense(s_hash, c->rk_dev, S->lin_pos[gr], rb + r * k, run, d, k);
            r += run;
        }
        HIP_TRY(c, hipEventRecord(S->ev_enc[cidx & 1], s_hash));
        return LIG_OK;
    };
    if (S->rounds) TRY(form_rand_chunk(0));
    for (size_t cidx = 0; cidx < S->rounds; cidx++) {
        const size_t lb = S->lrow0[cidx], nb = S->lrow0[cidx + 1] - lb;
        fr* rb = rand_buf(cidx);
        if (cidx + 1 < S->rounds) TRY(form_rand_chunk(cidx + 1));
        if (rands_by_thread) HIP_TRY(c, hipStreamWaitValue32(s, S->up_flag_dev + rflag0 + cidx, rseq, hipStreamWaitValueGte, 0xffffffffu));
        else HIP_TRY(c, hipStreamWaitEvent(s, S->ev_enc[cidx & 1], 0));
        if (nb) {
            if (c->fast) {      // as in lig_synth_prove: coset-2 values times the codewords' coset-2 plane inside the encoder's output kernel
                TRY(lig_internal_encode_dot(c, rb, nb, S->cw + lb * 3 * (size_t)k + k, 3 * (size_t)k, lig_tune::DOT_GROUP, p_linC));
            } else {
                TRY(lig_internal_encode_rows(c, rb, S->rhalf, nb, lig::ENC_HALF));
                lig::launch_rlc_accumulate29(s, S->cw + lb * 3 * (size_t)k + k, 3 * (size_t)k, 1, S->rhalf, k, nb, k, nullptr, nullptr, p_linC, lig_tune::DOT_GROUP);
            }
            if (!fused_rlc) lig::launch_rlc_accumulate29(s, S->msgs + lb * (size_t)k, k, 1, rb, k, nb, k, S->coef_dev + lb, p_code, p_linH, lig_tune::GROUP / 4);
        }
        if (rands_by_thread) HIP_TRY(c, hipStreamWriteValue32(s, S->up_flag_dev + uflag0 + cidx, rseq, 0));      // consumed: its half of the buffer is free
        else HIP_TRY(c, hipEventRecord(S->ev_comm[cidx & 1], s));
    }
    lig::launch_rlc_combine(s, tmp, p_code, pg, k);
    lig::launch_rlc_combine(s, linH, p_linH, pg, k);
    lig::launch_rlc_combine(s, linC, p_linC, dot_groups, k);
    lig::launch_lin_interleave(s, lin, linH, linC, k);       // see lig_synth_prove: even points of <w_n^2> = message domain
    const lig::CwView view{S->msgs, S->cw, k};
    lig::launch_quad_rows29_view(s, view, 2 * k, S->tri_dev, S->coef_dev + Rl, S->coef_dev + Rl + NTl, NTl, quad, S->parts, 2 * (size_t)pg * k);
    // partial sums [code (k message values) | lin (2k) | quad (2k)] -> every rank -> added mod p (one rank: they are the sums)
    if (W > 1 || S->exchange_even_alone) {
        HIP_TRY(c, hipMemcpyAsync(S->accp, tmp, (size_t)k * 32, hipMemcpyDeviceToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(S->accp + k, lin, 2 * (size_t)k * 32, hipMemcpyDeviceToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(S->accp + 3 * (size_t)k, quad, 2 * (size_t)k * 32, hipMemcpyDeviceToDevice, s));
        TRY(all_gather(S->accp, S->accg, 5 * (size_t)k * 32, s, "all_gather(partial accumulators)"));
        HIP_TRY(c, hipMemsetAsync(S->accp, 0, 5 * (size_t)k * 32, s));
        lig::launch_rlc_combine(s, S->accp, S->accg, W, 5 * k);
        HIP_TRY(c, hipMemsetAsync(S->acc, 0, 4 * (size_t)n * 32, s));
        HIP_TRY(c, hipMemcpyAsync(tmp, S->accp, (size_t)k * 32, hipMemcpyDeviceToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(lin, S->accp + k, 2 * (size_t)k * 32, hipMemcpyDeviceToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(quad, S->accp + 3 * (size_t)k, 2 * (size_t)k * 32, hipMemcpyDeviceToDevice, s));
    } else {
        HIP_TRY(c, hipMemsetAsync(lin + 2 * (size_t)k, 0, (size_t)(n - 2 * k) * 32, s));      // linH / linC scratch behind the 2k values
    }
    H::Fr* dots = reinterpret_cast<H::Fr*>(S->h_small);
    lig::launch_sum_elems(s, lin, k, 2, S->dots, nullptr);   // linear-test constant = -(sum of the message-domain half: the even points)
    HIP_TRY(c, hipMemcpyAsync(dots, S->dots, 32, hipMemcpyDeviceToHost, s));
    // as in lig_synth_prove: each accumulator is extended, masked and sent to the host as soon as it is final; the host absorbs
    // it into the stage-2 seed hash (sequential SHA-256 over 3 MiB) while the GPU extends the next one and runs the decodes
    uint8_t* enc = S->h_enc;
    const size_t vec_bytes = (size_t)n * 32, enc_bytes = 3 * vec_bytes;
    hipEvent_t ev_acc[3] = {S->ev_enc[0], S->ev_enc[1], S->ev_hash[0]};      // stage 1 is over: its events are free
    TRY(lig_internal_encode_rows(c, tmp, code, 1, false));
    lig::launch_eltwise(s, LIG_OP_ADD_ASSIGN, mask, nullptr, code, n, fr{}, 0);
    HIP_TRY(c, hipMemcpyAsync(enc, code, vec_bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipEventRecord(ev_acc[0], s));
    TRY(lig_internal_extend_2k(c, lin));
    lig::launch_eltwise(s, LIG_OP_ADD_ASSIGN, mlin, nullptr, lin, n, fr{}, 0);
    HIP_TRY(c, hipMemcpyAsync(enc + vec_bytes, lin, vec_bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipEventRecord(ev_acc[1], s));
    TRY(lig_internal_extend_2k(c, quad));
    lig::launch_eltwise(s, LIG_OP_ADD_ASSIGN, mquad, nullptr, quad, n, fr{}, 0);
    HIP_TRY(c, hipMemcpyAsync(enc + 2 * vec_bytes, quad, vec_bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipEventRecord(ev_acc[2], s));
    H::Fr* dec = reinterpret_cast<H::Fr*>(S->h_small + 32);
    const fr* accs[3] = {code, lin, quad};
    for (int a3 = 0; a3 < 3; a3++) {
        TRY(lig_internal_decode_to(c, accs[a3], tmp));
        HIP_TRY(c, hipMemcpyAsync(dec + (size_t)a3 * n, tmp, vec_bytes, hipMemcpyDeviceToHost, s));
    }
    const size_t n_nodes = lig_merkle_nodes(n);
    HIP_TRY(c, hipMemcpyAsync(S->h_nodes, S->nodes, n_nodes * 32, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipEventRecord(c->ev_join, s));
    {
        Sha256 h2;
        h2.add("LigetronStage2", 15).add(info->root, 32);
        for (int a3 = 0; a3 < 3; a3++) {
            TRY(drain_event(ev_acc[a3], "stage 2 (accumulators)"));
            h2.add(enc + (size_t)a3 * vec_bytes, vec_bytes);
        }
        h2.finish(info->stage2_seed);
    }
    (void)enc_bytes;
    if (const_sum_given) std::memcpy(info->const_sum, const_sum_given, 32);     // the caller's public constant (linear_sums)
    else {
        const H::Fr sum = H::neg(dots[0]);                    // (copied before ev_acc[0])
        std::memcpy(info->const_sum, sum.v, 32);
    }
    const std::vector<uint32_t> idx = sample_columns(info->stage2_seed, n, t);
    TRY(lig_sample_init(c, idx.data(), idx.size()));
    TRY(drain_event(c->ev_join, "stage 2 (decodes)"));     // decoded accumulators and Merkle nodes are on the host
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
    const std::vector<uint8_t> sib = decommit(S->h_nodes, (n_nodes + 1) / 2, idx);
    info->ms_stage2 = ms_since(t0);
    t0 = clk::now();

    // ---------------- stage 3
    lig::launch_gather_rows_planar(s, view, Rl, c->sample_idx, t, S->smp);                      // local rows, then the 3 masks
    lig::launch_gather_rows(s, S->maskcw, n, 3, c->sample_idx, t, S->smp + Rl * (size_t)t);
    TRY(all_gather(S->smp, S->smpg, RM * (size_t)t * 32, s, "all_gather(opened columns)"));
    char ver[17] = {0};
    std::memcpy(ver, S->version, 16);
    const size_t smp_bytes = (R + 3) * (size_t)t * 32;
    const EnvelopeLayout lay = write_envelope(S->h_proof, S->h_proof_cap, ver, S->program_hash, S->generated_at, k, n, t,
                                              info->root, sib, idx, enc, smp_bytes);
    if (lay.total > S->h_proof_cap) FAIL(c, LIG_E_NOMEM, "proof buffer too small");
    // opened columns in commit order: global chunk g = the (g / W)-th chunk of rank g mod W
    std::vector<size_t> taken(W, 0);
    uint8_t* dst = S->h_proof + lay.samples_off;
    for (size_t g = 0; g < S->G; g++) {
        const uint32_t h = (uint32_t)(g % W);
        const size_t rg = S->chunk_rows(g);
        if (rg) HIP_TRY(c, hipMemcpyAsync(dst, S->smpg + ((size_t)h * RM + taken[h]) * t, rg * (size_t)t * 32, hipMemcpyDeviceToHost, s));
        taken[h] += rg;
        dst += rg * (size_t)t * 32;
    }
    HIP_TRY(c, hipMemcpyAsync(dst, S->smp + Rl * (size_t)t, 3 * (size_t)t * 32, hipMemcpyDeviceToHost, s));
    TRY(drain(s, "stage 3 (opened columns)"));
    *proof = S->h_proof;
    *proof_len = lay.total;
    info->ms_stage3 = ms_since(t0);
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}

int lig_shard_prove(lig_shard* S, const uint8_t** proof, size_t* proof_len, lig_proof_info* info) {
    if (!S || !proof || !proof_len || !info) return LIG_E_ARG;
    lig_ctx* c = S->c;
    CHECK_CTX(c);
    if (S->poisoned) FAIL(c, LIG_E_STATE, "lig_shard_prove: the shard is poisoned (work queued behind a failed collective never drained): destroy it");
    if (S->from_rows) FAIL(c, LIG_E_STATE, "lig_shard_prove on a shard made by lig_shard_rows_begin (use lig_shard_rows_commit / _prove)");
    std::memset(info, 0, sizeof *info);
    info->rows = S->R + 3;
    const auto t_begin = clk::now();
    TRY(shard_stage1(S, info));
    info->ms_stage1 = ms_since(t_begin);
    TRY(shard_stage23(S, ShardRands{}, nullptr, proof, proof_len, info));
    info->ms_total = ms_since(t_begin);
    return LIG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The sharded prover over rows SUPPLIED BY THE CALLER (include/lig_hip.h: lig_shard_rows_*): what lig_rows_* is to
// lig_synth_*, for one trace over the GPUs of a node.  Every rank passes the kinds of ALL committed rows (the deal and the
// stream positions are global) and the message rows of ITS chunks only, in commit order; after the commit every rank's
// constraint generator derives the randomness rows of its own rows from the stage-1 seed.
// Replaces the per-row callbacks of include/zkp/nonbatch_context.hpp:445-471 (stage 1), :654-780 (stage 2), :924-970 (stage 3).
static bool kinds_to_rows(lig_ctx* c, const lig_rows_job* job, std::vector<RowDesc>& rows, std::vector<uint8_t>& draw, std::vector<uint64_t>& pos) {
    const uint32_t k = c->k, l = c->l, pad = k - l;
    const size_t R = job->rows;
    rows.resize(R); draw.assign(R, 0); pos.assign(R + 1, 0);
    for (size_t r = 0; r < R; r++) {
        const uint8_t kd = job->kinds[r] & 0x7f;
        if (kd > RK_BQZ) { c->err = "rows job: unknown row kind"; return false; }
        const bool first_of_3 = kd == 1 || kd == RK_BQX, first_of_2 = kd == RK_EQX;
        if (first_of_3 && !(r + 2 < R && (job->kinds[r + 1] & 0x7f) == kd + 1 && (job->kinds[r + 2] & 0x7f) == kd + 2)) { c->err = "rows job: incomplete x,y,z triple"; return false; }
        if (first_of_2 && !(r + 1 < R && (job->kinds[r + 1] & 0x7f) == RK_EQY)) { c->err = "rows job: incomplete equality pair"; return false; }
        const bool follower = kd == 2 || kd == 3 || kd == RK_EQY || kd == RK_BQY || kd == RK_BQZ;
        if (follower && !(r > 0 && (job->kinds[r - 1] & 0x7f) == kd - 1)) { c->err = "rows job: row of a group without its predecessor"; return false; }
        const bool draws = kd <= 3 || kd == RK_INIT;
        if (kd == RK_INIT && pad != 192) { c->err = "rows job: on_batch_init rows need k - l = 192 (params::sample_size)"; return false; }
        if ((job->kinds[r] & LIG_ROW_DRAW_PAD) && !draws) { c->err = "rows job: LIG_ROW_DRAW_PAD on a row kind that draws no padding upstream"; return false; }
        draw[r] = (job->kinds[r] & LIG_ROW_DRAW_PAD) ? 1 : 0;
        pos[r + 1] = pos[r] + (draws ? pad : 0);
        const uint32_t dense = job->dense_rands_per_row ? job->dense_rands_per_row[r] : 0;
        if (dense > k || (dense && kd > 3)) { c->err = "rows job: dense_rands_per_row out of range or on a batch row"; return false; }
        rows[r] = RowDesc{kd, dense};
    }
    return true;
}

int lig_shard_rows_plan(const uint8_t* kinds, size_t n_rows, uint32_t world, uint64_t* rounds_out, uint64_t* boundaries, size_t cap) {
    if ((n_rows && !kinds) || !world || !rounds_out) return LIG_E_ARG;
    std::vector<RowDesc> rows(n_rows);
    for (size_t r = 0; r < n_rows; r++) rows[r] = RowDesc{(uint8_t)(kinds[r] & 0x7f), 0};
    size_t rounds = 0;
    std::vector<size_t> gb;
    shard_chunks(rows, world, rounds, gb);
    *rounds_out = rounds;
    if (gb.size() > cap || !boundaries) return gb.size() > cap ? LIG_E_NOMEM : LIG_E_ARG;
    for (size_t i = 0; i < gb.size(); i++) boundaries[i] = gb[i];
    return LIG_OK;
}

static int shard_up_flags(lig_shard* S) {
    lig_ctx* c = S->c;
    if (S->up_flag) return LIG_OK;
    const size_t bytes = ((3 * S->rounds + 8) * 4 + 4095) & ~(size_t)4095;
    HIP_TRY(c, hipHostMalloc((void**)&S->up_flag, bytes, hipHostMallocDefault));
    std::memset((void*)S->up_flag, 0, bytes);
    HIP_TRY(c, hipHostGetDevicePointer((void**)&S->up_flag_dev, (void*)S->up_flag, 0));
    return LIG_OK;
}
static void shard_up_drain(lig_shard* S) { while (S->up_pending.load(std::memory_order_acquire) > 0) std::this_thread::yield(); }
// local rows of a rows job -> S->msgs.  Device rows: one copy on the main stream.  Host rows: round by round through the uploader
// thread -- lig_shard_rows_commit encodes my c-th chunk as soon as it has arrived (the caller's memory must stay valid until then).
static int shard_rows_load(lig_shard* S, const void* local_msgs, bool on_device) {
    lig_ctx* c = S->c;
    if (S->Rl && !local_msgs) FAIL(c, LIG_E_ARG, "sharded rows job: null local rows");
    shard_up_drain(S);                                 // an upload nobody committed
    HIP_TRY(c, hipStreamSynchronize(c->stream));      // the previous trace is done with S->msgs
    S->committed = false;
    S->rows_by_thread = false;
    if (!S->Rl) return LIG_OK;
    const size_t row_bytes = (size_t)c->k * 32;
    // Host rows are copied HERE, synchronously (as in round 3).  Round 4 had moved them onto the library's uploader thread with a stream
    // memory wait per round (what lig_rows_* does on ONE GPU, where it buys the link rate): with several PROCESSES on one device -- the only
    // way this entry runs on the one-GPU test box -- a DMA transfer that is started while the process's main stream holds a pending
    // hipStreamWaitValue32 sometimes never completes (profiles/r05_rows_entry_hang.md: 15 % of the runs, the GPUTEST hang of round 4).
    // LIG_SHARD_UPLOADER=1 restores the round-4 path (reproduction / a node where every rank has its own GPU).
    if (!on_device && lig::knobs().shard_uploader && lig::knobs().upload_mode == 2 && lig_internal_uploader_available(c)) {
        TRY(shard_up_flags(S));
        S->up_seq++;
        S->up_abort.store(0, std::memory_order_release);
        std::vector<UploadJob> jobs;
        for (size_t cidx = 0; cidx < S->rounds; cidx++) {
            const size_t lb = S->lrow0[cidx], nb = S->lrow0[cidx + 1] - lb;
            jobs.push_back(UploadJob{(uint8_t*)S->msgs + lb * row_bytes, (const uint8_t*)local_msgs + lb * row_bytes, nb * row_bytes, S->up_flag + cidx, S->up_seq, &S->up_failed});
        }
        lig_internal_uploader_submit(c->device, jobs, &S->up_pending);
        S->rows_by_thread = true;
        return LIG_OK;
    }
    HIP_TRY(c, hipMemcpyAsync(S->msgs, local_msgs, S->Rl * row_bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));      // the caller's memory is no longer referenced
    return LIG_OK;
}

int lig_shard_rows_begin(lig_ctx* c, const lig_rows_job* job, uint32_t rank, uint32_t world, const lig_comm* comm, lig_shard** out) {
    CHECK_CTX(c);
    if (!job || !out || !comm || world == 0 || rank >= world) return LIG_E_ARG;
    if (!comm->all_to_all || !comm->all_gather) return LIG_E_ARG;
    if (job->rows && !job->kinds) FAIL(c, LIG_E_ARG, "sharded rows job: null kinds");
    if (job->elem_bytes) FAIL(c, LIG_E_ARG, "sharded rows job: the narrow row format (elem_bytes) is not supported here, pass full-width rows");
    const uint32_t l = c->l, k = c->k, n = c->n, t = 192;
    if (l >= k || l < 2 || t > n || k - l < t || k % world) FAIL(c, LIG_E_ARG, "sharded trace: need 2 <= l <= k - 192 and world | k");
    *out = nullptr;
    lig_shard* S = new lig_shard();
    S->c = c; S->comm = *comm; S->rank = rank; S->world = world; S->from_rows = true;
    std::memset(&S->job, 0, sizeof S->job);
    std::memcpy(S->encoding_seed, job->encoding_seed, 32);
    std::memcpy(S->program_hash, job->program_hash, 32);
    std::memcpy(S->version, job->version, 16);
    S->generated_at = job->generated_at;
    S->dense_rands = job->dense_rands_per_row != nullptr;
    auto fail = [&](int rc) { lig_shard_destroy(S); return rc; };
    if (job->n_public_args && (!job->public_args || !job->public_arg_lens)) return fail(LIG_E_ARG);
    std::memset(S->ih, 0, 32);
    Sha256().add(S->ih, 32).add("Ligero", 7).finish(S->ih);
    const uint8_t* a = job->public_args;
    for (uint64_t i = 0; i < job->n_public_args; i++) {
        uint8_t prev[32];
        std::memcpy(prev, S->ih, 32);
        Sha256().add(prev, 32).add(a, job->public_arg_lens[i]).finish(S->ih);
        a += job->public_arg_lens[i];
    }
    if (!kinds_to_rows(c, job, S->rows, S->draw, S->enc_pos)) return fail(LIG_E_ARG);
    const size_t R = S->R = S->rows.size();
    S->lin_pos.assign(R + 1, 0); S->code_ord.assign(R + 1, 0); S->wit_pos.assign(R + 1, 0);
    for (size_t r = 0; r < R; r++) {
        S->lin_pos[r + 1] = S->lin_pos[r] + S->rows[r].data;
        S->code_ord[r + 1] = S->code_ord[r] + has_code_check(S->rows[r].kind);
    }
    int rc = shard_alloc(c, rank, world, S);
    if (rc == LIG_OK) rc = shard_rows_load(S, job->msgs, job->msgs_on_device != 0);
    if (rc != LIG_OK) return fail(rc);
    *out = S;
    return LIG_OK;
}
// the next trace of the same shape: new local rows into the same buffers
int lig_shard_rows_restart(lig_shard* S, const void* local_msgs, int msgs_on_device) {
    if (!S) return LIG_E_ARG;
    lig_ctx* c = S->c;
    CHECK_CTX(c);
    if (S->poisoned) FAIL(c, LIG_E_STATE, "lig_shard_rows_restart: the shard is poisoned (work queued behind a failed collective never drained): destroy it");
    if (!S->from_rows) FAIL(c, LIG_E_STATE, "lig_shard_rows_restart: not a rows shard");
    // (unlike lig_rows_restart there is no second message matrix here: stage 2 of a committed trace still reads the local rows)
    if (S->committed) FAIL(c, LIG_E_STATE, "lig_shard_rows_restart: the committed trace has not been proved yet");
    return shard_rows_load(S, local_msgs, msgs_on_device != 0);
}
int lig_shard_rows_commit(lig_shard* S, uint8_t root[32], uint8_t stage1_seed[32]) {
    if (!S) return LIG_E_ARG;
    lig_ctx* c = S->c;
    CHECK_CTX(c);
    if (S->poisoned) FAIL(c, LIG_E_STATE, "lig_shard_rows_commit: the shard is poisoned (work queued behind a failed collective never drained): destroy it");
    if (!S->from_rows) FAIL(c, LIG_E_STATE, "lig_shard_rows_commit: not a rows shard");
    if (S->committed) FAIL(c, LIG_E_STATE, "lig_shard_rows_commit: the committed trace has not been proved yet");
    std::memset(&S->info1, 0, sizeof S->info1);
    S->info1.rows = S->R + 3;
    const auto t_begin = clk::now();
    {
        const int rc = shard_stage1(S, &S->info1);
        if (rc != LIG_OK) {               // the caller is told it may free its rows
            const std::string why = c->err;
            shard_up_drain(S);
            const bool quiet = shard_quiesce(S, SHARD_QUIESCE_S);
            S->rows_by_thread = false;
            c->err = quiet ? why : why + SHARD_POISONED_MSG;
            return rc;
        }
    }
    S->info1.ms_stage1 = ms_since(t_begin);
    S->committed = true;
    if (root) std::memcpy(root, S->info1.root, 32);
    if (stage1_seed) std::memcpy(stage1_seed, S->info1.stage1_seed, 32);
    return LIG_OK;
}
int lig_shard_rows_prove(lig_shard* S, const void* local_rands, int rands_on_device, const uint8_t const_sum[32], const uint8_t** proof,
                         size_t* proof_len, lig_proof_info* info) {
    if (!S || !proof || !proof_len || !info) return LIG_E_ARG;
    lig_ctx* c = S->c;
    CHECK_CTX(c);
    if (S->poisoned) FAIL(c, LIG_E_STATE, "lig_shard_rows_prove: the shard is poisoned (work queued behind a failed collective never drained): destroy it");
    if (!S->from_rows || !S->committed) FAIL(c, LIG_E_STATE, "lig_shard_rows_prove: lig_shard_rows_commit has not run on this shard");
    if (S->Rl && !local_rands && !S->dense_rands) FAIL(c, LIG_E_ARG, "lig_shard_rows_prove: null randomness rows");
    if (const_sum) {
        H::Fr v;
        std::memcpy(v.v, const_sum, 32);
        if (H::geq(v, H::P)) FAIL(c, LIG_E_ARG, "lig_shard_rows_prove: constant not reduced mod p");
    }
    *info = S->info1;
    const auto t_begin = clk::now();
    ShardRands rs;
    if (local_rands && rands_on_device) rs.dev = (const fr*)local_rands; else if (local_rands) rs.host = (const uint8_t*)local_rands;
    {
        const int rc = shard_stage23(S, rs, const_sum, proof, proof_len, info);
        if (rc != LIG_OK) {               // randomness-row copies the uploader thread still holds read the caller's memory: drop them, wait
            const std::string why = c->err;
            S->up_abort.store(1, std::memory_order_release);
            shard_up_drain(S);
            const bool quiet = shard_quiesce(S, SHARD_QUIESCE_S);
            c->err = quiet ? why : why + SHARD_POISONED_MSG;
            return rc;
        }
        if (rs.host) {
            shard_up_drain(S);
            if (const int e = S->up_failed.exchange(0)) FAIL(c, LIG_E_HIP, std::string("randomness rows upload failed: ") + hipGetErrorString((hipError_t)e));
        }
    }
    info->ms_total = info->ms_stage1 + ms_since(t_begin);
    S->committed = false;
    return LIG_OK;
}

}  // extern "C"
