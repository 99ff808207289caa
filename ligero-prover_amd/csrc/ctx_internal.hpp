// ctx_internal.hpp -- the context object shared by lig_capi.hip and prover.hip (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <functional>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>
#include "../../include/lig_hip.h"
#include "kernels.hpp"

using lig::fr;


struct lig_ctx {
    int device = 0;
    uint32_t l = 0, k = 0, n = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;            // side stream: column hash and samplers, overlapped with the encodes on `stream`.  ONE per device and process,
                                              // shared by every context (side_shared; lig_capi.hip: why), unless LIG_SHARED_SIDE=0
    bool side_shared = false, copy_is_main = false;
    hipStream_t stream3 = nullptr;            // copy stream (stream-ordered host-row uploads, the sharded prover's exchange): created on first use, lig_internal_copy_stream()
    hipStream_t stream_sha = nullptr;         // experiment (LIG_SHA_CUMASK): a CU-masked stream for the stage-1 column hash; null: stream2
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    lig::NttPlan plan_half;                   // size 2k, root w_n^2
    std::string err;
    lig::NttPlan plan[3];
    lig::EncodePlan ep;
    // two-launch single-row transforms: [LIG_SIZE_K | LIG_SIZE_2K | LIG_SIZE_N][forward | inverse], the inverse on <w_n^2>
    lig::TiledPlan tplan[3][2], tplan_half_inv;
    bool tiled = false;
    fr* tiled_scratch[2] = {nullptr, nullptr};   // 3 rows of n elements each: [0] main stream, [1] side stream
    bool fast = false;
    std::vector<void*> owned;                 // device tables freed at destroy
    fr* scratch_y = nullptr; fr* scratch_z = nullptr; size_t scratch_rows = 0;
    std::unordered_map<void*, std::pair<size_t, uint64_t>> sha;   // state ptr -> (n_inst, rows absorbed)
    uint32_t* sample_idx = nullptr; size_t sample_count = 0, sample_cap = 0;
    std::vector<std::pair<void*, size_t>> vws; // verifier workspace: the device buffers of the last verification, reused slot by slot
    uint32_t* rk_dev = nullptr;               // 60 AES round-key words
    // small host -> device transfers on the proving path (round keys, coefficients, sample indices) go through a pinned
    // staging ring read by a copy KERNEL: a hipMemcpy H2D would queue behind the multi-hundred-MB row uploads of
    // lig_rows_* on the DMA engine and stall the proof for their whole duration (measured: stage 2 5.6 -> 15.4 ms)
    uint8_t* stage_host = nullptr; uint8_t* stage_dev = nullptr; size_t stage_cap = 0, stage_pos = 0;
    fr* small_dev = nullptr;                  // staging for per-call scalars (rc/rq/tables)
    size_t small_cap = 0;
    uint32_t* tri_dev = nullptr; size_t tri_cap = 0;
    // optional per-kernel timing (lig_profile_*): HIP events recorded on the ctx stream around the dominant kernel
    bool streams_shared = false;              // LIG_STREAM_MAP: the three streams are process-wide physical streams, not destroyed with the context
    bool prof_on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    size_t prof_used = 0;
    uint64_t prof_rows = 0;
    std::vector<uint32_t> prof_launch_rows;   // rows of bracketed launch i (lig_profile_read_launches)
    // communicators made on this context (comm_rccl.hip, comm_ipc.hip): (object, its finalizer); ended with the context
    std::vector<std::pair<void*, void (*)(void*)>> comms;
    // diagnostics: what the call in progress is waiting for (set by lig_shard_* around their queued work, read by a communicator's
    // watchdog thread when it declares the communicator dead: the reason a rank stopped is then on stderr, not guessed afterwards)
    std::mutex debug_mu;
    std::function<std::string()> debug_state;
};
// the context's copy stream, created on first use: a context that never uploads through stream-ordered copies (resident witness, the
// uploader thread) holds two streams, so that two contexts + the null stream fit the runtime's four hardware queues one to one
inline hipStream_t lig_internal_copy_stream(lig_ctx* c) {
    if (!c->stream3 && hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); c->stream3 = c->stream; c->copy_is_main = true; }
    return c->stream3;
}
inline void lig_internal_set_debug_state(lig_ctx* c, std::function<std::string()> f) { std::lock_guard<std::mutex> lk(c->debug_mu); c->debug_state = std::move(f); }
inline std::string lig_internal_debug_state(lig_ctx* c) { std::lock_guard<std::mutex> lk(c->debug_mu); return c->debug_state ? c->debug_state() : std::string(); }

// end every communicator that still lives on the context (each finalizer drains the context's streams first and clears the
// object's back pointer); called by lig_ctx_destroy
inline void lig_internal_comms_release(lig_ctx* c) {
    auto v = std::move(c->comms);
    c->comms.clear();
    for (auto& e : v) e.second(e.first);
}
inline void lig_internal_comm_unregister(lig_ctx* c, void* obj) {
    for (size_t i = 0; i < c->comms.size(); i++) if (c->comms[i].first == obj) { c->comms.erase(c->comms.begin() + i); return; }
}

// tests (LIG_FAULT_COMM): the library's communicators (comm_rccl.hip, comm_ipc.hip) ask this at the top of their all-to-all.
// 1: the stream-ordered form fails, 2: the host-synchronous form fails too, 3: the stream-ordered form never returns (the caller's
// watchdog has to end the process).  Used to show that bench.py's transport ladder falls through to the next rung.
int lig_internal_comm_fault(lig_ctx* c, bool stream_ordered);

// mode: lig::ENC_FULL (0, rows x n) / ENC_HALF (1, rows x k, coset 2) / ENC_PLANAR (2, rows x 3k, cosets 1..3 as planes) /
// ENC_ZRES (4, rows x 3k, cosets 1..3 as the tile kernel's Z tiles: no last radix-8 pass; fast encoder only)
// phases (fast encoder, rows <= one launch group): 1 = K1 only, 14 = everything after K1 (the Y scratch carries the rows in between)
int lig_internal_encode_rows(lig_ctx* c, const void* msgs, void* out, size_t rows, int mode, hipStream_t on = nullptr, int phases = 15,
                             void* y_scratch = nullptr, void* z_scratch = nullptr);      // (own Y / Z scratch of >= rows rows: a caller that pipelines chunks)
// cw2_z: the rows of cw2 are Z tiles of coset 2 (lig::ENC_ZRES) instead of coset values
int lig_internal_encode_dot(lig_ctx* c, const void* rands, size_t rows, const void* cw2, size_t cw2_stride, uint32_t group_rows, void* part, hipStream_t on = nullptr,
                            bool cw2_z = false);
int lig_internal_extend_2k(lig_ctx* c, void* buf);
// decode_ntt_device of `src` (n elements, left intact) into `dst` (n elements, != src), on the context stream
int lig_internal_decode_to(lig_ctx* c, const void* src, void* dst);
int lig_internal_encode_2k_rows(lig_ctx* c, void* buf, size_t rows, hipStream_t on = nullptr);
int lig_internal_encode_generic(lig_ctx* c, void* buf, hipStream_t on = nullptr);
// dst (device) <- src (host, any memory), `bytes` a multiple of 4, enqueued on `st`; src may be reused as soon as this returns
int lig_internal_upload_small(lig_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t st);
// host_pinned (hipHostMalloc'ed, 16-byte aligned) <- src (device), enqueued on `st` as a copy kernel (no DMA engine); the data is
// visible to the host once work queued behind it on `st` has been waited for (event / stream synchronize)
int lig_internal_download(lig_ctx* c, void* host_pinned, const void* src, size_t bytes, hipStream_t st);
// make the shared encode scratch large enough for `rows` rows per launch group (all context streams are drained first)
int lig_internal_reserve_scratch(lig_ctx* c, size_t rows);

// every entry point may be called from any host thread (bench.py proves from worker threads): the context's device is made
// current for the calling thread first -- HIP streams and allocations are only usable with their own device current
#define CHECK_CTX(c) do { if (!(c)) return LIG_E_ARG; if (hipSetDevice((c)->device) != hipSuccess) { (c)->err = "hipSetDevice failed"; return LIG_E_HIP; } } while (0)
#define HIP_TRY(c, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { (c)->err = std::string(#call) + ": " + hipGetErrorString(e__); return LIG_E_HIP; } } while (0)
#define FAIL(c, code, msg) do { (c)->err = (msg); return (code); } while (0)

