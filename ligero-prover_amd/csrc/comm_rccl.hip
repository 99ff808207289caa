// comm_rccl.hip -- the collectives of the sharded prover on RCCL (include/lig_hip.h: lig_rccl_*).
//
// One process per GPU; the communicator spans the GPUs of the node over xGMI.  The exchange of codeword column slices
// is a grouped ncclSend / ncclRecv (xGMI is point-to-point: every pair of GPUs has its own link, the W-1 transfers of a
// rank run on W-1 different links at once), leaves / partial sums / opened columns are ncclAllGather.  Everything is
// enqueued on the HIP stream the caller names -- the sharded prover overlaps the exchange of round c with the encode of
// round c+1 and the column hash of round c-1 -- and nothing here blocks the host.  The reference has no counterpart: it
// is single-device ("Collective call sites: none").
//
// librccl is NOT a link-time dependency of liblig_hip.so: it is resolved on the first lig_rccl_* call, and the copy that is
// already mapped into the process wins (a torch process has its own librccl.so.1 next to torch; a second copy from
// /opt/rocm/lib in the same process would be a version-skew risk on an 8-GPU node).  Single-GPU consumers and hosts that
// only use the transcript helpers load the library where RCCL is not installed; lig_rccl_* then return LIG_E_STATE.
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <type_traits>

#include "ctx_internal.hpp"

namespace {

struct RcclApi {
    void* handle = nullptr;
    std::string path, why;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;     // optional (failure reporting)
    decltype(&ncclCommAbort) CommAbort = nullptr;                     // optional
    bool ok = false;
};

RcclApi& api() {
    static RcclApi a;
    static std::once_flag once;
    std::call_once(once, [] {
        // 1. LIG_RCCL_LIB (explicit), 2. whatever librccl.so.1 is already mapped (RTLD_NOLOAD matches by SONAME: torch's copy
        // in a torch process), 3. the loader's search path, 4. the ROCm install
        const std::string& env = lig::knobs().rccl_lib;
        if (!env.empty()) a.handle = dlopen(env.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!a.handle) a.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!a.handle) a.handle = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
        if (!a.handle) a.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!a.handle) a.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!a.handle) { const char* e = dlerror(); a.why = std::string("librccl not found: ") + (e ? e : "?"); return; }
        bool all = true;
        auto sym = [&](auto& fn, const char* name) { fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(a.handle, name)); if (!fn) { all = false; a.why = std::string("librccl lacks ") + name; } };
        sym(a.GetUniqueId, "ncclGetUniqueId"); sym(a.CommInitRank, "ncclCommInitRank"); sym(a.CommDestroy, "ncclCommDestroy");
        sym(a.CommCount, "ncclCommCount"); sym(a.GetErrorString, "ncclGetErrorString"); sym(a.GetVersion, "ncclGetVersion");
        sym(a.GroupStart, "ncclGroupStart"); sym(a.GroupEnd, "ncclGroupEnd"); sym(a.Send, "ncclSend"); sym(a.Recv, "ncclRecv");
        sym(a.AllGather, "ncclAllGather");
        a.CommGetAsyncError = reinterpret_cast<decltype(a.CommGetAsyncError)>(dlsym(a.handle, "ncclCommGetAsyncError"));
        a.CommAbort = reinterpret_cast<decltype(a.CommAbort)>(dlsym(a.handle, "ncclCommAbort"));
        a.ok = all;
        Dl_info info;
        if (all && dladdr(reinterpret_cast<void*>(a.Send), &info) && info.dli_fname) a.path = info.dli_fname;
    });
    return a;
}

struct RcclComm {
    ncclComm_t comm = nullptr;
    lig_ctx* ctx = nullptr;                 // cleared by lig_ctx_destroy (lig_internal_comms_release): the context may die first
    uint32_t rank = 0, world = 1;
    bool in_sync = false;                   // a2a_sync is running a2a_on (fault injection tells the two forms apart)
    bool aborted = false;
};

int fail(RcclComm* r, const char* what, ncclResult_t e) {
    if (r && r->ctx) r->ctx->err = std::string(what) + ": " + api().GetErrorString(e);
    return 1;
}

int a2a_on(void* user, const void* send, void* recv, size_t block, void* stream) {
    RcclComm* r = static_cast<RcclComm*>(user);
    if (!r->comm) return 1;
    if (!r->in_sync && lig_internal_comm_fault(r->ctx, true)) return 1;
    RcclApi& A = api();
    hipStream_t st = static_cast<hipStream_t>(stream);
    ncclResult_t e = A.GroupStart();
    if (e != ncclSuccess) return fail(r, "ncclGroupStart", e);
    for (uint32_t h = 0; h < r->world; h++) {
        e = A.Send(static_cast<const uint8_t*>(send) + (size_t)h * block, block, ncclUint8, (int)h, r->comm, st);
        if (e != ncclSuccess) { (void)A.GroupEnd(); return fail(r, "ncclSend", e); }
        e = A.Recv(static_cast<uint8_t*>(recv) + (size_t)h * block, block, ncclUint8, (int)h, r->comm, st);
        if (e != ncclSuccess) { (void)A.GroupEnd(); return fail(r, "ncclRecv", e); }
    }
    e = A.GroupEnd();
    return e == ncclSuccess ? 0 : fail(r, "ncclGroupEnd", e);
}
int ag_on(void* user, const void* send, void* recv, size_t bytes, void* stream) {
    RcclComm* r = static_cast<RcclComm*>(user);
    if (!r->comm) return 1;
    const ncclResult_t e = api().AllGather(send, recv, bytes, ncclUint8, r->comm, static_cast<hipStream_t>(stream));
    return e == ncclSuccess ? 0 : fail(r, "ncclAllGather", e);
}
// host-synchronous forms: same collectives on the context stream, then wait
int a2a_sync(void* user, const void* send, void* recv, size_t block) {
    RcclComm* r = static_cast<RcclComm*>(user);
    if (!r->ctx || lig_internal_comm_fault(r->ctx, false)) return 1;
    r->in_sync = true;
    const int rc = a2a_on(user, send, recv, block, r->ctx->stream);
    r->in_sync = false;
    if (rc) return 1;
    return hipStreamSynchronize(r->ctx->stream) == hipSuccess ? 0 : 1;
}
int ag_sync(void* user, const void* send, void* recv, size_t bytes) {
    RcclComm* r = static_cast<RcclComm*>(user);
    if (!r->ctx || ag_on(user, send, recv, bytes, r->ctx->stream)) return 1;
    return hipStreamSynchronize(r->ctx->stream) == hipSuccess ? 0 : 1;
}

// lig_comm.failed / .abort (include/lig_hip.h): what RCCL itself knows about the communicator (asynchronous errors of its proxy
// threads and transports), and ncclCommAbort -- queued RCCL kernels of this rank exit, the streams drain
int comm_failed(void* user) {
    RcclComm* r = static_cast<RcclComm*>(user);
    if (r->aborted) { if (r->ctx) r->ctx->err = "nccl: communicator aborted (a wait for queued collectives timed out)"; return 1; }
    if (!r->comm || !api().CommGetAsyncError) return 0;
    ncclResult_t st = ncclSuccess;
    if (api().CommGetAsyncError(r->comm, &st) != ncclSuccess || st == ncclSuccess || st == ncclInProgress) return 0;
    if (r->ctx) r->ctx->err = std::string("nccl asynchronous error: ") + api().GetErrorString(st);
    return 1;
}
void comm_abort(void* user) {
    RcclComm* r = static_cast<RcclComm*>(user);
    if (!r->comm || r->aborted) return;
    r->aborted = true;
    if (api().CommAbort) { (void)api().CommAbort(r->comm); r->comm = nullptr; }       // (abort also frees the communicator)
}

// drain the context's streams and end the communicator; the RcclComm object itself stays (the caller's lig_comm points to it)
void finalize(RcclComm* r) {
    if (r->ctx) {
        (void)hipSetDevice(r->ctx->device);
        (void)hipStreamSynchronize(r->ctx->stream); (void)hipStreamSynchronize(r->ctx->stream2); if (r->ctx->stream3) (void)hipStreamSynchronize(r->ctx->stream3);
    }
    if (r->comm) (void)api().CommDestroy(r->comm);
    r->comm = nullptr;
    r->ctx = nullptr;
}

}  // namespace

// (lig_ctx_destroy ends the communicators made on its context through this finalizer -- their streams are about to be
// destroyed; a later lig_rccl_comm_destroy of the same lig_comm only frees the bookkeeping)
static void finalize_erased(void* p) { finalize(static_cast<RcclComm*>(p)); }

extern "C" {

int lig_rccl_available(char* path_out, size_t cap, int* version) {
    RcclApi& A = api();
    if (path_out && cap) { std::strncpy(path_out, (A.ok ? A.path : A.why).c_str(), cap - 1); path_out[cap - 1] = 0; }
    if (version) { *version = 0; if (A.ok) (void)A.GetVersion(version); }
    return A.ok ? LIG_OK : LIG_E_STATE;
}

int lig_rccl_unique_id(uint8_t out[LIG_RCCL_ID_BYTES]) {
    static_assert(LIG_RCCL_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!out) return LIG_E_ARG;
    if (!api().ok) return LIG_E_STATE;
    ncclUniqueId id;
    if (api().GetUniqueId(&id) != ncclSuccess) return LIG_E_HIP;
    std::memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return LIG_OK;
}

int lig_rccl_comm_create(lig_ctx* c, const uint8_t id[LIG_RCCL_ID_BYTES], uint32_t rank, uint32_t world, lig_comm* out) {
    CHECK_CTX(c);
    if (!id || !out || !world || rank >= world) return LIG_E_ARG;
    std::memset(out, 0, sizeof *out);
    if (!api().ok) FAIL(c, LIG_E_STATE, api().why);
    RcclComm* r = new RcclComm();
    r->ctx = c; r->rank = rank; r->world = world;
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
    const ncclResult_t e = api().CommInitRank(&r->comm, (int)world, uid, (int)rank);
    if (e != ncclSuccess) { c->err = std::string("ncclCommInitRank: ") + api().GetErrorString(e); delete r; return LIG_E_HIP; }
    c->comms.push_back({r, finalize_erased});
    out->user = r;
    out->all_to_all = a2a_sync;
    out->all_gather = ag_sync;
    out->all_to_all_on = a2a_on;
    out->all_gather_on = ag_on;
    out->failed = comm_failed;
    out->abort = comm_abort;
    return LIG_OK;
}

int lig_rccl_comm_count(const lig_comm* comm, uint32_t* ranks) {
    if (!comm || !comm->user || !ranks || comm->all_to_all_on != a2a_on) return LIG_E_ARG;
    RcclComm* r = static_cast<RcclComm*>(comm->user);
    int cnt = 0;
    if (!r->comm || api().CommCount(r->comm, &cnt) != ncclSuccess) return LIG_E_STATE;
    *ranks = (uint32_t)cnt;
    return LIG_OK;
}

void lig_rccl_comm_destroy(lig_comm* comm) {
    if (!comm || !comm->user || comm->all_to_all_on != a2a_on) return;
    RcclComm* r = static_cast<RcclComm*>(comm->user);
    if (r->ctx) lig_internal_comm_unregister(r->ctx, r);
    finalize(r);
    delete r;
    std::memset(comm, 0, sizeof *comm);
}

}  // extern "C"
