// comm_rccl.hip -- the collectives of the sharded prover on RCCL (include/lig_hip.h: lig_rccl_*).
//
// One process per GPU; the communicator spans the GPUs of the node over xGMI.  The exchange of codeword column slices
// is a grouped ncclSend / ncclRecv (xGMI is point-to-point: every pair of GPUs has its own link, the W-1 transfers of a
// rank run on W-1 different links at once), leaves / partial sums / opened columns are ncclAllGather.  Everything is
// enqueued on the HIP stream the caller names -- the sharded prover overlaps the exchange of round c with the encode of
// round c+1 and the column hash of round c-1 -- and nothing here blocks the host.  The reference has no counterpart: it
// is single-device ("Collective call sites: none").
#include <rccl/rccl.h>

#include <cstring>
#include <string>

#include "ctx_internal.hpp"

namespace {

struct RcclComm {
    ncclComm_t comm = nullptr;
    lig_ctx* ctx = nullptr;
    uint32_t rank = 0, world = 1;
};

int fail(RcclComm* r, const char* what, ncclResult_t e) {
    if (r && r->ctx) r->ctx->err = std::string(what) + ": " + ncclGetErrorString(e);
    return 1;
}

int a2a_on(void* user, const void* send, void* recv, size_t block, void* stream) {
    RcclComm* r = static_cast<RcclComm*>(user);
    hipStream_t st = static_cast<hipStream_t>(stream);
    ncclResult_t e = ncclGroupStart();
    if (e != ncclSuccess) return fail(r, "ncclGroupStart", e);
    for (uint32_t h = 0; h < r->world; h++) {
        e = ncclSend(static_cast<const uint8_t*>(send) + (size_t)h * block, block, ncclUint8, (int)h, r->comm, st);
        if (e != ncclSuccess) { (void)ncclGroupEnd(); return fail(r, "ncclSend", e); }
        e = ncclRecv(static_cast<uint8_t*>(recv) + (size_t)h * block, block, ncclUint8, (int)h, r->comm, st);
        if (e != ncclSuccess) { (void)ncclGroupEnd(); return fail(r, "ncclRecv", e); }
    }
    e = ncclGroupEnd();
    return e == ncclSuccess ? 0 : fail(r, "ncclGroupEnd", e);
}
int ag_on(void* user, const void* send, void* recv, size_t bytes, void* stream) {
    RcclComm* r = static_cast<RcclComm*>(user);
    const ncclResult_t e = ncclAllGather(send, recv, bytes, ncclUint8, r->comm, static_cast<hipStream_t>(stream));
    return e == ncclSuccess ? 0 : fail(r, "ncclAllGather", e);
}
// host-synchronous forms: same collectives on the context stream, then wait
int a2a_sync(void* user, const void* send, void* recv, size_t block) {
    RcclComm* r = static_cast<RcclComm*>(user);
    if (a2a_on(user, send, recv, block, r->ctx->stream)) return 1;
    return hipStreamSynchronize(r->ctx->stream) == hipSuccess ? 0 : 1;
}
int ag_sync(void* user, const void* send, void* recv, size_t bytes) {
    RcclComm* r = static_cast<RcclComm*>(user);
    if (ag_on(user, send, recv, bytes, r->ctx->stream)) return 1;
    return hipStreamSynchronize(r->ctx->stream) == hipSuccess ? 0 : 1;
}

}  // namespace

extern "C" {

int lig_rccl_unique_id(uint8_t out[LIG_RCCL_ID_BYTES]) {
    static_assert(LIG_RCCL_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!out) return LIG_E_ARG;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return LIG_E_HIP;
    std::memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return LIG_OK;
}

int lig_rccl_comm_create(lig_ctx* c, const uint8_t id[LIG_RCCL_ID_BYTES], uint32_t rank, uint32_t world, lig_comm* out) {
    CHECK_CTX(c);
    if (!id || !out || !world || rank >= world) return LIG_E_ARG;
    std::memset(out, 0, sizeof *out);
    RcclComm* r = new RcclComm();
    r->ctx = c; r->rank = rank; r->world = world;
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
    const ncclResult_t e = ncclCommInitRank(&r->comm, (int)world, uid, (int)rank);
    if (e != ncclSuccess) { c->err = std::string("ncclCommInitRank: ") + ncclGetErrorString(e); delete r; return LIG_E_HIP; }
    out->user = r;
    out->all_to_all = a2a_sync;
    out->all_gather = ag_sync;
    out->all_to_all_on = a2a_on;
    out->all_gather_on = ag_on;
    return LIG_OK;
}

void lig_rccl_comm_destroy(lig_comm* comm) {
    if (!comm || !comm->user) return;
    RcclComm* r = static_cast<RcclComm*>(comm->user);
    if (r->ctx) { (void)hipSetDevice(r->ctx->device); (void)hipStreamSynchronize(r->ctx->stream); (void)hipStreamSynchronize(r->ctx->stream2); (void)hipStreamSynchronize(r->ctx->stream3); }
    if (r->comm) (void)ncclCommDestroy(r->comm);
    delete r;
    std::memset(comm, 0, sizeof *comm);
}

}  // extern "C"
