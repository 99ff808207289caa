// comm_ipc.hip -- stream-ordered collectives between PROCESSES that can map each other's device memory (lig_ipc_*).
//
// The second implementation of lig_comm's stream-ordered forms (the product's is comm_rccl.hip).  It exists so that the
// sharded prover's double-buffered exchange pipeline (shard.hip: ev_enc / ev_comm / ev_hash) runs with REAL peers on a box
// that has one GPU: W processes share the device, every rank pulls its blocks straight out of the peers' send buffers
// (hipIpcMemHandle: the peer's allocation is mapped into this process), and all ordering is done by the GPU's command
// processors on flags in POSIX shared memory (hipStreamWriteValue32 / hipStreamWaitValue32 on a hipHostRegister'ed page):
// no host thread ever waits for GPU work.  Between GPUs of one node the same code moves the data over xGMI peer access
// (hipIpcMemLazyEnablePeerAccess); RCCL remains the product path there.
//
//   all_to_all_on(send, recv, block, st), call number c (the same on every rank: collectives are called in program order):
//     host   : publish {IPC handle of send's allocation, offset} as publication c; read the peers' publication c
//     stream : ready[me] <- c                                   (everything queued on `st` before: my send data is complete)
//              for each peer h: wait ready[h] >= c; recv[h] <- peer_h.send[me]      (device-to-device copy)
//              pulled[me] <- c; for each peer h: wait pulled[h] >= c               (my send buffer is free again: like a
//                                                                                  completed ncclSend)
// The host side only waits for the peers' HOSTS to reach the same call (a few microseconds of skew), never for a GPU.
// Verified on the target box by tools/ipc_probe.hip (memory handles: an interior pointer exports its allocation's BASE,
// hence the explicit offset; stream memory operations on registered shared memory; profiles/r03_ipc_probe.txt).
//
// What tools/soak_sharded.py taught this file (profiles/r03_soak_sharded_ipc.md):
//   * exports are cached per epoch -- every hipIpcGetMemHandle exports the allocation anew and fails after a few dozen;
//   * forget() is a host collective -- a peer must close its mapping BEFORE the owner frees the buffer;
//   * a send out of an allocation smaller than 2 MiB goes through a window exported once -- such allocations are fragments
//     of shared blocks and the importer is handed the block's base.
// Build: compiled in with -DLIG_WITH_IPC_COMM (the Makefile's default: the GPU test-suite needs it); `make RELEASE=1` leaves it
// out -- lig_ipc_comm_create then returns LIG_E_STATE and the library carries no shared-memory / IPC-handle code at all.
// Failure model (round 5): a stream wait has no timeout, so every communicator runs a host WATCHDOG thread.  It declares the
// communicator dead when (a) a peer has declared it dead (abort word in the segment: any rank whose collective fails on the
// host writes it), (b) a peer's process is gone (pids are in the segment; /proc/<pid>/stat: missing or a zombie), or (c) a
// collective has been outstanding without any flag of any rank changing for LIG_IPC_STALL_S seconds (default 60).  A dead
// communicator is POISONED: the watchdog keeps writing 0x7fffffff into every ready / pulled word, so every wait already in a
// queue of this rank is released and the streams drain; lig_comm.failed() then reports it (the data of collectives since is
// garbage) and lig_shard_* returns LIG_E_STATE with the reason instead of hanging.  Every later collective fails on entry.
#include <cstring>

#include "ctx_internal.hpp"

#ifndef LIG_WITH_IPC_COMM
extern "C" {
int lig_ipc_comm_create(lig_ctx* c, const char*, uint32_t, uint32_t, lig_comm* out) {
    CHECK_CTX(c);
    if (out) std::memset(out, 0, sizeof *out);
    FAIL(c, LIG_E_STATE, "ipc comm: this build of liblig_hip.so has no process-to-process test communicator (make without RELEASE=1)");
}
void lig_ipc_comm_destroy(lig_comm*) {}
}
#else
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <signal.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "ctx_internal.hpp"

namespace {

constexpr uint32_t MAXW = 16, RING = 4, MAGIC = 0x4c494743u;      // "LIGC"
// flags: one 64-byte line per (rank, call mod SLOTS).  Consecutive collectives may be enqueued on different streams of a
// rank (the exchange on the copy stream, the gathers on the main stream): with a slot per call a late "ready <- c" can
// never overwrite "ready <- c+1"
constexpr uint32_t SLOTS = 8, FLAG_STRIDE = 16, FLAGS_PER_KIND = MAXW * SLOTS * FLAG_STRIDE;      // u32 units

struct Pub {
    std::atomic<uint64_t> seq;
    hipIpcMemHandle_t handle;
    uint64_t offset;
    uint64_t epoch;               // bumped by the exporter's forget(): handles published earlier may name freed allocations
    uint32_t via_window;          // 1: the data is staged in the exporter's window (slot = call mod SLOTS), `handle` is unused
};
struct Shm {
    std::atomic<uint32_t> magic, arrived, departed;
    uint32_t world;
    std::atomic<uint32_t> abort_by;       // rank + 1 of the first rank that declared the communicator dead (0: alive)
    std::atomic<int32_t> pid[MAXW];       // process of every rank (0: not arrived yet / left in good order)
    std::atomic<uint64_t> pidns[MAXW];    // inode of that process's PID namespace (0: unknown): a pid means something only inside its own namespace
    std::atomic<uint64_t> forgot[MAXW];   // epoch up to which rank h has closed every mapping of its peers' buffers
    hipIpcMemHandle_t window[MAXW];       // every rank's staging window (exported once, at creation)
    Pub pub[MAXW][RING];
    alignas(4096) uint32_t ready[FLAGS_PER_KIND];
    uint32_t pulled[FLAGS_PER_KIND];
};
constexpr size_t FLAG_BYTES = 2 * FLAGS_PER_KIND * sizeof(uint32_t);
constexpr size_t SLOT_BYTES = (size_t)2 << 20, DIRECT_MIN_ALLOC = (size_t)2 << 20;
static_assert(FLAG_BYTES % 4096 == 0, "whole pages are registered");
static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");

struct Mapping { hipIpcMemHandle_t h; uint64_t epoch; uint8_t* base; };
struct Export { void* base; hipIpcMemHandle_t h; };

struct IpcComm {
    lig_ctx* ctx = nullptr;
    uint32_t rank = 0, world = 1;
    std::string name;
    Shm* shm = nullptr;
    uint32_t* ready_dev = nullptr;      // device view of shm->ready (pulled follows at + FLAGS_PER_KIND)
    uint64_t calls = 0, epoch = 0;
    std::vector<Mapping> maps[MAXW];
    // handles of my own allocations, valid for the current epoch: every hipIpcGetMemHandle exports the allocation anew (a
    // dmabuf descriptor each time -- after a few dozen collectives the call failed with "invalid argument" on the target box)
    std::vector<Export> exports;
    // Staging window for sends out of SMALL allocations.  The runtime carves device allocations below 2 MiB out of shared
    // blocks; the handle of such a fragment names its whole block and the importer is handed the block's base -- it would
    // read a neighbour's bytes (seen on the target box as wrong opened columns / wrong roots once addresses were recycled,
    // and as hipIpcGetMemHandle failures).  So a send whose allocation is smaller than 2 MiB is copied into slot (call mod
    // SLOTS) of a 16 MiB window that is exported once, and the peers pull from there.
    uint8_t* window = nullptr;
    const uint8_t* peer_window[MAXW] = {nullptr};
    bool registered = false;
    // watchdog (see the failure model in the header of this file)
    std::thread wd;
    std::atomic<int> wd_stop{0}, dead{0};
    std::atomic<uint64_t> calls_pub{0};   // = calls, for the watchdog
    std::mutex why_mu;
    std::string why;                      // first reason the communicator was declared dead
    int stall_s = 60;
};

using clk = std::chrono::steady_clock;
#define HOST_TIMEOUT_S (lig::knobs().ipc_host_s)      // LIG_IPC_HOST_S (default 120): how long a rank waits ON THE HOST for a peer to reach the same collective

constexpr uint32_t POISON = 0x7fffffffu;      // >= every call number, whichever signedness the comparison uses

// declare the communicator dead (idempotent): the watchdog of every rank sees the abort word and poisons its own view
void declare_dead(IpcComm* r, const std::string& msg) {
    {
        std::lock_guard<std::mutex> lk(r->why_mu);
        if (r->why.empty()) r->why = msg;
    }
    r->dead.store(1, std::memory_order_release);
    if (r->shm) {
        uint32_t none = 0;
        (void)r->shm->abort_by.compare_exchange_strong(none, r->rank + 1);
    }
}
std::string dead_reason(IpcComm* r) {
    std::lock_guard<std::mutex> lk(r->why_mu);
    return r->why;
}
// a host-side failure inside a collective: the peers must not wait for this rank
int fail(IpcComm* r, const std::string& msg) {
    if (r && r->ctx) r->ctx->err = "ipc comm: " + msg;
    if (r && r->shm) declare_dead(r, msg);
    return 1;
}
void poison(Shm* sh) {
    for (uint32_t i = 0; i < FLAGS_PER_KIND; i += FLAG_STRIDE) {
        __atomic_store_n(&sh->ready[i], POISON, __ATOMIC_RELEASE);
        __atomic_store_n(&sh->pulled[i], POISON, __ATOMIC_RELEASE);
    }
}
// is the process of a peer still running?  (kill(pid, 0) also succeeds for a zombie nobody has reaped yet: ask /proc for the state)
uint64_t my_pid_namespace() {
    struct stat st;
    return stat("/proc/self/ns/pid", &st) == 0 ? (uint64_t)st.st_ino : 0;
}
// (a peer in ANOTHER PID namespace -- one container per rank sharing /dev/shm -- cannot be probed from here: unknown is not dead, the
// stall rule below is what catches it)
bool process_alive(int32_t pid, uint64_t pidns) {
    if (pid <= 0) return true;                   // not arrived yet, or left in good order
    static const uint64_t mine = my_pid_namespace();
    if (!pidns || !mine || pidns != mine) return true;
    char path[64], buf[256];
    std::snprintf(path, sizeof path, "/proc/%d/stat", (int)pid);
    FILE* f = std::fopen(path, "r");
    if (!f) return kill(pid, 0) == 0 || errno == EPERM;      // no /proc: fall back on the signal probe
    const size_t got = std::fread(buf, 1, sizeof buf - 1, f);
    std::fclose(f);
    buf[got] = 0;
    const char* p = std::strrchr(buf, ')');      // "pid (comm) S ..."
    return !(p && p[1] == ' ' && (p[2] == 'Z' || p[2] == 'X' || p[2] == 'x'));
}

void watchdog(IpcComm* r) {
    Shm* sh = r->shm;
    const uint32_t W = r->world;
    uint64_t last_sum = 0;
    auto last_change = clk::now();
    constexpr uint32_t RS = SLOTS * FLAG_STRIDE;
    while (!r->wd_stop.load(std::memory_order_acquire)) {
        if (r->dead.load(std::memory_order_acquire)) { poison(sh); usleep(2000); continue; }     // late "ready <- c" writes of queued work must not re-arm a wait
        std::string why;
        if (const uint32_t by = sh->abort_by.load(std::memory_order_acquire)) why = "rank " + std::to_string(by - 1) + " declared the communicator dead";
        for (uint32_t h = 0; h < W && why.empty(); h++)
            if (h != r->rank && !process_alive(sh->pid[h].load(std::memory_order_acquire), sh->pidns[h].load(std::memory_order_acquire)))
                why = "the process of rank " + std::to_string(h) + " (pid " + std::to_string(sh->pid[h].load()) + ") is gone";
        if (why.empty()) {
            const uint64_t c = r->calls_pub.load(std::memory_order_acquire);
            bool complete = true, all_published = true;
            uint64_t sum = c;
            for (uint32_t h = 0; h < W; h++) {
                // host skew (a rank still forming its rows before it enters collective c) is bounded by host_wait's HOST_TIMEOUT_S in the
                // ranks that wait for it: the stall clock runs only once every rank has published c
                if (c && sh->pub[h][c % RING].seq.load(std::memory_order_acquire) < c) all_published = false;
                for (uint32_t sl = 0; sl < SLOTS; sl++)
                    sum = sum * 1315423911u + __atomic_load_n(&sh->ready[h * RS + sl * FLAG_STRIDE], __ATOMIC_ACQUIRE) + ((uint64_t)__atomic_load_n(&sh->pulled[h * RS + sl * FLAG_STRIDE], __ATOMIC_ACQUIRE) << 32);
                if (c && __atomic_load_n(&sh->pulled[h * RS + (c % SLOTS) * FLAG_STRIDE], __ATOMIC_ACQUIRE) < (uint32_t)c) complete = false;
            }
            const auto now = clk::now();
            // a rank that has LEFT (its streams were drained first) will never write the flags an incomplete collective still waits for
            if (c && !complete && sh->departed.load(std::memory_order_acquire)) why = "a rank left the communicator while collective " + std::to_string(c) + " was outstanding";
            else if (!c || complete || !all_published || sum != last_sum) { last_sum = sum; last_change = now; }
            else if (now - last_change > std::chrono::seconds(r->stall_s)) {
                why = "collective " + std::to_string(c) + " outstanding and no flag of any rank changed for " + std::to_string(r->stall_s) + " s; flags [ready/pulled per rank, slot " + std::to_string(c % SLOTS) + "]:";
                for (uint32_t h = 0; h < W; h++)
                    why += " " + std::to_string(__atomic_load_n(&sh->ready[h * RS + (c % SLOTS) * FLAG_STRIDE], __ATOMIC_ACQUIRE)) + "/" +
                           std::to_string(__atomic_load_n(&sh->pulled[h * RS + (c % SLOTS) * FLAG_STRIDE], __ATOMIC_ACQUIRE));
            }
        }
        if (!why.empty()) {
            declare_dead(r, why);
            // diagnostics on stderr: why this rank's queues stand still (the call in progress registers a reporter with the context)
            const std::string st = r->ctx ? lig_internal_debug_state(r->ctx) : std::string();
            std::fprintf(stderr, "[lig ipc comm] rank %u: %s\n%s%s", r->rank, dead_reason(r).c_str(), st.c_str(), st.empty() ? "" : "\n");
            std::fflush(stderr);
            continue;
        }
        usleep(10000);
    }
}

// host-side wait for the peers' HOSTS; gives up at once when the communicator is dead
template <class Pred>
bool host_wait(IpcComm* r, Pred done, int timeout_s = -1) {
    if (timeout_s < 0) timeout_s = HOST_TIMEOUT_S;
    const auto t0 = clk::now();
    for (unsigned spins = 0; !done(); spins++) {
        if (spins > 2000) usleep(50);
        if ((spins & 63) == 63) {
            if (r && r->dead.load(std::memory_order_acquire)) return false;
            if (clk::now() - t0 > std::chrono::seconds(timeout_s)) return false;
        }
    }
    return true;
}

// peer h's pointer for publication `c` (mapped on first sight of the handle)
int peer_pointer(IpcComm* r, uint32_t h, uint64_t c, const uint8_t** out) {
    Pub& p = r->shm->pub[h][c % RING];
    if (!host_wait(r, [&] { return p.seq.load(std::memory_order_acquire) == c; }))
        return fail(r, r->dead.load() ? dead_reason(r) : "peer " + std::to_string(h) + " never reached collective " + std::to_string(c));
    if (p.via_window) { *out = r->peer_window[h] + p.offset; return 0; }
    const hipIpcMemHandle_t hd = p.handle;
    const uint64_t off = p.offset, ep = p.epoch;
    auto& maps = r->maps[h];
    // mappings of an older epoch name allocations the peer has freed since (it drained its streams first, and its collectives
    // only complete once every rank has pulled: nothing of ours still reads them) -- the same handle bytes may come back for a
    // new allocation, so they must go
    for (size_t i = 0; i < maps.size();)
        if (maps[i].epoch != ep) { (void)hipIpcCloseMemHandle(maps[i].base); maps.erase(maps.begin() + i); } else i++;
    for (const Mapping& m : maps)
        if (!std::memcmp(&m.h, &hd, sizeof hd)) { *out = m.base + off; return 0; }
    void* base = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&base, hd, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return fail(r, std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e));
    maps.push_back({hd, ep, static_cast<uint8_t*>(base)});
    *out = static_cast<uint8_t*>(base) + off;
    return 0;
}

// common part: publish my send buffer (or that it goes through the window), return the call number
int publish(IpcComm* r, const void* send, size_t total_bytes, uint64_t* call, bool* via_window) {
    const uint64_t c = ++r->calls;
    r->calls_pub.store(c, std::memory_order_release);
    Pub& p = r->shm->pub[r->rank][c % RING];
    void* base = nullptr;
    size_t size = 0;
    hipError_t e = hipMemGetAddressRange((hipDeviceptr_t*)&base, &size, (hipDeviceptr_t)send);
    if (e != hipSuccess) return fail(r, std::string("hipMemGetAddressRange: ") + hipGetErrorString(e));
    *via_window = size < DIRECT_MIN_ALLOC;
    *call = c;
    if (*via_window) {
        if (total_bytes > SLOT_BYTES) return fail(r, "small allocation with a send larger than a window slot");
        p.via_window = 1; p.offset = (c % SLOTS) * SLOT_BYTES; p.epoch = r->epoch;
        p.seq.store(c, std::memory_order_release);
        return 0;
    }
    p.via_window = 0;
    const Export* known = nullptr;
    for (const Export& x : r->exports) if (x.base == base) { known = &x; break; }
    if (!known) {
        Export x{base, {}};
        e = hipIpcGetMemHandle(&x.h, base);
        if (e != hipSuccess) return fail(r, std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e));
        r->exports.push_back(x);
        known = &r->exports.back();
    }
    p.handle = known->h;
    p.offset = (uint64_t)(static_cast<const uint8_t*>(send) - static_cast<const uint8_t*>(base));
    p.epoch = r->epoch;
    p.seq.store(c, std::memory_order_release);
    return 0;
}

#define IPC_TRY(r, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return fail((r), std::string(#call) + ": " + hipGetErrorString(e__)); } while (0)

// what = 0: all-to-all (recv block h <- peer h's send block `me`), 1: all-gather (recv block h <- peer h's send)
int collective_on(IpcComm* r, int what, const void* send, void* recv, size_t bytes, hipStream_t st) {
    if (!r->shm || !r->ctx) return 1;
    if (!r->dead.load(std::memory_order_acquire))         // (the shared abort word first: this rank's watchdog may not have seen it yet)
        if (const uint32_t by = r->shm->abort_by.load(std::memory_order_acquire)) declare_dead(r, "rank " + std::to_string(by - 1) + " declared the communicator dead");
    if (r->dead.load(std::memory_order_acquire)) { r->ctx->err = "ipc comm: " + dead_reason(r); return 1; }
    if (hipSetDevice(r->ctx->device) != hipSuccess) return fail(r, "hipSetDevice");
    const uint32_t W = r->world, me = r->rank;
    uint64_t c = 0;
    bool via_window = false;
    const size_t total = what == 0 ? (size_t)W * bytes : bytes;
    if (publish(r, send, total, &c, &via_window)) return 1;
    const uint8_t* src[MAXW];
    for (uint32_t i = 1; i < W; i++) {
        const uint32_t h = (me + i) % W;
        if (peer_pointer(r, h, c, &src[h])) return 1;
    }
    src[me] = static_cast<const uint8_t*>(send);
    const uint32_t slot = (uint32_t)(c % SLOTS);
    uint32_t* ready = r->ready_dev + slot * FLAG_STRIDE;                 // + rank * SLOTS * FLAG_STRIDE
    uint32_t* pulled = ready + FLAGS_PER_KIND;
    constexpr uint32_t RS = SLOTS * FLAG_STRIDE;
    const uint32_t v = (uint32_t)c;
    if (via_window) {
        // the slot was last used by call c - SLOTS, possibly on another stream of mine: every peer must have pulled that one
        if (c > SLOTS)
            for (uint32_t i = 1; i < W; i++) IPC_TRY(r, hipStreamWaitValue32(st, pulled + ((me + i) % W) * RS, (uint32_t)(c - SLOTS), hipStreamWaitValueGte, 0xffffffffu));
        if (total) IPC_TRY(r, hipMemcpyAsync(r->window + slot * SLOT_BYTES, send, total, hipMemcpyDeviceToDevice, st));
    }
    // slot reuse across my own streams: call c - SLOTS (same slot, possibly another stream) must have written its `pulled` --
    // which follows its `ready` -- before this call writes `ready <- c`, or a late "ready <- c - SLOTS" would overwrite it
    if (c > SLOTS) IPC_TRY(r, hipStreamWaitValue32(st, pulled + me * RS, (uint32_t)(c - SLOTS), hipStreamWaitValueGte, 0xffffffffu));
    // (LIG_FAULT_COMM=4, tests: rank 1 never raises its `ready` flag -- every rank has published the collective on the host, the peers' queued
    // waits never complete: the GPU-side stall the watchdog's LIG_IPC_STALL_S rule exists for)
    if (!(lig::knobs().fault_comm == 4 && me == 1)) IPC_TRY(r, hipStreamWriteValue32(st, ready + me * RS, v, 0));
    for (uint32_t i = 0; i < W; i++) {
        const uint32_t h = (me + i) % W;                    // own block first, then the peers in a rotated order (no hot spot)
        if (h != me) IPC_TRY(r, hipStreamWaitValue32(st, ready + h * RS, v, hipStreamWaitValueGte, 0xffffffffu));
        const uint8_t* from = src[h] + (what == 0 ? (size_t)me * bytes : 0);
        if (bytes) IPC_TRY(r, hipMemcpyAsync(static_cast<uint8_t*>(recv) + (size_t)h * bytes, from, bytes, hipMemcpyDeviceToDevice, st));
    }
    IPC_TRY(r, hipStreamWriteValue32(st, pulled + me * RS, v, 0));
    for (uint32_t i = 1; i < W; i++) {
        const uint32_t h = (me + i) % W;
        IPC_TRY(r, hipStreamWaitValue32(st, pulled + h * RS, v, hipStreamWaitValueGte, 0xffffffffu));
    }
    return 0;
}

int a2a_on(void* user, const void* send, void* recv, size_t block, void* stream) {
    IpcComm* r = static_cast<IpcComm*>(user);
    if (lig_internal_comm_fault(r->ctx, true)) return 1;
    return collective_on(r, 0, send, recv, block, static_cast<hipStream_t>(stream));
}
int ag_on(void* user, const void* send, void* recv, size_t bytes, void* stream) { return collective_on(static_cast<IpcComm*>(user), 1, send, recv, bytes, static_cast<hipStream_t>(stream)); }
// lig_comm.failed: non-zero once the communicator is dead -- the caller has drained its streams and asks whether what the
// collectives delivered is data or the leftovers of a poisoned wait
int comm_failed(void* user) {
    IpcComm* r = static_cast<IpcComm*>(user);
    // A PEER's watchdog may have declared the communicator dead and poisoned the shared flags -- which releases this rank's queued waits too --
    // a moment before this rank's own watchdog notices the abort word: streams that drained over poisoned waits must not be taken for a
    // completed collective (round 6: seen once in three runs of the GPU suite with LIG_FAULT_COMM=4).  The abort word is the shared truth.
    if (!r->dead.load(std::memory_order_acquire) && r->shm) {
        if (const uint32_t by = r->shm->abort_by.load(std::memory_order_acquire)) declare_dead(r, "rank " + std::to_string(by - 1) + " declared the communicator dead");
    }
    if (!r->dead.load(std::memory_order_acquire)) return 0;
    if (r->ctx) r->ctx->err = "ipc comm: " + dead_reason(r);
    return 1;
}
// the caller is about to free device buffers it has used as send buffers
// Collective on the host (every rank calls it at the same point of the program, lig_shard_destroy): the caller's streams are
// drained, so its own pulls are finished and -- a collective only completes once every rank has pulled -- so are the peers'
// pulls from its buffers.  Each rank closes its mappings of the peers' buffers, then waits until every peer has done the
// same: an allocation must not be freed (and its address recycled and exported again) while a peer still has it mapped --
// on the target box the export of the new allocation then failed with "invalid argument".
void forget(void* user) {
    IpcComm* r = static_cast<IpcComm*>(user);
    if (!r->shm) return;
    for (uint32_t h = 0; h < MAXW; h++) {
        for (const Mapping& m : r->maps[h]) (void)hipIpcCloseMemHandle(m.base);
        r->maps[h].clear();
    }
    r->exports.clear();
    const uint64_t ep = ++r->epoch;
    r->shm->forgot[r->rank].store(ep, std::memory_order_release);
    for (uint32_t h = 0; h < r->world; h++)
        (void)host_wait(r, [&] { return r->shm->forgot[h].load(std::memory_order_acquire) >= ep; }, 60);
}
void comm_abort(void* user) { declare_dead(static_cast<IpcComm*>(user), "aborted by the caller (its wait for queued collectives timed out)"); }
int a2a_sync(void* user, const void* send, void* recv, size_t block) {
    IpcComm* r = static_cast<IpcComm*>(user);
    if (!r->ctx || lig_internal_comm_fault(r->ctx, false) || collective_on(r, 0, send, recv, block, r->ctx->stream)) return 1;
    return hipStreamSynchronize(r->ctx->stream) == hipSuccess ? 0 : 1;
}
int ag_sync(void* user, const void* send, void* recv, size_t bytes) {
    IpcComm* r = static_cast<IpcComm*>(user);
    if (!r->ctx || ag_on(user, send, recv, bytes, r->ctx->stream)) return 1;
    return hipStreamSynchronize(r->ctx->stream) == hipSuccess ? 0 : 1;
}

void finalize(IpcComm* r) {
    // a dead communicator: the watchdog keeps the flags poisoned while this rank's streams drain (below), and is stopped after
    if (r->ctx) {
        (void)hipSetDevice(r->ctx->device);
        (void)hipStreamSynchronize(r->ctx->stream); (void)hipStreamSynchronize(r->ctx->stream2); if (r->ctx->stream3) (void)hipStreamSynchronize(r->ctx->stream3);
    }
    for (uint32_t h = 0; h < MAXW; h++) {
        for (const Mapping& m : r->maps[h]) (void)hipIpcCloseMemHandle(m.base);
        r->maps[h].clear();
        if (r->peer_window[h] && h != r->rank) (void)hipIpcCloseMemHandle(const_cast<uint8_t*>(r->peer_window[h]));
        r->peer_window[h] = nullptr;
    }
    r->wd_stop.store(1, std::memory_order_release);
    if (r->wd.joinable()) r->wd.join();
    if (r->shm) {
        // nobody may unmap / unlink while a peer's stream still polls the flags: leave together (a dead communicator: nobody waits)
        r->shm->pid[r->rank].store(0, std::memory_order_release);       // left in good order: not a crash
        r->shm->departed.fetch_add(1);
        if (!r->dead.load()) (void)host_wait(r, [&] { return r->shm->departed.load() >= r->world; }, 15);
        if (r->registered) (void)hipHostUnregister(r->shm->ready);
        (void)munmap(r->shm, sizeof(Shm));
        if (r->rank == 0) (void)shm_unlink(r->name.c_str());
    }
    if (r->window) (void)hipFree(r->window);         // after the departure barrier: no peer maps it any more
    r->window = nullptr;
    r->shm = nullptr;
    r->ctx = nullptr;
}
void finalize_erased(void* p) { finalize(static_cast<IpcComm*>(p)); }

}  // namespace

extern "C" {

int lig_ipc_comm_create(lig_ctx* c, const char* shm_name, uint32_t rank, uint32_t world, lig_comm* out) {
    CHECK_CTX(c);
    if (!shm_name || shm_name[0] != '/' || !out || !world || world > MAXW || rank >= world) return LIG_E_ARG;
    std::memset(out, 0, sizeof *out);
    int can = 0;
    (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, c->device);
    if (!can) FAIL(c, LIG_E_STATE, "ipc comm: the device has no stream memory operations");
    IpcComm* r = new IpcComm();
    r->ctx = c; r->rank = rank; r->world = world; r->name = shm_name;
    auto bail = [&](const std::string& msg, int code) { c->err = "ipc comm: " + msg; if (r->shm) (void)munmap(r->shm, sizeof(Shm)); delete r; return code; };
    // Rank 0 builds the segment under a private name, initialises it and only then gives it the agreed name (rename replaces a
    // stale segment of that name atomically); a peer that still catches a stale one -- every rank of an earlier communicator has
    // left it (departed != 0) or it is already full (arrived >= world) -- lets go of it and opens the name again.
    if (rank == 0) {
        const std::string tmp = std::string(shm_name) + ".new" + std::to_string((long)getpid());
        (void)shm_unlink(tmp.c_str());
        int fd = shm_open(tmp.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(Shm)) != 0) { if (fd >= 0) close(fd); (void)shm_unlink(tmp.c_str()); return bail("shm_open(create) failed", LIG_E_STATE); }
        void* m = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m == MAP_FAILED) { (void)shm_unlink(tmp.c_str()); return bail("mmap failed", LIG_E_STATE); }
        r->shm = static_cast<Shm*>(m);                // a fresh segment is zero-filled: publication counters and flags start at 0
        r->shm->world = world;
        r->shm->magic.store(MAGIC, std::memory_order_release);
        const std::string from = "/dev/shm" + tmp, to = std::string("/dev/shm") + shm_name;
        if (rename(from.c_str(), to.c_str()) != 0) {   // (no /dev/shm view of POSIX shared memory here: fall back to unlink + link by name)
            (void)shm_unlink(shm_name);
            (void)shm_unlink(tmp.c_str());
            return bail("cannot publish the shared segment under its name", LIG_E_STATE);
        }
    } else {
        const auto t0 = clk::now();
        for (;;) {
            int fd = shm_open(shm_name, O_RDWR, 0600);
            struct stat sb;
            if (fd >= 0 && fstat(fd, &sb) == 0 && (size_t)sb.st_size >= sizeof(Shm)) {
                void* m = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                close(fd);
                if (m == MAP_FAILED) return bail("mmap failed", LIG_E_STATE);
                Shm* sh = static_cast<Shm*>(m);
                if (sh->magic.load(std::memory_order_acquire) == MAGIC && sh->departed.load() == 0 && sh->arrived.load() < sh->world) {
                    if (sh->world != world) { (void)munmap(m, sizeof(Shm)); return bail("segment made for another world size", LIG_E_STATE); }
                    r->shm = sh;
                    break;
                }
                (void)munmap(m, sizeof(Shm));         // stale: rank 0 has not replaced it yet
            } else if (fd >= 0) close(fd);
            if (clk::now() - t0 > std::chrono::seconds(HOST_TIMEOUT_S)) return bail("the shared segment never appeared", LIG_E_STATE);
            usleep(1000);
        }
    }
    if (hipHostRegister(r->shm->ready, FLAG_BYTES, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess ||
        hipHostGetDevicePointer((void**)&r->ready_dev, r->shm->ready, 0) != hipSuccess)
        return bail("hipHostRegister of the flag page failed", LIG_E_HIP);
    r->registered = true;
    if (hipMalloc((void**)&r->window, SLOTS * SLOT_BYTES) != hipSuccess || hipIpcGetMemHandle(&r->shm->window[rank], r->window) != hipSuccess) {
        (void)hipHostUnregister(r->shm->ready); if (r->window) (void)hipFree(r->window);
        return bail("window allocation / export failed", LIG_E_HIP);
    }
    r->shm->pidns[rank].store(my_pid_namespace(), std::memory_order_release);
    r->shm->pid[rank].store((int32_t)getpid(), std::memory_order_release);
    r->shm->arrived.fetch_add(1);
    if (!host_wait(nullptr, [&] { return r->shm->arrived.load() >= world; })) { (void)hipHostUnregister(r->shm->ready); (void)hipFree(r->window); return bail("not all ranks arrived", LIG_E_STATE); }
    for (uint32_t h = 0; h < world; h++) {
        if (h == rank) { r->peer_window[h] = r->window; continue; }
        void* pw = nullptr;
        if (hipIpcOpenMemHandle(&pw, r->shm->window[h], hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            (void)hipHostUnregister(r->shm->ready); (void)hipFree(r->window);
            return bail("cannot map the window of rank " + std::to_string(h), LIG_E_HIP);
        }
        r->peer_window[h] = static_cast<const uint8_t*>(pw);
    }
    r->stall_s = lig::knobs().ipc_stall_s;
    r->wd = std::thread([r] { watchdog(r); });
    c->comms.push_back({r, finalize_erased});
    out->user = r;
    out->all_to_all = a2a_sync;
    out->all_gather = ag_sync;
    out->all_to_all_on = a2a_on;
    out->all_gather_on = ag_on;
    out->forget = forget;
    out->failed = comm_failed;
    out->abort = comm_abort;
    return LIG_OK;
}

void lig_ipc_comm_destroy(lig_comm* comm) {
    if (!comm || !comm->user || comm->all_to_all_on != a2a_on) return;
    IpcComm* r = static_cast<IpcComm*>(comm->user);
    if (r->ctx) lig_internal_comm_unregister(r->ctx, r);
    finalize(r);
    delete r;
    std::memset(comm, 0, sizeof *comm);
}

}  // extern "C"
#endif  // LIG_WITH_IPC_COMM
