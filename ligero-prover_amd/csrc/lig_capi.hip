// lig_capi.hip -- the C ABI of include/lig_hip.h: context, tables, buffer plumbing, launch sequences.
// Host-side counterpart of webgpu_context (include/wgpu.hpp, src/webgpu/engine.cpp) and device_context
// (src/webgpu/device_context.cpp): one HIP stream instead of a WebGPU queue, plain device pointers instead of
// WGPUBuffer/bind groups, twiddle tables generated with host_field.hpp instead of GMP.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include <unistd.h>
#include "../../include/lig_hip.h"
#include "host_field.hpp"
#include "kernels.hpp"
#include "fr29.hpp"
#include "ctx_internal.hpp"

namespace lig {
void aes_upload_tables();
void launch_rng_fill_rows_dense(hipStream_t s, const uint32_t* rk60_dev, uint64_t first, fr* out, size_t rows, uint32_t per_row, uint32_t k);
}

namespace H = lig::host;

static fr to_dev(const H::Fr& a) {
    fr r;
    std::memcpy(r.v, a.v, 32);
    return r;
}
static uint32_t ilog2u(uint32_t x) { uint32_t r = 0; while ((1u << r) < x) r++; return r; }

static int upload(lig_ctx* c, const std::vector<fr>& host, fr** dev) {
    void* p = nullptr;
    HIP_TRY(c, hipMalloc(&p, host.size() * sizeof(fr)));
    c->owned.push_back(p);
    HIP_TRY(c, hipMemcpy(p, host.data(), host.size() * sizeof(fr), hipMemcpyHostToDevice));
    *dev = (fr*)p;
    return LIG_OK;
}

// ntt_precompute_omegas (src/webgpu/engine.cpp:1382-1503): w^i * R, w^-i * R (i < N/2), N^-1 * R
static int make_plan(lig_ctx* c, lig::NttPlan& pl, uint32_t N, const H::Fr& root) {
    pl.N = N; pl.log2N = ilog2u(N);
    std::vector<fr> w(N / 2), wi(N / 2);
    const H::Fr rootm = H::to_mont(root), rinvm = H::to_mont(H::inv(root));
    H::Fr cur = H::R1, curi = H::R1;    // 1 in Montgomery form
    for (uint32_t i = 0; i < N / 2; i++) {
        w[i] = to_dev(cur); wi[i] = to_dev(curi);
        cur = H::montmul(cur, rootm); curi = H::montmul(curi, rinvm);
    }
    pl.ninv = to_dev(H::to_mont(H::inv(H::from_u64(N))));
    int rc;
    if ((rc = upload(c, w, &pl.w)) != LIG_OK) return rc;
    return upload(c, wi, &pl.winv);
}

// powers table in Montgomery form: out[i] = base^(f(i)) generated incrementally
static std::vector<H::Fr> powers_mont(const H::Fr& base, size_t count) {
    std::vector<H::Fr> out(count);
    const H::Fr bm = H::to_mont(base);
    H::Fr cur = H::R1;
    for (size_t i = 0; i < count; i++) { out[i] = cur; cur = H::montmul(cur, bm); }
    return out;
}

// ---- 29-bit-limb tables of the batched encode path (fr29.hpp): entry = x * 2^261 mod p as 9 limbs, 48-byte stride
static const H::Fr R261 = {{0x2fd4e1568fffff57ull, 0x75bba827a494b01aull, 0x5301fa84819caa80ull, 0x0dc83629563d4475ull}};   // 2^261 mod p
static lig::f29s to_f29s(const H::Fr& plain) {
    const H::Fr m = H::mul(plain, R261);
    lig::f29s o;
    std::memset(&o, 0, sizeof o);
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, w = bit >> 6, sh = bit & 63;
        uint64_t v = m.v[w] >> sh;
        if (sh > 35 && w < 3) v |= m.v[w + 1] << (64 - sh);
        o.v[i] = (uint32_t)(i < 8 ? (v & 0x1FFFFFFFull) : v);
    }
    return o;
}
// windowed table entry (fr29.hpp: f29_mulw): W_g = x * 2^(116 + 87 g) mod p for g = 0, 1, 2, nine 29-bit limbs each
static lig::f29w to_f29w(const H::Fr& plain) {
    static const std::vector<H::Fr> shift = [] {
        std::vector<H::Fr> s(3);
        H::Fr c = H::from_u64(1);
        for (int i = 0; i < 116; i++) c = H::add(c, c);
        for (int g = 0; g < 3; g++) { s[g] = c; for (int i = 0; i < 87; i++) c = H::add(c, c); }
        return s;
    }();
    lig::f29w o;
    std::memset(&o, 0, sizeof o);
    for (int g = 0; g < 3; g++) {
        const H::Fr m = H::mul(plain, shift[g]);
        for (int i = 0; i < 9; i++) {
            const int bit = 29 * i, w = bit >> 6, sh = bit & 63;
            uint64_t v = m.v[w] >> sh;
            if (sh > 35 && w < 3) v |= m.v[w + 1] << (64 - sh);
            o.v[9 * g + i] = (uint32_t)(i < 8 ? (v & 0x1FFFFFFFull) : v);
        }
    }
    return o;
}
// windowed table -> device planes (fr29.hpp: f29wt): plane i = words 4i .. 4i+3 of every entry
static int upload29w(lig_ctx* c, const std::vector<lig::f29w>& host, lig::f29wt* dev) {
    const size_t E = host.size();
    std::vector<uint32_t> planes(7 * E * 4);
    for (size_t e = 0; e < E; e++)
        for (int i = 0; i < 7; i++) std::memcpy(&planes[((size_t)i * E + e) * 4], &host[e].v[4 * i], 16);
    void* p = nullptr;
    HIP_TRY(c, hipMalloc(&p, planes.size() * 4));
    c->owned.push_back(p);
    HIP_TRY(c, hipMemcpy(p, planes.data(), planes.size() * 4, hipMemcpyHostToDevice));
    dev->base = (const uint4*)p;
    dev->stride = (uint32_t)E;
    dev->idx = 0;
    return LIG_OK;
}
template <class T>
static int upload29(lig_ctx* c, const std::vector<T>& host, T** dev) {
    void* p = nullptr;
    HIP_TRY(c, hipMalloc(&p, host.size() * sizeof(T)));
    c->owned.push_back(p);
    HIP_TRY(c, hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *dev = (T*)p;
    return LIG_OK;
}
// plain powers base^0 .. base^(count-1)
static std::vector<H::Fr> powers_plain(const H::Fr& base, size_t count) {
    std::vector<H::Fr> out(count);
    const H::Fr bm = H::to_mont(base);
    H::Fr cur = H::R1;
    for (size_t i = 0; i < count; i++) { out[i] = H::from_mont(cur); cur = H::montmul(cur, bm); }
    return out;
}

static int make_encode_plan(lig_ctx* c, const H::Fr& wk, const H::Fr& w4k) {
    lig::EncodePlan& ep = c->ep;
    const uint32_t k = c->k, A = 8, B = k / 8;
    ep.k = k; ep.n = c->n; ep.log2k = ilog2u(k); ep.A = A; ep.B = B; ep.log2B = ilog2u(B);
    const H::Fr wk_inv = H::inv(wk);
    const H::Fr psi = H::pow_u64(w4k, 4);                       // order k
    int rc;
    // DIT stage twiddles of the size-B transforms: the stage of span M starts at entry M/2-1, rho^(j*B/M), j < M/2
    auto stage_table = [&](const H::Fr& rho) {
        std::vector<H::Fr> pw = powers_plain(rho, B / 2);
        std::vector<lig::f29w> t(B);                            // B-1 used
        for (uint32_t M = 2; M <= B; M <<= 1)
            for (uint32_t j = 0; j < M / 2; j++) t[(M / 2 - 1) + j] = to_f29w(pw[(size_t)j * (B / M)]);
        t[B - 1] = to_f29w(H::from_u64(1));
        return t;
    };
    if ((rc = upload29w(c, stage_table(H::pow_u64(wk_inv, A)), &ep.tw_b_inv)) != LIG_OK) return rc;
    if ((rc = upload29w(c, stage_table(H::pow_u64(psi, A)), &ep.tw_b)) != LIG_OK) return rc;
    // seams: seam_inv[j1][i2] = w_k^(-i2*j1), seam_fwd[i1][q2] = psi^(i1*q2)
    {
        std::vector<lig::f29w> si((size_t)A * B), sf((size_t)A * B);
        for (uint32_t j1 = 0; j1 < A; j1++) {
            std::vector<H::Fr> a = powers_plain(H::pow_u64(wk_inv, j1), B), b = powers_plain(H::pow_u64(psi, j1), B);
            for (uint32_t i = 0; i < B; i++) { si[(size_t)j1 * B + i] = to_f29w(a[i]); sf[(size_t)j1 * B + i] = to_f29w(b[i]); }
        }
        if ((rc = upload29w(c, si, &ep.seam_inv)) != LIG_OK) return rc;
        if ((rc = upload29w(c, sf, &ep.seam_fwd)) != LIG_OK) return rc;
    }
    // twist[r-1][j1][i2] = k^-1 * w_n^(r*(j1 + 8*i2)), r = 1..3.  The 1/k of the inverse transform rides on the twist (every
    // computed coset has one; coset 0 is copied from the message), so the inverse tile transforms keep unscaled coefficients.
    {
        std::vector<lig::f29w> tw((size_t)3 * k);
        const H::Fr kinv = H::inv(H::from_u64(k));
        for (uint32_t r = 1; r < 4; r++) {
            std::vector<H::Fr> pw = powers_plain(H::pow_u64(w4k, r), k);
            for (uint32_t j1 = 0; j1 < A; j1++)
                for (uint32_t t = 0; t < B / 4; t++)
                    for (uint32_t q = 0; q < 4; q++) {          // thread order inside a tile: entry q*B/4 + t <-> position brev(4t + q)
                        uint32_t i2 = 0;
                        for (uint32_t bit = 0, v = 4 * t + q; bit < ep.log2B; bit++, v >>= 1) i2 = (i2 << 1) | (v & 1);
                        tw[((size_t)(r - 1) * A + j1) * B + q * (B / 4) + t] = to_f29w(H::mul(kinv, pw[j1 + (size_t)A * i2]));
                    }
        }
        if ((rc = upload29w(c, tw, &ep.twist)) != LIG_OK) return rc;
    }
    // radix-8 constants: w8[i] = w^i, w = w_k^(-k/8) (inverse) / psi^(k/8) (forward); k^-1
    {
        std::vector<H::Fr> a = powers_plain(H::pow_u64(wk_inv, B), 8), b = powers_plain(H::pow_u64(psi, B), 8);
        std::vector<lig::f29w> da(8), db(8);
        std::vector<lig::f29s> ki(1);
        for (int i = 0; i < 8; i++) { da[i] = to_f29w(a[i]); db[i] = to_f29w(b[i]); }
        ki[0] = to_f29s(H::inv(H::from_u64(k)));
        if ((rc = upload29w(c, da, &ep.w8_inv)) != LIG_OK) return rc;
        if ((rc = upload29w(c, db, &ep.w8_fwd)) != LIG_OK) return rc;
        if ((rc = upload29(c, ki, &ep.kinv)) != LIG_OK) return rc;
    }
    return LIG_OK;
}

// ---- small uploads without the DMA engine (ctx_internal.hpp)
__global__ void k_copy_u32(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int lig_internal_upload_small(lig_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (!bytes) return LIG_OK;
    if (bytes & 3) FAIL(c, LIG_E_ARG, "upload_small: size not a multiple of 4");
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (need > c->stage_cap) {               // larger than the ring (or no ring): plain copy
        HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        return LIG_OK;
    }
    if (c->stage_pos + need > c->stage_cap) {
        // wrap: everything that still reads the ring must have finished (proofs are synchronous per context, so this is
        // a formality -- once per ~40 proofs)
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->stream2) HIP_TRY(c, hipStreamSynchronize(c->stream2));
        c->stage_pos = 0;
    }
    std::memcpy(c->stage_host + c->stage_pos, src, bytes);
    const size_t words = bytes / 4;
    const uint32_t blocks = (uint32_t)((words + 255) / 256 < 64 ? (words + 255) / 256 : 64);
    hipLaunchKernelGGL(k_copy_u32, dim3(blocks ? blocks : 1), dim3(256), 0, st, (uint32_t*)dst, (const uint32_t*)(c->stage_dev + c->stage_pos), words);
    c->stage_pos += need;
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}

// ---- device -> pinned host without the DMA engines (ctx_internal.hpp): the proof's downloads (accumulators, decoded
// accumulators, Merkle nodes, opened columns: 1 .. 13 MB each) as stores of a copy kernel into the mapped host buffer.  With a
// 550 MB witness upload of the NEXT trace in flight, hipMemcpyAsync D2H of the current proof was seen to finish only when the
// upload did (stage 2 4.9 -> 11.0 ms in three of five proofs, tools/time_rows2.py, profiles/r03_h2d_pipeline.md).
__global__ void __launch_bounds__(256) k_copy_u128(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// dst at any byte address (the opened columns land inside the protobuf envelope): aligned dword stores assembled from two
// source dwords, the up to three bytes on either edge stored one by one.  src 4-byte aligned, bytes % 4 == 0.
__global__ void __launch_bounds__(256) k_copy_to_unaligned(uint8_t* __restrict__ dst, const uint32_t* __restrict__ src, size_t words) {
    const uint32_t sh = (uint32_t)((uintptr_t)dst & 3u);                  // 1..3
    uint32_t* D = reinterpret_cast<uint32_t*>(dst - sh);                  // D[w] = dst bytes [4w - sh, 4w - sh + 4)
    const uint32_t lo = 8u * (4u - sh), hi = 8u * sh;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t w = gid + 1; w < words; w += (size_t)gridDim.x * blockDim.x) D[w] = (src[w - 1] >> lo) | (src[w] << hi);
    if (gid == 0) {
        const uint8_t* sb = reinterpret_cast<const uint8_t*>(src);
        for (uint32_t b = 0; b < 4u - sh; b++) dst[b] = sb[b];                                  // head: the upper part of D[0]
        for (uint32_t b = 0; b < sh; b++) dst[4 * words - sh + b] = sb[4 * words - sh + b];     // tail: the lower part of D[words]
    }
}
int lig_internal_download(lig_ctx* c, void* host_pinned, const void* src, size_t bytes, hipStream_t st) {
    if (!bytes) return LIG_OK;
    const bool by_kernel = lig::knobs().d2h_kernel;
    if (!by_kernel || (bytes & 3) || ((uintptr_t)src & 15)) {
        HIP_TRY(c, hipMemcpyAsync(host_pinned, src, bytes, hipMemcpyDeviceToHost, st));
        return LIG_OK;
    }
    if (!((uintptr_t)host_pinned & 15) && !(bytes & 15)) {
        const size_t n16 = bytes / 16;
        const uint32_t blocks = (uint32_t)((n16 + 255) / 256 < 128 ? (n16 + 255) / 256 : 128);
        hipLaunchKernelGGL(k_copy_u128, dim3(blocks), dim3(256), 0, st, (uint4*)host_pinned, (const uint4*)src, n16);
    } else if (!((uintptr_t)host_pinned & 3)) {
        const size_t words = bytes / 4;
        const uint32_t blocks = (uint32_t)((words + 255) / 256 < 256 ? (words + 255) / 256 : 256);
        hipLaunchKernelGGL(k_copy_u32, dim3(blocks), dim3(256), 0, st, (uint32_t*)host_pinned, (const uint32_t*)src, words);
    } else {
        const size_t words = bytes / 4;
        const uint32_t blocks = (uint32_t)((words + 255) / 256 < 256 ? (words + 255) / 256 : 256);     // (64 workgroups: 339 us for 13 MB, 256: 260 us)
        hipLaunchKernelGGL(k_copy_to_unaligned, dim3(blocks), dim3(256), 0, st, (uint8_t*)host_pinned, (const uint32_t*)src, words);
    }
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}

// ---- tables of the two-launch single-row transforms (ntt_tiled.hip)
static int make_tiled_plan(lig_ctx* c, lig::TiledPlan& tp, uint32_t N, const H::Fr& root, bool inverse) {
    const H::Fr w = inverse ? H::inv(root) : root;
    tp.N = N; tp.log2N = ilog2u(N); tp.log2A = tp.log2N / 2; tp.log2B = tp.log2N - tp.log2A;
    const uint32_t A = 1u << tp.log2A, B = 1u << tp.log2B;
    auto stage_table = [&](const H::Fr& rho, uint32_t S) {      // span M at entry M/2-1: rho^(j*S/M), j < M/2
        std::vector<H::Fr> pw = powers_plain(rho, S / 2);
        std::vector<lig::f29s> t(S);
        for (uint32_t M = 2; M <= S; M <<= 1)
            for (uint32_t j = 0; j < M / 2; j++) t[(M / 2 - 1) + j] = to_f29s(pw[(size_t)j * (S / M)]);
        t[S - 1] = to_f29s(H::from_u64(1));
        return t;
    };
    int rc;
    if ((rc = upload29(c, stage_table(H::pow_u64(w, B), A), &tp.tw_a)) != LIG_OK) return rc;
    if ((rc = upload29(c, stage_table(H::pow_u64(w, A), B), &tp.tw_b)) != LIG_OK) return rc;
    std::vector<lig::f29s> mid((size_t)N);
    const H::Fr scale = inverse ? H::inv(H::from_u64(N)) : H::from_u64(1);
    H::Fr base = H::from_u64(1);                                  // w^k1
    for (uint32_t k1 = 0; k1 < A; k1++) {
        H::Fr cur = scale;
        for (uint32_t n2 = 0; n2 < B; n2++) { mid[(size_t)k1 * B + n2] = to_f29s(cur); cur = H::mul(cur, base); }
        base = H::mul(base, w);
    }
    return upload29(c, mid, &tp.mid);
}

// the one place the library reads its environment (ctx_internal.hpp: lig::Knobs)
const lig::Knobs& lig::knobs() {
    static const lig::Knobs k = [] {
        lig::Knobs t;
        auto num = [](const char* name, long dflt) { const char* e = std::getenv(name); return e && *e ? std::atol(e) : dflt; };
        auto pos = [&](const char* name, long dflt) { const long v = num(name, 0); return v > 0 ? v : dflt; };
        t.encode_kmask = (int)num("LIG_ENCODE_KMASK", 15);
        { const long v = num("LIG_K13_BLOCK", 0); t.k13_block = (v == 64 || v == 128) ? (uint32_t)v : 256u; }
        t.k2_dyn_lds = (uint32_t)num("LIG_K2_DYN_LDS", 0);
        t.encode_chunk = (size_t)pos("LIG_ENCODE_CHUNK", 512);
        { const char* e = std::getenv("LIG_ENCODE_GENERIC"); t.encode_generic = e && e[0] == '1'; }
        t.sha_block = (uint32_t)pos("LIG_SHA_BLOCK", 256);
        { const long v = num("LIG_SHA_WS", 2); t.sha_ws = (v == 0 || v == 1 || v == 4) ? (int)v : 2; }
        t.sha_gate = (int)num("LIG_SHA_GATE", 1);
        { const long v = num("LIG_AES_BLOCKS", 0); t.aes_blocks = v >= 64 && v <= 4096 ? (uint32_t)v : 0u; }
        t.aes_layout = (int)num("LIG_AES_LAYOUT", 1);
        t.shared_side = num("LIG_SHARED_SIDE", 1) != 0;
        t.sha_gate_rows = (size_t)pos("LIG_SHA_GATE_ROWS", 2);
#ifdef LIG_EXPERIMENTS      // measured and rejected in round 4 (profiles/r04_sha_priority_ab.md, r04_sha_cumask_ab.md, r04_filler_proof_ab.md): only an
        // A/B build (`make EXPERIMENTS=1`) reads them; the product build ignores the variables
        t.sha_prio = (int)num("LIG_SHA_PRIO", 0);
        if (const char* v = std::getenv("LIG_STREAM_MAP")) t.stream_map = v;      // round 6: profiles/r06_stream_map_ab.md
        if (const char* v = std::getenv("LIG_STREAM_PAD")) t.stream_pad = v;
        t.sha_cumask = (int)num("LIG_SHA_CUMASK", 0);
        t.ctx_low_prio_every = (int)num("LIG_CTX_LOW_PRIO_EVERY", 0);
#endif
        t.s1_head = (size_t)num("LIG_S1_HEAD", 128); t.s1_tail = (size_t)num("LIG_S1_TAIL", 96); t.s2_head = (size_t)num("LIG_S2_HEAD", 192);
        t.fused_rlc = std::getenv("LIG_NO_FUSED_RLC") == nullptr;
        t.early_code = num("LIG_EARLY_CODE", 1) != 0;
        t.upload_mode = (int)num("LIG_UPLOAD_MODE", 2);
        t.rands_upload_mode = (int)num("LIG_RANDS_UPLOAD_MODE", 2);
        t.upload_prio = num("LIG_UPLOAD_PRIO", 1) != 0;
        t.shard_uploader = num("LIG_SHARD_UPLOADER", 0) != 0;
        t.upload_timeout_s = (int)pos("LIG_UPLOAD_TIMEOUT_S", 5);
        t.fault_upload = (int)num("LIG_FAULT_UPLOAD", 0);
        t.d2h_kernel = num("LIG_D2H_KERNEL", 1) != 0;
        t.spin_wait = num("LIG_SPIN_WAIT", 1) != 0;
        t.spin_wait_ms = (int)pos("LIG_SPIN_WAIT_MS", 50);
        t.shard_force_exchange = std::getenv("LIG_SHARD_FORCE_EXCHANGE") != nullptr;
        t.trace = std::getenv("LIG_TRACE") != nullptr;
        t.zres = num("LIG_ZRES", 0) != 0;
        t.fault_comm = (int)num("LIG_FAULT_COMM", 0);
        t.ipc_stall_s = (int)pos("LIG_IPC_STALL_S", 120);
        t.ipc_host_s = (int)pos("LIG_IPC_HOST_S", 120);
        t.comm_timeout_s = (int)pos("LIG_COMM_TIMEOUT_S", 300);
        { const char* e = std::getenv("LIG_RCCL_LIB"); if (e) t.rccl_lib = e; }
        return t;
    }();
    return k;
}

int lig_internal_comm_fault(lig_ctx* c, bool stream_ordered) {
    const int f = lig::knobs().fault_comm;
    if (!f) return 0;
    if (f == 3 && stream_ordered) for (;;) sleep(3600);
    if ((stream_ordered && (f == 1 || f == 2)) || (!stream_ordered && f == 2)) {
        if (c) c->err = std::string("injected fault (LIG_FAULT_COMM) in the ") + (stream_ordered ? "stream-ordered" : "host-synchronous") + " all-to-all";
        return 1;
    }
    return 0;
}

extern "C" {

const char* lig_version(void) { return "lig_hip 0.1 (gfx950)"; }
void lig_abi_sizes(uint32_t out[LIG_ABI_STRUCTS]) {
    const uint32_t v[LIG_ABI_STRUCTS] = {sizeof(lig_batch_op), sizeof(lig_synth_job), sizeof(lig_proof_info), sizeof(lig_verify_info), sizeof(lig_rows_job), sizeof(lig_comm)};
    if (out) std::memcpy(out, v, sizeof v);
}

int lig_ctx_create(lig_ctx** out, int device, uint32_t l, uint32_t k, uint32_t n) {
    if (!out) return LIG_E_ARG;
    *out = nullptr;
    if (k < 512 || (k & (k - 1)) || n != 4 * k || l > k || (uint64_t)n > (1ull << 28)) return LIG_E_ARG;
    lig_ctx* c = new lig_ctx();
    c->device = device; c->l = l; c->k = k; c->n = n;
    *out = c;    // returned even on failure so that lig_last_error works; caller destroys
    HIP_TRY(c, hipSetDevice(device));
    // experiment (profiles/r04_filler_proof_ab.md): every n-th context as a low-priority "filler" whose kernels only take what the others leave
    static std::atomic<uint32_t> n_created{0};
    const uint32_t ctx_index = n_created.fetch_add(1);
    const int every = lig::knobs().ctx_low_prio_every;
    const bool low = every > 0 && (ctx_index % (uint32_t)every) == (uint32_t)every - 1;
    int prio_lo = 0, prio_hi = 0;
    if (low) HIP_TRY(c, hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    // EXPERIMENT (LIG_STREAM_MAP, round 6; profiles/r06_stream_map_ab.md): which streams of the proofs in flight share a hardware queue decides
    // 15 % of the throughput (GPU_MAX_HW_QUEUES sweep) and is an accident of creation order.  The knob makes it explicit: six digits =
    // the PHYSICAL stream of [main, side, copy] of even contexts, then of odd contexts; physical streams live for the process and are
    // shared by every context that names them (the runtime must then give each its own queue: GPU_MAX_HW_QUEUES >= their number).
    static hipStream_t g_phys[64][10] = {};
    static std::mutex g_phys_mu;
    const std::string& smap = lig::knobs().stream_map;
    int role = 0;
    auto make_stream = [&](hipStream_t* st) -> hipError_t {
        const int my_role = role++;
        // "a" + groups of FOUR digits: [main, side, copy, hash] -- the stage-1 column hash on a physical stream of its own (c->stream_sha)
        const bool four = !smap.empty() && smap[0] == 'a';
        const size_t per = four ? 4 : 3, off0 = four ? 1 : 0;
        if (smap.size() >= off0 + 2 * per && (smap.size() - off0) % per == 0 && device >= 0 && device < 64 && !low) {       // contexts cycle through the groups
            const int idx = smap[off0 + (ctx_index % (uint32_t)((smap.size() - off0) / per)) * per + my_role] - '0';
            if (idx >= 0 && idx < 10) {
                std::lock_guard<std::mutex> lk(g_phys_mu);
                if (!g_phys[device][idx]) { const hipError_t e = hipStreamCreateWithFlags(&g_phys[device][idx], hipStreamNonBlocking); if (e != hipSuccess) return e; }
                *st = g_phys[device][idx];
                c->streams_shared = true;
                return hipSuccess;
            }
        }
        return low ? hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio_lo) : hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    };
    // EXPERIMENT (LIG_STREAM_PAD = "even,odd[,order]"): dummy streams created (and kept) before the streams of an even / odd context, and
    // the creation order of {main, side, copy} as a permutation string ("012" = main, side, copy): shifts which streams the runtime's
    // least-used-queue rule puts together when there are more streams than GPU_MAX_HW_QUEUES
    hipStream_t* slots3[3] = {&c->stream, &c->stream2, &c->stream3};
    int order3[3] = {0, 1, 2};
    {
        const std::string& pad = lig::knobs().stream_pad;
        if (!pad.empty()) {
            int a = 0, b = 0; char ord[8] = "012";
            std::sscanf(pad.c_str(), "%d,%d,%3s", &a, &b, ord);
            const int n_pad = (ctx_index & 1u) ? b : a;
            for (int i = 0; i < n_pad && i < 8; i++) { hipStream_t d; (void)hipStreamCreateWithFlags(&d, hipStreamNonBlocking); }
            for (int i = 0; i < 3; i++) if (ord[i] >= '0' && ord[i] <= '2') order3[i] = ord[i] - '0';
        }
    }
    const bool permuted = !(order3[0] == 0 && order3[1] == 1 && order3[2] == 2);
    if (permuted) {
        for (int i = 0; i < 3; i++) { role = order3[i]; HIP_TRY(c, make_stream(slots3[order3[i]])); }
    } else
    HIP_TRY(c, make_stream(&c->stream));
    H::Fr wk, w2k, w4k;
    H::omegas(k, wk, w2k, w4k);
    int rc;
    if ((rc = make_plan(c, c->plan[LIG_SIZE_K], k, wk)) != LIG_OK) return rc;
    if ((rc = make_plan(c, c->plan[LIG_SIZE_2K], 2 * k, w2k)) != LIG_OK) return rc;
    if ((rc = make_plan(c, c->plan[LIG_SIZE_N], n, w4k)) != LIG_OK) return rc;
    if ((rc = make_plan(c, c->plan_half, 2 * k, H::mul(w4k, w4k))) != LIG_OK) return rc;     // the subgroup <w_n^2>, order 2k
    if (lig::knobs().sha_prio) {              // experiment (profiles/r04_sha_priority_ab.md): the hash stream in the high-priority queue class
        int lo = 0, hi = 0;
        HIP_TRY(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIP_TRY(c, hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, hi));
    } else if (!permuted) {
        // The side stream (column hash of stage 1, samplers of stage 2) is ONE stream per device for the whole process: with several proofs in
        // flight NO two side kernels may run next to each other -- not two column hashes, not a hash next to the other proof's sampler.  Measured
        // (profiles/r06_stream_map_ab.md): side work of the proofs serialised 2.04 x 10^9 constraints/s, on streams of their own 1.75-1.92 x 10^9.
        // Rounds 2-5 had this by an accident of the runtime (7 streams on 4 hardware queues: the two side streams happened to land in one queue).
        static hipStream_t g_side[64] = {};
        static std::mutex g_side_mu;
        const bool own_side = !lig::knobs().shared_side || low || smap.size() >= 6 || device < 0 || device >= 64;
        if (own_side) HIP_TRY(c, make_stream(&c->stream2));
        else {
            std::lock_guard<std::mutex> lk(g_side_mu);
            if (!g_side[device]) HIP_TRY(c, hipStreamCreateWithFlags(&g_side[device], hipStreamNonBlocking));
            c->stream2 = g_side[device];
            c->side_shared = true;
        }
    }
    // (the copy stream is created on first use: lig_internal_copy_stream)
    if (!permuted && smap.size() >= 6) HIP_TRY(c, make_stream(&c->stream3));
    if (!smap.empty() && smap[0] == 'a') { role = 3; HIP_TRY(c, make_stream(&c->stream_sha)); }
    if (lig::knobs().sha_cumask) {
        // experiment (profiles/r04_sha_cumask_ab.md): with two proofs in flight the hash kernels of both may be placed on the same
        // CUs (two hash waves per SIMD: both chains at half speed); even / odd contexts hash on disjoint halves of the chip
        static std::atomic<uint32_t> n_ctx{0};
        const uint32_t parity = n_ctx.fetch_add(1) & 1u;
        hipDeviceProp_t prop;
        HIP_TRY(c, hipGetDeviceProperties(&prop, device));
        const uint32_t cus = (uint32_t)prop.multiProcessorCount, words = (cus + 31) / 32;
        std::vector<uint32_t> mask(words, 0);
        const int mode = lig::knobs().sha_cumask;         // 1: lower / upper half of the CU numbers, 2: even / odd CU numbers
        for (uint32_t i = 0; i < cus; i++) {
            const bool mine = mode == 2 ? (i & 1u) == parity : (i < cus / 2) == (parity == 0);
            if (mine) mask[i / 32] |= 1u << (i % 32);
        }
        HIP_TRY(c, hipExtStreamCreateWithCUMask(&c->stream_sha, words, mask.data()));
    }
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    // LIG_ENCODE_GENERIC=1 (tests): take the generic radix-2 row path even where the tiled encoder exists
    c->fast = lig::encode_fast_supported(k) && !lig::knobs().encode_generic;
    if (c->fast && (rc = make_encode_plan(c, wk, w4k)) != LIG_OK) return rc;
    c->tiled = lig::tiled_supported(ilog2u(k)) && lig::tiled_supported(ilog2u(n));
    if (c->tiled) {
        const H::Fr roots[3] = {wk, w2k, w4k};
        const uint32_t sizes[3] = {k, 2 * k, n};
        for (int w3 = 0; w3 < 3; w3++)
            for (int inv = 0; inv < 2; inv++)
                if ((rc = make_tiled_plan(c, c->tplan[w3][inv], sizes[w3], roots[w3], inv != 0)) != LIG_OK) return rc;
        if ((rc = make_tiled_plan(c, c->tplan_half_inv, 2 * k, H::mul(w4k, w4k), true)) != LIG_OK) return rc;
        for (int i = 0; i < 2; i++) {
            HIP_TRY(c, hipMalloc((void**)&c->tiled_scratch[i], 3 * (size_t)n * sizeof(fr)));
            c->owned.push_back(c->tiled_scratch[i]);
        }
    }
    HIP_TRY(c, hipMalloc((void**)&c->rk_dev, 60 * sizeof(uint32_t)));
    c->stage_cap = (size_t)16 << 20;
    HIP_TRY(c, hipHostMalloc((void**)&c->stage_host, c->stage_cap, hipHostMallocDefault));
    HIP_TRY(c, hipHostGetDevicePointer((void**)&c->stage_dev, c->stage_host, 0));
    lig::aes_upload_tables();
    HIP_TRY(c, hipDeviceSynchronize());
    return LIG_OK;
}

void lig_ctx_destroy(lig_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    // the provers and the verifier queue work on all three streams (hash / sampler / RCCL gathers on stream2, uploads and the
    // exchange on stream3): an early error return may have left some of it in flight
    for (hipStream_t st : {c->stream, c->stream2, c->stream3}) if (st) (void)hipStreamSynchronize(st);
    lig_internal_comms_release(c);
    for (void* p : c->owned) (void)hipFree(p);
    (void)hipFree(c->scratch_y); (void)hipFree(c->scratch_z); (void)hipFree(c->sample_idx);
    (void)hipFree(c->rk_dev); (void)hipFree(c->small_dev); (void)hipFree(c->tri_dev);
    if (c->stage_host) (void)hipHostFree(c->stage_host);
    for (auto& w : c->vws) (void)hipFree(w.first);
    for (auto& e : c->prof_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->stream_sha) { (void)hipStreamSynchronize(c->stream_sha); if (!c->streams_shared) (void)hipStreamDestroy(c->stream_sha); }
    if (!c->streams_shared) {       // (LIG_STREAM_MAP: the physical streams belong to the process)
        if (c->stream3 && !c->copy_is_main) (void)hipStreamDestroy(c->stream3);
        if (c->stream2 && !c->side_shared) (void)hipStreamDestroy(c->stream2);      // (the shared side stream lives for the process)
        if (c->stream) (void)hipStreamDestroy(c->stream);
    }
    delete c;
}

// the verifier's device workspace (~2.5 GB after a 2^24-constraint verification) is kept between calls; a service that is done
// verifying for a while gives it back with this
int lig_verify_release(lig_ctx* c) {
    CHECK_CTX(c);
    for (hipStream_t st : {c->stream, c->stream2, c->stream3}) if (st) HIP_TRY(c, hipStreamSynchronize(st));
    for (auto& w : c->vws) if (w.first) (void)hipFree(w.first);
    c->vws.clear();
    return LIG_OK;
}

int lig_device_pci_bus_id(int device, char* out, size_t cap) {
    if (!out || cap < 16) return LIG_E_ARG;
    return hipDeviceGetPCIBusId(out, (int)cap, device) == hipSuccess ? LIG_OK : LIG_E_HIP;
}
int lig_device_peer_access(int device, int peer, int* can) {
    if (!can) return LIG_E_ARG;
    return hipDeviceCanAccessPeer(can, device, peer) == hipSuccess ? LIG_OK : LIG_E_HIP;
}

int lig_sync(lig_ctx* c) { CHECK_CTX(c); HIP_TRY(c, hipStreamSynchronize(c->stream)); return LIG_OK; }
const char* lig_last_error(const lig_ctx* c) { return c ? c->err.c_str() : "null context"; }
uint32_t lig_message_size(const lig_ctx* c) { return c ? c->l : 0; }
uint32_t lig_padding_size(const lig_ctx* c) { return c ? c->k : 0; }
uint32_t lig_encoding_size(const lig_ctx* c) { return c ? c->n : 0; }
void* lig_stream(lig_ctx* c) { return c ? (void*)c->stream : nullptr; }

// ---------------------------------------------------------------- buffers
int lig_malloc(lig_ctx* c, size_t bytes, void** dptr) {
    CHECK_CTX(c);
    if (!dptr) return LIG_E_ARG;
    HIP_TRY(c, hipMalloc(dptr, bytes ? bytes : 1));
    HIP_TRY(c, hipMemsetAsync(*dptr, 0, bytes, c->stream));     // WebGPU buffers are zero-initialised
    return LIG_OK;
}
int lig_free(lig_ctx* c, void* dptr) {
    CHECK_CTX(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->sha.erase(dptr);
    HIP_TRY(c, hipFree(dptr));
    return LIG_OK;
}
int lig_host_alloc(lig_ctx* c, size_t bytes, void** host_ptr) {
    CHECK_CTX(c);
    if (!host_ptr) return LIG_E_ARG;
    HIP_TRY(c, hipHostMalloc(host_ptr, bytes ? bytes : 16, hipHostMallocDefault));
    return LIG_OK;
}
int lig_host_free(lig_ctx* c, void* host_ptr) {
    CHECK_CTX(c);
    if (host_ptr) HIP_TRY(c, hipHostFree(host_ptr));
    return LIG_OK;
}
int lig_write(lig_ctx* c, void* dst, const void* src, size_t bytes) {
    CHECK_CTX(c);
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));    // the host buffer may be reused right away (wgpuQueueWriteBuffer semantics)
    return LIG_OK;
}
int lig_write_async(lig_ctx* c, void* dst, const void* pinned_src, size_t bytes) {
    CHECK_CTX(c);
    if (!bytes) return LIG_OK;
    if (!dst || !pinned_src) return LIG_E_ARG;
    HIP_TRY(c, hipMemcpyAsync(dst, pinned_src, bytes, hipMemcpyHostToDevice, c->stream));
    return LIG_OK;
}
int lig_fence_record(lig_ctx* c, void** fence) {
    CHECK_CTX(c);
    if (!fence) return LIG_E_ARG;
    if (!*fence) {
        hipEvent_t ev;
        HIP_TRY(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        *fence = (void*)ev;
    }
    HIP_TRY(c, hipEventRecord((hipEvent_t)*fence, c->stream));
    return LIG_OK;
}
int lig_fence_wait(lig_ctx* c, void* fence) {
    CHECK_CTX(c);
    if (!fence) return LIG_OK;
    HIP_TRY(c, hipEventSynchronize((hipEvent_t)fence));
    return LIG_OK;
}
void lig_fence_destroy(lig_ctx* c, void* fence) {
    if (!c || !fence) return;
    if (hipSetDevice(c->device) != hipSuccess) return;
    (void)hipEventDestroy((hipEvent_t)fence);
}
int lig_write_clear(lig_ctx* c, void* dst, size_t dst_bytes, const void* src, size_t bytes) {
    CHECK_CTX(c);
    if (bytes > dst_bytes) FAIL(c, LIG_E_ARG, "write_clear: source larger than destination");
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    if (dst_bytes > bytes) HIP_TRY(c, hipMemsetAsync((char*)dst + bytes, 0, dst_bytes - bytes, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return LIG_OK;
}
int lig_clear(lig_ctx* c, void* dst, size_t bytes) { CHECK_CTX(c); HIP_TRY(c, hipMemsetAsync(dst, 0, bytes, c->stream)); return LIG_OK; }
int lig_copy(lig_ctx* c, void* dst, const void* src, size_t bytes) {
    CHECK_CTX(c);
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return LIG_OK;
}
int lig_read(lig_ctx* c, void* host_dst, const void* src, size_t bytes) {
    CHECK_CTX(c);
    HIP_TRY(c, hipMemcpyAsync(host_dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return LIG_OK;
}

// ---------------------------------------------------------------- transforms
static int ensure_scratch(lig_ctx* c, size_t rows) {
    if (rows <= c->scratch_rows) return LIG_OK;
    // the scratch is shared by every stream of the context (encodes may run on `on` streams): drain them all before freeing
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->stream2) HIP_TRY(c, hipStreamSynchronize(c->stream2));
    if (c->stream3) HIP_TRY(c, hipStreamSynchronize(c->stream3));
    (void)hipFree(c->scratch_y); (void)hipFree(c->scratch_z);
    c->scratch_y = c->scratch_z = nullptr; c->scratch_rows = 0;
    HIP_TRY(c, hipMalloc((void**)&c->scratch_y, rows * (size_t)c->k * sizeof(fr)));       // Y
    HIP_TRY(c, hipMalloc((void**)&c->scratch_z, rows * (size_t)c->n * sizeof(fr)));
    c->scratch_rows = rows;
    return LIG_OK;
}

}  // extern "C"

int lig_internal_reserve_scratch(lig_ctx* c, size_t rows) { return c->fast ? ensure_scratch(c, rows) : ensure_scratch(c, rows < 64 ? rows : 64); }

// shared by lig_encode_rows and the batched prover (msgs and out must not overlap).  half = false: out = rows x n
// codewords.  half = true: out = rows x k, out[q] = P(w_n^(4q + 2)): the odd points of the order-2k subgroup <w_n^2>
// (its even points are the message row itself, reversed: w_n^4 = w_k^-1).
int lig_internal_encode_rows(lig_ctx* c, const void* msgs, void* out, size_t rows, int mode, hipStream_t on, int phases, void* y_scratch, void* z_scratch) {
    const bool half = mode == lig::ENC_HALF;
    const size_t out_stride = half ? (size_t)c->k : (mode == lig::ENC_PLANAR || mode == lig::ENC_ZRES) ? 3 * (size_t)c->k : (size_t)c->n;
    hipStream_t st = on ? on : c->stream;
    if (mode == lig::ENC_ZRES && !c->fast) return LIG_E_STATE;       // Z tiles exist only in the tiled encoder
    if (phases != 15 && (!c->fast || rows > lig::knobs().encode_chunk)) return LIG_E_STATE;      // split phases share ONE Y scratch
    if (c->fast) {
        // rows per launch group: the Y/Z scratch (1 MiB/row) should stay inside the 256 MiB L3 so that K3's
        // re-read of Z does not go to HBM.  LIG_ENCODE_CHUNK overrides for experiments.
        const size_t chunk = lig::knobs().encode_chunk;
        int rc = ensure_scratch(c, rows < chunk ? rows : chunk);
        if (rc != LIG_OK) return rc;
        for (size_t r0 = 0; r0 < rows; r0 += chunk) {
            const size_t nr = rows - r0 < chunk ? rows - r0 : chunk;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (c->prof_on && !half && nr > 1 && (phases & 6)) {      // only the full (3 computed cosets) multi-row launches of the dominant kernel are bracketed
                if (c->prof_used == c->prof_events.size()) {
                    hipEvent_t a, b;
                    HIP_TRY(c, hipEventCreate(&a)); HIP_TRY(c, hipEventCreate(&b));
                    c->prof_events.push_back({a, b});
                }
                e0 = c->prof_events[c->prof_used].first; e1 = c->prof_events[c->prof_used].second;
                if (c->prof_launch_rows.size() < c->prof_used + 1) c->prof_launch_rows.resize(c->prof_used + 1);
                c->prof_launch_rows[c->prof_used] = (uint32_t)nr;
                c->prof_used++; c->prof_rows += nr;
            }
            lig::encode_rows_fast(st, c->ep, (const fr*)msgs + r0 * c->k, (fr*)out + r0 * out_stride, y_scratch ? (fr*)y_scratch : c->scratch_y,
                                  z_scratch ? (fr*)z_scratch : c->scratch_z, nr, e0, e1, mode, nullptr, phases);
        }
    } else if (mode == lig::ENC_PLANAR) {
        // generic path (k > 32768), planar: codewords of a few rows at a time in the Z scratch, then one strided copy per plane
        const size_t n = c->n, k = c->k, chunk = 64;
        int rc = ensure_scratch(c, rows < chunk ? rows : chunk);
        if (rc != LIG_OK) return rc;
        for (size_t r0 = 0; r0 < rows; r0 += chunk) {
            const size_t nr = rows - r0 < chunk ? rows - r0 : chunk;
            fr* z = c->scratch_z;
            HIP_TRY(c, hipMemsetAsync(z, 0, nr * n * sizeof(fr), st));
            HIP_TRY(c, hipMemcpy2DAsync(z, n * sizeof(fr), (const fr*)msgs + r0 * k, k * sizeof(fr), k * sizeof(fr), nr, hipMemcpyDeviceToDevice, st));
            lig::ntt_generic_inverse(st, c->plan[LIG_SIZE_K], z, nr, n);
            lig::ntt_generic_forward(st, c->plan[LIG_SIZE_N], z, nr, n);
            for (size_t r = 0; r < nr; r++)
                for (uint32_t cs = 1; cs < 4; cs++)
                    HIP_TRY(c, hipMemcpy2DAsync((fr*)out + (r0 + r) * 3 * k + (cs - 1) * k, sizeof(fr), z + r * n + cs, 4 * sizeof(fr), sizeof(fr), k,
                                                hipMemcpyDeviceToDevice, st));
        }
    } else if (!half) {
        // generic path: copy + zero-pad each row, INTT_k, then NTT_n with the radix-2 kernels
        HIP_TRY(c, hipMemsetAsync(out, 0, rows * out_stride * sizeof(fr), st));
        HIP_TRY(c, hipMemcpy2DAsync(out, out_stride * sizeof(fr), msgs, (size_t)c->k * sizeof(fr), (size_t)c->k * sizeof(fr),
                                    rows, hipMemcpyDeviceToDevice, st));
        lig::ntt_generic_inverse(st, c->plan[LIG_SIZE_K], (fr*)out, rows, out_stride);
        lig::ntt_generic_forward(st, c->plan[LIG_SIZE_N], (fr*)out, rows, out_stride);
    } else {
        // generic path, half: NTT_2k on <w_n^2> into the Z scratch (2k per row), then keep the odd points
        const size_t k2 = 2 * (size_t)c->k, chunk = 64;
        int rc = ensure_scratch(c, rows < chunk ? rows : chunk);
        if (rc != LIG_OK) return rc;
        for (size_t r0 = 0; r0 < rows; r0 += chunk) {
            const size_t nr = rows - r0 < chunk ? rows - r0 : chunk;
            fr* z = c->scratch_z;
            HIP_TRY(c, hipMemsetAsync(z, 0, nr * k2 * sizeof(fr), st));
            HIP_TRY(c, hipMemcpy2DAsync(z, k2 * sizeof(fr), (const fr*)msgs + r0 * c->k, (size_t)c->k * sizeof(fr), (size_t)c->k * sizeof(fr),
                                        nr, hipMemcpyDeviceToDevice, st));
            lig::ntt_generic_inverse(st, c->plan[LIG_SIZE_K], z, nr, k2);
            lig::ntt_generic_forward(st, c->plan_half, z, nr, k2);
            HIP_TRY(c, hipMemcpy2DAsync((fr*)out + r0 * c->k, sizeof(fr), z + 1, 2 * sizeof(fr), sizeof(fr), nr * (size_t)c->k,
                                        hipMemcpyDeviceToDevice, st));
        }
    }
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}
// stage-2 linear test on the coset w_n^2 <w_n^4>: for every row r the coset values of rands[r] (k values, never stored) times
// cw2[r] (k values of another matrix on the same coset, rows cw2_stride elements apart), summed per group of group_rows rows
// into part[group][k] (added to what is there).  Fast encoder only; rows <= the reserved scratch.
int lig_internal_encode_dot(lig_ctx* c, const void* rands, size_t rows, const void* cw2, size_t cw2_stride, uint32_t group_rows, void* part, hipStream_t on, bool cw2_z) {
    if (!c->fast) return LIG_E_STATE;
    hipStream_t st = on ? on : c->stream;
    int rc = ensure_scratch(c, rows);
    if (rc != LIG_OK) return rc;
    const lig::EncodeDot dot{(const fr*)cw2, cw2_stride, group_rows, (fr*)part, cw2_z};
    lig::encode_rows_fast(st, c->ep, (const fr*)rands, nullptr, c->scratch_y, c->scratch_z, rows, nullptr, nullptr, lig::ENC_DOT, &dot);
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}
// values of a degree-<2k polynomial on <w_n^2> (buf[0..2k), rest of the n-buffer zero) -> its values on all n points
// encode_2k of `rows` consecutive n-element buffers in one pass of the radix-2 kernels (the two degree-<2k mask rows)
// scratch of the tiled transforms: one per context stream that runs them
static fr* tiled_scratch_for(lig_ctx* c, hipStream_t st) { return c->tiled_scratch[st == c->stream2 ? 1 : 0]; }

int lig_internal_encode_2k_rows(lig_ctx* c, void* buf, size_t rows, hipStream_t on) {
    hipStream_t st = on ? on : c->stream;
    if (c->tiled && rows <= 3) {
        lig::ntt_tiled(st, c->tplan[LIG_SIZE_2K][1], (fr*)buf, c->n, (fr*)buf, c->n, rows, tiled_scratch_for(c, st));
        lig::ntt_tiled(st, c->tplan[LIG_SIZE_N][0], (fr*)buf, c->n, (fr*)buf, c->n, rows, tiled_scratch_for(c, st));
        HIP_TRY(c, hipGetLastError());
        return LIG_OK;
    }
    lig::ntt_generic_inverse(st, c->plan[LIG_SIZE_2K], (fr*)buf, rows, c->n);
    lig::ntt_generic_forward(st, c->plan[LIG_SIZE_N], (fr*)buf, rows, c->n);
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}
// encode of one n-element buffer with the radix-2 kernels only (no shared scratch: safe next to a batched encode on another stream)
int lig_internal_encode_generic(lig_ctx* c, void* buf, hipStream_t on) {
    hipStream_t st = on ? on : c->stream;
    if (c->tiled) {
        lig::ntt_tiled(st, c->tplan[LIG_SIZE_K][1], (fr*)buf, c->n, (fr*)buf, c->n, 1, tiled_scratch_for(c, st));
        lig::ntt_tiled(st, c->tplan[LIG_SIZE_N][0], (fr*)buf, c->n, (fr*)buf, c->n, 1, tiled_scratch_for(c, st));
        HIP_TRY(c, hipGetLastError());
        return LIG_OK;
    }
    lig::ntt_generic_inverse(st, c->plan[LIG_SIZE_K], (fr*)buf, 1, c->n);
    lig::ntt_generic_forward(st, c->plan[LIG_SIZE_N], (fr*)buf, 1, c->n);
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}
int lig_internal_decode_to(lig_ctx* c, const void* src, void* dst) {
    if (c->tiled) {
        lig::ntt_tiled(c->stream, c->tplan[LIG_SIZE_N][1], (const fr*)src, c->n, (fr*)dst, c->n, 1, c->tiled_scratch[0]);
        lig::ntt_tiled(c->stream, c->tplan[LIG_SIZE_K][0], (fr*)dst, c->n, (fr*)dst, c->n, 1, c->tiled_scratch[0], c->k);
        HIP_TRY(c, hipGetLastError());
        return LIG_OK;
    }
    HIP_TRY(c, hipMemcpyAsync(dst, src, (size_t)c->n * sizeof(fr), hipMemcpyDeviceToDevice, c->stream));
    return lig_decode(c, dst);
}
int lig_internal_extend_2k(lig_ctx* c, void* buf) {
    if (c->tiled) {
        lig::ntt_tiled(c->stream, c->tplan_half_inv, (fr*)buf, c->n, (fr*)buf, c->n, 1, c->tiled_scratch[0]);
        lig::ntt_tiled(c->stream, c->tplan[LIG_SIZE_N][0], (fr*)buf, c->n, (fr*)buf, c->n, 1, c->tiled_scratch[0]);
        HIP_TRY(c, hipGetLastError());
        return LIG_OK;
    }
    lig::ntt_generic_inverse(c->stream, c->plan_half, (fr*)buf, 1, c->n);
    lig::ntt_generic_forward(c->stream, c->plan[LIG_SIZE_N], (fr*)buf, 1, c->n);
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}

extern "C" {

int lig_encode_rows(lig_ctx* c, const void* msgs, void* codewords, size_t rows) {
    CHECK_CTX(c);
    if (!rows) return LIG_OK;
    if (!msgs || !codewords) return LIG_E_ARG;
    return lig_internal_encode_rows(c, msgs, codewords, rows, false);
}

int lig_encode(lig_ctx* c, void* buf) {
    CHECK_CTX(c);
    if (!buf) return LIG_E_ARG;
    if (c->fast) {
        // in place: K3 copies coset 0 of the codeword straight from the message row while it overwrites buf, so the
        // message is first moved to the unused tail of the Z scratch (one row of Z holds 3 cosets, the buffer has room for 4)
        int rc = ensure_scratch(c, 1);
        if (rc != LIG_OK) return rc;
        fr* mcopy = c->scratch_z + 3 * (size_t)c->k;
        HIP_TRY(c, hipMemcpyAsync(mcopy, buf, (size_t)c->k * sizeof(fr), hipMemcpyDeviceToDevice, c->stream));
        lig::encode_rows_fast(c->stream, c->ep, mcopy, (fr*)buf, c->scratch_y, c->scratch_z, 1, nullptr, nullptr);
    } else return lig_internal_encode_generic(c, buf, nullptr);
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}
int lig_encode_2k(lig_ctx* c, void* buf) {
    CHECK_CTX(c);
    if (!buf) return LIG_E_ARG;
    return lig_internal_encode_2k_rows(c, buf, 1, nullptr);
}
int lig_decode(lig_ctx* c, void* buf) {
    CHECK_CTX(c);
    if (!buf) return LIG_E_ARG;
    if (c->tiled) {
        // INTT_n; coefficients k..2k-1 folded onto 0..k-1 while the size-k forward transform loads them (N of the fold taken
        // from the 2k config: half = k, engine.cpp:782-786); buf[k..n) keeps the raw coefficients
        lig::ntt_tiled(c->stream, c->tplan[LIG_SIZE_N][1], (fr*)buf, c->n, (fr*)buf, c->n, 1, c->tiled_scratch[0]);
        lig::ntt_tiled(c->stream, c->tplan[LIG_SIZE_K][0], (fr*)buf, c->n, (fr*)buf, c->n, 1, c->tiled_scratch[0], c->k);
        HIP_TRY(c, hipGetLastError());
        return LIG_OK;
    }
    lig::ntt_generic_inverse(c->stream, c->plan[LIG_SIZE_N], (fr*)buf, 1, c->n);
    lig::ntt_generic_fold(c->stream, (fr*)buf, c->k, 1, c->n);      // N taken from the 2k config: half = k (engine.cpp:782-786)
    lig::ntt_generic_forward(c->stream, c->plan[LIG_SIZE_K], (fr*)buf, 1, c->n);
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}
int lig_ntt(lig_ctx* c, void* buf, int which, int inverse) {
    CHECK_CTX(c);
    if (!buf || which < 0 || which > 2) return LIG_E_ARG;
    if (c->tiled) {
        lig::ntt_tiled(c->stream, c->tplan[which][inverse ? 1 : 0], (fr*)buf, c->n, (fr*)buf, c->n, 1, c->tiled_scratch[0]);
        HIP_TRY(c, hipGetLastError());
        return LIG_OK;
    }
    if (inverse) lig::ntt_generic_inverse(c->stream, c->plan[which], (fr*)buf, 1, c->n);
    else lig::ntt_generic_forward(c->stream, c->plan[which], (fr*)buf, 1, c->n);
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}

// ---------------------------------------------------------------- eltwise
static bool canonical32(const uint8_t* s) {
    H::Fr v; std::memcpy(v.v, s, 32);
    return !H::geq(v, H::P);
}
int lig_eltwise(lig_ctx* c, int op, const void* x, const void* y, void* out, size_t count, const uint8_t* scalar32, uint32_t bit) {
    CHECK_CTX(c);
    if (op < LIG_OP_ADD || op > LIG_OP_BIT_DECOMPOSE || !out) return LIG_E_ARG;
    const bool needs_x = op != LIG_OP_ADD_ASSIGN ? true : true;
    const bool needs_y = op == LIG_OP_ADD || op == LIG_OP_SUB || op == LIG_OP_MUL || op == LIG_OP_FMA || op == LIG_OP_DIV;
    const bool needs_c = op == LIG_OP_ADD_CONST || op == LIG_OP_SUB_CONST || op == LIG_OP_CONST_SUB || op == LIG_OP_MUL_CONST ||
                         op == LIG_OP_MONTMUL_CONST || op == LIG_OP_FMA_CONST;
    if ((needs_x && !x) || (needs_y && !y) || (needs_c && !scalar32)) FAIL(c, LIG_E_ARG, "eltwise: missing operand");
    fr sc = lig::fr{};
    if (needs_c) {
        if (!canonical32(scalar32)) FAIL(c, LIG_E_ARG, "eltwise: scalar not reduced mod p");
        H::Fr v; std::memcpy(v.v, scalar32, 32);
        if (op == LIG_OP_MUL_CONST || op == LIG_OP_FMA_CONST) v = H::to_mont(v);
        sc = to_dev(v);
    }
    if (!count) return LIG_OK;
    lig::launch_eltwise(c->stream, op, (const fr*)x, (const fr*)y, (fr*)out, count, sc, bit);
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}

static int ensure_small(lig_ctx* c, size_t elems) {
    if (elems <= c->small_cap) return LIG_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->small_dev); c->small_dev = nullptr; c->small_cap = 0;
    size_t cap = elems < 1024 ? 1024 : elems;
    HIP_TRY(c, hipMalloc((void**)&c->small_dev, cap * sizeof(fr)));
    c->small_cap = cap;
    return LIG_OK;
}

int lig_powmod(lig_ctx* c, const uint8_t* base32, const void* exp_u32, const void* coeff, void* out, size_t count, int add) {
    CHECK_CTX(c);
    if (!base32 || !exp_u32 || !coeff || !out) return LIG_E_ARG;
    if (!canonical32(base32)) FAIL(c, LIG_E_ARG, "powmod: base not reduced mod p");
    // powmod_context::set_base (src/webgpu/powmod_context.cpp:245-268): table[i] = base^(2^i) * R
    H::Fr b; std::memcpy(b.v, base32, 32);
    std::vector<fr> table(32);
    H::Fr cur = H::to_mont(b);
    for (int i = 0; i < 32; i++) { table[i] = to_dev(cur); cur = H::montmul(cur, cur); }
    int rc = ensure_small(c, 32);
    if (rc != LIG_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->small_dev, table.data(), 32 * sizeof(fr), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (count) lig::launch_powmod(c->stream, c->small_dev, (const uint32_t*)exp_u32, (const fr*)coeff, (fr*)out, count, add);
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}

// ---------------------------------------------------------------- column SHA-256 + Merkle
size_t lig_sha_state_bytes(size_t n_inst) { return n_inst * 16 * sizeof(uint32_t); }
int lig_sha_init(lig_ctx* c, void* state, size_t n_inst) {
    CHECK_CTX(c);
    if (!state || !n_inst) return LIG_E_ARG;
    lig::launch_sha_init(c->stream, (uint32_t*)state, n_inst);
    c->sha[state] = {n_inst, 0};
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}
int lig_sha_update_rows(lig_ctx* c, void* state, const void* codewords, size_t rows) {
    CHECK_CTX(c);
    auto it = c->sha.find(state);
    if (it == c->sha.end()) FAIL(c, LIG_E_STATE, "sha_update: state was not initialised with lig_sha_init");
    if (!rows) return LIG_OK;
    if (!codewords) return LIG_E_ARG;
    const size_t n_inst = it->second.first;
    lig::launch_sha_update_rows(c->stream, (uint32_t*)state, n_inst, (const fr*)codewords, n_inst, rows, it->second.second);
    it->second.second += rows;
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}
int lig_sha_update(lig_ctx* c, void* state, const void* row) { return lig_sha_update_rows(c, state, row, 1); }
int lig_sha_final(lig_ctx* c, void* state, void* digests) {
    CHECK_CTX(c);
    auto it = c->sha.find(state);
    if (it == c->sha.end()) FAIL(c, LIG_E_STATE, "sha_final: state was not initialised with lig_sha_init");
    if (!digests) return LIG_E_ARG;
    lig::launch_sha_final(c->stream, (const uint32_t*)state, it->second.first, it->second.second, (uint32_t*)digests);
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}
size_t lig_merkle_nodes(size_t n_leaves) { size_t P = 1; while (P < n_leaves) P <<= 1; return 2 * P - 1; }
int lig_merkle_build(lig_ctx* c, const void* leaves, size_t n_leaves, void* nodes) {
    CHECK_CTX(c);
    if (!leaves || !nodes || !n_leaves) return LIG_E_ARG;
    lig::launch_merkle_build(c->stream, (const uint32_t*)leaves, n_leaves, (uint32_t*)nodes);
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}

// ---------------------------------------------------------------- sampling
int lig_sample_init(lig_ctx* c, const uint32_t* host_idx, size_t count) {
    CHECK_CTX(c);
    if (!host_idx || !count) return LIG_E_ARG;
    for (size_t i = 0; i < count; i++) if (host_idx[i] >= c->n) FAIL(c, LIG_E_ARG, "sample_init: index out of range");
    if (count > c->sample_cap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->stream2) HIP_TRY(c, hipStreamSynchronize(c->stream2));
        (void)hipFree(c->sample_idx); c->sample_idx = nullptr; c->sample_cap = 0;
        HIP_TRY(c, hipMalloc((void**)&c->sample_idx, count * sizeof(uint32_t)));
        c->sample_cap = count;
    }
    int rc_up = lig_internal_upload_small(c, c->sample_idx, host_idx, count * sizeof(uint32_t), c->stream);
    if (rc_up != LIG_OK) return rc_up;
    c->sample_count = count;
    return LIG_OK;
}
int lig_gather_rows(lig_ctx* c, const void* codewords, size_t rows, void* out) {
    CHECK_CTX(c);
    if (!c->sample_idx) FAIL(c, LIG_E_STATE, "gather: lig_sample_init has not been called");
    if (!rows) return LIG_OK;
    if (!codewords || !out) return LIG_E_ARG;
    lig::launch_gather_rows(c->stream, (const fr*)codewords, c->n, rows, c->sample_idx, (uint32_t)c->sample_count, (fr*)out);
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}
int lig_sample_gather(lig_ctx* c, const void* from, void* to, size_t slot) {
    CHECK_CTX(c);
    if (!to) return LIG_E_ARG;
    return lig_gather_rows(c, from, 1, (fr*)to + slot * c->sample_count);
}

// ---------------------------------------------------------------- stage-2 accumulators
int lig_rlc_rows(lig_ctx* c, const void* U, const void* Rn, size_t rows, const uint8_t* rc_host, void* code, void* lin,
                 const uint32_t* triples_host, const uint8_t* rq_host, size_t n_triples, void* quad) {
    CHECK_CTX(c);
    if (!rows) return LIG_OK;
    if (!U) return LIG_E_ARG;
    if (code && !rc_host) FAIL(c, LIG_E_ARG, "rlc: code accumulator without coefficients");
    if (n_triples && (!quad || !triples_host || !rq_host)) FAIL(c, LIG_E_ARG, "rlc: incomplete quadratic arguments");
    std::vector<fr> sc(rows + n_triples);
    for (size_t r = 0; r < rows && code; r++) {
        if (!canonical32(rc_host + 32 * r)) FAIL(c, LIG_E_ARG, "rlc: coefficient not reduced mod p");
        H::Fr v; std::memcpy(v.v, rc_host + 32 * r, 32); sc[r] = to_dev(H::to_mont(v));
    }
    for (size_t t = 0; t < n_triples; t++) {
        if (!canonical32(rq_host + 32 * t)) FAIL(c, LIG_E_ARG, "rlc: coefficient not reduced mod p");
        for (int q = 0; q < 3; q++) if (triples_host[3 * t + q] >= rows) FAIL(c, LIG_E_ARG, "rlc: triple row index out of range");
        H::Fr v; std::memcpy(v.v, rq_host + 32 * t, 32); sc[rows + t] = to_dev(H::to_mont(v));
    }
    int rc = ensure_small(c, rows + n_triples);
    if (rc != LIG_OK) return rc;
    if (n_triples > c->tri_cap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->tri_dev); c->tri_dev = nullptr; c->tri_cap = 0;
        HIP_TRY(c, hipMalloc((void**)&c->tri_dev, 3 * n_triples * sizeof(uint32_t)));
        c->tri_cap = n_triples;
    }
    // sc / triples_host are host temporaries: through the pinned staging ring + a copy kernel, so that the call does not wait for
    // what is queued on the stream before it (a deferred-row flush of hip_context queues a 134 MB upload and an encode first)
    rc = lig_internal_upload_small(c, c->small_dev, sc.data(), sc.size() * sizeof(fr), c->stream);
    if (rc != LIG_OK) return rc;
    if (n_triples) { rc = lig_internal_upload_small(c, c->tri_dev, triples_host, 3 * n_triples * sizeof(uint32_t), c->stream); if (rc != LIG_OK) return rc; }
    lig::launch_rlc_rows(c->stream, (const fr*)U, (const fr*)Rn, rows, c->n, c->small_dev, (fr*)code, (fr*)lin, c->tri_dev,
                         c->small_dev + rows, n_triples, (fr*)quad);
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}

// ---------------------------------------------------------------- per-kernel timing of the dominant kernel
int lig_profile_enable(lig_ctx* c, int on) {
    CHECK_CTX(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->prof_on = on != 0; c->prof_used = 0; c->prof_rows = 0;
    return LIG_OK;
}
int lig_profile_read(lig_ctx* c, uint64_t* launches, uint64_t* rows, double* total_ms) {
    CHECK_CTX(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double ms = 0;
    for (size_t i = 0; i < c->prof_used; i++) {
        float t = 0;
        HIP_TRY(c, hipEventElapsedTime(&t, c->prof_events[i].first, c->prof_events[i].second));
        ms += t;
    }
    if (launches) *launches = c->prof_used;
    if (rows) *rows = c->prof_rows;
    if (total_ms) *total_ms = ms;
    return LIG_OK;
}

// the same for the launches of exactly `rows_in_launch` rows (bench.py: the 512-row launches, the figure a rocprofv3 kernel table shows per
// launch size -- DESIGN.md section 6's reconciliation needs no arithmetic over the six launch sizes of a proof)
int lig_profile_read_launches(lig_ctx* c, uint32_t rows_in_launch, uint64_t* launches, double* total_ms) {
    CHECK_CTX(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double ms = 0;
    uint64_t cnt = 0;
    for (size_t i = 0; i < c->prof_used; i++) {
        if (c->prof_launch_rows[i] != rows_in_launch) continue;
        float t = 0;
        HIP_TRY(c, hipEventElapsedTime(&t, c->prof_events[i].first, c->prof_events[i].second));
        ms += t; cnt++;
    }
    if (launches) *launches = cnt;
    if (total_ms) *total_ms = ms;
    return LIG_OK;
}

// ---------------------------------------------------------------- AES-CTR field sampler
int lig_rng_fill(lig_ctx* c, const uint8_t* key32, uint64_t first_elem, void* out, size_t count) {
    CHECK_CTX(c);
    if (!key32 || (!out && count)) return LIG_E_ARG;
    uint32_t rk[60];
    lig::aes256_expand_host(key32, rk);
    { int rc_up = lig_internal_upload_small(c, c->rk_dev, rk, sizeof rk, c->stream); if (rc_up != LIG_OK) return rc_up; }
    lig::launch_rng_fill(c->stream, c->rk_dev, first_elem, (fr*)out, count);
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}
int lig_rng_fill_rows(lig_ctx* c, const uint8_t* key32, uint64_t first_elem, const uint32_t* per_row_host, size_t rows, void* out) {
    CHECK_CTX(c);
    if (!key32 || (rows && (!out || !per_row_host))) return LIG_E_ARG;
    for (size_t r = 0; r < rows; r++) if (per_row_host[r] > c->k) FAIL(c, LIG_E_ARG, "rng_fill_rows: more elements than a row holds");
    uint32_t rk[60];
    lig::aes256_expand_host(key32, rk);
    { int rc_up = lig_internal_upload_small(c, c->rk_dev, rk, sizeof rk, c->stream); if (rc_up != LIG_OK) return rc_up; }
    uint64_t pos = first_elem;
    for (size_t r = 0; r < rows;) {           // runs of rows with equal fill are one launch
        size_t run = 1;
        while (r + run < rows && per_row_host[r + run] == per_row_host[r]) run++;
        lig::launch_rng_fill_rows_dense(c->stream, c->rk_dev, pos, (fr*)out + r * (size_t)c->k, run, per_row_host[r], c->k);
        pos += (uint64_t)run * per_row_host[r]; r += run;
    }
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}

}  // extern "C"
