// tile_dft.hpp -- the in-LDS transform primitives shared by the batched row encoder (ntt_encode.hip) and the tiled
// single-row transforms (ntt_tiled.hip): radix-8 DIT butterfly in registers, the LDS exchange tile, radix-2^2 DIT steps.
// All arithmetic on 29-bit-limb lazy values (fr29.hpp); bounds are stated next to each operation.
#pragma once
#include "fr29.hpp"

namespace lig {

__device__ __forceinline__ constexpr int brev3(int p) { return ((p & 1) << 2) | (p & 2) | ((p >> 2) & 1); }

// Radix-8 DIT butterfly in registers.  In: a[p] = input number brev3(p), normalised limbs, value < 3p.
// Out: a[j] = sum_i in[i] * w^(i*j), lazy (limbs < 2^31 + 8, value < 28p).  w8[1..3] = w, w^2, w^3 (either table format; the
// entries are uniform, so their loads are broadcasts).
// SCALAR: the three constants as scalar operands (s_load + SGPR source, fr29.hpp: f29ws) -- for straight-line kernels (K1, K3);
// a kernel that runs the butterfly in a loop keeps them in vector registers instead, loaded once (81 SGPRs do not stay live
// next to the loop's own, and re-loading them per iteration costs more than it saves: k_encode_out_dot 76 -> 108 us).
template <bool SCALAR = true, class TW>
__device__ __forceinline__ void radix8_dit(f29 (&a)[8], const TW w8) {
    f29 t, u;
    auto konst = [](const TW p) { if constexpr (SCALAR) return tab_get_uniform(p); else return tab_get(p); };
    const auto w2 = konst(w8 + 2);
    // span 2, twiddle 1.  u: limbs < 2^30, < 6p.  v = x - y + 4p: limbs < 2^31, < 7p.
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        u = f29_add(a[i], a[i + 1]);
        a[i + 1] = f29_sub_k4(a[i], a[i + 1]);
        a[i] = u;
    }
    // span 4.  pairs (0,2),(4,6): twiddle 1, subtrahend limbs < 2^30, < 6p -> + 8p.  pairs (1,3),(5,7): twiddle w^2.
#pragma unroll
    for (int b = 0; b < 8; b += 4) {
        u = f29_add(a[b], a[b + 2]);               // limbs < 2^31, < 12p
        a[b + 2] = f29_sub_k8(a[b], a[b + 2]);     // limbs < 3.5 * 2^30, < 14p
        a[b] = u;
        t = tab_mul(a[b + 3], w2);                 // input limbs < 2^31
        u = f29_add(a[b + 1], t);                  // limbs < 2.5 * 2^30, < 8.2p
        a[b + 3] = f29_sub_k2(a[b + 1], t);        // limbs < 3 * 2^30, < 9p
        a[b + 1] = u;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = f29_qnorm(a[i]);
    // span 8.  pair (0,4): twiddle 1, subtrahend limbs < 2^29+8, < 12p -> + 16p.  others: w, w^2, w^3.
    u = f29_add(a[0], a[4]); a[4] = f29_sub_k16(a[0], a[4]); a[0] = u;                 // < 24p | limbs < 2^31+8, < 28p
    t = tab_mul(a[5], konst(w8 + 1)); u = f29_add(a[1], t); a[5] = f29_sub_k2(a[1], t); a[1] = u;
    t = tab_mul(a[6], w2); u = f29_add(a[2], t); a[6] = f29_sub_k2(a[2], t); a[2] = u;
    t = tab_mul(a[7], konst(w8 + 3)); u = f29_add(a[3], t); a[7] = f29_sub_k2(a[3], t); a[3] = u;
}

// ---------------------------------------------------------------------------------------------------- tile transform
// LDS exchange buffer: element `pos` of the tile = limbs 0-3 | limbs 4-7 | limb 8 in three planes.
template <int LOG2B>
struct TileLds {
    static constexpr uint32_t B = 1u << LOG2B;
    uint4 lo[B];
    uint4 hi[B];
    uint32_t top[B];
};
template <int LOG2B>
__device__ __forceinline__ void lds_put(TileLds<LOG2B>& L, uint32_t pos, const f29& x) {
    L.lo[pos] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    L.hi[pos] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
    L.top[pos] = x.v[8];
}
template <int LOG2B>
__device__ __forceinline__ f29 lds_get(const TileLds<LOG2B>& L, uint32_t pos) {
    const uint4 a = L.lo[pos], b = L.hi[pos];
    f29 r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    r.v[8] = L.top[pos];
    return r;
}

// Synchronisation between a step's LDS writes and the next step's reads.  The next step reads inside blocks of SPAN
// elements; a wave owns 256 consecutive elements (64 lanes x 4), so for SPAN <= 256 -- or a workgroup that is a single
// wave -- the exchange never leaves the wave: LDS instructions of one wave execute in order, only the compiler has to be
// kept from reordering them.  (A step re-writes exactly the positions it read, so nothing is needed between its own reads
// and writes.)  For B = 1024 this leaves ONE s_barrier per tile transform instead of eight.
template <uint32_t SPAN, uint32_t THREADS>
__device__ __forceinline__ void tile_sync() {
    if constexpr (SPAN <= 256 || THREADS <= 64) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    else __syncthreads();
}

// EXPERIMENT (-DLIG_TILE_DPP, not the default; result in profiles/r02_dpp_exchange_ab.md): the first exchange of a tile
// transform (after spans 2 and 4: thread t holds positions 4t + q and needs 16(t/4) + (t%4) + 4q) is a 4x4 transpose inside
// a quad of adjacent lanes, done in registers with DPP quad_perm moves instead of through LDS: two butterfly rounds
// (lane^1, lane^2), per dword one select of what to send, one v_mov_dpp, two selects to place it.
#ifdef LIG_TILE_DPP
__device__ __forceinline__ void quad_transpose(f29 (&x)[4], const uint32_t lane) {
    const bool odd = lane & 1u, hi = lane & 2u;
#pragma unroll
    for (int w = 0; w < 9; w++) {
#pragma unroll
        for (int a = 0; a < 4; a += 2) {                  // pairs (0,1), (2,3) against lane ^ 1
            const uint32_t send = odd ? x[a].v[w] : x[a + 1].v[w];
            const uint32_t recv = (uint32_t)__builtin_amdgcn_mov_dpp((int)send, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
            x[a].v[w] = odd ? recv : x[a].v[w];
            x[a + 1].v[w] = odd ? x[a + 1].v[w] : recv;
        }
#pragma unroll
        for (int a = 0; a < 2; a++) {                      // pairs (0,2), (1,3) against lane ^ 2
            const uint32_t send = hi ? x[a].v[w] : x[a + 2].v[w];
            const uint32_t recv = (uint32_t)__builtin_amdgcn_mov_dpp((int)send, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
            x[a].v[w] = hi ? recv : x[a].v[w];
            x[a + 2].v[w] = hi ? x[a + 2].v[w] : recv;
        }
    }
}
#endif

// One radix-2^2 DIT step (spans M/2 and M, M = 4^(S+1)) of the size-B transform.  Thread t owns positions
// base + q*Q (Q = M/4).  tw: the stage of span M' starts at entry M'/2 - 1 (M'/2 entries rho^(j*B/M')).
template <int LOG2B, int S, class TW>
__device__ __forceinline__ void tile_step(f29 (&x)[4], const TW tw, TileLds<LOG2B>& L, const uint32_t t) {
    constexpr int STEPS = LOG2B / 2;
    constexpr uint32_t M = 4u << (2 * S), Q = M >> 2;
    const uint32_t p = t & (Q - 1);
    const uint32_t base = (t / Q) * M + p;
#ifdef LIG_TILE_DPP
    if constexpr (S > 1) {
#else
    if constexpr (S > 0) {
#endif
#pragma unroll
        for (int q = 0; q < 4; q++) x[q] = lds_get(L, base + q * Q);
    }
    f29 t1, t3, u;
    if constexpr (S == 0) {
        // spans 2 and 4: twiddles 1 | 1, W_4^1.  inputs normalised, < 3p.
        u = f29_add(x[0], x[1]); x[1] = f29_sub_k4(x[0], x[1]); x[0] = u;         // u: limbs < 2^30, < 6p; v: limbs < 2^31, < 7p
        u = f29_add(x[2], x[3]); x[3] = f29_sub_k4(x[2], x[3]); x[2] = u;
        t3 = tab_mul(x[3], tab_get(tw + 2));                               // W_4^1
        u = f29_add(x[0], x[2]); x[2] = f29_sub_k8(x[0], x[2]); x[0] = u;           // < 12p | limbs < 3.5*2^30, < 14p
        u = f29_add(x[1], t3); x[3] = f29_sub_k2(x[1], t3); x[1] = u;               // limbs < 3*2^30, < 9p
    } else {
        // operands at rest: limbs < 2^29 + 8.  every output gains at most 4p.
        const auto wa = tab_get(tw + (Q - 1) + p);                                   // W_{M/2}^p
        t1 = tab_mul(x[1], wa);
        t3 = tab_mul(x[3], wa);
        u = f29_add(x[0], t1); x[1] = f29_sub_k2(x[0], t1); x[0] = u;               // limbs < 2^30+8 | < 1.5*2^30+8
        u = f29_add(x[2], t3); x[3] = f29_sub_k2(x[2], t3); x[2] = u;
        t1 = tab_mul(x[2], tab_get(tw + (2 * Q - 1) + p));                 // W_M^p
        t3 = tab_mul(x[3], tab_get(tw + (2 * Q - 1) + p + Q));             // W_M^(p+Q)
        u = f29_add(x[0], t1); x[2] = f29_sub_k2(x[0], t1); x[0] = u;               // limbs < 2^31 + 8
        u = f29_add(x[1], t3); x[3] = f29_sub_k2(x[1], t3); x[1] = u;               // limbs < 2.5*2^30 + 8
    }
    if constexpr (S + 1 < STEPS) {
#ifdef LIG_TILE_DPP
        if constexpr (S == 0) {
#pragma unroll
            for (int q = 0; q < 4; q++) x[q] = f29_qnorm(x[q]);
            quad_transpose(x, t);
            tile_step<LOG2B, S + 1, TW>(x, tw, L, t);
            return;
        }
#endif
#pragma unroll
        for (int q = 0; q < 4; q++) lds_put(L, base + q * Q, f29_qnorm(x[q]));
        tile_sync<4 * M, (1u << LOG2B) / 4>();
        tile_step<LOG2B, S + 1, TW>(x, tw, L, t);
    } else if constexpr (LOG2B & 1) {
        // B = 2 * 4^STEPS: the radix-4 steps have built the two half-size transforms; one radix-2 stage of span B joins
        // them.  Thread t owns positions t + q*B/4 from here on (the exit ownership), i.e. the pairs (t, t + B/2) and
        // (t + B/4, t + 3B/4) with twiddles W_B^t and W_B^(t + B/4).
        constexpr uint32_t B = 1u << LOG2B, T = B / 4;
#pragma unroll
        for (int q = 0; q < 4; q++) lds_put(L, base + q * Q, f29_qnorm(x[q]));
        tile_sync<B, T>();
#pragma unroll
        for (int q = 0; q < 4; q++) x[q] = lds_get(L, t + q * T);
        const f29 ta = tab_mul(x[2], tab_get(tw + (B / 2 - 1) + t));
        const f29 tb = tab_mul(x[3], tab_get(tw + (B / 2 - 1) + t + T));
        u = f29_add(x[0], ta); x[2] = f29_sub_k2(x[0], ta); x[0] = u;               // at-rest operands: + at most 2p
        u = f29_add(x[1], tb); x[3] = f29_sub_k2(x[1], tb); x[1] = u;
    }
}
// Size-B DIT transform.  Entry: x[q] = input number brev(4t + q) (bit-reversed load), normalised, < 3p.
// Exit: x[q] = output number t + q*B/4 (natural order, the coalesced ownership pattern), lazy:
// limbs < 2.5*2^30 + 8, value < 14p + 4p*(STEPS-1) + 2p <= 30p.
template <int LOG2B, class TW>
__device__ __forceinline__ void tile_dft(f29 (&x)[4], const TW tw, TileLds<LOG2B>& L, const uint32_t t) {
    static_assert(LOG2B >= 4 && LOG2B <= 12, "tile length 16 .. 4096 (36 bytes of LDS per element: 144 KiB of the 160 KiB of a CU at most)");
    tile_step<LOG2B, 0, TW>(x, tw, L, t);
}

}  // namespace lig
