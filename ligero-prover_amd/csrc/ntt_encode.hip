// ntt_encode.hip -- the hot loop: Reed-Solomon encode of a BATCH of rows on gfx950.
//
// Replaces encode_ntt_device (src/webgpu/engine.cpp:755-770: 15 dispatches of <=64 workgroups per row,
// radix-2 stages through global memory, bit-reversal passes) with four launches per row batch.
//
// Math.  codeword[j] = P(w_n^j), j < n = 4k, P = the degree-<k interpolant of the message on
// the w_k domain.  With c = INTT_k(msg), psi = w_n^4 (order k) and j = 4q + r:
//     codeword[4q + r] = sum_i (c[i] * w_n^(r*i)) * psi^(i*q)                       (4 coset NTTs of size k;
// the 3k zero-padded coefficients of the reference's size-n transform are never touched).  Each size-k
// transform is split k = 8*B so that every global store is a contiguous >= 512-byte run per wave and the
// size-B part runs out of LDS:
//
//   K1  encode_in  : thread = (row, i2).  Radix-8 butterfly across the strided elements msg[B*i1 + i2], seam
//                    twiddle w_k^(-i2*j1) -> Y[j1][i2].
//   K2a encode_coef: workgroup = (row, j1).  Size-B inverse transform of Y[j1][.] in LDS (the 1/k factor rides on the K2b twist)
//                    -> C[j1][j2] = coefficient c[j1 + 8*j2].
//   K2b encode_mid : workgroup = (row, j1, coset r).  C[j1][.] * w_n^(r*i) -> size-B forward transform in LDS
//                    -> times seam twiddle psi^(j1*q2) -> Z[r][j1][q2].                 (72% of all multiplies)
//   K3  encode_out : thread = (row, q2, r).  Radix-8 butterfly across Z[r][0..8)[q2], exact canonical
//                    reduction, codeword[4*(q2 + B*q1) + r]: 4 adjacent lanes = 4 cosets = 128 contiguous bytes,
//                    a wave writes 2 KiB runs in natural order (no bit-reversal pass anywhere).
//
// All butterflies are decimation-in-time on 29-bit-limb lazy values (fr29.hpp): x' = x + w*y, y' = x - w*y + 2p,
// so magnitudes grow additively (<= 4p per radix-4 step) and no modular reduction is needed inside a transform;
// limbs are renormalised once per radix-4 step.  Y, C, Z hold values < 2^256 in the canonical 8 x u32 layout
// (not necessarily < p); only K3 produces canonical residues, which is all the reference guarantees as well.
//
// Limb/value bounds are stated next to each operation; the invariants are
//   "at rest" (LDS / loaded from HBM): limbs < 2^29 + 8, value < 30p;
//   Montgomery-product input a: limbs <= 2.5 * 2^30 + 8; every subtraction adds a borrow-proof multiple of p
//   whose limbs dominate the subtrahend's limbs and whose value dominates its value.
#include "fr29.hpp"
#include "kernels.hpp"
#include <cstdlib>

namespace lig {

__device__ __forceinline__ constexpr int brev3(int p) { return ((p & 1) << 2) | (p & 2) | ((p >> 2) & 1); }

// Radix-8 DIT butterfly in registers.  In: a[p] = input number brev3(p), normalised limbs, value < 3p.
// Out: a[j] = sum_i in[i] * w^(i*j), lazy (limbs < 2^31 + 8, value < 28p).  w1,w2,w3 = w, w^2, w^3 (Montgomery form).
__device__ __forceinline__ void radix8_dit(f29 (&a)[8], const f29& w1, const f29& w2, const f29& w3) {
    f29 t, u;
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
        t = f29_montmul(a[b + 3], w2);             // input limbs < 2^31
        u = f29_add(a[b + 1], t);                  // limbs < 2.5 * 2^30, < 8.2p
        a[b + 3] = f29_sub_k2(a[b + 1], t);        // limbs < 3 * 2^30, < 9p
        a[b + 1] = u;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = f29_qnorm(a[i]);
    // span 8.  pair (0,4): twiddle 1, subtrahend limbs < 2^29+8, < 12p -> + 16p.  others: w, w^2, w^3.
    u = f29_add(a[0], a[4]); a[4] = f29_sub_k16(a[0], a[4]); a[0] = u;                 // < 24p | limbs < 2^31+8, < 28p
    t = f29_montmul(a[5], w1); u = f29_add(a[1], t); a[5] = f29_sub_k2(a[1], t); a[1] = u;
    t = f29_montmul(a[6], w2); u = f29_add(a[2], t); a[6] = f29_sub_k2(a[2], t); a[2] = u;
    t = f29_montmul(a[7], w3); u = f29_add(a[3], t); a[7] = f29_sub_k2(a[3], t); a[3] = u;
}

// ---------------------------------------------------------------------------------------------------- K1
template <int LOG2B>
__global__ void __launch_bounds__(256, 2) k_encode_in(const fr* __restrict__ msgs, fr* __restrict__ Y, const f29s* __restrict__ seam_inv,
                                                   const f29s* __restrict__ w8, size_t rows) {
    constexpr uint32_t B = 1u << LOG2B, K = 8u * B;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t row = gid >> LOG2B;
    const uint32_t i2 = (uint32_t)gid & (B - 1);
    if (row >= rows) return;
    const fr* m = msgs + row * K;
    f29 a[8];
#pragma unroll
    for (int p = 0; p < 8; p++) a[p] = unpack29(fr_load(m + (size_t)brev3(p) * B + i2));      // canonical inputs
    radix8_dit(a, f29_load_tab(w8 + 1), f29_load_tab(w8 + 2), f29_load_tab(w8 + 3));
    fr* y = Y + row * K;
    fr_store(y + i2, pack29(f29_reduce_2p(a[0])));
    // explicit unrolling: hipcc leaves a `#pragma unroll` loop over 7 Montgomery products rolled and then keeps a[] in scratch
#define LIG_SEAM(J1) fr_store(y + (size_t)(J1) * B + i2, pack29(f29_montmul(a[J1], f29_load_tab(seam_inv + (size_t)(J1) * B + i2))))
    LIG_SEAM(1); LIG_SEAM(2); LIG_SEAM(3); LIG_SEAM(4); LIG_SEAM(5); LIG_SEAM(6); LIG_SEAM(7);
#undef LIG_SEAM
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

// One radix-2^2 DIT step (spans M/2 and M, M = 4^(S+1)) of the size-B transform.  Thread t owns positions
// base + q*Q (Q = M/4).  tw: the stage of span M' starts at entry M'/2 - 1 (M'/2 entries rho^(j*B/M')).
template <int LOG2B, int S>
__device__ __forceinline__ void tile_step(f29 (&x)[4], const f29s* __restrict__ tw, TileLds<LOG2B>& L, const uint32_t t) {
    constexpr int STEPS = LOG2B / 2;
    constexpr uint32_t M = 4u << (2 * S), Q = M >> 2;
    const uint32_t p = t & (Q - 1);
    const uint32_t base = (t / Q) * M + p;
    if constexpr (S > 0) {
#pragma unroll
        for (int q = 0; q < 4; q++) x[q] = lds_get(L, base + q * Q);
    }
    f29 t1, t3, u;
    if constexpr (S == 0) {
        // spans 2 and 4: twiddles 1 | 1, W_4^1.  inputs normalised, < 3p.
        u = f29_add(x[0], x[1]); x[1] = f29_sub_k4(x[0], x[1]); x[0] = u;         // u: limbs < 2^30, < 6p; v: limbs < 2^31, < 7p
        u = f29_add(x[2], x[3]); x[3] = f29_sub_k4(x[2], x[3]); x[2] = u;
        t3 = f29_montmul(x[3], f29_load_tab(tw + 2));                               // W_4^1
        u = f29_add(x[0], x[2]); x[2] = f29_sub_k8(x[0], x[2]); x[0] = u;           // < 12p | limbs < 3.5*2^30, < 14p
        u = f29_add(x[1], t3); x[3] = f29_sub_k2(x[1], t3); x[1] = u;               // limbs < 3*2^30, < 9p
    } else {
        // operands at rest: limbs < 2^29 + 8.  every output gains at most 4p.
        const f29 wa = f29_load_tab(tw + (Q - 1) + p);                               // W_{M/2}^p
        t1 = f29_montmul(x[1], wa);
        t3 = f29_montmul(x[3], wa);
        u = f29_add(x[0], t1); x[1] = f29_sub_k2(x[0], t1); x[0] = u;               // limbs < 2^30+8 | < 1.5*2^30+8
        u = f29_add(x[2], t3); x[3] = f29_sub_k2(x[2], t3); x[2] = u;
        t1 = f29_montmul(x[2], f29_load_tab(tw + (2 * Q - 1) + p));                 // W_M^p
        t3 = f29_montmul(x[3], f29_load_tab(tw + (2 * Q - 1) + p + Q));             // W_M^(p+Q)
        u = f29_add(x[0], t1); x[2] = f29_sub_k2(x[0], t1); x[0] = u;               // limbs < 2^31 + 8
        u = f29_add(x[1], t3); x[3] = f29_sub_k2(x[1], t3); x[1] = u;               // limbs < 2.5*2^30 + 8
    }
    if constexpr (S + 1 < STEPS) {
#pragma unroll
        for (int q = 0; q < 4; q++) lds_put(L, base + q * Q, f29_qnorm(x[q]));
        tile_sync<4 * M, (1u << LOG2B) / 4>();
        tile_step<LOG2B, S + 1>(x, tw, L, t);
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
        const f29 ta = f29_montmul(x[2], f29_load_tab(tw + (B / 2 - 1) + t));
        const f29 tb = f29_montmul(x[3], f29_load_tab(tw + (B / 2 - 1) + t + T));
        u = f29_add(x[0], ta); x[2] = f29_sub_k2(x[0], ta); x[0] = u;               // at-rest operands: + at most 2p
        u = f29_add(x[1], tb); x[3] = f29_sub_k2(x[1], tb); x[1] = u;
    }
}
// Size-B DIT transform.  Entry: x[q] = input number brev(4t + q) (bit-reversed load), normalised, < 3p.
// Exit: x[q] = output number t + q*B/4 (natural order, the coalesced ownership pattern), lazy:
// limbs < 2.5*2^30 + 8, value < 14p + 4p*(STEPS-1) + 2p <= 30p.
template <int LOG2B>
__device__ __forceinline__ void tile_dft(f29 (&x)[4], const f29s* __restrict__ tw, TileLds<LOG2B>& L, const uint32_t t) {
    static_assert(LOG2B >= 4 && LOG2B <= 10, "tile length 16 .. 1024 (36 bytes of LDS per element, static LDS <= 64 KiB)");
    tile_step<LOG2B, 0>(x, tw, L, t);
}

// ---------------------------------------------------------------------------------------------------- K2a
template <int LOG2B>
__global__ void __launch_bounds__((1 << LOG2B) / 4) k_encode_coef(const fr* __restrict__ Y, fr* __restrict__ Cc,
                                                                  const f29s* __restrict__ tw_inv, const f29s* __restrict__ kinv) {
    constexpr uint32_t B = 1u << LOG2B, T = B / 4;
    __shared__ TileLds<LOG2B> L;
    const uint32_t t = threadIdx.x;
    const fr* y = Y + (size_t)blockIdx.x * B;          // tile (row, j1) = blockIdx.x
    f29 x[4];
#pragma unroll
    for (int q = 0; q < 4; q++) x[q] = unpack29(fr_load(y + (__brev(4 * t + q) >> (32 - LOG2B))));
    tile_dft<LOG2B>(x, tw_inv, L, t);
    (void)kinv;                                    // the 1/k factor is folded into the twist tables of K2b (lig_capi.hip)
    fr* c = Cc + (size_t)blockIdx.x * B;
#pragma unroll
    for (int q = 0; q < 4; q++) fr_store(c + t + q * T, pack29(f29_reduce_2p(x[q])));
}

// ---------------------------------------------------------------------------------------------------- K2b
// Coset 0 of the codeword needs no arithmetic: w_4k comes from root2 = root1^(2^61 - 1) (src/bn254.cpp:36-43), so
// w_4k^4 = w_k^(2^61 - 1) = w_k^(-1) and codeword[4q] = P(w_k^(-q)) = msg[(k - q) mod k] -- a reversed copy of the message.
// FULL = true : cosets r = 1, 2, 3 are computed here (3 workgroups per tile), K3 copies coset 0 from the message row.
// FULL = false: only coset r = 2 (the odd points of the order-2k subgroup <w_n^2>) -- all a stage-2 randomness row
//               needs: its values on the even points are the row itself (prover.hip).
template <int LOG2B, bool FULL>
__global__ void __launch_bounds__((1 << LOG2B) / 4) k_encode_mid(const fr* __restrict__ Cc, fr* __restrict__ Z,
                                                                 const f29s* __restrict__ tw_fwd, const f29s* __restrict__ twist,
                                                                 const f29s* __restrict__ seam_fwd) {
    constexpr uint32_t B = 1u << LOG2B, T = B / 4, NC = FULL ? 3 : 1;
    __shared__ TileLds<LOG2B> L;
    const uint32_t t = threadIdx.x;
    // XCD-aware mapping: workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), each with a private L2.  The NC
    // coset workgroups of one tile read the same 32 KiB of coefficients, so they get block ids 8 apart (same XCD)
    // and the tile index is the fastest-varying part: per group of 8*NC blocks, b = j1 + 8*ci.
    const uint32_t j1 = blockIdx.x & 7u, ci = (blockIdx.x >> 3) % NC;      // tile, coset slot
    const uint32_t r = FULL ? ci + 1 : 2;                                  // coset number
    const size_t row = blockIdx.x / (8 * NC);
    const fr* c = Cc + (row * 8 + j1) * (size_t)B;
    f29 x[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t pos = __brev(4 * t + q) >> (32 - LOG2B);
        x[q] = f29_montmul(unpack29(fr_load(c + pos)), f29_load_tab(twist + ((size_t)(r - 1) * 8 + j1) * B + pos));   // w_n^(r*(j1 + 8*pos))
    }
    tile_dft<LOG2B>(x, tw_fwd, L, t);
    fr* z = Z + ((row * NC + ci) * 8 + j1) * (size_t)B;
    if (j1 != 0) {
        const f29s* sf = seam_fwd + (size_t)j1 * B;
#pragma unroll
        for (int q = 0; q < 4; q++) fr_store(z + t + q * T, pack29(f29_montmul(x[q], f29_load_tab(sf + t + q * T))));
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) fr_store(z + t + q * T, pack29(f29_reduce_2p(x[q])));
    }
}

// ---------------------------------------------------------------------------------------------------- K3
// FULL : out row = the codeword, element 4*(q2 + B*q1) + r; the lanes with r = 0 copy the reversed message instead of
//        running the radix-8 step, so every 128-byte line of the codeword is still written by four adjacent lanes.
// !FULL: out row = k elements, element q = q2 + B*q1 is P(w_n^(4q + 2)).
template <int LOG2B, bool FULL>
__global__ void __launch_bounds__(256) k_encode_out(const fr* __restrict__ Z, fr* __restrict__ cw, const f29s* __restrict__ w8,
                                                    const fr* __restrict__ msgs, size_t rows) {
    constexpr uint32_t B = 1u << LOG2B, K = 8u * B;
    constexpr int LOGNC = FULL ? 2 : 0;
    constexpr uint32_t NC = FULL ? 3 : 1, OS = FULL ? 4 : 1;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t row = gid >> (LOG2B + LOGNC);
    if (row >= rows) return;
    const uint32_t r = FULL ? (uint32_t)gid & 3u : 0u;
    const uint32_t q2 = ((uint32_t)gid >> LOGNC) & (B - 1);
    fr* out = cw + row * (OS * (size_t)K);
    fr v[8];
    if (FULL && r == 0) {
        // loads only: the stores below are shared with the other three lanes of the 128-byte line (one full-line write)
        const fr* m = msgs + row * (size_t)K;
#pragma unroll
        for (int q1 = 0; q1 < 8; q1++) v[q1] = fr_load(m + ((K - (q2 + B * q1)) & (K - 1)));
    } else {
        const uint32_t ci = FULL ? r - 1 : 0;
        const fr* z = Z + ((row * NC + ci) * 8) * (size_t)B + q2;
        f29 a[8];
#pragma unroll
        for (int p = 0; p < 8; p++) a[p] = unpack29(fr_load(z + (size_t)brev3(p) * B));
        radix8_dit(a, f29_load_tab(w8 + 1), f29_load_tab(w8 + 2), f29_load_tab(w8 + 3));
#pragma unroll
        for (int q1 = 0; q1 < 8; q1++) v[q1] = pack29(f29_canon(a[q1]));
    }
#pragma unroll
    for (int q1 = 0; q1 < 8; q1++) fr_store(out + OS * ((size_t)q2 + (size_t)B * q1) + r, v[q1]);
}

bool encode_fast_supported(uint32_t k) { return k == 512 || k == 1024 || k == 2048 || k == 4096 || k == 8192; }

template <int LOG2B, bool FULL>
static void encode_rows_t(hipStream_t s, const EncodePlan& ep, const fr* msgs, fr* cw, fr* Y, fr* Z, size_t rows,
                          hipEvent_t ev0, hipEvent_t ev1) {
    constexpr uint32_t B = 1u << LOG2B;
    fr* Cc = Y + rows * (size_t)(8 * B);      // second half of the Y scratch (2 * rows * k elements)
    const size_t th1 = rows * B;
    static const int kmask = [] { const char* e = std::getenv("LIG_ENCODE_KMASK"); return e ? std::atoi(e) : 15; }();   // experiments only
    if (kmask & 1) hipLaunchKernelGGL(k_encode_in<LOG2B>, dim3((uint32_t)((th1 + 255) / 256)), dim3(256), 0, s, msgs, Y, ep.seam_inv, ep.w8_inv, rows);
    if (kmask & 2) hipLaunchKernelGGL(k_encode_coef<LOG2B>, dim3((uint32_t)(rows * 8)), dim3(B / 4), 0, s, Y, Cc, ep.tw_b_inv, ep.kinv);
    if (ev0) (void)hipEventRecord(ev0, s);
    if (kmask & 4) hipLaunchKernelGGL((k_encode_mid<LOG2B, FULL>), dim3((uint32_t)(rows * 8 * (FULL ? 3 : 1))), dim3(B / 4), 0, s, Cc, Z, ep.tw_b, ep.twist, ep.seam_fwd);
    if (ev1) (void)hipEventRecord(ev1, s);
    const size_t th3 = rows * B * (FULL ? 4 : 1);
    if (kmask & 8) hipLaunchKernelGGL((k_encode_out<LOG2B, FULL>), dim3((uint32_t)((th3 + 255) / 256)), dim3(256), 0, s, Z, cw, ep.w8_fwd, msgs, rows);
}

// half = false: codewords (rows x n).  half = true: rows x k values on the coset w_n^2 <w_n^4>, out[q] = P(w_n^(4q + 2)).
void encode_rows_fast(hipStream_t s, const EncodePlan& ep, const fr* msgs, fr* out, fr* scratch_y, fr* scratch_z, size_t rows,
                      hipEvent_t ev0, hipEvent_t ev1, bool half) {
    switch (ep.log2B * 2 + (half ? 1 : 0)) {
        case 12: encode_rows_t<6, true>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1); break;
        case 13: encode_rows_t<6, false>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1); break;
        case 14: encode_rows_t<7, true>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1); break;
        case 15: encode_rows_t<7, false>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1); break;
        case 18: encode_rows_t<9, true>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1); break;
        case 19: encode_rows_t<9, false>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1); break;
        case 16: encode_rows_t<8, true>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1); break;
        case 17: encode_rows_t<8, false>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1); break;
        case 20: encode_rows_t<10, true>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1); break;
        case 21: encode_rows_t<10, false>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1); break;
        default: break;
    }
}

}  // namespace lig
