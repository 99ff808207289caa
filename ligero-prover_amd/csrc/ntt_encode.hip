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
#include "tile_dft.hpp"
#include <cstdlib>

namespace lig {

// ---------------------------------------------------------------------------------------------------- K1
#ifndef LIG_K1_WAVES
#define LIG_K1_WAVES 2
#endif
template <int LOG2B>
__global__ void __launch_bounds__(256, LIG_K1_WAVES) k_encode_in(const fr* __restrict__ msgs, fr* __restrict__ Y, const f29s* __restrict__ seam_inv,
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
                                                    const fr* __restrict__ msgs, size_t rows, fr* __restrict__ coset2) {
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
    // optional compact copy of coset 2 (the odd points of <w_n^2>): the stage-2 linear test reads it k-contiguous instead of
    // fetching every fourth element of the codeword (4x the bytes it uses)
    if (FULL && coset2 != nullptr && r == 2) {
        fr* c2 = coset2 + row * (size_t)K;
#pragma unroll
        for (int q1 = 0; q1 < 8; q1++) fr_store(c2 + q2 + (size_t)B * q1, v[q1]);
    }
}

bool encode_fast_supported(uint32_t k) { return k == 512 || k == 1024 || k == 2048 || k == 4096 || k == 8192; }

template <int LOG2B, bool FULL>
static void encode_rows_t(hipStream_t s, const EncodePlan& ep, const fr* msgs, fr* cw, fr* Y, fr* Z, size_t rows,
                          hipEvent_t ev0, hipEvent_t ev1, fr* coset2) {
    constexpr uint32_t B = 1u << LOG2B;
    fr* Cc = Y + rows * (size_t)(8 * B);      // second half of the Y scratch (2 * rows * k elements)
    const size_t th1 = rows * B;
    static const int kmask = [] { const char* e = std::getenv("LIG_ENCODE_KMASK"); return e ? std::atoi(e) : 15; }();   // experiments only
    if (kmask & 1) hipLaunchKernelGGL(k_encode_in<LOG2B>, dim3((uint32_t)((th1 + 255) / 256)), dim3(256), 0, s, msgs, Y, ep.seam_inv, ep.w8_inv, rows);
    if (kmask & 2) hipLaunchKernelGGL(k_encode_coef<LOG2B>, dim3((uint32_t)(rows * 8)), dim3(B / 4), 0, s, Y, Cc, ep.tw_b_inv, ep.kinv);
    if (ev0) (void)hipEventRecord(ev0, s);
    // LIG_K2B_DYN_LDS (experiments only): extra dynamic LDS bytes per workgroup = fewer workgroups per CU (profiles/r02_occupancy_ab.md)
    static const uint32_t dyn_lds = [] { const char* e = std::getenv("LIG_K2B_DYN_LDS"); return e ? (uint32_t)std::atoi(e) : 0u; }();
    if (kmask & 4) hipLaunchKernelGGL((k_encode_mid<LOG2B, FULL>), dim3((uint32_t)(rows * 8 * (FULL ? 3 : 1))), dim3(B / 4), dyn_lds, s, Cc, Z, ep.tw_b, ep.twist, ep.seam_fwd);
    if (ev1) (void)hipEventRecord(ev1, s);
    const size_t th3 = rows * B * (FULL ? 4 : 1);
    if (kmask & 8) hipLaunchKernelGGL((k_encode_out<LOG2B, FULL>), dim3((uint32_t)((th3 + 255) / 256)), dim3(256), 0, s, Z, cw, ep.w8_fwd, msgs, rows, coset2);
}

// half = false: codewords (rows x n).  half = true: rows x k values on the coset w_n^2 <w_n^4>, out[q] = P(w_n^(4q + 2)).
void encode_rows_fast(hipStream_t s, const EncodePlan& ep, const fr* msgs, fr* out, fr* scratch_y, fr* scratch_z, size_t rows,
                      hipEvent_t ev0, hipEvent_t ev1, bool half, fr* coset2) {
    switch (ep.log2B * 2 + (half ? 1 : 0)) {
        case 12: encode_rows_t<6, true>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1, coset2); break;
        case 13: encode_rows_t<6, false>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1, coset2); break;
        case 14: encode_rows_t<7, true>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1, coset2); break;
        case 15: encode_rows_t<7, false>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1, coset2); break;
        case 18: encode_rows_t<9, true>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1, coset2); break;
        case 19: encode_rows_t<9, false>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1, coset2); break;
        case 16: encode_rows_t<8, true>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1, coset2); break;
        case 17: encode_rows_t<8, false>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1, coset2); break;
        case 20: encode_rows_t<10, true>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1, coset2); break;
        case 21: encode_rows_t<10, false>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1, coset2); break;
        default: break;
    }
}

}  // namespace lig
