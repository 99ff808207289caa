// ntt_encode.hip -- the hot loop: Reed-Solomon encode of a BATCH of rows on gfx950.
//
// Replaces encode_ntt_device (src/webgpu/engine.cpp:755-770: 15 dispatches of <=64 workgroups per row,
// radix-2 stages through global memory, bit-reversal passes) with three launches per row batch.
//
// Math.  codeword[j] = P(w_n^j), j < n = 4k, P = the degree-<k interpolant of the message on
// the w_k domain.  With c = INTT_k(msg), psi = w_n^4 (order k) and j = 4q + r:
//     codeword[4q + r] = sum_i (c[i] * w_n^(r*i)) * psi^(i*q)                       (4 coset NTTs of size k;
// the 3k zero-padded coefficients of the reference's size-n transform are never touched).  Each size-k
// transform is split k = A*B (A = 8, B = k/8) so that every global access is a contiguous >= 512-byte run
// per wave and the size-B part lives in LDS:
//
//   K1  encode_in : thread = (row, i2).  Radix-8 butterfly across the 8 strided elements msg[B*i1 + i2],
//                   seam twiddle w_k^(-i2*j1), writes Y[j1][i2].                     (global -> regs -> global)
//   K2  encode_mid: workgroup = (row, j1).  Size-B inverse transform of Y[j1][.] in LDS gives the strided
//                   coefficients c[j1 + 8*i2]; kept in registers, they are twisted by k^-1 * w_n^(r*i) and pushed
//                   through a size-B forward transform once per coset r, times the seam twiddle psi^(j1*q2),
//                   written to Z[r][j1][q2].                                         (87% of all multiplies)
//   K3  encode_out: thread = (row, q2, r).  Radix-8 butterfly across Z[r][0..8)[q2], canonical reduction,
//                   codeword[4*(q2 + B*q1) + r]: 4 adjacent lanes = 4 cosets = 128 contiguous bytes, a wave
//                   writes 2 KiB runs in natural order (no bit-reversal pass).
//
// Values stay lazily in [0,2p) between butterflies (Montgomery products of a < 4p by a canonical twiddle
// are < 2p); only K3 produces canonical residues, which is all the reference guarantees too.
#include "kernels.hpp"

namespace lig {

// a, b in [0,2p): s = a + b mod 2p-lazy ([0,2p)), d = a - b + 2p in (0,4p)
__device__ __forceinline__ fr lazy_add(const fr& a, const fr& b) {
    fr s, t, r;
    add256(s, a, b);                         // < 4p < 2^256
    fr p2 = fr_const(FR_2P);
    uint32_t borrow = sub256(t, s, p2);
    select256(r, borrow == 0, s, t);
    return r;
}
__device__ __forceinline__ fr lazy_sub(const fr& a, const fr& b) {   // result in (0,4p), input of a Montgomery product
    fr t, r;
    fr p2 = fr_const(FR_2P);
    add256(t, a, p2);
    sub256(r, t, b);
    return r;
}
__device__ __forceinline__ fr lazy_sub_red(const fr& a, const fr& b) {  // a - b mod 2p-lazy, result [0,2p)
    fr d, e, r;
    uint32_t borrow = sub256(d, a, b);
    fr p2 = fr_const(FR_2P);
    add256(e, d, p2);
    select256(r, borrow != 0, d, e);
    return r;
}
__device__ __forceinline__ fr canon(const fr& a) {   // [0,2p) -> [0,p)
    return fr_reduce_once(a);
}

// radix-8 DIF butterfly in registers: a[p] <- sum_i a[i] * w^(i * brev3(p)); w1,w2,w3 = w, w^2, w^3 (w of order 8,
// Montgomery form).  5 Montgomery products.  Inputs/outputs lazy [0,2p).
__device__ __forceinline__ void radix8_dif(fr (&a)[8], const fr& w1, const fr& w2, const fr& w3) {
    fr t;
    // stage M = 8
    t = lazy_sub(a[0], a[4]); a[0] = lazy_add(a[0], a[4]); a[4] = lazy_add(t, fr_zero());            // (a0-a4)*1: fold (0,4p) -> [0,2p)
    t = lazy_sub(a[1], a[5]); a[1] = lazy_add(a[1], a[5]); a[5] = fr_montmul_lazy(t, w1);
    t = lazy_sub(a[2], a[6]); a[2] = lazy_add(a[2], a[6]); a[6] = fr_montmul_lazy(t, w2);
    t = lazy_sub(a[3], a[7]); a[3] = lazy_add(a[3], a[7]); a[7] = fr_montmul_lazy(t, w3);
    // stage M = 4 (two halves)
    t = lazy_sub_red(a[0], a[2]); a[0] = lazy_add(a[0], a[2]); a[2] = t;
    t = lazy_sub(a[1], a[3]);     a[1] = lazy_add(a[1], a[3]); a[3] = fr_montmul_lazy(t, w2);
    t = lazy_sub_red(a[4], a[6]); a[4] = lazy_add(a[4], a[6]); a[6] = t;
    t = lazy_sub(a[5], a[7]);     a[5] = lazy_add(a[5], a[7]); a[7] = fr_montmul_lazy(t, w2);
    // stage M = 2
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        t = lazy_sub_red(a[i], a[i + 1]);
        a[i] = lazy_add(a[i], a[i + 1]);
        a[i + 1] = t;
    }
}
__device__ __forceinline__ constexpr int brev3(int p) { return ((p & 1) << 2) | (p & 2) | ((p >> 2) & 1); }

// ---------------------------------------------------------------------------------------------------- K1
template <int LOG2B>
__global__ void __launch_bounds__(256, 4) k_encode_in(const fr* __restrict__ msgs, fr* __restrict__ Y, const fr* __restrict__ seam_inv,
                                                   const fr* __restrict__ w8, size_t rows) {
    constexpr uint32_t B = 1u << LOG2B, K = 8u * B;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t row = gid >> LOG2B;
    const uint32_t i2 = (uint32_t)gid & (B - 1);
    if (row >= rows) return;
    const fr* m = msgs + row * K;
    fr a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = fr_load(m + (size_t)i * B + i2);       // canonical inputs
    const fr w1 = fr_load(w8 + 1), w2 = fr_load(w8 + 2), w3 = fr_load(w8 + 3);
    radix8_dif(a, w1, w2, w3);
    fr* y = Y + row * K;
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const int j1 = brev3(p);
        fr v = a[p];
        if (j1 != 0) v = fr_montmul_lazy(v, fr_load(seam_inv + (size_t)j1 * B + i2));
        fr_store(y + (size_t)j1 * B + i2, v);
    }
}

// ---------------------------------------------------------------------------------------------------- K2
// LDS exchange: element `pos` of the tile lives as two 16-byte halves in two planes.
template <int LOG2B>
struct TileLds {
    static constexpr uint32_t B = 1u << LOG2B;
    uint4 lo[B];
    uint4 hi[B];
};
__device__ __forceinline__ uint32_t swz(uint32_t pos) { return pos ^ ((pos >> 5) & 7u); }

template <int LOG2B>
__device__ __forceinline__ void lds_put(TileLds<LOG2B>& L, uint32_t pos, const fr& x) {
    const uint32_t q = swz(pos);
    L.lo[q] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    L.hi[q] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
}
template <int LOG2B>
__device__ __forceinline__ fr lds_get(const TileLds<LOG2B>& L, uint32_t pos) {
    const uint32_t q = swz(pos);
    const uint4 a = L.lo[q], b = L.hi[q];
    fr r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}

// Size-B DIF transform of the tile, radix-2^2 steps in registers, LDS exchange between steps.
// Entry/exit: thread t holds positions t + q*(B/4), q = 0..3, natural order, values lazy [0,2p).
// tw = per-stage twiddles, stage of span M at offset B - M (M/2 entries rho^(idx*B/M)), Montgomery form.
// one radix-2^2 step (two DIF stages, spans M and M/2) of the size-B tile transform
template <int LOG2B, int S>
__device__ __forceinline__ void tile_step(fr (&x)[4], const fr* __restrict__ tw, TileLds<LOG2B>& L, const uint32_t t) {
    constexpr uint32_t B = 1u << LOG2B;
    constexpr int STEPS = LOG2B / 2;
    constexpr uint32_t M = B >> (2 * S), Q = M >> 2;
    const uint32_t b = t / Q, p = t % Q;            // Q is a power of two: shifts/masks
    const uint32_t base = b * M + p;
    if constexpr (S > 0) {   // fetch this step's operands
#pragma unroll
        for (int q = 0; q < 4; q++) x[q] = lds_get(L, base + q * Q);
        __syncthreads();
    }
    fr d0 = lazy_sub(x[0], x[2]), d1 = lazy_sub(x[1], x[3]);
    fr b0 = lazy_add(x[0], x[2]), b1 = lazy_add(x[1], x[3]);
    fr b2, b3;
    if constexpr (Q > 1) {
        b2 = fr_montmul_lazy(d0, fr_load(tw + (B - M) + p));
        b3 = fr_montmul_lazy(d1, fr_load(tw + (B - M) + p + Q));
    } else {
        b2 = lazy_add(d0, fr_zero());                                   // W_4^0 = 1
        b3 = fr_montmul_lazy(d1, fr_load(tw + (B - 4) + 1));            // W_4^1
    }
    fr e0 = lazy_sub(b0, b1), e1 = lazy_sub(b2, b3);
    x[0] = lazy_add(b0, b1);
    x[2] = lazy_add(b2, b3);
    if constexpr (Q > 1) {
        const fr w = fr_load(tw + (B - M / 2) + p);
        x[1] = fr_montmul_lazy(e0, w);
        x[3] = fr_montmul_lazy(e1, w);
    } else {
        x[1] = lazy_add(e0, fr_zero());
        x[3] = lazy_add(e1, fr_zero());
    }
    // publish: after the last step scatter to bit-reversed positions so that the tile is in natural order
    if constexpr (S + 1 < STEPS) {
#pragma unroll
        for (int q = 0; q < 4; q++) lds_put(L, base + q * Q, x[q]);
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) lds_put(L, __brev(base + q) >> (32 - LOG2B), x[q]);
    }
    __syncthreads();
    if constexpr (S + 1 < STEPS) tile_step<LOG2B, S + 1>(x, tw, L, t);
}

// Size-B DIF transform of the tile, radix-2^2 steps in registers, LDS exchange between steps.
// Entry/exit: thread t holds positions t + q*(B/4), q = 0..3, natural order, values lazy [0,2p).
// tw = per-stage twiddles, stage of span M at offset B - M (M/2 entries rho^(idx*B/M)), Montgomery form.
template <int LOG2B>
__device__ __forceinline__ void tile_dft(fr (&x)[4], const fr* __restrict__ tw, TileLds<LOG2B>& L, const uint32_t t) {
    constexpr uint32_t B = 1u << LOG2B;
    static_assert(LOG2B % 2 == 0, "tile length must be a power of 4");
    tile_step<LOG2B, 0>(x, tw, L, t);
    // natural order back into the entry ownership pattern
#pragma unroll
    for (int q = 0; q < 4; q++) x[q] = lds_get(L, t + q * (B / 4));
    __syncthreads();
}

template <int LOG2B>
__global__ void __launch_bounds__((1 << LOG2B) / 4, 4) k_encode_mid(const fr* __restrict__ Y, fr* __restrict__ Z,
                                                                 const fr* __restrict__ tw_inv, const fr* __restrict__ tw_fwd,
                                                                 const fr* __restrict__ twist, const fr* __restrict__ seam_fwd) {
    constexpr uint32_t B = 1u << LOG2B, K = 8u * B, T = B / 4;
    __shared__ TileLds<LOG2B> L;
    const uint32_t t = threadIdx.x;
    const uint32_t j1 = blockIdx.x & 7u;
    const size_t row = blockIdx.x >> 3;
    const fr* y = Y + row * K + (size_t)j1 * B;
    fr c[4], x[4];
#pragma unroll
    for (int q = 0; q < 4; q++) c[q] = fr_load(y + t + q * T);
    tile_dft<LOG2B>(c, tw_inv, L, t);                   // c[q] = k * coefficient[j1 + 8*(t + q*T)]  (lazy)
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
        const fr* tws = twist + ((size_t)r * 8 + j1) * B;   // k^-1 * w_n^(r*(j1 + 8*i2)), layout [r][j1][i2]
#pragma unroll
        for (int q = 0; q < 4; q++) x[q] = fr_montmul_lazy(c[q], fr_load(tws + t + q * T));
        tile_dft<LOG2B>(x, tw_fwd, L, t);
        fr* z = Z + ((row * 4 + r) * 8 + j1) * (size_t)B;
        if (j1 != 0) {
            const fr* sf = seam_fwd + (size_t)j1 * B;
#pragma unroll
            for (int q = 0; q < 4; q++) x[q] = fr_montmul_lazy(x[q], fr_load(sf + t + q * T));
        }
#pragma unroll
        for (int q = 0; q < 4; q++) fr_store(z + t + q * T, x[q]);
    }
}

// ---------------------------------------------------------------------------------------------------- K3
template <int LOG2B>
__global__ void __launch_bounds__(256, 4) k_encode_out(const fr* __restrict__ Z, fr* __restrict__ cw, const fr* __restrict__ w8, size_t rows) {
    constexpr uint32_t B = 1u << LOG2B, K = 8u * B;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t row = gid >> (LOG2B + 2);
    if (row >= rows) return;
    const uint32_t r = (uint32_t)gid & 3u;
    const uint32_t q2 = ((uint32_t)gid >> 2) & (B - 1);
    const fr* z = Z + ((row * 4 + r) * 8) * (size_t)B + q2;
    fr a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = fr_load(z + (size_t)i * B);
    const fr w1 = fr_load(w8 + 1), w2 = fr_load(w8 + 2), w3 = fr_load(w8 + 3);
    radix8_dif(a, w1, w2, w3);
    fr* out = cw + row * (4 * (size_t)K);
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const int q1 = brev3(p);
        fr_store(out + 4 * ((size_t)q2 + (size_t)B * q1) + r, canon(a[p]));
    }
}

bool encode_fast_supported(uint32_t k) { return k == 512 || k == 2048 || k == 8192; }

template <int LOG2B>
static void encode_rows_t(hipStream_t s, const EncodePlan& ep, const fr* msgs, fr* cw, fr* Y, fr* Z, size_t rows,
                          hipEvent_t ev0, hipEvent_t ev1) {
    constexpr uint32_t B = 1u << LOG2B;
    const size_t th1 = rows * B;
    hipLaunchKernelGGL(k_encode_in<LOG2B>, dim3((uint32_t)((th1 + 255) / 256)), dim3(256), 0, s, msgs, Y, ep.seam_inv, ep.w8_inv, rows);
    if (ev0) (void)hipEventRecord(ev0, s);
    hipLaunchKernelGGL(k_encode_mid<LOG2B>, dim3((uint32_t)(rows * 8)), dim3(B / 4), 0, s, Y, Z, ep.tw_b_inv, ep.tw_b, ep.twist, ep.seam_fwd);
    if (ev1) (void)hipEventRecord(ev1, s);
    const size_t th3 = rows * B * 4;
    hipLaunchKernelGGL(k_encode_out<LOG2B>, dim3((uint32_t)((th3 + 255) / 256)), dim3(256), 0, s, Z, cw, ep.w8_fwd, rows);
}

void encode_rows_fast(hipStream_t s, const EncodePlan& ep, const fr* msgs, fr* codewords, fr* scratch_y, fr* scratch_z, size_t rows,
                      hipEvent_t ev0, hipEvent_t ev1) {
    switch (ep.log2B) {
        case 6: encode_rows_t<6>(s, ep, msgs, codewords, scratch_y, scratch_z, rows, ev0, ev1); break;
        case 8: encode_rows_t<8>(s, ep, msgs, codewords, scratch_y, scratch_z, rows, ev0, ev1); break;
        case 10: encode_rows_t<10>(s, ep, msgs, codewords, scratch_y, scratch_z, rows, ev0, ev1); break;
        default: break;
    }
}

}  // namespace lig
