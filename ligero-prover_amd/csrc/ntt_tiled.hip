// ntt_tiled.hip -- single-row transforms of any size N = 2^8 .. 2^20 in TWO launches, out of LDS.
//
// Serves everything that is one transform of one (or two, three) rows: ntt_{forward,inverse}_{k,2k,n}, the degree-<2k
// mask rows (ntt_inverse_2k + ntt_forward_n), the 2k -> n extension of the stage-2 accumulators, decode_ntt_device
// (INTT_n, fold, NTT_k).  The reference runs these as one dispatch per radix-2 stage through global memory plus
// bit-reversal passes (src/webgpu/engine.cpp:844-968: ~15 dispatches per transform); the generic radix-2 kernels of
// ntt_generic.hip mirror that and remain the fallback for N > 2^20.
//
// Four-step decomposition N = A * B (A = 2^floor(log2N / 2)), natural order in and out, no bit-reversal pass:
//     X[k1 + A*k2] = sum_{n2 < B} ( w^(n2*k1) * sum_{n1 < A} x[B*n1 + n2] * (w^B)^(n1*k1) ) * (w^A)^(n2*k2)
//   pass 1: workgroup = 1024/A columns n2; size-A transforms over the stride-B elements in LDS (tile_dft, bit-reversed
//           load), times the middle twiddle w^(n2*k1) (which also carries 1/N of an inverse transform) -> Y[k1][n2]
//   pass 2: workgroup = 1024/B rows k1 of Y; size-B transforms, exact canonical reduction -> X[k1 + A*k2]
// Runs of 1024/A (1024/B) consecutive 32-byte elements per global access; the data of one transform (<= 32 MiB) lives in L2.
#include "kernels.hpp"
#include "tile_dft.hpp"

namespace lig {

// FOLD: the input element is in[i] + in[i + fold] (decode_ntt_device folds coefficients k..2k-1 onto 0..k-1 before the
// size-k forward transform, shader/kernels.wgsl.in:105-116)
template <int LOG2A, bool FOLD>
__global__ void __launch_bounds__(256) k_tiled_pass1(const fr* __restrict__ in, size_t in_stride, fr* __restrict__ Y, size_t y_stride,
                                                     const f29s* __restrict__ tw_a, const f29s* __restrict__ mid, uint32_t log2B,
                                                     uint32_t fold) {
    constexpr uint32_t A = 1u << LOG2A, TA = A / 4;
    __shared__ TileLds<LOG2A> L[1024 >> LOG2A];
    const uint32_t tiles = blockDim.x / TA;
    const uint32_t tl = threadIdx.x / TA, t = threadIdx.x % TA;
    const uint32_t n2 = blockIdx.x * tiles + tl;
    const fr* x_in = in + (size_t)blockIdx.y * in_stride;
    f29 x[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t n1 = __brev(4 * t + q) >> (32 - LOG2A);
        const size_t idx = ((size_t)n1 << log2B) + n2;
        x[q] = unpack29(fr_load(x_in + idx));
        if (FOLD) x[q] = f29_qnorm(f29_add(x[q], unpack29(fr_load(x_in + idx + fold))));      // < 2p, normalised
    }
    tile_dft<LOG2A>(x, tw_a, L[tl], t);
    fr* y = Y + (size_t)blockIdx.y * y_stride;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t k1 = t + q * TA;
        const size_t o = ((size_t)k1 << log2B) + n2;
        fr_store(y + o, pack29(f29_montmul(x[q], f29_load_tab(mid + o))));                  // < 1.2p
    }
}

template <int LOG2B>
__global__ void __launch_bounds__(256) k_tiled_pass2(const fr* __restrict__ Y, size_t y_stride, fr* __restrict__ out, size_t out_stride,
                                                     const f29s* __restrict__ tw_b, uint32_t log2A) {
    constexpr uint32_t B = 1u << LOG2B, TB = B / 4;
    __shared__ TileLds<LOG2B> L[1024 >> LOG2B];
    const uint32_t tiles = blockDim.x / TB;
    const uint32_t tl = threadIdx.x / TB, t = threadIdx.x % TB;
    const uint32_t k1 = blockIdx.x * tiles + tl;
    const fr* y = Y + (size_t)blockIdx.y * y_stride + (size_t)k1 * B;
    f29 x[4];
#pragma unroll
    for (int q = 0; q < 4; q++) x[q] = unpack29(fr_load(y + (__brev(4 * t + q) >> (32 - LOG2B))));
    tile_dft<LOG2B>(x, tw_b, L[tl], t);
    fr* o = out + (size_t)blockIdx.y * out_stride;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t k2 = t + q * TB;
        fr_store(o + k1 + ((size_t)k2 << log2A), pack29(f29_canon(x[q])));
    }
}

bool tiled_supported(uint32_t log2N) { return log2N >= 8 && log2N <= 20; }

template <int LOG2A>
static void launch_pass1(hipStream_t s, const TiledPlan& tp, const fr* in, size_t in_stride, fr* Y, size_t rows, uint32_t fold) {
    const uint32_t A = 1u << LOG2A, B = 1u << tp.log2B;
    const uint32_t tiles = (1024 / A) < B ? (1024 / A) : B;
    const dim3 grid(B / tiles, (uint32_t)rows), block(tiles * (A / 4));
    if (fold) hipLaunchKernelGGL((k_tiled_pass1<LOG2A, true>), grid, block, 0, s, in, in_stride, Y, (size_t)tp.N, tp.tw_a, tp.mid, tp.log2B, fold);
    else hipLaunchKernelGGL((k_tiled_pass1<LOG2A, false>), grid, block, 0, s, in, in_stride, Y, (size_t)tp.N, tp.tw_a, tp.mid, tp.log2B, 0u);
}
template <int LOG2B>
static void launch_pass2(hipStream_t s, const TiledPlan& tp, const fr* Y, fr* out, size_t out_stride, size_t rows) {
    const uint32_t A = 1u << tp.log2A, B = 1u << LOG2B;
    const uint32_t tiles = (1024 / B) < A ? (1024 / B) : A;
    hipLaunchKernelGGL(k_tiled_pass2<LOG2B>, dim3(A / tiles, (uint32_t)rows), dim3(tiles * (B / 4)), 0, s, Y, (size_t)tp.N, out, out_stride,
                       tp.tw_b, tp.log2A);
}

// `rows` transforms of size tp.N: row r reads in + r*in_stride (first N elements, or with fold != 0 the sums
// in[i] + in[i + fold]) and writes out + r*out_stride; in == out is allowed (everything goes through `scratch`,
// rows * N elements).  Inputs canonical (or any value < 2p); outputs canonical.
void ntt_tiled(hipStream_t s, const TiledPlan& tp, const fr* in, size_t in_stride, fr* out, size_t out_stride, size_t rows, fr* scratch,
               uint32_t fold) {
    switch (tp.log2A) {
        case 4: launch_pass1<4>(s, tp, in, in_stride, scratch, rows, fold); break;
        case 5: launch_pass1<5>(s, tp, in, in_stride, scratch, rows, fold); break;
        case 6: launch_pass1<6>(s, tp, in, in_stride, scratch, rows, fold); break;
        case 7: launch_pass1<7>(s, tp, in, in_stride, scratch, rows, fold); break;
        case 8: launch_pass1<8>(s, tp, in, in_stride, scratch, rows, fold); break;
        case 9: launch_pass1<9>(s, tp, in, in_stride, scratch, rows, fold); break;
        default: launch_pass1<10>(s, tp, in, in_stride, scratch, rows, fold); break;
    }
    switch (tp.log2B) {
        case 4: launch_pass2<4>(s, tp, scratch, out, out_stride, rows); break;
        case 5: launch_pass2<5>(s, tp, scratch, out, out_stride, rows); break;
        case 6: launch_pass2<6>(s, tp, scratch, out, out_stride, rows); break;
        case 7: launch_pass2<7>(s, tp, scratch, out, out_stride, rows); break;
        case 8: launch_pass2<8>(s, tp, scratch, out, out_stride, rows); break;
        case 9: launch_pass2<9>(s, tp, scratch, out, out_stride, rows); break;
        default: launch_pass2<10>(s, tp, scratch, out, out_stride, rows); break;
    }
}

}  // namespace lig
