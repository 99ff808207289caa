// ntt_generic.hip -- size-generic radix-2 NTT for gfx950: any power-of-two N, any batch of rows.
// Serves the executor calls that are NOT on the per-row hot loop: ntt_{forward,inverse}_{k,2k,n},
// the two degree-<2k mask rows (lig_encode_2k) and the three decodes per proof (lig_decode).
// Replaces ntt_bit_reverse / ntt_forward_radix2 / ntt_inverse_radix2 / ntt_adjust_inverse_reduce / ntt_fold
// (shader/kernels.wgsl.in:58-262) and their drivers (src/webgpu/engine.cpp:844-968).  The hot per-row
// encode lives in ntt_encode.hip.
#include "kernels.hpp"

namespace lig {

// in-place bit reversal permutation of N elements per row (kernels.wgsl.in:58-74)
__global__ void k_bitrev(fr* __restrict__ buf, uint32_t N, uint32_t bits, size_t row_stride) {
    fr* row = buf + (size_t)blockIdx.y * row_stride;
    for (uint32_t id = blockIdx.x * blockDim.x + threadIdx.x; id < N; id += gridDim.x * blockDim.x) {
        uint32_t rev = __brev(id) >> (32 - bits);
        if (id < rev) {
            fr a = fr_load(row + id), b = fr_load(row + rev);
            fr_store(row + id, b);
            fr_store(row + rev, a);
        }
    }
}

// one DIF stage of span M: (x, y) -> (x + y, (x - y) * w^(idx * N/M))      (kernels.wgsl.in:125-153)
__global__ void k_dif_stage(fr* __restrict__ buf, const fr* __restrict__ tw, uint32_t N, uint32_t M,
                            uint32_t tw_stride, size_t row_stride) {
    fr* row = buf + (size_t)blockIdx.y * row_stride;
    const uint32_t M2 = M >> 1;
    for (uint32_t inst = blockIdx.x * blockDim.x + threadIdx.x; inst < (N >> 1); inst += gridDim.x * blockDim.x) {
        const uint32_t index = inst & (M2 - 1), k = (inst / M2) * M + index;
        fr x = fr_load(row + k), y = fr_load(row + k + M2);
        fr_store(row + k, fr_add(x, y));
        fr d = fr_sub(x, y);
        if (M2 == 1) fr_store(row + k + M2, d);                 // w^0 = 1: the reference's last shared stage
        else fr_store(row + k + M2, fr_montmul(d, fr_load(tw + (size_t)index * tw_stride)));
    }
}

// one DIT stage of span M: (x, y) -> (x + w*y, x - w*y)                    (kernels.wgsl.in:230-262)
__global__ void k_dit_stage(fr* __restrict__ buf, const fr* __restrict__ tw, uint32_t N, uint32_t M,
                            uint32_t tw_stride, size_t row_stride) {
    fr* row = buf + (size_t)blockIdx.y * row_stride;
    const uint32_t M2 = M >> 1;
    for (uint32_t inst = blockIdx.x * blockDim.x + threadIdx.x; inst < (N >> 1); inst += gridDim.x * blockDim.x) {
        const uint32_t index = inst & (M2 - 1), k = (inst / M2) * M + index;
        fr x = fr_load(row + k), y = fr_load(row + k + M2);
        if (M2 != 1) y = fr_montmul(y, fr_load(tw + (size_t)index * tw_stride));
        fr_store(row + k, fr_add(x, y));
        fr_store(row + k + M2, fr_sub(x, y));
    }
}

// x *= N^-1 (Montgomery-form constant)                                   (kernels.wgsl.in:93-103)
__global__ void k_scale(fr* __restrict__ buf, fr ninv, uint32_t N, size_t row_stride) {
    fr* row = buf + (size_t)blockIdx.y * row_stride;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x)
        fr_store(row + i, fr_montmul(fr_load(row + i), ninv));
}

// buf[i] += buf[i + half], i < half                                       (kernels.wgsl.in:105-116)
__global__ void k_fold(fr* __restrict__ buf, uint32_t half, size_t row_stride) {
    fr* row = buf + (size_t)blockIdx.y * row_stride;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < half; i += gridDim.x * blockDim.x)
        fr_store(row + i, fr_add(fr_load(row + i), fr_load(row + i + half)));
}

static inline dim3 grid_for(uint32_t work, size_t rows) {
    uint32_t gx = (work + 255) / 256;
    if (gx > 1024) gx = 1024;
    if (gx == 0) gx = 1;
    return dim3(gx, (uint32_t)rows, 1);
}

void ntt_generic_forward(hipStream_t s, const NttPlan& pl, fr* buf, size_t rows, size_t row_stride) {
    for (uint32_t iter = pl.log2N; iter >= 1; iter--) {
        const uint32_t M = 1u << iter;
        hipLaunchKernelGGL(k_dif_stage, grid_for(pl.N / 2, rows), dim3(256), 0, s, buf, pl.w, pl.N, M, pl.N / M, row_stride);
    }
    hipLaunchKernelGGL(k_bitrev, grid_for(pl.N, rows), dim3(256), 0, s, buf, pl.N, pl.log2N, row_stride);
}

void ntt_generic_inverse(hipStream_t s, const NttPlan& pl, fr* buf, size_t rows, size_t row_stride) {
    hipLaunchKernelGGL(k_bitrev, grid_for(pl.N, rows), dim3(256), 0, s, buf, pl.N, pl.log2N, row_stride);
    for (uint32_t iter = 1; iter <= pl.log2N; iter++) {
        const uint32_t M = 1u << iter;
        hipLaunchKernelGGL(k_dit_stage, grid_for(pl.N / 2, rows), dim3(256), 0, s, buf, pl.winv, pl.N, M, pl.N / M, row_stride);
    }
    hipLaunchKernelGGL(k_scale, grid_for(pl.N, rows), dim3(256), 0, s, buf, pl.ninv, pl.N, row_stride);
}

void ntt_generic_fold(hipStream_t s, fr* buf, uint32_t half, size_t rows, size_t row_stride) {
    hipLaunchKernelGGL(k_fold, grid_for(half, rows), dim3(256), 0, s, buf, half, row_stride);
}

}  // namespace lig
