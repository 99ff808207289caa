// eltwise.hip -- element-wise field kernels, stage-2 random-linear-combination accumulators and the
// stage-3 column gather for gfx950.
// Replaces the 14 Eltwise* entry points, EltwisePowMod/PowAddMod and sample_gather of
// shader/kernels.wgsl.in:326-549.  All kernels are grid-stride over 32-byte elements with 2 x 16-byte
// accesses per lane (a wave touches 2 KiB contiguous per operand).
#include "kernels.hpp"
#include "fr29.hpp"
#include "../../include/lig_hip.h"

namespace lig {

// a^(p-2): x/0 = 0 like the reference's extended-Euclid bn254fr_invmod (shader/bn254fr.wgsl.in:128-153)
__device__ fr fr_inv_mont(const fr& a_mont) {
    // exponent p - 2, processed MSB first; a and result in Montgomery form
    fr acc = fr_const(FR_R);
    uint32_t e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = fr_p_limb(i);
    e[0] -= 2;
    for (int bit = 253; bit >= 0; bit--) {
        acc = fr_montmul(acc, acc);
        if ((e[bit >> 5] >> (bit & 31)) & 1) acc = fr_montmul(acc, a_mont);
    }
    return acc;
}

template <int OP>
__global__ void k_eltwise(const fr* __restrict__ x, const fr* __restrict__ y, fr* __restrict__ out, size_t count,
                          fr scalar, uint32_t bit) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        fr r;
        if constexpr (OP == LIG_OP_ADD) r = fr_add(fr_load(x + i), fr_load(y + i));
        else if constexpr (OP == LIG_OP_SUB) r = fr_sub(fr_load(x + i), fr_load(y + i));
        else if constexpr (OP == LIG_OP_ADD_ASSIGN) r = fr_add(fr_load(out + i), fr_load(x + i));
        else if constexpr (OP == LIG_OP_ADD_CONST) r = fr_add(fr_load(x + i), scalar);
        else if constexpr (OP == LIG_OP_SUB_CONST) r = fr_sub(fr_load(x + i), scalar);
        else if constexpr (OP == LIG_OP_CONST_SUB) r = fr_sub(scalar, fr_load(x + i));
        else if constexpr (OP == LIG_OP_MUL) r = fr_mul(fr_load(x + i), fr_load(y + i));
        else if constexpr (OP == LIG_OP_MUL_CONST) r = fr_montmul(fr_load(x + i), scalar);        // scalar pre-scaled by R on the host
        else if constexpr (OP == LIG_OP_MONTMUL_CONST) r = fr_montmul(fr_load(x + i), scalar);
        else if constexpr (OP == LIG_OP_FMA) r = fr_add(fr_load(out + i), fr_mul(fr_load(x + i), fr_load(y + i)));
        else if constexpr (OP == LIG_OP_FMA_CONST) r = fr_add(fr_load(out + i), fr_montmul(fr_load(x + i), scalar));  // scalar*R
        else if constexpr (OP == LIG_OP_DIV) {
            fr ym = fr_to_mont(fr_load(y + i));
            r = fr_montmul(fr_load(x + i), fr_inv_mont(ym));     // x * (y^-1 * R) / R
        } else {   // LIG_OP_BIT_DECOMPOSE (kernels.wgsl.in:502-510; bounds-checked here)
            fr v = fr_load(x + i);
            uint32_t limb = 0;
#pragma unroll
            for (int l = 0; l < 8; l++) limb = ((bit >> 5) == (uint32_t)l) ? v.v[l] : limb;
            r = fr_zero();
            r.v[0] = (bit < 256) ? ((limb >> (bit & 31)) & 1u) : 0u;
        }
        fr_store(out + i, r);
    }
}

// EltwiseDivMod with one inversion per M elements (Montgomery's trick) on the 29-bit-limb core: out = x / y, x / 0 = 0 like
// the reference's per-element extended Euclid (shader/bn254fr.wgsl.in:128-153, kernels.wgsl.in:453-466).  Thread t owns
// elements t, t + T, ..., t + (M-1)T (T = threads in the grid: coalesced), multiplies the M denominators together (zeros
// replaced by one), inverts the product once by Fermat (253 squarings + 109 products) and unwinds: 5M + 362 products per
// thread instead of 363 per element.
template <int M>
// (no __restrict__: the ABI lets x, y and out alias, and the unwind loop reads y[i] again right before it stores out[i])
__global__ void __launch_bounds__(256) k_div_batched(const fr* x, const fr* y, fr* out, size_t count) {
    const size_t T = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const f29 r2 = f29_const_r2();
    f29 one_m = f29_zero();
    one_m.v[0] = 1;
    one_m = f29_montmul(one_m, r2);                       // R' mod p: the Montgomery form of 1
    f29 pre[M];                                            // pre[j] = z_0 ... z_j  (Montgomery form)
    bool zero[M];
    f29 run = one_m;
    for (int j = 0; j < M; j++) {          // kept rolled: pre[] lives in scratch, 36 bytes per step
        const size_t i = t + (size_t)j * T;
        fr yv = fr_zero();
        if (i < count) yv = fr_load(y + i);
        zero[j] = !(yv.v[0] | yv.v[1] | yv.v[2] | yv.v[3] | yv.v[4] | yv.v[5] | yv.v[6] | yv.v[7]);
        const f29 z = zero[j] ? one_m : f29_montmul(unpack29(yv), r2);
        run = f29_montmul(run, z);
        pre[j] = run;
    }
    // run^(p-2), MSB first; bit 253 of p - 2 is set, so start from run itself
    f29 inv = run;
    for (int bit = 252; bit >= 0; bit--) {
        inv = f29_montmul(inv, inv);
        uint32_t limb = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) limb = ((bit >> 5) == w) ? fr_p_limb(w) - (w == 0 ? 2u : 0u) : limb;
        if ((limb >> (bit & 31)) & 1) inv = f29_montmul(inv, run);
    }
    for (int j = M - 1; j >= 0; j--) {
        const size_t i = t + (size_t)j * T;
        fr yv = fr_zero(), xv = fr_zero();
        if (i < count) { yv = fr_load(y + i); xv = fr_load(x + i); }
        const f29 z = zero[j] ? one_m : f29_montmul(unpack29(yv), r2);
        const f29 inv_j = j ? f29_montmul(inv, pre[j - 1]) : inv;      // (z_0..z_j)^-1 * (z_0..z_{j-1}) = z_j^-1
        inv = f29_montmul(inv, z);
        if (i < count) fr_store(out + i, zero[j] ? fr_zero() : pack29(f29_canon(f29_montmul(unpack29(xv), inv_j))));
    }
}

static inline uint32_t blocks_for(size_t count) {
    size_t b = (count + 255) / 256;
    if (b > 2048) b = 2048;
    if (b == 0) b = 1;
    return (uint32_t)b;
}

void launch_eltwise(hipStream_t s, int op, const fr* x, const fr* y, fr* out, size_t count, fr scalar, uint32_t bit) {
    dim3 g(blocks_for(count)), b(256);
    if (op == LIG_OP_DIV && count) {
        // one k-element row: the Fermat chain is the latency floor, so keep M small and the grid wide; big batches: M = 16
        if (count >= (1u << 20)) hipLaunchKernelGGL(k_div_batched<16>, dim3((uint32_t)((count + 16 * 256 - 1) / (16 * 256))), b, 0, s, x, y, out, count);
        else hipLaunchKernelGGL(k_div_batched<4>, dim3((uint32_t)((count + 4 * 256 - 1) / (4 * 256))), b, 0, s, x, y, out, count);
        return;
    }
#define LIG_CASE(OPC) case OPC: hipLaunchKernelGGL(k_eltwise<OPC>, g, b, 0, s, x, y, out, count, scalar, bit); break;
    switch (op) {
        LIG_CASE(LIG_OP_ADD) LIG_CASE(LIG_OP_SUB) LIG_CASE(LIG_OP_ADD_ASSIGN) LIG_CASE(LIG_OP_ADD_CONST)
        LIG_CASE(LIG_OP_SUB_CONST) LIG_CASE(LIG_OP_CONST_SUB) LIG_CASE(LIG_OP_MUL) LIG_CASE(LIG_OP_MUL_CONST)
        LIG_CASE(LIG_OP_MONTMUL_CONST) LIG_CASE(LIG_OP_FMA) LIG_CASE(LIG_OP_FMA_CONST) LIG_CASE(LIG_OP_DIV)
        LIG_CASE(LIG_OP_BIT_DECOMPOSE)
    }
#undef LIG_CASE
}

// EltwisePowMod / EltwisePowAddMod (kernels.wgsl.in:513-538, bn254fr_powmod bn254fr.wgsl.in:157-168):
// table[i] = base^(2^i) * R
__global__ void k_powmod(const fr* __restrict__ table, const uint32_t* __restrict__ exp, const fr* __restrict__ coeff,
                         fr* __restrict__ out, size_t count, int add) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t e = exp[i];
        fr acc = fr_const(FR_R);
        for (int b = 0; b < 32; b++)
            if ((e >> b) & 1u) acc = fr_montmul(acc, fr_load(table + b));
        fr r = fr_montmul(fr_load(coeff + i), acc);
        if (add) r = fr_add(r, fr_load(out + i));
        fr_store(out + i, r);
    }
}
void launch_powmod(hipStream_t s, const fr* table32, const uint32_t* exp, const fr* coeff, fr* out, size_t count, int add) {
    hipLaunchKernelGGL(k_powmod, dim3(blocks_for(count)), dim3(256), 0, s, table32, exp, coeff, out, count, add);
}

// stage 3: out[r][i] = cw[r][idx[i]]   (sample_gather, kernels.wgsl.in:541-549; one launch for a row batch)
__global__ void k_gather_rows(const fr* __restrict__ cw, size_t row_stride, size_t rows, const uint32_t* __restrict__ idx,
                              uint32_t count, fr* __restrict__ out) {
    const size_t total = rows * count;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t r = t / count;
        const uint32_t i = (uint32_t)(t - r * count);
        fr_store(out + t, fr_load(cw + r * row_stride + idx[i]));
    }
}
__global__ void k_gather_rows_planar(CwView cw, size_t rows, const uint32_t* __restrict__ idx, uint32_t count, fr* __restrict__ out) {
    const size_t total = rows * count;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t r = t / count;
        fr_store(out + t, fr_load(cw.at(r, idx[(uint32_t)(t - r * count)])));
    }
}
void launch_gather_rows_planar(hipStream_t s, CwView cw, size_t rows, const uint32_t* idx, uint32_t count, fr* out) {
    if (rows) hipLaunchKernelGGL(k_gather_rows_planar, dim3(blocks_for(rows * count)), dim3(256), 0, s, cw, rows, idx, count, out);
}
void launch_gather_rows(hipStream_t s, const fr* cw, size_t row_stride, size_t rows, const uint32_t* idx, uint32_t count, fr* out) {
    hipLaunchKernelGGL(k_gather_rows, dim3(blocks_for(rows * count)), dim3(256), 0, s, cw, row_stride, rows, idx, count, out);
}

// stage 2: one thread per codeword column j accumulates over the rows of the batch
//   code[j] += sum_r rc[r] * U[r][j];  lin[j] += sum_r U[r][j] * R[r][j];
//   quad[j] += sum_t rq[t] * (U[x_t][j] * U[y_t][j] - U[z_t][j])
// (check_code / check_linear / check_quadratic, include/zkp/nonbatch_context.hpp:756-780, which the reference
// runs as 2..9 full-vector launches per row).  rc/rq are Montgomery form (scaled on the host).
__global__ void k_rlc_rows(const fr* __restrict__ U, const fr* __restrict__ Rn, size_t rows, uint32_t n,
                           const fr* __restrict__ rc, fr* __restrict__ code, fr* __restrict__ lin,
                           const uint32_t* __restrict__ triples, const fr* __restrict__ rq, size_t n_triples,
                           fr* __restrict__ quad) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    if (code != nullptr) {
        fr acc = fr_load(code + j);
        for (size_t r = 0; r < rows; r++) acc = fr_add(acc, fr_montmul(fr_load(U + r * n + j), fr_load(rc + r)));
        fr_store(code + j, acc);
    }
    if (lin != nullptr && Rn != nullptr) {
        fr acc = fr_load(lin + j);
        for (size_t r = 0; r < rows; r++) acc = fr_add(acc, fr_mul(fr_load(U + r * n + j), fr_load(Rn + r * n + j)));
        fr_store(lin + j, acc);
    }
    if (quad != nullptr && n_triples) {
        fr acc = fr_load(quad + j);
        for (size_t t = 0; t < n_triples; t++) {
            fr x = fr_load(U + (size_t)triples[3 * t] * n + j), y = fr_load(U + (size_t)triples[3 * t + 1] * n + j);
            fr z = fr_load(U + (size_t)triples[3 * t + 2] * n + j);
            fr d = fr_sub(fr_mul(x, y), z);
            acc = fr_add(acc, fr_montmul(d, fr_load(rq + t)));
        }
        fr_store(quad + j, acc);
    }
}
void launch_rlc_rows(hipStream_t s, const fr* U, const fr* Rn, size_t rows, uint32_t n, const fr* rc_dev, fr* code,
                     fr* lin, const uint32_t* triples_dev, const fr* rq_dev, size_t n_triples, fr* quad) {
    hipLaunchKernelGGL(k_rlc_rows, dim3((n + 127) / 128), dim3(128), 0, s, U, Rn, rows, n, rc_dev, code, lin, triples_dev,
                       rq_dev, n_triples, quad);
}

}  // namespace lig
