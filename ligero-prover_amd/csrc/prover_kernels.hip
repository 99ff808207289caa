// prover_kernels.hip -- device side of the batched three-stage prover (prover.hip): row forming from the AES
// streams, stage-2 random-linear-combination accumulators on the 29-bit-limb core, witness/randomness inner
// products.  Counterparts in the reference: witness_manager::pad_encoding_random / process_masks
// (include/zkp/backend/witness_manager.hpp:271-336, host OpenSSL one element at a time), check_code /
// check_linear / check_quadratic (include/zkp/nonbatch_context.hpp:756-780: 2-9 full-vector launches per row).
#include "fr29.hpp"
#include "kernels.hpp"

namespace lig {

// ---------------------------------------------------------------------------------------------- stage 2
// Partial accumulators over one group of rows, one thread per position j < count:
//   code_part[g][j] = sum_r rc[r] * U[r][j*ues]        (rc given as rc*R' -> plain products)      if rc  != null
//   lin_part[g][j]  = sum_r U[r][j*ues] * Rn[r][j]                                                if Rn  != null
// U rows are urs elements apart and read with element stride ues (ues = 2 picks the codeword values on the
// order-2k subgroup <w_n^2>), Rn rows are rrs elements apart.  Products are added lazily (limbs renormalised every
// 6 terms, value < 1.2p * group <= 2^261 for group <= 128).
__global__ void __launch_bounds__(256) k_rlc_partial(const fr* __restrict__ U, size_t urs, uint32_t ues, const fr* __restrict__ Rn,
                                                     size_t rrs, size_t rows, uint32_t count, const f29s* __restrict__ rc,
                                                     uint32_t group_rows, fr* __restrict__ code_part, fr* __restrict__ lin_part, int accumulate) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    const size_t r0 = (size_t)blockIdx.y * group_rows;
    const size_t r1 = r0 + group_rows < rows ? r0 + group_rows : rows;
    f29 ac = f29_zero(), al = f29_zero();
    if (accumulate && rc != nullptr) ac = unpack29(fr_load(code_part + (size_t)blockIdx.y * count + j));     // running partial, < 2p
    int since = 0;
    for (size_t r = r0; r < r1; r++) {
        const f29 u = unpack29(fr_load(U + r * urs + (size_t)j * ues));
        if (rc != nullptr) ac = f29_add(ac, f29_montmul(u, f29_load_tab(rc + r)));
        if (Rn != nullptr) al = f29_add(al, f29_montmul(u, unpack29(fr_load(Rn + r * rrs + j))));
        if (++since == 6) { ac = f29_qnorm(ac); al = f29_qnorm(al); since = 0; }
    }
    if (rc != nullptr) fr_store(code_part + (size_t)blockIdx.y * count + j, pack29(f29_reduce_2p(ac)));
    if (Rn != nullptr) {
        f29 v = f29_montmul(f29_qnorm(al), f29_const_r2());                                                     // plain value, < 1.2p
        if (accumulate) v = f29_reduce_2p(f29_add(v, unpack29(fr_load(lin_part + (size_t)blockIdx.y * count + j))));
        fr_store(lin_part + (size_t)blockIdx.y * count + j, pack29(v));
    }
}

// acc[j] = (acc[j] + sum_g part[g][j]) mod p, canonical
__global__ void __launch_bounds__(256) k_rlc_combine(fr* __restrict__ acc, const fr* __restrict__ part, uint32_t groups, uint32_t n) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    f29 a = unpack29(fr_load(acc + j));
    for (uint32_t g = 0; g < groups; g++) {
        a = f29_add(a, unpack29(fr_load(part + (size_t)g * n + j)));     // each term < 2p, limbs normalised
        if ((g & 3) == 3) a = f29_qnorm(a);
    }
    fr_store(acc + j, pack29(f29_canon(f29_qnorm(a))));
}

// quad[j] += sum_t rq[t] * (U[x_t][j*ues] * U[y_t][j*ues] - U[z_t][j*ues]);  rq2 = rq*R'^2, rq1 = rq*R' (per triple)
// Src: where element j of a row lives.  StridedRows = rows of any stride / element stride; EvenOfView = the order-2k subgroup
// <w_n^2> (codeword elements 2j) of a CwView: even j from the message row, odd j from the coset-2 plane.
struct StridedRows { const fr* U; size_t urs; uint32_t ues; __device__ const fr* at(size_t row, uint32_t j) const { return U + row * urs + (size_t)j * ues; } };
struct EvenOfView { CwView v; __device__ const fr* at(size_t row, uint32_t j) const { return v.at(row, 2 * j); } };
// gridDim.y > 1: the terms are cut into gridDim.y groups of `per_group`, group g writes its sum (< 2p) to part[g * count + j] and
// k_rlc_combine adds the groups to quad -- a trace with a thousand triples is a thousand dependent iterations per thread on
// 64 workgroups otherwise (2.5 ms of a 23 ms proof with half of the constraints quadratic, profiles/r04_quad_mix.md)
template <class Src>
__global__ void __launch_bounds__(256) k_quad_rows(Src src, uint32_t count,
                                                   const uint32_t* __restrict__ triples, const f29s* __restrict__ rq2,
                                                   const f29s* __restrict__ rq1, size_t n_triples, fr* __restrict__ quad,
                                                   fr* __restrict__ part, size_t per_group) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    const bool grouped = part != nullptr;
    const size_t t0 = grouped ? (size_t)blockIdx.y * per_group : 0;
    const size_t t1 = grouped ? (t0 + per_group < n_triples ? t0 + per_group : n_triples) : n_triples;
    f29 a = grouped ? f29_zero() : unpack29(fr_load(quad + j));
    for (size_t t = t0; t < t1; t++) {
        const uint32_t yi = triples[3 * t + 1];
        const f29 x = unpack29(fr_load(src.at(triples[3 * t], j)));
        const f29 z = unpack29(fr_load(src.at(triples[3 * t + 2], j)));
        // y index 0xFFFFFFFF: the equality term rq * (x - z) of on_batch_equal (nonbatch_context.hpp:811-825), i.e. y = 1
        const f29 xy = yi == 0xFFFFFFFFu ? f29_montmul(x, f29_load_tab(rq1 + t))
                                         : f29_montmul(f29_montmul(x, unpack29(fr_load(src.at(yi, j)))), f29_load_tab(rq2 + t));      // x*y*rq  (< 1.2p)
        const f29 zq = f29_montmul(z, f29_load_tab(rq1 + t));                       // z*rq    (< 1.2p)
        a = f29_add(a, f29_add(xy, f29_sub_k2(f29_zero(), zq)));                    // + xy + (2p - zq)
        a = f29_reduce_2p(a);
    }
    if (grouped) fr_store(part + (size_t)blockIdx.y * count + j, pack29(a));        // normalised, < 2p: what k_rlc_combine adds up
    else fr_store(quad + j, pack29(f29_canon(a)));
}

// one group-partial pass + combine.  Either rc_dev (code-type: acc_code += sum rc*U) or Rn (linear-type:
// acc_lin += sum U*Rn) or both.
void launch_rlc_rows29(hipStream_t s, const fr* U, size_t urs, uint32_t ues, const fr* Rn, size_t rrs, size_t rows, uint32_t count,
                       const f29s* rc_dev, fr* code, fr* lin, fr* part_code, fr* part_lin, uint32_t group_rows) {
    const uint32_t groups = (uint32_t)((rows + group_rows - 1) / group_rows);
    dim3 g((count + 255) / 256, groups);
    hipLaunchKernelGGL(k_rlc_partial, g, dim3(256), 0, s, U, urs, ues, Rn, rrs, rows, count, rc_dev, group_rows, part_code, part_lin, 0);
    if (rc_dev != nullptr) hipLaunchKernelGGL(k_rlc_combine, dim3((count + 255) / 256), dim3(256), 0, s, code, part_code, groups, count);
    if (Rn != nullptr) hipLaunchKernelGGL(k_rlc_combine, dim3((count + 255) / 256), dim3(256), 0, s, lin, part_lin, groups, count);
}
// partial pass only, ADDING to the group partials already in part_code / part_lin (zeroed by the caller before the first
// chunk): the prover combines once per proof instead of once per chunk.  Group g of every chunk adds to slot g.
void launch_rlc_accumulate29(hipStream_t s, const fr* U, size_t urs, uint32_t ues, const fr* Rn, size_t rrs, size_t rows, uint32_t count,
                             const f29s* rc_dev, fr* part_code, fr* part_lin, uint32_t group_rows) {
    const uint32_t groups = (uint32_t)((rows + group_rows - 1) / group_rows);
    dim3 g((count + 255) / 256, groups);
    hipLaunchKernelGGL(k_rlc_partial, g, dim3(256), 0, s, U, urs, ues, Rn, rrs, rows, count, rc_dev, group_rows, part_code, part_lin, 1);
}
// acc[j] = (acc[j] + sum_g part[g*count + j]) mod p  (used to add the all-gathered per-GPU partial accumulators)
void launch_rlc_combine(hipStream_t s, fr* acc, const fr* part, uint32_t groups, uint32_t count) {
    hipLaunchKernelGGL(k_rlc_combine, dim3((count + 255) / 256), dim3(256), 0, s, acc, part, groups, count);
}
// part (optional): scratch of part_elems elements for group partials; with it and enough terms the sum runs in up to 64 groups
template <class Src>
static void launch_quad_any(hipStream_t s, Src src, uint32_t count, const uint32_t* triples_dev, const f29s* rq2, const f29s* rq1, size_t n_triples,
                            fr* quad, fr* part, size_t part_elems) {
    if (!n_triples) return;
    size_t groups = part ? std::min<size_t>(std::min<size_t>(64, part_elems / count), (n_triples + 15) / 16) : 1;     // >= 16 terms per group
    if (groups < 2) {
        hipLaunchKernelGGL(k_quad_rows<Src>, dim3((count + 255) / 256), dim3(256), 0, s, src, count, triples_dev, rq2, rq1, n_triples, quad, (fr*)nullptr, (size_t)0);
        return;
    }
    const size_t per_group = (n_triples + groups - 1) / groups;
    groups = (n_triples + per_group - 1) / per_group;
    hipLaunchKernelGGL(k_quad_rows<Src>, dim3((count + 255) / 256, (uint32_t)groups), dim3(256), 0, s, src, count, triples_dev, rq2, rq1, n_triples, quad, part, per_group);
    hipLaunchKernelGGL(k_rlc_combine, dim3((count + 255) / 256), dim3(256), 0, s, quad, part, (uint32_t)groups, count);
}
void launch_quad_rows29(hipStream_t s, const fr* U, size_t urs, uint32_t ues, uint32_t count, const uint32_t* triples_dev,
                        const f29s* rq2, const f29s* rq1, size_t n_triples, fr* quad, fr* part, size_t part_elems) {
    launch_quad_any(s, StridedRows{U, urs, ues}, count, triples_dev, rq2, rq1, n_triples, quad, part, part_elems);
}
// the same sum over the 2k even codeword positions of a planar codeword matrix
void launch_quad_rows29_view(hipStream_t s, CwView cw, uint32_t count, const uint32_t* triples_dev, const f29s* rq2, const f29s* rq1,
                             size_t n_triples, fr* quad, fr* part, size_t part_elems) {
    launch_quad_any(s, EvenOfView{cw}, count, triples_dev, rq2, rq1, n_triples, quad, part, part_elems);
}

// ---------------------------------------------------------------------------------------------- linear-test constant
// out[0] = sum_{i < count} in[i * stride]  (canonical), one workgroup.  The constant of the linear test of the synthetic
// constraint stream is minus the sum of all inner products <witness row, randomness row> (the reference accumulates
// it on the host while generating constraints: linear_sums(), src/webgpu_prover.cpp:307).  Randomness rows are zero
// outside their data slots, so that double sum is just the sum of the message-domain accumulator sum_r msg_r o rand_r
// over its k positions -- no per-row inner products are needed.
__global__ void __launch_bounds__(256) k_sum_elems(const fr* __restrict__ in, uint32_t count, uint32_t stride, fr* __restrict__ out, fr* neg_out) {
    __shared__ uint32_t sh[256 * 9];
    f29 a = f29_zero();
    int since = 0;
    for (uint32_t i = threadIdx.x; i < count; i += 256) {
        a = f29_add(a, unpack29(fr_load(in + (size_t)i * stride)));
        if (++since == 6) { a = f29_qnorm(a); since = 0; }
    }
    a = f29_reduce_2p(f29_qnorm(a));
#pragma unroll
    for (int i = 0; i < 9; i++) sh[i * 256 + threadIdx.x] = a.v[i];
    __syncthreads();
    for (int half = 128; half >= 1; half >>= 1) {
        if ((int)threadIdx.x < half) {
            f29 x, y;
#pragma unroll
            for (int i = 0; i < 9; i++) { x.v[i] = sh[i * 256 + threadIdx.x]; y.v[i] = sh[i * 256 + threadIdx.x + half]; }
            x = f29_reduce_2p(f29_add(x, y));
#pragma unroll
            for (int i = 0; i < 9; i++) sh[i * 256 + threadIdx.x] = x.v[i];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        f29 x;
#pragma unroll
        for (int i = 0; i < 9; i++) x.v[i] = sh[i * 256];
        const f29 sum = f29_canon(x);
        fr_store(out, pack29(sum));
        // neg_out (optional, may lie inside the buffer `in` strides over -- hence no __restrict__): (p - sum) mod p, the
        // closing slot of the linear mask row (witness_manager.hpp:283-297)
        if (neg_out != nullptr) fr_store(neg_out, pack29(f29_canon(f29_sub_k2(f29_zero(), sum))));
    }
}
void launch_sum_elems(hipStream_t s, const fr* in, uint32_t count, uint32_t stride, fr* out, fr* neg_out) {
    hipLaunchKernelGGL(k_sum_elems, dim3(1), dim3(256), 0, s, in, count, stride, out, neg_out);
}

// The linear-test accumulator lives on the order-2k subgroup <w_n^2> (index m <-> w_n^(2m)).  Its even points w_n^(4q) =
// w_k^(-q) are message positions, where it is accumulated in message order (accH, straight from the message and
// randomness rows, no transform); its odd points are accumulated from codeword coset 2 (accC).  out[2q] = accH[(k - q) % k],
// out[2q + 1] = accC[q].
__global__ void k_lin_interleave(fr* __restrict__ out, const fr* __restrict__ accH, const fr* __restrict__ accC, uint32_t k) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= 2 * k) return;
    const uint32_t q = m >> 1;
    fr_store(out + m, (m & 1) ? fr_load(accC + q) : fr_load(accH + ((k - q) & (k - 1))));
}
void launch_lin_interleave(hipStream_t s, fr* out, const fr* accH, const fr* accC, uint32_t k) {
    hipLaunchKernelGGL(k_lin_interleave, dim3((2 * k + 255) / 256), dim3(256), 0, s, out, accH, accC, k);
}

// ---------------------------------------------------------------------------------------------- row upload
// dst (HBM) <- src (pinned host memory mapped into the device address space), 16-byte aligned, bytes % 16 == 0.
// 64 workgroups keep ~1 MiB of reads in flight, enough to cover the PCIe round trip; the kernel sits on 64 of the 256 CUs
// with one wave slot each and is issue-idle almost all the time.
__global__ void __launch_bounds__(256) k_copy_from_host(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {      // four independent loads per thread in flight
        const uint4 a = src[i], b = src[i + stride];
        const uint4 c2 = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c2; dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
void launch_copy_from_host(hipStream_t s, uint8_t* dst_dev, const uint8_t* src_mapped, size_t bytes) {
    if (!bytes) return;
    hipLaunchKernelGGL(k_copy_from_host, dim3(64), dim3(256), 0, s, (uint4*)dst_dev, (const uint4*)src_mapped, bytes / 16);
}

}  // namespace lig
