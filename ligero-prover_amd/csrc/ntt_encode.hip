// ntt_encode.hip -- the hot loop: Reed-Solomon encode of a BATCH of rows on gfx950.
//
// Replaces encode_ntt_device (src/webgpu/engine.cpp:755-770: 15 dispatches of <=64 workgroups per row,
// radix-2 stages through global memory, bit-reversal passes) with three launches per row batch.
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
//   K2  encode_tiles: workgroup = (row, j1).  Size-B inverse transform of Y[j1][.] in LDS -> the coefficients c[j1 + 8*j2] (the
//                    1/k factor rides on the twist tables), kept in registers; then per coset r: times w_n^(r*i) -> size-B
//                    forward transform in LDS -> times seam twiddle psi^(j1*q2) -> Z[r][j1][q2].   (90% of all multiplies)
//   K3  encode_out : thread = (row, q2, r).  Radix-8 butterfly across Z[r][0..8)[q2], exact canonical
//                    reduction, codeword[4*(q2 + B*q1) + r]: 4 adjacent lanes = 4 cosets = 128 contiguous bytes,
//                    a wave writes 2 KiB runs in natural order (no bit-reversal pass anywhere).
//
// All butterflies are decimation-in-time on 29-bit-limb lazy values (fr29.hpp): x' = x + w*y, y' = x - w*y + 2p,
// so magnitudes grow additively (<= 4p per radix-4 step) and no modular reduction is needed inside a transform;
// limbs are renormalised once per radix-4 step.  Y, Z hold values < 2^256 in the canonical 8 x u32 layout
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
__global__ void __launch_bounds__(256, LIG_K1_WAVES) k_encode_in(const fr* __restrict__ msgs, fr* __restrict__ Y, const f29wt seam_inv,
                                                   const f29wt w8, size_t rows) {
    constexpr uint32_t B = 1u << LOG2B, K = 8u * B;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t row = gid >> LOG2B;
    const uint32_t i2 = (uint32_t)gid & (B - 1);
    if (row >= rows) return;
    const fr* m = msgs + row * K;
    f29 a[8];
#pragma unroll
    for (int p = 0; p < 8; p++) a[p] = unpack29(fr_load(m + (size_t)brev3(p) * B + i2));      // canonical inputs
    radix8_dit(a, w8);
    fr* y = Y + row * K;
    fr_store(y + i2, pack29(f29_reduce_2p(a[0])));
    // explicit unrolling: hipcc leaves a `#pragma unroll` loop over 7 Montgomery products rolled and then keeps a[] in scratch
#define LIG_SEAM(J1) fr_store(y + (size_t)(J1) * B + i2, pack29(f29_mulw(a[J1], f29_load_w(seam_inv + (size_t)(J1) * B + i2))))
    LIG_SEAM(1); LIG_SEAM(2); LIG_SEAM(3); LIG_SEAM(4); LIG_SEAM(5); LIG_SEAM(6); LIG_SEAM(7);
#undef LIG_SEAM
}

// ---------------------------------------------------------------------------------------------------- K2
// workgroup = (row, j1): the coefficient tile C[j1][.] is produced (inverse tile transform of Y[j1][.]) and consumed (one
// forward tile transform per coset) without leaving the CU.  Between the two, the tile changes ownership once through
// LDS: the inverse transform leaves thread t with coefficients t + q*T, the forward transforms start from the bit-reversed
// positions brev(4t + q).  The C scratch matrix (k*32 bytes written + read per row) and one launch are gone; the price is
// 36 more live registers (the coefficients stay in VGPRs across the NC transforms): 134 VGPRs = 3 waves per SIMD instead of
// 4 (measured: forcing 128 registers, with 3 spilled dwords, is not faster; profiles/r02_fused_tiles_ab.md).
#ifndef LIG_K2_WAVES
#define LIG_K2_WAVES 3
#endif
// EXPERIMENT (-DLIG_K2_CB_GLOBAL, not the default; profiles/r05_tile_cb_global_ab.md): the coefficients between the inverse and the
// forward transforms go through the workgroup's own Y tile (dead once loaded; L2-resident) instead of staying in 36 registers
// across the coset loop -- the full kernel then needs the half kernel's registers and runs 4 waves per SIMD (-DLIG_K2_WAVES=4).
#ifdef LIG_K2_CB_GLOBAL
#define LIG_K2_Y_QUAL
#else
#define LIG_K2_Y_QUAL __restrict__
#endif
template <int LOG2B, bool FULL>
__global__ void __launch_bounds__((1 << LOG2B) / 4, FULL ? LIG_K2_WAVES : 4) k_encode_tiles(const fr* LIG_K2_Y_QUAL Y, fr* LIG_K2_Y_QUAL Z,
                                                                   const f29wt tw_inv, const f29wt tw_fwd, const f29wt twist, const f29wt seam_fwd) {
    constexpr uint32_t B = 1u << LOG2B, T = B / 4, NC = FULL ? 3 : 1;
#ifdef LIG_K2_CB_GLOBAL
    constexpr bool CBG = FULL;
#else
    constexpr bool CBG = false;
#endif
    __shared__ TileLds<LOG2B> L;
#ifdef LIG_K2_SETPRIO            // A/B (profiles/r04_tile_setprio_ab.md): tile waves ahead of the other streams' waves in the SIMD's arbiter
    __builtin_amdgcn_s_setprio(LIG_K2_SETPRIO);
#endif
    const uint32_t t = threadIdx.x;
    const uint32_t j1 = blockIdx.x & 7u;
    const size_t row = blockIdx.x >> 3;
    const fr* y = Y + (size_t)blockIdx.x * B;
    f29 x[4], cb[4];
    auto pos = [t](int q) { return __brev(4 * t + q) >> (32 - LOG2B); };
#pragma unroll
    for (int q = 0; q < 4; q++) x[q] = unpack29(fr_load(y + pos(q)));
    tile_dft<LOG2B>(x, tw_inv, L, t);
    fr* const cg = const_cast<fr*>(y);
    if constexpr (CBG) {
#pragma unroll
        for (int q = 0; q < 4; q++) fr_store(cg + t + q * T, pack29(f29_reduce_2p(x[q])));      // < 2p, 8 x u32
        __syncthreads();                               // the stores are visible to the workgroup; the exchange buffer is free
    } else {
        __syncthreads();                               // every wave is done reading the last exchange of the transform
#pragma unroll
        for (int q = 0; q < 4; q++) lds_put(L, t + q * T, f29_reduce_2p(x[q]));      // < 2p, normalised
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) cb[q] = lds_get(L, pos(q));
    }
#pragma unroll 1
    for (uint32_t ci = 0; ci < NC; ci++) {
        const uint32_t r = FULL ? ci + 1 : 2;
        const f29wt tws = twist + ((size_t)(r - 1) * 8 + j1) * B;
        // the thread index is made opaque once per coset: otherwise every t-dependent table / store address of the loop body is
        // hoisted into 64-bit register pairs that do not fit next to the coefficients and end up in scratch memory
        uint32_t tt = t;
        asm volatile("" : "+v"(tt));
        if constexpr (CBG) {
#pragma unroll
            for (int q = 0; q < 4; q++) x[q] = f29_mulw(unpack29(fr_load(cg + (__brev(4 * tt + q) >> (32 - LOG2B)))), f29_load_w(tws + (q * T + tt)));
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) x[q] = f29_mulw(cb[q], f29_load_w(tws + (q * T + tt)));   // w_n^(r*(j1 + 8*pos))
        }
        __syncthreads();                               // the exchange buffer is free again (ownership change / previous coset)
        tile_dft<LOG2B>(x, tw_fwd, L, tt);
        fr* z = Z + ((row * NC + ci) * 8 + j1) * (size_t)B;
        if (j1 != 0) {
            const f29wt sf = seam_fwd + (size_t)j1 * B;
#pragma unroll
            for (int q = 0; q < 4; q++) fr_store(z + tt + q * T, pack29(f29_mulw(x[q], f29_load_w(sf + tt + q * T))));
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) fr_store(z + tt + q * T, pack29(f29_reduce_2p(x[q])));
        }
    }
}

// ---------------------------------------------------------------------------------------------------- K3
// MODE 0 (interleaved): out row = the codeword as the reference lays it out, element 4*(q2 + B*q1) + r; the lanes with r = 0
//        copy the reversed message instead of running the radix-8 step, so every 128-byte line is written by four
//        adjacent lanes.
// MODE 1 (half): out row = k elements, element q = q2 + B*q1 is P(w_n^(4q + 2)).
// MODE 2 (planar): out row = 3k elements, the computed cosets as planes: element (r-1)*k + q is P(w_n^(4q + r)), r = 1, 2, 3.
//        Coset 0 is not stored at all (it IS the message row, reversed); the batched prover reads its columns from there.
//        Per row this writes 3k*32 bytes instead of 4k*32 and does not read the message.
template <int LOG2B, int MODE>
__global__ void __launch_bounds__(256) k_encode_out(const fr* __restrict__ Z, fr* __restrict__ cw, const f29wt w8,
                                                    const fr* __restrict__ msgs, size_t rows) {
    constexpr uint32_t B = 1u << LOG2B, K = 8u * B;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (MODE == 2) {
        const uint32_t q2 = (uint32_t)gid & (B - 1);
        const size_t rc = gid >> LOG2B;                 // row * 3 + coset slot
        if (rc >= rows * 3) return;
        const fr* z = Z + (rc * 8) * (size_t)B + q2;
        f29 a[8];
#pragma unroll
        for (int p = 0; p < 8; p++) a[p] = unpack29(fr_load(z + (size_t)brev3(p) * B));
        radix8_dit(a, w8);
        fr* out = cw + rc * (size_t)K + q2;
#pragma unroll
        for (int q1 = 0; q1 < 8; q1++) fr_store(out + (size_t)B * q1, pack29(f29_canon(a[q1])));
    } else {
        constexpr bool FULL = MODE == 0;
        constexpr int LOGNC = FULL ? 2 : 0;
        constexpr uint32_t NC = FULL ? 3 : 1, OS = FULL ? 4 : 1;
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
            radix8_dit(a, w8);
#pragma unroll
            for (int q1 = 0; q1 < 8; q1++) v[q1] = pack29(f29_canon(a[q1]));
        }
#pragma unroll
        for (int q1 = 0; q1 < 8; q1++) fr_store(out + OS * ((size_t)q2 + (size_t)B * q1) + r, v[q1]);
    }
}

// ---------------------------------------------------------------------------------------------------- K3 . codeword
// The stage-2 linear test needs, per randomness row R_r, its values on the coset w_n^2 <w_n^4> only to multiply them with the
// same coset of the codeword U_r and add the products over the rows.  This output kernel does that in place of writing the
// k values: thread = (q2, group of rows); per row the radix-8 butterfly across Z[0..8)[q2] (no canonical reduction: the
// lazy values go straight into the products), eight products with U_r's coset-2 plane, lazy accumulation; one partial sum
// per group and position, ADDED to part[g][.] (the partials persist across chunks, prover_kernels.hip: k_rlc_partial has
// the same contract).  Saves the k*32-byte write and re-read of the coset values per row and one launch.
template <int LOG2B>
__global__ void __launch_bounds__(256, 2) k_encode_out_dot(const fr* __restrict__ Z, const f29wt w8, const fr* __restrict__ cw2,
                                                           size_t cws, size_t rows, uint32_t group_rows, fr* __restrict__ part) {
    constexpr uint32_t B = 1u << LOG2B, K = 8u * B;
    const uint32_t q2 = blockIdx.x * blockDim.x + threadIdx.x;
    if (q2 >= B) return;
    const size_t r0 = (size_t)blockIdx.y * group_rows;
    const size_t r1 = r0 + group_rows < rows ? r0 + group_rows : rows;
    f29 acc[8];
#pragma unroll
    for (int q1 = 0; q1 < 8; q1++) acc[q1] = f29_zero();
    int since = 0;
    for (size_t r = r0; r < r1; r++) {
        const fr* z = Z + (r * 8) * (size_t)B + q2;
        const fr* u = cw2 + r * cws + q2;
        f29 a[8];
#pragma unroll
        for (int p = 0; p < 8; p++) a[p] = unpack29(fr_load(z + (size_t)brev3(p) * B));
        radix8_dit<false>(a, w8);                                                 // limbs < 2^31 + 8, value < 28p
#pragma unroll
        for (int q1 = 0; q1 < 8; q1++) acc[q1] = f29_add(acc[q1], f29_montmul(a[q1], unpack29(fr_load(u + (size_t)B * q1))));   // each term < 1.2p
        if (++since == 6) {
#pragma unroll
            for (int q1 = 0; q1 < 8; q1++) acc[q1] = f29_qnorm(acc[q1]);
            since = 0;
        }
    }
    fr* out = part + (size_t)blockIdx.y * K + q2;
#pragma unroll
    for (int q1 = 0; q1 < 8; q1++) {
        f29 v = f29_montmul(f29_qnorm(acc[q1]), f29_const_r2());          // plain value, < 1.2p
        v = f29_reduce_2p(f29_add(v, unpack29(fr_load(out + (size_t)B * q1))));
        fr_store(out + (size_t)B * q1, pack29(v));
    }
}

// The same with the codeword side given as Z tiles (ENC_ZRES: K3 was never run for the resident matrix): per row one more radix-8
// butterfly -- across cwz[r][0..8)[q2], the coset-2 tiles of U_r -- whose eight outputs wait in LDS (this thread's own 256 bytes; no
// synchronisation: written and read by the same thread) while the butterfly of the randomness row runs in the same registers.
template <int LOG2B>
__global__ void __launch_bounds__(256, 2) k_encode_out_dot_z(const fr* __restrict__ Z, const f29wt w8, const fr* __restrict__ cwz,
                                                             size_t cws, size_t rows, uint32_t group_rows, fr* __restrict__ part) {
    constexpr uint32_t B = 1u << LOG2B, K = 8u * B;
    __shared__ uint4 park[16][256];
    const uint32_t tid = threadIdx.x;
    const uint32_t q2 = blockIdx.x * blockDim.x + threadIdx.x;
    if (q2 >= B) return;
    const size_t r0 = (size_t)blockIdx.y * group_rows;
    const size_t r1 = r0 + group_rows < rows ? r0 + group_rows : rows;
    f29 acc[8];
#pragma unroll
    for (int q1 = 0; q1 < 8; q1++) acc[q1] = f29_zero();
    int since = 0;
    for (size_t r = r0; r < r1; r++) {
        const fr* z = Z + (r * 8) * (size_t)B + q2;
        const fr* uz = cwz + r * cws + q2;
        f29 a[8];
#pragma unroll
        for (int p = 0; p < 8; p++) a[p] = unpack29(fr_load(uz + (size_t)brev3(p) * B));
        radix8_dit<false>(a, w8);
#pragma unroll
        for (int q1 = 0; q1 < 8; q1++) {
            const fr c = pack29(f29_reduce_2p(a[q1]));                            // U_r on the coset, < 2p
            park[2 * q1][tid] = make_uint4(c.v[0], c.v[1], c.v[2], c.v[3]);
            park[2 * q1 + 1][tid] = make_uint4(c.v[4], c.v[5], c.v[6], c.v[7]);
        }
#pragma unroll
        for (int p = 0; p < 8; p++) a[p] = unpack29(fr_load(z + (size_t)brev3(p) * B));
        radix8_dit<false>(a, w8);                                                 // limbs < 2^31 + 8, value < 28p
#pragma unroll
        for (int q1 = 0; q1 < 8; q1++) {
            const uint4 lo = park[2 * q1][tid], hi = park[2 * q1 + 1][tid];
            fr c;
            c.v[0] = lo.x; c.v[1] = lo.y; c.v[2] = lo.z; c.v[3] = lo.w; c.v[4] = hi.x; c.v[5] = hi.y; c.v[6] = hi.z; c.v[7] = hi.w;
            acc[q1] = f29_add(acc[q1], f29_montmul(a[q1], unpack29(c)));           // each term < 28p * 2p / 2^261 + p < 1.4p
        }
        if (++since == 6) {
#pragma unroll
            for (int q1 = 0; q1 < 8; q1++) acc[q1] = f29_qnorm(acc[q1]);
            since = 0;
        }
    }
    fr* out = part + (size_t)blockIdx.y * K + q2;
#pragma unroll
    for (int q1 = 0; q1 < 8; q1++) {
        f29 v = f29_montmul(f29_qnorm(acc[q1]), f29_const_r2());
        v = f29_reduce_2p(f29_add(v, unpack29(fr_load(out + (size_t)B * q1))));
        fr_store(out + (size_t)B * q1, pack29(v));
    }
}

// Column gather over a resident matrix of Z tiles (ENC_ZRES): out[r*count + i] = codeword element idx[i] of row r.  Coset 0 from the
// message row (reversed); an element of cosets 1..3 is output q1 of the radix-8 butterfly across Z[r][coset-1][0..8)[q2], reduced
// exactly as K3 would have (canonical residues are unique: the same bytes as the planar matrix).
template <int LOG2B>
__global__ void __launch_bounds__(256) k_gather_rows_z(CwView cw, const f29wt w8, size_t rows, const uint32_t* __restrict__ idx, uint32_t count,
                                                       fr* __restrict__ out) {
    constexpr uint32_t B = 1u << LOG2B, K = 8u * B;
    const size_t total = rows * count;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t r = t / count;
        const uint32_t col = idx[(uint32_t)(t - r * count)];
        const uint32_t cs = col & 3u, q = col >> 2;
        if (cs == 0) { fr_store(out + t, fr_load(cw.msgs + r * K + ((K - q) & (K - 1)))); continue; }
        const uint32_t q2 = q & (B - 1), q1 = q >> LOG2B;
        const fr* z = cw.planes + r * 3 * (size_t)K + (size_t)(cs - 1) * K + q2;
        f29 a[8];
#pragma unroll
        for (int p = 0; p < 8; p++) a[p] = unpack29(fr_load(z + (size_t)brev3(p) * B));
        radix8_dit(a, w8);
        f29 v = a[0];
#pragma unroll
        for (int i = 1; i < 8; i++)
            if (q1 == (uint32_t)i) v = a[i];
        fr_store(out + t, pack29(f29_canon(v)));
    }
}

// tile length B = k/8 = 64 .. 4096: 36 bytes of LDS per element, i.e. 2.25 .. 144 KiB per workgroup of B/4 threads
bool encode_fast_supported(uint32_t k) { return k >= 512 && k <= 32768 && (k & (k - 1)) == 0; }

// mode: 0 = codewords rows x n (reference layout), 1 = rows x k (coset 2 only), 2 = rows x 3k (cosets 1..3 as planes),
// 3 = coset 2 only, not stored: its products with the rows of dot.cw2 are added to the group partials dot.part
template <int LOG2B>
static void encode_rows_t(hipStream_t s, const EncodePlan& ep, const fr* msgs, fr* cw, fr* Y, fr* Z, size_t rows,
                          hipEvent_t ev0, hipEvent_t ev1, int mode, const EncodeDot* dot, int phases) {
    constexpr uint32_t B = 1u << LOG2B;
    const size_t th1 = rows * B;
    const int kmask = lig::knobs().encode_kmask & phases;       // 1 = K1, 6 = K2, 8 = K3 (the knob: experiments only; phases: a caller that pipelines K1 ahead)
    // K1 / K3 have no LDS and no barrier: their workgroup size is free (LIG_K13_BLOCK, experiments)
    const uint32_t bs13 = lig::knobs().k13_block;
    if (kmask & 1) hipLaunchKernelGGL(k_encode_in<LOG2B>, dim3((uint32_t)((th1 + bs13 - 1) / bs13)), dim3(bs13), 0, s, msgs, Y, ep.seam_inv, ep.w8_inv, rows);
    if (ev0) (void)hipEventRecord(ev0, s);
    if (kmask & 6) {
        // LIG_K2_DYN_LDS (experiments): unused dynamic LDS per workgroup, to lower the tile kernel's workgroups per CU below what its
        // registers allow (3) and leave room for the waves of the other stream's kernels
        const uint32_t dyn = lig::knobs().k2_dyn_lds;
        if (mode == 1 || mode == 3) hipLaunchKernelGGL((k_encode_tiles<LOG2B, false>), dim3((uint32_t)(rows * 8)), dim3(B / 4), dyn, s, Y, Z, ep.tw_b_inv, ep.tw_b, ep.twist, ep.seam_fwd);
        else hipLaunchKernelGGL((k_encode_tiles<LOG2B, true>), dim3((uint32_t)(rows * 8)), dim3(B / 4), dyn, s, Y, mode == ENC_ZRES ? cw : Z, ep.tw_b_inv, ep.tw_b, ep.twist, ep.seam_fwd);
    }
    if (ev1) (void)hipEventRecord(ev1, s);
    if (mode == ENC_ZRES) return;                       // the tiles ARE the output: no last pass
    if (!(kmask & 8)) return;
    if (mode == 3) {
        const uint32_t groups = (uint32_t)((rows + dot->group_rows - 1) / dot->group_rows);
        if (dot->cw2_z) {
            hipLaunchKernelGGL(k_encode_out_dot_z<LOG2B>, dim3((B + 255) / 256, groups), dim3(B < 256 ? B : 256), 0, s, Z, ep.w8_fwd, dot->cw2, dot->cw2_stride, rows,
                               dot->group_rows, dot->part);
            return;
        }
        hipLaunchKernelGGL(k_encode_out_dot<LOG2B>, dim3((B + 255) / 256, groups), dim3(B < 256 ? B : 256), 0, s, Z, ep.w8_fwd, dot->cw2, dot->cw2_stride, rows,
                           dot->group_rows, dot->part);
        return;
    }
    const size_t th3 = rows * B * (mode == 0 ? 4 : mode == 1 ? 1 : 3);
    const dim3 g3((uint32_t)((th3 + bs13 - 1) / bs13));
    if (mode == 0) hipLaunchKernelGGL((k_encode_out<LOG2B, 0>), g3, dim3(bs13), 0, s, Z, cw, ep.w8_fwd, msgs, rows);
    else if (mode == 1) hipLaunchKernelGGL((k_encode_out<LOG2B, 1>), g3, dim3(bs13), 0, s, Z, cw, ep.w8_fwd, msgs, rows);
    else hipLaunchKernelGGL((k_encode_out<LOG2B, 2>), g3, dim3(bs13), 0, s, Z, cw, ep.w8_fwd, msgs, rows);
}

bool launch_gather_rows_z(hipStream_t s, const EncodePlan& ep, CwView cw, size_t rows, const uint32_t* idx, uint32_t count, fr* out) {
    if (!rows || !count) return true;
    const size_t total = rows * count;
    const dim3 grid((uint32_t)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096));
    switch (ep.log2B) {
#define LIG_GZ(L) case L: hipLaunchKernelGGL(k_gather_rows_z<L>, grid, dim3(256), 0, s, cw, ep.w8_fwd, rows, idx, count, out); return true;
        LIG_GZ(6) LIG_GZ(7) LIG_GZ(8) LIG_GZ(9) LIG_GZ(10) LIG_GZ(11) LIG_GZ(12)
#undef LIG_GZ
        default: return false;
    }
}

void encode_rows_fast(hipStream_t s, const EncodePlan& ep, const fr* msgs, fr* out, fr* scratch_y, fr* scratch_z, size_t rows,
                      hipEvent_t ev0, hipEvent_t ev1, int mode, const EncodeDot* dot, int phases) {
    switch (ep.log2B) {
        case 6: encode_rows_t<6>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1, mode, dot, phases); break;
        case 7: encode_rows_t<7>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1, mode, dot, phases); break;
        case 8: encode_rows_t<8>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1, mode, dot, phases); break;
        case 9: encode_rows_t<9>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1, mode, dot, phases); break;
        case 10: encode_rows_t<10>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1, mode, dot, phases); break;
        case 11: encode_rows_t<11>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1, mode, dot, phases); break;
        case 12: encode_rows_t<12>(s, ep, msgs, out, scratch_y, scratch_z, rows, ev0, ev1, mode, dot, phases); break;
        default: break;
    }
}

}  // namespace lig
