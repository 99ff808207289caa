// ubench_shoup.hip -- A/B of the two ways to multiply a lazy 9 x 29-bit-limb value by a FIXED table operand on gfx950:
//   montmul : Montgomery product a * (w R') / R'  (csrc/fr29.hpp: 81 + 81 chained v_mad_u64_u32, 9 v_mul_lo, 17 shifts)
//   shoup   : r = a*w - floor(a*w'/2^261) * p with w' = floor(w 2^261 / p) stored beside w
//             (45 mads for the low limbs of a*w, 53 for the quotient with one guard column, 45 for the low limbs of q*p,
//              a borrow-proof 9-limb subtraction with a carry pass)
// Both are checked against each other (same canonical residue) and timed as dependent chains, 4 waves per SIMD.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I ligero-prover_amd/csrc tools/ubench_shoup.hip -o tools/ubench_shoup
// constants: tools/ubench_shoup_consts.hpp (16 pseudo-random w; w R' mod p; floor(w 2^261 / p))
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "fr29.hpp"
#include "ubench_shoup_consts.hpp"

using namespace lig;

// plain-C++ Montgomery product (what hipcc makes of the column loop without the generated asm blocks)
__device__ __forceinline__ f29 montmul_plain(const f29& a, const f29& b) {
    uint32_t m[9];
    f29 t;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = 0; i < 9; i++) { const int j = k - i; if (j >= 0 && j < 9) acc = mad64(a.v[i], b.v[j], acc); }
#pragma unroll
        for (int i = 0; i < 9; i++) { const int j = k - i; if (j >= 1 && j < 9) acc = mad64(m[i], F29_P(j), acc); }      // m[i] with i < k: known
        if (k < 9) { m[k] = ((uint32_t)acc * F29_N0) & F29_MASK; acc = mad64(m[k], F29_P(0), acc); }
        if (k >= 9) t.v[k - 9] = (uint32_t)acc & F29_MASK;
        acc >>= 29;
    }
    t.v[8] = (uint32_t)acc;
    return t;
}

// Shoup-style product with the precomputed quotient operand wq = floor(w * 2^261 / p)
__device__ __forceinline__ f29 mulshoup(const f29& a, const f29& w, const f29& wq) {
    // quotient q = floor(a * wq / 2^261), from columns 7 (guard) .. 16
    uint32_t q[9];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 7; k < 17; k++) {
#pragma unroll
        for (int i = 0; i < 9; i++) { const int j = k - i; if (j >= 0 && j < 9) acc = mad64(a.v[i], wq.v[j], acc); }
        if (k >= 9) q[k - 9] = (uint32_t)acc & F29_MASK;
        acc >>= 29;
    }
    q[8] = (uint32_t)acc;
    // low nine limbs of a * w and of q * p
    uint32_t x[9], y[9];
    acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc = mad64(a.v[i], w.v[k - i], acc);
        x[k] = (uint32_t)acc & F29_MASK;
        acc >>= 29;
    }
    acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc = mad64(q[i], F29_P(k - i), acc);
        y[k] = (uint32_t)acc & F29_MASK;
        acc >>= 29;
    }
    // r = x - y mod 2^261 (the true value is < 3p < 2^256): limb-wise with the complement, one carry pass, top limb masked
    f29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = x[i] + (F29_MASK - y[i]) + (i == 0 ? 1u : 0u);
    r = f29_qnorm(r);
    r.v[8] &= F29_MASK;
    r = f29_qnorm(r);          // a carry out of the first pass may have left limb 7 at 2^29
    r.v[8] &= F29_MASK;
    return r;
}

__device__ __forceinline__ f29 tab(const uint32_t (*t)[9], int i) { f29 r; for (int j = 0; j < 9; j++) r.v[j] = t[i][j]; return r; }

__device__ uint32_t d_w[16][9], d_wm[16][9], d_wq[16][9];

template <int MODE>
__global__ void __launch_bounds__(256) k_chain(uint32_t* out, int iters) {
    f29 a = f29_zero();
    a.v[0] = threadIdx.x + 7 * blockIdx.x + 3; a.v[3] = 0x1234567u ^ threadIdx.x; a.v[8] = 0x00F0F0Fu;
    f29 w[4], wq[4];
    for (int i = 0; i < 4; i++) { w[i] = tab(MODE == 2 ? d_w : d_wm, (threadIdx.x + i) & 15); wq[i] = tab(d_wq, (threadIdx.x + i) & 15); }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (MODE == 0) a = f29_montmul(a, w[i]);
            if (MODE == 1) a = montmul_plain(a, w[i]);
            if (MODE == 2) a = mulshoup(a, w[i], wq[i]);
        }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 9; i++) acc ^= a.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// correctness: montmul(a, w R') and mulshoup(a, w, wq) have the same canonical residue, for lazy a
__global__ void k_check(uint32_t* bad) {
    const int t = threadIdx.x;
    f29 a = f29_zero();
    for (int i = 0; i < 9; i++) a.v[i] = (0x9E3779B9u * (t * 9 + i + 1)) & (i < 8 ? 0x7FFFFFFFu : 0x00FFFFFFu);   // lazy limbs < 2^31
    const f29 r0 = f29_canon(f29_montmul(a, tab(d_wm, t & 15)));
    const f29 r1 = f29_canon(montmul_plain(a, tab(d_wm, t & 15)));
    const f29 r2 = f29_canon(mulshoup(a, tab(d_w, t & 15), tab(d_wq, t & 15)));
    for (int i = 0; i < 9; i++) if (r0.v[i] != r2.v[i] || r0.v[i] != r1.v[i]) atomicAdd(bad, 1u);
}

template <int MODE>
double run(const char* name, int waves) {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * waves, iters = 512;
    uint32_t* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_chain<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_chain<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double products = (double)blocks * 256 * iters * 4;
    std::printf("%-34s %d waves/SIMD: %7.3f ms  %.3e products/s  (%.1f ps per product per lane-slot)\n", name, waves, ms, products / (ms * 1e-3), ms * 1e9 / products);
    hipFree(out);
    return ms;
}

int main() {
    hipMemcpyToSymbol(HIP_SYMBOL(d_w), SH_W, sizeof SH_W);
    hipMemcpyToSymbol(HIP_SYMBOL(d_wm), SH_WM, sizeof SH_WM);
    hipMemcpyToSymbol(HIP_SYMBOL(d_wq), SH_WQ, sizeof SH_WQ);
    uint32_t* bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(256), 0, 0, bad);
    uint32_t hb = 1; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    std::printf("agreement of montmul (asm), montmul (plain C++) and the Shoup product on 256 lazy inputs: %s\n", hb ? "MISMATCH" : "ok");
    for (int waves : {4, 2}) {
        run<0>("montmul, generated asm columns", waves);
        run<1>("montmul, plain C++ columns", waves);
        run<2>("shoup, plain C++ columns", waves);
    }
    return hb ? 1 : 0;
}
