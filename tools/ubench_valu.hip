// ubench_valu.hip -- VALU issue-rate microbenchmark for the integer/fp64 instructions a 256-bit modular
// multiply can be built from on gfx950.  Prints ops/clk/CU for each.  Build:
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITERS 4096
#define UNROLL 16

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
    uint32_t a[UNROLL]; uint64_t q[UNROLL]; double d[UNROLL];
    const uint32_t t = threadIdx.x + seed;
#pragma unroll
    for (int i = 0; i < UNROLL; i++) { a[i] = t * (i + 3) + 1; q[i] = ((uint64_t)a[i] << 20) + i; d[i] = (double)(a[i] & 1023) + 0.5; }
    uint32_t m = t | 0x10001u;
    double dm = 1.0000001 + 1e-9 * (t & 7);
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(m), "v"(a[i]) : "vcc");
            if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 3) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(m) : "vcc");
            if (OP == 4) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
            if (OP == 5) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(dm));
            if (OP == 6) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m));
            if (OP == 7) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[i]) : "v"(q[(i + 1) % UNROLL]));
            if (OP == 8) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 9) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 10) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 11) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 12) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(q[i]));
            if (OP == 13) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dm));
            if (OP == 14) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
            if (OP == 15) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(m));
            if (OP == 16) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(q[i]) : "v"(q[(i + 1) % UNROLL]));
            if (OP == 17) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 18) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "s"(seed), "v"(a[i]) : "vcc");
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; i++) acc ^= a[i] ^ (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32) ^ (uint32_t)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OP>
double run(const char* name, int waves_per_simd) {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * waves_per_simd;       // 256 threads = 4 waves = 1 wave per SIMD per block
    uint32_t* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 2u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 256 * ITERS * UNROLL;
    const double clk_hz = (double)prop.clockRate * 1e3;
    const double per_clk_cu = ops / (ms * 1e-3) / clk_hz / cus;
    printf("%-22s waves/SIMD=%d  %8.3f ms  %7.2f Gop/s  %6.2f lane-ops/clk/CU (at %.0f MHz nominal)\n", name, waves_per_simd, ms,
           ops / ms * 1e-6, per_clk_cu, clk_hz * 1e-6);
    hipFree(out);
    return per_clk_cu;
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_mad_u64_u32", w);
        run<18>("v_mad_u64_u32 sgpr", w);
        run<1>("v_mul_lo_u32", w);
        run<2>("v_mul_hi_u32", w);
        run<3>("v_add_co_u32", w);
        run<4>("v_addc_co_u32", w);
        run<9>("v_add_u32", w);
        run<17>("v_xor_b32", w);
        run<8>("v_mov_b32", w);
        run<14>("v_cndmask_b32", w);
        run<15>("v_alignbit_b32", w);
        run<7>("v_lshl_add_u64", w);
        run<12>("v_lshrrev_b64", w);
        run<6>("v_mad_u32_u24", w);
        run<10>("v_mul_u32_u24", w);
        run<11>("v_mul_hi_u32_u24", w);
        run<5>("v_fma_f64", w);
        run<13>("v_mul_f64", w);
        run<16>("v_pk_fma_f32", w);
        printf("\n");
    }
    return 0;
}
