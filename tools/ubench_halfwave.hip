// ubench_halfwave.hip -- does a wave64 VALU instruction with only 32 (or 16) active lanes issue faster on gfx950?
// Also: issue rate of a single wave per SIMD with a realistic dependent instruction mix (SHA-256 round-like).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_halfwave.hip -o tools/ubench_halfwave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 8192
#define UNROLL 16

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed, uint32_t active) {
    if ((threadIdx.x & 63) >= active) return;
    uint32_t a[UNROLL];
    const uint32_t t = threadIdx.x + seed;
#pragma unroll
    for (int i = 0; i < UNROLL; i++) a[i] = t * (i + 3) + 1;
    uint32_t m = t | 0x10001u;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 1) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(m));
            if (OP == 2) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) % UNROLL]));
            if (OP == 3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) % UNROLL]));
            if (OP == 4) asm volatile("v_lshl_or_b32 %0, %0, 7, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 5) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) % UNROLL]));
            if (OP == 6) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) % UNROLL]));
            if (OP == 7) asm volatile("v_lshrrev_b32 %0, 7, %0" : "+v"(a[i]));
            if (OP == 8) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) % UNROLL]));
            if (OP == 9) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(uint64_t*)&a[i & ~1]) : "v"(m), "v"(a[(i + 3) % UNROLL]) : "vcc");
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; i++) acc ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OP>
void run(const char* name, int waves_per_simd, uint32_t active) {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * waves_per_simd;
    uint32_t* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u, active);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 2u, active);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * ITERS * UNROLL;      // wave-instructions
    printf("%-18s waves/SIMD=%d active=%2u  %8.3f ms  %6.2f cycles/wave-instr/SIMD at 2.4 GHz\n", name, waves_per_simd, active, ms,
           ms * 1e-3 * 2.4e9 / (winstr / (cus * 4.0)));
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) {
        for (uint32_t act : {64u, 32u, 16u}) {
            run<0>("v_add_u32", w, act);
            run<1>("v_alignbit_b32", w, act);
        }
        run<2>("v_bitop3_b32", w, 64);
        run<3>("v_add3_u32", w, 64);
        run<4>("v_lshl_or_b32", w, 64);
        run<5>("v_xad_u32", w, 64);
        run<6>("v_perm_b32", w, 64);
        run<7>("v_lshrrev_b32", w, 64);
        run<8>("v_and_or_b32", w, 64);
        run<9>("v_mad_u64_u32", w, 64);
        run<9>("v_mad_u64_u32", w, 32);
        printf("\n");
    }
    return 0;
}
