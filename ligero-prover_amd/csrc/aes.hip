// aes.hip -- AES-256-CTR field sampler on gfx950.
//
// Device counterpart of mpz_random_engine (include/util/csprng.hpp:28-110: EVP_aes_256_ctr, IV = 0,
// zero plaintext, 16 KiB refills) + bn254_gmp::generate_random (include/zkp/finite_field_gmp.hpp:66-78:
// 32 keystream bytes as 4 little-endian u64 -> >>2 -> subtract p if >= p).  The reference draws these one at
// a time on the host (k-l pads per row, one per constraint in stage 2); on the GPU element e of a stream is
// computed independently from keystream blocks 2e and 2e+1, so a whole batch of rows is one launch.
// T-table AES with the four rotated tables staged in LDS (4 KiB per workgroup): the sampler runs next to the VALU-bound
// encode kernels, so it is written for the fewest VALU instructions -- no rotates (v_alignbit is a half-rate
// instruction on gfx950), three-input xors (v_bitop3_b32), and the last round masks S-box bytes out of the same tables.
#include "kernels.hpp"

namespace lig {

namespace {
struct AesTables { uint32_t te0[256]; uint8_t sbox[256]; };

uint8_t xtime_h(uint8_t x) { return (uint8_t)((x << 1) ^ ((x >> 7) * 0x1b)); }

const AesTables& host_tables() {
    static AesTables T;
    static bool ready = false;
    if (!ready) {
        // S-box from its definition: inverse in GF(2^8) then the affine map (FIPS-197 5.1.1)
        uint8_t p = 1, q = 1;
        do {
            p = (uint8_t)(p ^ (p << 1) ^ ((p & 0x80) ? 0x1b : 0));
            q ^= (uint8_t)(q << 1); q ^= (uint8_t)(q << 2); q ^= (uint8_t)(q << 4);
            if (q & 0x80) q ^= 0x09;
            auto rol = [](uint8_t v, int s) { return (uint8_t)((v << s) | (v >> (8 - s))); };
            T.sbox[p] = (uint8_t)(q ^ rol(q, 1) ^ rol(q, 2) ^ rol(q, 3) ^ rol(q, 4) ^ 0x63);
        } while (p != 1);
        T.sbox[0] = 0x63;
        for (int i = 0; i < 256; i++) {
            const uint8_t s = T.sbox[i], s2 = xtime_h(s), s3 = (uint8_t)(s2 ^ s);
            T.te0[i] = ((uint32_t)s2 << 24) | ((uint32_t)s << 16) | ((uint32_t)s << 8) | s3;
        }
        ready = true;
    }
    return T;
}
}  // namespace

void aes256_expand_host(const uint8_t key[32], uint32_t rk[60]) {
    const AesTables& T = host_tables();
    auto subw = [&](uint32_t t) {
        return ((uint32_t)T.sbox[t >> 24] << 24) | ((uint32_t)T.sbox[(t >> 16) & 255] << 16) |
               ((uint32_t)T.sbox[(t >> 8) & 255] << 8) | T.sbox[t & 255];
    };
    for (int i = 0; i < 8; i++)
        rk[i] = ((uint32_t)key[4 * i] << 24) | ((uint32_t)key[4 * i + 1] << 16) | ((uint32_t)key[4 * i + 2] << 8) | key[4 * i + 3];
    uint8_t rcon = 1;
    for (int i = 8; i < 60; i++) {
        uint32_t t = rk[i - 1];
        if (i % 8 == 0) { t = subw((t << 8) | (t >> 24)) ^ ((uint32_t)rcon << 24); rcon = xtime_h(rcon); }
        else if (i % 8 == 4) t = subw(t);
        rk[i] = rk[i - 8] ^ t;
    }
}

__device__ uint32_t g_te[4 * 256];     // Te0 | Te1 = ror(Te0, 8) | Te2 = ror(Te0, 16) | Te3 = ror(Te0, 24)

__device__ __forceinline__ uint32_t x3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }

// te = the four tables back to back (LDS).  Te0[x] = (2s, s, s, 3s) from the most significant byte down, so the plain
// S-box byte sits in bits 24-31 of Te2, 16-23 of Te3, 8-15 of Te0 and 0-7 of Te1: the last round needs no S-box table.
__device__ __forceinline__ void aes256_block(const uint32_t* __restrict__ rk, const uint32_t* te, uint64_t block_index, uint32_t out[4]) {
    const uint32_t *t0 = te, *t1 = te + 256, *t2 = te + 512, *t3 = te + 768;
    uint32_t s0 = rk[0], s1 = rk[1], s2 = (uint32_t)(block_index >> 32) ^ rk[2], s3 = (uint32_t)block_index ^ rk[3];
#pragma unroll
    for (int r = 1; r < 14; r++) {
        const uint32_t a0 = x3(t0[s0 >> 24], t1[(s1 >> 16) & 255], t2[(s2 >> 8) & 255]), a1 = x3(t0[s1 >> 24], t1[(s2 >> 16) & 255], t2[(s3 >> 8) & 255]);
        const uint32_t a2 = x3(t0[s2 >> 24], t1[(s3 >> 16) & 255], t2[(s0 >> 8) & 255]), a3 = x3(t0[s3 >> 24], t1[(s0 >> 16) & 255], t2[(s1 >> 8) & 255]);
        const uint32_t b0 = x3(a0, t3[s3 & 255], rk[4 * r]), b1 = x3(a1, t3[s0 & 255], rk[4 * r + 1]);
        const uint32_t b2 = x3(a2, t3[s1 & 255], rk[4 * r + 2]), b3 = x3(a3, t3[s2 & 255], rk[4 * r + 3]);
        s0 = b0; s1 = b1; s2 = b2; s3 = b3;
    }
    out[0] = x3(t2[s0 >> 24] & 0xff000000u, t3[(s1 >> 16) & 255] & 0x00ff0000u, t0[(s2 >> 8) & 255] & 0x0000ff00u) ^ (t1[s3 & 255] & 0xffu) ^ rk[56];
    out[1] = x3(t2[s1 >> 24] & 0xff000000u, t3[(s2 >> 16) & 255] & 0x00ff0000u, t0[(s3 >> 8) & 255] & 0x0000ff00u) ^ (t1[s0 & 255] & 0xffu) ^ rk[57];
    out[2] = x3(t2[s2 >> 24] & 0xff000000u, t3[(s3 >> 16) & 255] & 0x00ff0000u, t0[(s0 >> 8) & 255] & 0x0000ff00u) ^ (t1[s1 & 255] & 0xffu) ^ rk[58];
    out[3] = x3(t2[s3 >> 24] & 0xff000000u, t3[(s0 >> 16) & 255] & 0x00ff0000u, t0[(s1 >> 8) & 255] & 0x0000ff00u) ^ (t1[s2 & 255] & 0xffu) ^ rk[59];
}

__global__ void k_rng_fill(const uint32_t* __restrict__ rk_g, uint64_t first_elem, fr* __restrict__ out, size_t count) {
    __shared__ uint32_t te[4 * 256];
    __shared__ uint32_t rk[60];
    for (int i = threadIdx.x; i < 4 * 256; i += blockDim.x) te[i] = g_te[i];
    if (threadIdx.x < 60) rk[threadIdx.x] = rk_g[threadIdx.x];
    __syncthreads();
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (size_t)gridDim.x * blockDim.x) {
        const uint64_t blk = 2 * (first_elem + e);
        uint32_t o[8];
        aes256_block(rk, te, blk, o);
        aes256_block(rk, te, blk + 1, o + 4);
        fr v;
#pragma unroll
        for (int i = 0; i < 8; i++) v.v[i] = __builtin_bswap32(o[i]);   // keystream bytes -> little-endian limbs
#pragma unroll
        for (int i = 0; i < 8; i++) v.v[i] = (v.v[i] >> 2) | (i < 7 ? (v.v[i + 1] << 30) : 0u);
        fr_store(out + e, fr_reduce_once(v));   // v < 2^254 < 2p
    }
}

// Row-structured fill: out[r*row_stride + col_off + i*elem_stride] = stream element (first + r*stream_stride + i),
// r < rows, i < per_row.  One launch forms the k-l pad columns of a whole row batch, a dense randomness row
// batch, or the (0, r, 0, r, ...) pattern of a mask row (elem_stride = 2).
__global__ void k_rng_fill_rows(const uint32_t* __restrict__ rk_g, uint64_t first, fr* __restrict__ out, size_t rows,
                                uint32_t per_row, size_t row_stride, uint32_t col_off, uint32_t elem_stride, uint64_t stream_stride) {
    __shared__ uint32_t te[4 * 256];
    __shared__ uint32_t rk[60];
    for (int i = threadIdx.x; i < 4 * 256; i += blockDim.x) te[i] = g_te[i];
    if (threadIdx.x < 60) rk[threadIdx.x] = rk_g[threadIdx.x];
    __syncthreads();
    const size_t total = rows * per_row;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / per_row;
        const uint32_t i = (uint32_t)(e - r * per_row);
        const uint64_t blk = 2 * (first + r * stream_stride + i);
        uint32_t o[8];
        aes256_block(rk, te, blk, o);
        aes256_block(rk, te, blk + 1, o + 4);
        fr v;
#pragma unroll
        for (int w = 0; w < 8; w++) v.v[w] = __builtin_bswap32(o[w]);
#pragma unroll
        for (int w = 0; w < 8; w++) v.v[w] = (v.v[w] >> 2) | (w < 7 ? (v.v[w + 1] << 30) : 0u);
        fr_store(out + r * row_stride + col_off + (size_t)i * elem_stride, fr_reduce_once(v));
    }
}
// Dense variant for whole randomness rows: row r of `out` (k elements) = per_row stream elements followed by zeros, so the
// buffer needs no memset beforehand (the runtime's fill kernel reaches only ~1.4 TB/s).
__global__ void k_rng_fill_rows_dense(const uint32_t* __restrict__ rk_g, uint64_t first, fr* __restrict__ out, size_t rows,
                                      uint32_t per_row, uint32_t k) {
    __shared__ uint32_t te[4 * 256];
    __shared__ uint32_t rk[60];
    for (int i = threadIdx.x; i < 4 * 256; i += blockDim.x) te[i] = g_te[i];
    if (threadIdx.x < 60) rk[threadIdx.x] = rk_g[threadIdx.x];
    __syncthreads();
    const size_t total = rows * k;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / k;
        const uint32_t i = (uint32_t)(e - r * k);
        fr v = fr_zero();
        if (i < per_row) {
            const uint64_t blk = 2 * (first + r * (uint64_t)per_row + i);
            uint32_t o[8];
            aes256_block(rk, te, blk, o);
            aes256_block(rk, te, blk + 1, o + 4);
#pragma unroll
            for (int w = 0; w < 8; w++) v.v[w] = __builtin_bswap32(o[w]);
#pragma unroll
            for (int w = 0; w < 8; w++) v.v[w] = (v.v[w] >> 2) | (w < 7 ? (v.v[w + 1] << 30) : 0u);
            v = fr_reduce_once(v);
        }
        fr_store(out + e, v);
    }
}
void launch_rng_fill_rows_dense(hipStream_t s, const uint32_t* rk60_dev, uint64_t first, fr* out, size_t rows, uint32_t per_row, uint32_t k) {
    const size_t total = rows * k;
    if (!total) return;
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_rng_fill_rows_dense, dim3((uint32_t)blocks), dim3(256), 0, s, rk60_dev, first, out, rows, per_row, k);
}

void launch_rng_fill_rows(hipStream_t s, const uint32_t* rk60_dev, uint64_t first, fr* out, size_t rows, uint32_t per_row,
                          size_t row_stride, uint32_t col_off, uint32_t elem_stride, uint64_t stream_stride) {
    const size_t total = rows * per_row;
    if (!total) return;
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_rng_fill_rows, dim3((uint32_t)blocks), dim3(256), 0, s, rk60_dev, first, out, rows, per_row, row_stride,
                       col_off, elem_stride, stream_stride);
}

// per-device table upload, called from lig_ctx_create
void aes_upload_tables() {
    const AesTables& T = host_tables();
    uint32_t te[4 * 256];
    for (int t = 0; t < 4; t++)
        for (int i = 0; i < 256; i++) te[256 * t + i] = t ? (T.te0[i] >> (8 * t)) | (T.te0[i] << (32 - 8 * t)) : T.te0[i];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_te), te, sizeof(te));
}

void launch_rng_fill(hipStream_t s, const uint32_t* rk60_dev, uint64_t first_elem, fr* out, size_t count) {
    if (!count) return;
    size_t blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_rng_fill, dim3((uint32_t)blocks), dim3(256), 0, s, rk60_dev, first_elem, out, count);
}

}  // namespace lig
