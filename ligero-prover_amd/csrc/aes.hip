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
#include "fr29.hpp"
#include "kernels.hpp"
#include <cstdlib>

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
//
// LDS layout (LOGR): every table word is replicated R = 2^LOGR times, word (t, i, r) at index ((t*256 + i) << LOGR) + r,
// and lane l reads replica r = l mod R.  The 16 lookups per round are random: with one copy (LOGR = 0) the 32 lanes of a
// half-wave spread over the 32 banks with a maximum load of ~3.5, and the sampler is bound by exactly that (measured:
// 13.8k SIMD-cycles per wave of 64 elements, 448 lookups each).  With R = 16 the bank of a lookup is 16*(i mod 2) + r: only
// the two lanes of a half-wave that share a replica can collide (probability 1/2 when their bytes differ) -- ~1.25 cycles
// per half-wave instead of ~3.5.  64 KiB of LDS per workgroup, so the big fills run as <= 2 persistent workgroups per CU;
// the round keys are read through scalar loads (uniform addresses), not from LDS.
//
// LAYOUT 1 (round 6, LIG_AES_LAYOUT=1; A/B in profiles/r06_sampler_layout_ab.md): the same 64 KiB ordered entry-major -- byte address of
// (table t, entry i, replica r) = i * 256 + t * 64 + r * 4, 16 replicas -- so that the whole address of a lookup is ONE v_perm_b32:
// {0, 0, byte n of the state word, the lane's r * 4}, the table in the instruction's immediate offset.  One VALU instruction per lookup
// instead of two (extract + scale-and-add): 16 of the 36 VALU instructions of an AES round go.  Price: the bank of a lookup is
// (16 t + r) mod 32 whatever the entry, so the two lanes of a half-wave that share a replica always collide (2 LDS cycles per
// half-wave instead of ~1.25).
struct TeView { const uint32_t* base; uint32_t off; };      // LAYOUT 0: base = the lane's replica of word (0, 0); LAYOUT 1: base = the tables, off = r * 4
template <int LOGR, int LAYOUT>
__device__ __forceinline__ uint32_t te_at(const TeView& tv, uint32_t table, uint32_t word, int shift) {
    if constexpr (LAYOUT == 1) {
        static_assert(LOGR == 4 || LAYOUT == 0, "entry-major layout: 16 replicas");
        // selector bytes (most significant first): 0x0c = constant 0, 4 + n = byte n of src0 (the state word), 0 = byte 0 of src1 (the lane offset)
        const uint32_t a = __builtin_amdgcn_perm(word, tv.off, 0x0c0c0000u | ((4u + (uint32_t)(shift >> 3)) << 8));
        return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(tv.base) + a + table * 64u);
    } else {
        // byte `shift/8` of `word` -> table index scaled by R
        const uint32_t idx = shift == 24 ? (word >> 24) : ((word >> shift) & 255u);
        return tv.base[((table << 8) + idx) << LOGR];
    }
}
template <int LOGR, int LAYOUT>
__device__ __forceinline__ void aes256_block(const uint32_t* __restrict__ rk, const TeView& tl, uint64_t block_index, uint32_t out[4]) {
    uint32_t s0 = rk[0], s1 = rk[1], s2 = (uint32_t)(block_index >> 32) ^ rk[2], s3 = (uint32_t)block_index ^ rk[3];
#define T_(t, w, sh) te_at<LOGR, LAYOUT>(tl, t, w, sh)
#pragma unroll
    for (int r = 1; r < 14; r++) {
        const uint32_t a0 = x3(T_(0, s0, 24), T_(1, s1, 16), T_(2, s2, 8)), a1 = x3(T_(0, s1, 24), T_(1, s2, 16), T_(2, s3, 8));
        const uint32_t a2 = x3(T_(0, s2, 24), T_(1, s3, 16), T_(2, s0, 8)), a3 = x3(T_(0, s3, 24), T_(1, s0, 16), T_(2, s1, 8));
        const uint32_t b0 = x3(a0, T_(3, s3, 0), rk[4 * r]), b1 = x3(a1, T_(3, s0, 0), rk[4 * r + 1]);
        const uint32_t b2 = x3(a2, T_(3, s1, 0), rk[4 * r + 2]), b3 = x3(a3, T_(3, s2, 0), rk[4 * r + 3]);
        s0 = b0; s1 = b1; s2 = b2; s3 = b3;
    }
    out[0] = x3(T_(2, s0, 24) & 0xff000000u, T_(3, s1, 16) & 0x00ff0000u, T_(0, s2, 8) & 0x0000ff00u) ^ (T_(1, s3, 0) & 0xffu) ^ rk[56];
    out[1] = x3(T_(2, s1, 24) & 0xff000000u, T_(3, s2, 16) & 0x00ff0000u, T_(0, s3, 8) & 0x0000ff00u) ^ (T_(1, s0, 0) & 0xffu) ^ rk[57];
    out[2] = x3(T_(2, s2, 24) & 0xff000000u, T_(3, s3, 16) & 0x00ff0000u, T_(0, s0, 8) & 0x0000ff00u) ^ (T_(1, s1, 0) & 0xffu) ^ rk[58];
    out[3] = x3(T_(2, s3, 24) & 0xff000000u, T_(3, s0, 16) & 0x00ff0000u, T_(0, s1, 8) & 0x0000ff00u) ^ (T_(1, s2, 0) & 0xffu) ^ rk[59];
#undef T_
}
// stage the replicated tables; returns the calling lane's view
template <int LOGR, int LAYOUT>
__device__ __forceinline__ TeView te_stage(uint32_t* te) {
    if constexpr (LAYOUT == 1) {
        for (uint32_t i = threadIdx.x; i < (1024u << LOGR); i += blockDim.x) te[i] = g_te[(((i >> 4) & 3u) << 8) + (i >> 6)];     // i = entry * 64 + table * 16 + replica
        __syncthreads();
        return TeView{te, (threadIdx.x & 15u) * 4u};
    } else {
        for (uint32_t i = threadIdx.x; i < (1024u << LOGR); i += blockDim.x) te[i] = g_te[i >> LOGR];
        __syncthreads();
        return TeView{te + (threadIdx.x & ((1u << LOGR) - 1)), 0u};
    }
}
// keystream element -> field element (finite_field_gmp.hpp:66-78)
template <int LOGR, int LAYOUT>
__device__ __forceinline__ fr aes_field_elem(const uint32_t* __restrict__ rk, const TeView& tl, uint64_t elem) {
    uint32_t o[8];
    aes256_block<LOGR, LAYOUT>(rk, tl, 2 * elem, o);
    aes256_block<LOGR, LAYOUT>(rk, tl, 2 * elem + 1, o + 4);
    fr v;
#pragma unroll
    for (int i = 0; i < 8; i++) v.v[i] = __builtin_bswap32(o[i]);   // keystream bytes -> little-endian limbs
#pragma unroll
    for (int i = 0; i < 8; i++) v.v[i] = (v.v[i] >> 2) | (i < 7 ? (v.v[i + 1] << 30) : 0u);
    return fr_reduce_once(v);   // v < 2^254 < 2p
}

template <int LOGR, int LAYOUT = 0>
__global__ void __launch_bounds__(256) k_rng_fill(const uint32_t* __restrict__ rk, uint64_t first_elem, fr* __restrict__ out, size_t count) {
    __shared__ uint32_t te[1024 << LOGR];
    const TeView tl = te_stage<LOGR, LAYOUT>(te);
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (size_t)gridDim.x * blockDim.x)
        fr_store(out + e, aes_field_elem<LOGR, LAYOUT>(rk, tl, first_elem + e));
}

// Row-structured fill: out[r*row_stride + col_off + i*elem_stride] = stream element (first + r*stream_stride + i),
// r < rows, i < per_row.  One launch forms the k-l pad columns of a whole row batch, a dense randomness row
// batch, or the (0, r, 0, r, ...) pattern of a mask row (elem_stride = 2).
template <int LOGR, int LAYOUT = 0>
__global__ void __launch_bounds__(256) k_rng_fill_rows(const uint32_t* __restrict__ rk, uint64_t first, fr* __restrict__ out, size_t rows,
                                                       uint32_t per_row, size_t row_stride, uint32_t col_off, uint32_t elem_stride, uint64_t stream_stride) {
    __shared__ uint32_t te[1024 << LOGR];
    const TeView tl = te_stage<LOGR, LAYOUT>(te);
    const size_t total = rows * per_row;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / per_row;
        const uint32_t i = (uint32_t)(e - r * per_row);
        fr_store(out + r * row_stride + col_off + (size_t)i * elem_stride, aes_field_elem<LOGR, LAYOUT>(rk, tl, first + r * stream_stride + i));
    }
}
// Dense variant for whole randomness rows: row r of `out` (k elements) = per_row stream elements followed by zeros, so the
// buffer needs no memset beforehand (the runtime's fill kernel reaches only ~1.4 TB/s).
template <int LOGR, int LAYOUT = 0>
__global__ void __launch_bounds__(256) k_rng_fill_rows_dense(const uint32_t* __restrict__ rk, uint64_t first, fr* __restrict__ out, size_t rows,
                                                             uint32_t per_row, uint32_t k) {
    __shared__ uint32_t te[1024 << LOGR];
    const TeView tl = te_stage<LOGR, LAYOUT>(te);
    const size_t total = rows * k;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / k;
        const uint32_t i = (uint32_t)(e - r * k);
        fr_store(out + e, i < per_row ? aes_field_elem<LOGR, LAYOUT>(rk, tl, first + r * (uint64_t)per_row + i) : fr_zero());
    }
}
// Sampler + stage-2 accumulation in one pass (the dense randomness rows of the synthetic stream, nonbatch_context.hpp:756-780
// check_code / check_linear on the message domain): thread = position j of a group of rows.  Per row it draws the row's
// randomness element (AES, LDS-bound), stores it for the encoder (K1 reads the row next), and -- while the element is in
// registers -- adds rc_r * msg_r[j] to the code partial and msg_r[j] * rand_r[j] to the linear partial of its group
// (VALU-bound: the two halves of k_rng_fill_rows_dense + k_rlc_partial fill each other's idle slots, and the randomness row
// is read once less).  Partials have k_rlc_partial's contract: code_part = running lazy sums (< 2p), lin_part plain values
// (< 2p), both ADDED to (zeroed by the caller before the first chunk).  Persistent workgroups over (group, 256 positions) tiles.
#ifndef LIG_RLC_THREADS
#define LIG_RLC_THREADS 256
#endif
template <int LOGR, int LAYOUT = 0>
__global__ void __launch_bounds__(LIG_RLC_THREADS) k_rand_rlc(const uint32_t* __restrict__ rk, uint64_t first, fr* __restrict__ rand_out,
                                                  const fr* __restrict__ msgs, size_t rows, uint32_t per_row, uint32_t k,
                                                  const f29s* __restrict__ rc, uint32_t group_rows, fr* __restrict__ code_part,
                                                  fr* __restrict__ lin_part) {
    __shared__ uint32_t te[1024 << LOGR];
    const TeView tl = te_stage<LOGR, LAYOUT>(te);
    // (-DLIG_RLC_THREADS=512: 512 threads share one 64 KiB set of replicated tables = four waves per SIMD instead of two to hide the LDS
    // latency of the 448 lookups per element behind the products; measured equal, profiles/r03_fused_rand_rlc_ab.md)
    const uint32_t jblocks = k / blockDim.x, groups = (uint32_t)((rows + group_rows - 1) / group_rows), tiles = jblocks * groups;
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint32_t g = tile / jblocks, j = (tile - g * jblocks) * blockDim.x + threadIdx.x;
        const size_t r0 = (size_t)g * group_rows, r1 = r0 + group_rows < rows ? r0 + group_rows : rows;
        const bool live = j < per_row;                         // positions >= per_row of a dense row are zero
        f29 ac = rc != nullptr ? unpack29(fr_load(code_part + (size_t)g * k + j)) : f29_zero(), al = f29_zero();
        int since = 0;
        for (size_t r = r0; r < r1; r++) {
            const f29 u = unpack29(fr_load(msgs + r * k + j));
            if (rc != nullptr) ac = f29_add(ac, f29_montmul(u, f29_load_tab(rc + r)));      // (rc == nullptr: the code test was accumulated up front)
            fr v = fr_zero();
            if (live) {
                v = aes_field_elem<LOGR, LAYOUT>(rk, tl, first + r * (uint64_t)per_row + j);
                al = f29_add(al, f29_montmul(u, unpack29(v)));
            }
            fr_store(rand_out + r * k + j, v);
            if (++since == 6) { ac = f29_qnorm(ac); al = f29_qnorm(al); since = 0; }
        }
        if (rc != nullptr) fr_store(code_part + (size_t)g * k + j, pack29(f29_reduce_2p(ac)));
        f29 w = f29_montmul(f29_qnorm(al), f29_const_r2());                                   // plain value, < 1.2p
        w = f29_reduce_2p(f29_add(w, unpack29(fr_load(lin_part + (size_t)g * k + j))));
        fr_store(lin_part + (size_t)g * k + j, pack29(w));
    }
}

// launch shape: big fills = persistent workgroups (<= 2 per CU, 64 KiB of replicated tables each); small ones = one copy of
// the tables (4 KiB), many workgroups
static constexpr size_t BIG_FILL = (size_t)1 << 20;      // below this the 64 KiB table staging per workgroup does not pay
#ifndef LIG_AES_REP
#define LIG_AES_REP 4
#endif
static constexpr int REP = LIG_AES_REP;
static constexpr uint32_t BIG_BLOCKS_MAX = 256u * (REP >= 4 ? 2u : REP == 3 ? 4u : 8u);     // persistent workgroups: as many as the LDS of 256 CUs holds
#define BIG_BLOCKS (lig::knobs().aes_blocks ? lig::knobs().aes_blocks : BIG_BLOCKS_MAX)
#define PERM_LAYOUT (REP == 4 && lig::knobs().aes_layout == 1)      // LIG_AES_LAYOUT=1: the entry-major tables of the big launches (one v_perm_b32 per lookup)
static inline uint32_t small_blocks(size_t total, size_t cap) { size_t b = (total + 255) / 256; return (uint32_t)(b > cap ? cap : b); }
void launch_rng_fill_rows_dense(hipStream_t s, const uint32_t* rk60_dev, uint64_t first, fr* out, size_t rows, uint32_t per_row, uint32_t k) {
    const size_t total = rows * k;
    if (!total) return;
    if (total >= BIG_FILL && PERM_LAYOUT) hipLaunchKernelGGL((k_rng_fill_rows_dense<4, 1>), dim3(BIG_BLOCKS), dim3(256), 0, s, rk60_dev, first, out, rows, per_row, k);
    else if (total >= BIG_FILL) hipLaunchKernelGGL(k_rng_fill_rows_dense<REP>, dim3(BIG_BLOCKS), dim3(256), 0, s, rk60_dev, first, out, rows, per_row, k);
    else hipLaunchKernelGGL(k_rng_fill_rows_dense<0>, dim3(small_blocks(total, 8192)), dim3(256), 0, s, rk60_dev, first, out, rows, per_row, k);
}

// rows x k dense randomness rows into `out` + the message-domain partials of the code and linear tests (k % 256 == 0)
void launch_rand_rlc(hipStream_t s, const uint32_t* rk60_dev, uint64_t first, fr* out, const fr* msgs, size_t rows, uint32_t per_row, uint32_t k,
                     const f29s* rc_dev, uint32_t group_rows, fr* code_part, fr* lin_part) {
    if (!rows) return;
    const uint32_t th = (k % LIG_RLC_THREADS == 0) ? LIG_RLC_THREADS : 256;      // k is a multiple of 256 (checked by the callers)
    const size_t tiles = (size_t)(k / th) * ((rows + group_rows - 1) / group_rows);
    if (rows * k >= BIG_FILL && PERM_LAYOUT)
        hipLaunchKernelGGL((k_rand_rlc<4, 1>), dim3((uint32_t)(tiles < BIG_BLOCKS ? tiles : BIG_BLOCKS)), dim3(th), 0, s, rk60_dev, first, out, msgs, rows, per_row, k,
                           rc_dev, group_rows, code_part, lin_part);
    else if (rows * k >= BIG_FILL)
        hipLaunchKernelGGL(k_rand_rlc<REP>, dim3((uint32_t)(tiles < BIG_BLOCKS ? tiles : BIG_BLOCKS)), dim3(th), 0, s, rk60_dev, first, out, msgs, rows, per_row, k,
                           rc_dev, group_rows, code_part, lin_part);
    else
        hipLaunchKernelGGL(k_rand_rlc<0>, dim3((uint32_t)(tiles < 8192 ? tiles : 8192)), dim3(th), 0, s, rk60_dev, first, out, msgs, rows, per_row, k, rc_dev,
                           group_rows, code_part, lin_part);
}

void launch_rng_fill_rows(hipStream_t s, const uint32_t* rk60_dev, uint64_t first, fr* out, size_t rows, uint32_t per_row,
                          size_t row_stride, uint32_t col_off, uint32_t elem_stride, uint64_t stream_stride) {
    const size_t total = rows * per_row;
    if (!total) return;
    if (total >= BIG_FILL && PERM_LAYOUT)
        hipLaunchKernelGGL((k_rng_fill_rows<4, 1>), dim3(BIG_BLOCKS), dim3(256), 0, s, rk60_dev, first, out, rows, per_row, row_stride, col_off, elem_stride, stream_stride);
    else if (total >= BIG_FILL)
        hipLaunchKernelGGL(k_rng_fill_rows<REP>, dim3(BIG_BLOCKS), dim3(256), 0, s, rk60_dev, first, out, rows, per_row, row_stride, col_off, elem_stride, stream_stride);
    else
        hipLaunchKernelGGL(k_rng_fill_rows<0>, dim3(small_blocks(total, 8192)), dim3(256), 0, s, rk60_dev, first, out, rows, per_row, row_stride, col_off,
                           elem_stride, stream_stride);
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
    if (count >= BIG_FILL && PERM_LAYOUT) hipLaunchKernelGGL((k_rng_fill<4, 1>), dim3(BIG_BLOCKS), dim3(256), 0, s, rk60_dev, first_elem, out, count);
    else if (count >= BIG_FILL) hipLaunchKernelGGL(k_rng_fill<REP>, dim3(BIG_BLOCKS), dim3(256), 0, s, rk60_dev, first_elem, out, count);
    else hipLaunchKernelGGL(k_rng_fill<0>, dim3(small_blocks(count, 4096)), dim3(256), 0, s, rk60_dev, first_elem, out, count);
}

}  // namespace lig
