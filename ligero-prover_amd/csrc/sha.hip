// sha.hip -- per-column SHA-256 commitment and GPU Merkle tree for gfx950.
//
// Replaces shader/sha256.wgsl (sha256_init/update/final, one instance per codeword column) and the host
// merkle_tree::build_tree (include/zkp/merkle_tree.hpp:361-375).
//
// Layout.  The reference keeps a 300-byte context per instance with one message BYTE per u32 in global
// memory (include/wgpu.hpp:63-68).  Here the chaining value lives in registers for a whole row batch and the
// persistent state is 2 x 8 x n_inst u32, struct-of-arrays: h[8][n_inst] and pend[8][n_inst] (the buffered
// element when an odd number of rows has been absorbed: a SHA block is exactly two 32-byte elements).
// The number of absorbed rows is the same for all instances and is tracked by the host context.
//
// Message word order (shader/sha256.wgsl:148-163): every u32 limb is fed most-significant byte first, limbs
// least-significant first, so the 16 message words of a block are exactly the 8 limbs of row 2q followed by
// the 8 limbs of row 2q+1.  The leaf is the 8 state words stored as native u32 (:226-228).
#include "fr29.hpp"
#include "kernels.hpp"
#include "tile_dft.hpp"
#include <cstdlib>

namespace lig {

__device__ __constant__ static const uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }
// gfx950 v_bitop3_b32: any 3-input boolean function in one instruction (truth table over a = 0xF0, b = 0xCC, c = 0xAA).
// The compiler does not form it for x ^ y ^ z by itself; a lone hash wave is issue-latency bound (~5 cycles per
// instruction whatever its rate, tools/ubench_halfwave.hip), so instruction count is what the column chain pays for.
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ uint32_t ch(uint32_t e, uint32_t f, uint32_t g) { return __builtin_amdgcn_bitop3_b32(e, f, g, 0xCA); }
__device__ __forceinline__ uint32_t maj(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }

// one compression; w[16] is consumed (rolling schedule in registers)
__device__ __forceinline__ void sha256_compress(uint32_t h[8], uint32_t w[16]) {
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        uint32_t wi;
        if (i < 16) wi = w[i];
        else {
            const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            const uint32_t s0 = xor3(rotr(w15, 7), rotr(w15, 18), w15 >> 3);
            const uint32_t s1 = xor3(rotr(w2, 17), rotr(w2, 19), w2 >> 10);
            wi = (w[i & 15] + s0 + w[(i - 7) & 15]) + s1;
            w[i & 15] = wi;
        }
        const uint32_t t1 = (hh + xor3(rotr(e, 6), rotr(e, 11), rotr(e, 25)) + ch(e, f, g)) + (SHA_K[i] + wi);
        const uint32_t t2 = xor3(rotr(a, 2), rotr(a, 13), rotr(a, 22)) + maj(a, b, c);
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

__global__ void k_sha_init(uint32_t* __restrict__ st, size_t n_inst) {
    const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_inst; j += (size_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int i = 0; i < 8; i++) { st[(size_t)i * n_inst + j] = iv[i]; st[(size_t)(8 + i) * n_inst + j] = 0; }
    }
}

// absorb nrows rows; rows_before = rows absorbed so far.  Instance j hashes one codeword column:
//   pk == 0                : column j of `rows` (row r at rows + r*row_stride)                      -- the ABI form
//   pk == k, msgs == null  : PLANE-MAJOR instances: j = r*k + q is column 4q + r of `rows` (interleaved rows, e.g. the masks)
//   pk == k, msgs != null  : plane-major instances over a CwView: coset 0 from the message rows (reversed), cosets 1..3 from
//                            the planes in `rows` (rows x 3k); a wave reads 2 KiB contiguous per row
__global__ void k_sha_update_rows(uint32_t* __restrict__ st, size_t n_inst, const fr* __restrict__ rows, size_t row_stride,
                                  size_t nrows, uint64_t rows_before, uint32_t pk, const fr* __restrict__ msgs) {
    // The column chain is sequential in rows: this kernel is latency-bound with only n_inst/64 waves and usually runs
    // next to an encode kernel on the side stream.
    // (s_setprio(3) measured: no gain stand-alone, slower when co-resident with the encode kernels.  A two-wave variant --
    // one wave expanding the message schedule of block b+1 into LDS while the other runs the rounds of block b -- was
    // measured too: same stand-alone time, 2 % slower proofs; the per-block barrier and LDS round trip eat the shorter chain.)
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_inst) return;
    uint32_t h[8], w[16];
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = st[(size_t)i * n_inst + j];
    // virtual element sequence: the pending half block (if any) followed by the new rows; one compression per pair.
    // The two elements of the NEXT pair are requested before the current compression starts, so that their HBM latency
    // (about a quarter of a compression for a lone wave) is covered by the ~1470 instructions of the rounds.
    const size_t pend = (size_t)(rows_before & 1);
    const size_t total = nrows + pend;
    const fr* p0 = rows + j;                               // this column's element of row 0, and its row stride
    size_t rs = row_stride;
    if (pk) {
        const uint32_t r = (uint32_t)j / pk, q = (uint32_t)j - r * pk;
        if (msgs == nullptr) p0 = rows + 4 * (size_t)q + r;
        else if (r == 0) { p0 = msgs + ((pk - q) & (pk - 1)); rs = pk; }
        else { p0 = rows + (size_t)(r - 1) * pk + q; rs = 3 * (size_t)pk; }
    }
    auto elem = [&](size_t v) -> fr {                      // virtual element v
        if (v == 0 && pend) {
            fr e;
#pragma unroll
            for (int i = 0; i < 8; i++) e.v[i] = st[(size_t)(8 + i) * n_inst + j];
            return e;
        }
        return fr_load(p0 + (v - pend) * rs);
    };
    fr n0 = fr_zero(), n1 = fr_zero();
    if (total >= 2) { n0 = elem(0); n1 = elem(1); }
    for (size_t v = 0; v + 1 < total; v += 2) {
#pragma unroll
        for (int i = 0; i < 8; i++) { w[i] = n0.v[i]; w[8 + i] = n1.v[i]; }
        if (v + 3 < total) { n0 = elem(v + 2); n1 = elem(v + 3); }
        sha256_compress(h, w);
    }
    if (total & 1) {         // odd tail: keep the element for the next call / final
        if (total == 1 && pend) {
            // nothing new absorbed (nrows == 0 is filtered by the launcher); unreachable, kept for clarity
        } else {
            const fr e = fr_load(p0 + (total - 1 - pend) * rs);
#pragma unroll
            for (int i = 0; i < 8; i++) st[(size_t)(8 + i) * n_inst + j] = e.v[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) st[(size_t)i * n_inst + j] = h[i];
}

// ---------------------------------------------------------------------------------------------------------------------
// Wave-specialised column hash (round 5).  leaf_j is a Merkle-Damgard chain over all rows of column j: nothing shortens it except
// taking work OUT of the chain.  Of the 1760 instructions of a compression only the 64 rounds (14 instructions each) depend on the
// chaining value; the message schedule sigma0 / sigma1 and the K[t] + W[t] additions (a third of the instructions) do not.  So a
// group of 64 columns is served by TWO waves of one workgroup:
//   producer  loads the two rows of block b + 1 from HBM, expands the schedule, adds K and leaves the 64 words K[t] + W[t] of every
//             column in LDS (16 x ds_write_b128 per lane, 16 KiB per group);
//   consumer  copies them into registers (16 x ds_read_b128, issued behind round 15 of block b, i.e. hidden under its remaining
//             rounds) and runs nothing but the rounds: ~930 instructions per block on the critical path instead of 1760.
// Hand-off without barriers: two LDS words per group -- `ready` = blocks the producer has published, `taken` = blocks the consumer
// has copied out -- written with release, polled with acquire (s_sleep between polls; in steady state the producer is a block
// ahead: its block costs ~620 instructions).  ONE slot per group suffices because the consumer holds block b + 1 in registers while it
// works on block b.  The consumer keeps two register sets (even / odd blocks): no register copies between blocks.
// Workgroup = GROUPS consumer waves followed by GROUPS producer waves; with GROUPS = 2 the four waves of a workgroup land on the four
// SIMDs of a CU (32768 columns = 256 workgroups = one per CU: every wave alone on its SIMD when nothing else runs).
// A two-wave variant with a barrier per block was measured in round 3 and lost (the barrier and the LDS round trip ate the shorter
// chain); this one has no barrier after the prologue and no LDS latency on the chain.
struct Kw4 { uint32_t x, y, z, w; };

__device__ __forceinline__ uint32_t lds_acquire(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_release(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// rounds [R0, R1) of a compression on (a..hh) with the words K[t] + W[t] in kw[]
template <int R0, int R1>
__device__ __forceinline__ void sha256_rounds(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d, uint32_t& e, uint32_t& f, uint32_t& g, uint32_t& hh,
                                              const uint32_t (&kw)[64]) {
#pragma unroll
    for (int i = R0; i < R1; i++) {
        const uint32_t t1 = (hh + xor3(rotr(e, 6), rotr(e, 11), rotr(e, 25)) + ch(e, f, g)) + kw[i];
        const uint32_t t2 = xor3(rotr(a, 2), rotr(a, 13), rotr(a, 22)) + maj(a, b, c);
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
}

// Residency: the producer and the consumer wave of a group spin on LDS words of their own workgroup.  A workgroup is dispatched to a CU
// only when ALL of its waves (and its LDS) fit -- the hardware never starts half a workgroup -- so both waves of every group are resident
// for the kernel's whole life; nothing else (no second workgroup, no other kernel) is waited for.  GROUPS * 2 waves x <= 168 VGPRs fit the
// 512-register file of a SIMD for GROUPS <= 4 (one or two of these waves per SIMD).  Every variant and every ragged shape is checked against
// the oracle in tests/test_gpu_sha_ws.py; LIG_SHA_WS=0 (the one-wave kernel above) is the documented fallback.
template <int GROUPS>
__global__ void __launch_bounds__(GROUPS * 128) k_sha_update_rows_ws(uint32_t* __restrict__ st, size_t n_inst, const fr* __restrict__ rows, size_t row_stride,
                                                                      size_t nrows, uint64_t rows_before, uint32_t pk, const fr* __restrict__ msgs) {
    __shared__ uint4 ring[GROUPS][16][64];
    __shared__ uint32_t ready[GROUPS], taken[GROUPS];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool producer = wave >= GROUPS;
    const uint32_t g = producer ? wave - GROUPS : wave;
    if (threadIdx.x < GROUPS) { ready[threadIdx.x] = 0; taken[threadIdx.x] = 0; }
    __syncthreads();                                       // the only barrier: before any wave leaves
    const size_t group = (size_t)blockIdx.x * GROUPS + g;
    if (group * 64 >= n_inst) return;                      // (a whole group beyond the columns: both of its waves leave)
    const size_t j = group * 64 + lane;
    const bool live = j < n_inst;                          // a ragged last group: dead lanes take part in the hand-off, touch no memory
    const size_t pend = (size_t)(rows_before & 1);
    const size_t total = nrows + pend;
    const uint32_t nblocks = (uint32_t)(total / 2);
    uint4 (*slot)[64] = ring[g];

    if (producer) {
        const fr* p0 = rows + j;
        size_t rs = row_stride;
        if (pk && live) {
            const uint32_t r = (uint32_t)j / pk, q = (uint32_t)j - r * pk;
            if (msgs == nullptr) p0 = rows + 4 * (size_t)q + r;
            else if (r == 0) { p0 = msgs + ((pk - q) & (pk - 1)); rs = pk; }
            else { p0 = rows + (size_t)(r - 1) * pk + q; rs = 3 * (size_t)pk; }
        }
        auto elem = [&](size_t v) -> fr {                  // virtual element v: the pending half block (if any), then the new rows
            if (!live) return fr_zero();
            if (v == 0 && pend) {
                fr e;
#pragma unroll
                for (int i = 0; i < 8; i++) e.v[i] = st[(size_t)(8 + i) * n_inst + j];
                return e;
            }
            return fr_load(p0 + (v - pend) * rs);
        };
        fr n0 = fr_zero(), n1 = fr_zero();
        if (nblocks) { n0 = elem(0); n1 = elem(1); }
        for (uint32_t b = 0; b < nblocks; b++) {
            uint32_t w[64];
#pragma unroll
            for (int i = 0; i < 8; i++) { w[i] = n0.v[i]; w[8 + i] = n1.v[i]; }
            if (b + 1 < nblocks) { n0 = elem(2 * (size_t)b + 2); n1 = elem(2 * (size_t)b + 3); }      // the next block's rows: in flight under the expansion
#pragma unroll
            for (int i = 16; i < 64; i++) {
                const uint32_t w15 = w[i - 15], w2 = w[i - 2];
                w[i] = (w[i - 16] + xor3(rotr(w15, 7), rotr(w15, 18), w15 >> 3) + w[i - 7]) + xor3(rotr(w2, 17), rotr(w2, 19), w2 >> 10);
            }
#pragma unroll
            for (int i = 0; i < 64; i++) w[i] += SHA_K[i];
            while (__builtin_amdgcn_readfirstlane(lds_acquire(&taken[g])) < b) __builtin_amdgcn_s_sleep(2);      // the slot is free: block b - 1 is in the consumer's registers
#pragma unroll
            for (int i = 0; i < 16; i++) slot[i][lane] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
            lds_release(&ready[g], b + 1);
        }
        if ((total & 1) && live && !(total == 1 && pend)) {  // odd tail: keep the element for the next call / final
            const fr e = fr_load(p0 + (total - 1 - pend) * rs);
#pragma unroll
            for (int i = 0; i < 8; i++) st[(size_t)(8 + i) * n_inst + j] = e.v[i];
        }
        return;
    }

    // ---- consumer
    if (!nblocks) return;
    uint32_t h[8];
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = live ? st[(size_t)i * n_inst + j] : 0u;
    uint32_t ka[64], kb[64];
    auto fetch = [&](uint32_t (&dst)[64], uint32_t upto) {   // block upto - 1 out of the slot (its 16 reads are asynchronous: first use waits)
        while (__builtin_amdgcn_readfirstlane(lds_acquire(&ready[g])) < upto) __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int i = 0; i < 16; i++) { const uint4 q = slot[i][lane]; dst[4 * i] = q.x; dst[4 * i + 1] = q.y; dst[4 * i + 2] = q.z; dst[4 * i + 3] = q.w; }
    };
    auto block = [&](const uint32_t (&cur)[64], uint32_t (&nxt)[64], uint32_t b) {
        uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], gg = h[6], hh = h[7];
        sha256_rounds<0, 16>(a, bb, c, d, e, f, gg, hh, cur);
        const bool more = b + 1 < nblocks;
        if (more) fetch(nxt, b + 2);
        sha256_rounds<16, 64>(a, bb, c, d, e, f, gg, hh, cur);
        if (more) lds_release(&taken[g], b + 2);             // (release: the reads above have landed) the producer may overwrite the slot
        h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += gg; h[7] += hh;
    };
    fetch(ka, 1);
    lds_release(&taken[g], 1);
    for (uint32_t b = 0; b < nblocks; b += 2) {
        block(ka, kb, b);
        if (b + 1 < nblocks) block(kb, ka, b + 1);
    }
    if (live) {
#pragma unroll
        for (int i = 0; i < 8; i++) st[(size_t)i * n_inst + j] = h[i];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Column hash straight from the encoder's Z matrix (round 6, LIG_ZRES=1): the last radix-8 pass of the row encoder (K3,
// ntt_encode.hip) runs INSIDE the hash's producer wave, so the codeword planes of stage 1 are never written to HBM and never
// read back (-1.5 MB of traffic per committed row); the Z tiles K2 wrote stay resident in their place and stages 2 / 3 take
// single radix-8 outputs from them (k_encode_out_dot_z, k_gather_rows_z).
//
// K3's unit of work is (row, coset, q2) -> the 8 columns q2 + B*q1; the hash's is (column) x all rows.  A group of 64 columns
// of plane r >= 1 is therefore 8 consecutive q2 x all 8 q1 (column-lane = q1*8 + q2'), and its producer wave works in batches
// of 8 rows: butterfly-lane = (row-in-batch, q2'), 8 canonical outputs each -> a 16 KiB LDS stage (wave-private: the transpose
// from (row, q2') x q1 to (q1, q2') x row is LDS write + read of the same wave, ordered by a wave-local fence) -> per row the
// column-lane picks up its element and, every second row, expands the message schedule of a block exactly as the producer of
// k_sha_update_rows_ws does.  The loads of the next batch are issued before the four schedule expansions of the current one.
// Plane 0 (coset 0 = the message row reversed) keeps the contiguous column-lanes and the plain producer.  The consumer wave is
// the one of k_sha_update_rows_ws: rounds only.  Handles every (nrows, rows_before) combination (a pending half block is
// picked up from / left in the state like everywhere else).
template <int LOG2B, int GROUPS>
__global__ void __launch_bounds__(GROUPS * 128) k_sha_update_rows_z(uint32_t* __restrict__ st, const fr* __restrict__ Z, size_t nrows, uint64_t rows_before,
                                                                     const fr* __restrict__ msgs, const f29wt w8) {
    constexpr uint32_t B = 1u << LOG2B, K = 8u * B, GPP = B / 8;      // groups per plane (k / 64)
    constexpr size_t n_inst = 4 * (size_t)K;
    __shared__ uint4 ring[GROUPS][16][64];
    __shared__ uint4 stage[GROUPS][16][64];
    __shared__ uint32_t ready[GROUPS], taken[GROUPS];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool producer = wave >= GROUPS;
    const uint32_t g = producer ? wave - GROUPS : wave;
    if (threadIdx.x < GROUPS) { ready[threadIdx.x] = 0; taken[threadIdx.x] = 0; }
    __syncthreads();
    const uint32_t group = blockIdx.x * GROUPS + g;
    if (group >= 4 * GPP) return;
    const uint32_t plane = group / GPP, gi = group - plane * GPP;
    const uint32_t q = plane == 0 ? gi * 64 + lane : 8 * gi + (lane & 7u) + B * (lane >> 3);
    const size_t j = (size_t)plane * K + q;
    const size_t pend = (size_t)(rows_before & 1);
    const size_t total = nrows + pend;
    const uint32_t nblocks = (uint32_t)(total / 2);
    uint4 (*slot)[64] = ring[g];

    if (producer) {
        // block b of this group: expand the schedule of w[0..16), add K, hand the 64 words to the consumer
        auto publish = [&](uint32_t (&w)[64], uint32_t b) {
#pragma unroll
            for (int i = 16; i < 64; i++) {
                const uint32_t w15 = w[i - 15], w2 = w[i - 2];
                w[i] = (w[i - 16] + xor3(rotr(w15, 7), rotr(w15, 18), w15 >> 3) + w[i - 7]) + xor3(rotr(w2, 17), rotr(w2, 19), w2 >> 10);
            }
#pragma unroll
            for (int i = 0; i < 64; i++) w[i] += SHA_K[i];
            while (__builtin_amdgcn_readfirstlane(lds_acquire(&taken[g])) < b) __builtin_amdgcn_s_sleep(2);
#pragma unroll
            for (int i = 0; i < 16; i++) slot[i][lane] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
            lds_release(&ready[g], b + 1);
        };
        fr half = fr_zero();
        bool have_half = pend != 0;
        if (have_half) {
#pragma unroll
            for (int i = 0; i < 8; i++) half.v[i] = st[(size_t)(8 + i) * n_inst + j];
        }
        uint32_t b = 0;
        if (plane == 0) {
            const fr* p0 = msgs + ((K - q) & (K - 1));
            fr nx = fr_load(p0);
            for (size_t r = 0; r < nrows; r++) {
                const fr e = nx;
                if (r + 1 < nrows) nx = fr_load(p0 + (r + 1) * (size_t)K);
                if (have_half) {
                    uint32_t w[64];
#pragma unroll
                    for (int i = 0; i < 8; i++) { w[i] = half.v[i]; w[8 + i] = e.v[i]; }
                    publish(w, b++);
                    have_half = false;
                } else {
                    half = e;
                    have_half = true;
                }
            }
        } else {
            uint4 (*stg)[64] = stage[g];
            const uint32_t rho = lane >> 3;
            const fr* zb = Z + (size_t)(plane - 1) * K + 8 * gi + (lane & 7u);      // + row * 3K + p * B
            fr in[8];
            auto load_batch = [&](size_t m0) {
                const size_t row = m0 + rho;
                // the base is made opaque HERE: otherwise the loads of the next batch are hoisted above the butterfly of the current one
                // (read-only, no-alias memory) and 64 more registers are live across it
                const fr* zr = zb + row * (3 * (size_t)K);
                asm volatile("" : "+v"(zr) : : "memory");
                if (row < nrows) {
#pragma unroll
                    for (int p = 0; p < 8; p++) in[p] = fr_load(zr + (size_t)brev3(p) * B);
                } else {
#pragma unroll
                    for (int p = 0; p < 8; p++) in[p] = fr_zero();
                }
            };
            load_batch(0);
#pragma unroll 1
            for (size_t m0 = 0; m0 < nrows; m0 += 8) {
                {
                    f29 a[8];
#pragma unroll
                    for (int p = 0; p < 8; p++) a[p] = unpack29(in[p]);
                    radix8_dit(a, w8);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the previous batch's stage reads are done (same wave, in order)
#pragma unroll
                    for (int q1 = 0; q1 < 8; q1++) {
                        const fr c = pack29(f29_canon(a[q1]));
                        stg[2 * rho][q1 * 8 + (lane & 7u)] = make_uint4(c.v[0], c.v[1], c.v[2], c.v[3]);
                        stg[2 * rho + 1][q1 * 8 + (lane & 7u)] = make_uint4(c.v[4], c.v[5], c.v[6], c.v[7]);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
                load_batch(m0 + 8);              // in flight under the schedule expansions below (unconditional: rows past the end load
                                                 // nothing, and `in` is dead across the butterfly instead of loop-carried)
                const uint32_t cnt = nrows - m0 < 8 ? (uint32_t)(nrows - m0) : 8u;
#pragma unroll 1
                for (uint32_t s = 0; s < cnt; s++) {
                    const uint4 lo = stg[2 * s][lane], hi = stg[2 * s + 1][lane];
                    if (have_half) {
                        uint32_t w[64];
#pragma unroll
                        for (int i = 0; i < 8; i++) w[i] = half.v[i];
                        w[8] = lo.x; w[9] = lo.y; w[10] = lo.z; w[11] = lo.w; w[12] = hi.x; w[13] = hi.y; w[14] = hi.z; w[15] = hi.w;
                        publish(w, b++);
                        have_half = false;
                    } else {
                        half.v[0] = lo.x; half.v[1] = lo.y; half.v[2] = lo.z; half.v[3] = lo.w;
                        half.v[4] = hi.x; half.v[5] = hi.y; half.v[6] = hi.z; half.v[7] = hi.w;
                        have_half = true;
                    }
                }
            }
        }
        if (have_half && nrows) {         // odd tail: keep the element for the next call / final
#pragma unroll
            for (int i = 0; i < 8; i++) st[(size_t)(8 + i) * n_inst + j] = half.v[i];
        }
        return;
    }

    // ---- consumer (as in k_sha_update_rows_ws)
    if (!nblocks) return;
    uint32_t h[8];
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = st[(size_t)i * n_inst + j];
    uint32_t ka[64], kb[64];
    auto fetch = [&](uint32_t (&dst)[64], uint32_t upto) {
        while (__builtin_amdgcn_readfirstlane(lds_acquire(&ready[g])) < upto) __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int i = 0; i < 16; i++) { const uint4 v = slot[i][lane]; dst[4 * i] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w; }
    };
    auto block = [&](const uint32_t (&cur)[64], uint32_t (&nxt)[64], uint32_t b) {
        uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], gg = h[6], hh = h[7];
        sha256_rounds<0, 16>(a, bb, c, d, e, f, gg, hh, cur);
        const bool more = b + 1 < nblocks;
        if (more) fetch(nxt, b + 2);
        sha256_rounds<16, 64>(a, bb, c, d, e, f, gg, hh, cur);
        if (more) lds_release(&taken[g], b + 2);
        h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += gg; h[7] += hh;
    };
    fetch(ka, 1);
    lds_release(&taken[g], 1);
    for (uint32_t b = 0; b < nblocks; b += 2) {
        block(ka, kb, b);
        if (b + 1 < nblocks) block(kb, ka, b + 1);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) st[(size_t)i * n_inst + j] = h[i];
}

// padding + length (shader/sha256.wgsl:180-224); does not modify the state
// pk != 0: plane-major instances (see k_sha_update_rows): instance j = r*pk + q is leaf 4q + r
__global__ void k_sha_final(const uint32_t* __restrict__ st, size_t n_inst, uint64_t rows_total, uint32_t* __restrict__ digests, uint32_t pk) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_inst) return;
    uint32_t h[8], w[16];
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = st[(size_t)i * n_inst + j];
    const uint64_t bits = rows_total * 256ull;
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = 0;
    if (rows_total & 1) {
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = st[(size_t)(8 + i) * n_inst + j];
        w[8] = 0x80000000u;
    } else {
        w[0] = 0x80000000u;
    }
    w[14] = (uint32_t)(bits >> 32);
    w[15] = (uint32_t)bits;
    sha256_compress(h, w);
    const size_t leaf = pk ? 4 * (j % pk) + j / pk : j;
    uint4* out = reinterpret_cast<uint4*>(digests + 8 * leaf);
    out[0] = make_uint4(h[0], h[1], h[2], h[3]);
    out[1] = make_uint4(h[4], h[5], h[6], h[7]);
}

void launch_sha_init(hipStream_t s, uint32_t* state, size_t n_inst) {
    hipLaunchKernelGGL(k_sha_init, dim3((uint32_t)((n_inst + 255) / 256)), dim3(256), 0, s, state, n_inst);
}
void launch_sha_update_rows(hipStream_t s, uint32_t* state, size_t n_inst, const fr* rows, size_t row_stride, size_t nrows,
                            uint64_t rows_before, uint32_t plane_k, const fr* msgs) {
    if (!nrows) return;
    // 256-thread workgroups = one hash wave on each of the 4 SIMDs of a CU: a hash wave keeps ~65% of its SIMD's VALU busy,
    // and the encode workgroups that share the chip finish with their slowest wave, so the hash load has to be the same
    // on all SIMDs of a CU (measured: encode kernels 2-3x slower next to 64-thread hash workgroups, 1.1-1.8x next to
    // 256-thread ones; tools/corun_bench.py).  LIG_SHA_BLOCK overrides for experiments.
    const uint32_t bs = lig::knobs().sha_block;
    const int ws = lig::knobs().sha_ws;                  // LIG_SHA_WS: 0 = one wave per 64 columns (rounds 1-4), 1 / 2 / 4 = wave-specialised, groups per workgroup
    if (ws && nrows + (rows_before & 1) >= 2) {
        const size_t groups = (n_inst + 63) / 64;
        if (ws == 1) hipLaunchKernelGGL(k_sha_update_rows_ws<1>, dim3((uint32_t)groups), dim3(128), 0, s, state, n_inst, rows, row_stride, nrows, rows_before, plane_k, msgs);
        else if (ws == 4) hipLaunchKernelGGL(k_sha_update_rows_ws<4>, dim3((uint32_t)((groups + 3) / 4)), dim3(512), 0, s, state, n_inst, rows, row_stride, nrows, rows_before, plane_k, msgs);
        else hipLaunchKernelGGL(k_sha_update_rows_ws<2>, dim3((uint32_t)((groups + 1) / 2)), dim3(256), 0, s, state, n_inst, rows, row_stride, nrows, rows_before, plane_k, msgs);
        return;
    }
    hipLaunchKernelGGL(k_sha_update_rows, dim3((uint32_t)((n_inst + bs - 1) / bs)), dim3(bs), 0, s, state, n_inst, rows, row_stride,
                       nrows, rows_before, plane_k, msgs);
}
// column hash of `nrows` rows whose cosets 1..3 are Z tiles (rows x 3k, the layout k_encode_tiles writes) and whose coset 0 is the
// message row; plane-major instances like launch_sha_update_rows(.., plane_k = k, msgs).  false: k outside the tiled encoder's range
template <int LOG2B>
static void sha_update_rows_z_t(hipStream_t s, uint32_t* state, const fr* z, size_t nrows, uint64_t rows_before, const fr* msgs, const f29wt w8) {
    constexpr uint32_t groups = 4u * ((1u << LOG2B) / 8u);
    if (lig::knobs().sha_ws == 1) hipLaunchKernelGGL((k_sha_update_rows_z<LOG2B, 1>), dim3(groups), dim3(128), 0, s, state, z, nrows, rows_before, msgs, w8);
    else hipLaunchKernelGGL((k_sha_update_rows_z<LOG2B, 2>), dim3(groups / 2), dim3(256), 0, s, state, z, nrows, rows_before, msgs, w8);
}
bool launch_sha_update_rows_z(hipStream_t s, uint32_t* state, const EncodePlan& ep, const fr* z, size_t nrows, uint64_t rows_before, const fr* msgs) {
    if (!nrows) return true;
    switch (ep.log2B) {
        case 6: sha_update_rows_z_t<6>(s, state, z, nrows, rows_before, msgs, ep.w8_fwd); return true;
        case 7: sha_update_rows_z_t<7>(s, state, z, nrows, rows_before, msgs, ep.w8_fwd); return true;
        case 8: sha_update_rows_z_t<8>(s, state, z, nrows, rows_before, msgs, ep.w8_fwd); return true;
        case 9: sha_update_rows_z_t<9>(s, state, z, nrows, rows_before, msgs, ep.w8_fwd); return true;
        case 10: sha_update_rows_z_t<10>(s, state, z, nrows, rows_before, msgs, ep.w8_fwd); return true;
        case 11: sha_update_rows_z_t<11>(s, state, z, nrows, rows_before, msgs, ep.w8_fwd); return true;
        case 12: sha_update_rows_z_t<12>(s, state, z, nrows, rows_before, msgs, ep.w8_fwd); return true;
        default: return false;
    }
}
void launch_sha_final(hipStream_t s, const uint32_t* state, size_t n_inst, uint64_t rows_total, uint32_t* digests, uint32_t plane_k) {
    hipLaunchKernelGGL(k_sha_final, dim3((uint32_t)((n_inst + 63) / 64)), dim3(64), 0, s, state, n_inst, rows_total, digests, plane_k);
}

// ---- Merkle tree: node[i] = SHA256(node[2i+1] || node[2i+2]) over canonical digest BYTES
// (include/zkp/merkle_tree.hpp:361-375); one launch per level, heap layout, missing leaves = zero digests.
__global__ void k_merkle_level(uint32_t* __restrict__ nodes, size_t first, size_t count) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const size_t i = first + t;
    const uint4* ch = reinterpret_cast<const uint4*>(nodes + 8 * (2 * i + 1));
    uint4 q[4] = {ch[0], ch[1], ch[2], ch[3]};
    uint32_t w[16] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w,
                      q[2].x, q[2].y, q[2].z, q[2].w, q[3].x, q[3].y, q[3].z, q[3].w};
#pragma unroll
    for (int k = 0; k < 16; k++) w[k] = __builtin_bswap32(w[k]);   // bytes -> big-endian message words
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    sha256_compress(h, w);
#pragma unroll
    for (int k = 0; k < 16; k++) w[k] = 0;
    w[0] = 0x80000000u;
    w[15] = 512;
    sha256_compress(h, w);
    uint4* out = reinterpret_cast<uint4*>(nodes + 8 * i);
    out[0] = make_uint4(__builtin_bswap32(h[0]), __builtin_bswap32(h[1]), __builtin_bswap32(h[2]), __builtin_bswap32(h[3]));
    out[1] = make_uint4(__builtin_bswap32(h[4]), __builtin_bswap32(h[5]), __builtin_bswap32(h[6]), __builtin_bswap32(h[7]));
}

// the levels of at most 256 nodes in ONE launch: a single workgroup walks up the tree, a barrier between levels (the nodes go
// through global memory: the writes of a level are made visible to the whole workgroup by the fence + barrier)
__global__ void __launch_bounds__(256) k_merkle_top(uint32_t* __restrict__ nodes, uint32_t top_width) {
    for (uint32_t width = top_width; width >= 1; width >>= 1) {
        if (threadIdx.x < width) {
            const size_t i = (size_t)(width - 1) + threadIdx.x;
            const uint4* ch = reinterpret_cast<const uint4*>(nodes + 8 * (2 * i + 1));
            uint4 q[4] = {ch[0], ch[1], ch[2], ch[3]};
            uint32_t w[16] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w,
                              q[2].x, q[2].y, q[2].z, q[2].w, q[3].x, q[3].y, q[3].z, q[3].w};
#pragma unroll
            for (int k = 0; k < 16; k++) w[k] = __builtin_bswap32(w[k]);
            uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
            sha256_compress(h, w);
#pragma unroll
            for (int k = 0; k < 16; k++) w[k] = 0;
            w[0] = 0x80000000u;
            w[15] = 512;
            sha256_compress(h, w);
            uint4* out = reinterpret_cast<uint4*>(nodes + 8 * i);
            out[0] = make_uint4(__builtin_bswap32(h[0]), __builtin_bswap32(h[1]), __builtin_bswap32(h[2]), __builtin_bswap32(h[3]));
            out[1] = make_uint4(__builtin_bswap32(h[4]), __builtin_bswap32(h[5]), __builtin_bswap32(h[6]), __builtin_bswap32(h[7]));
        }
        __threadfence();
        __syncthreads();
    }
}

// `leaves` may already BE the leaf level of `nodes` (nodes + 8 * (P - 1), P = bit_ceil(n_leaves)): then nothing is copied
void launch_merkle_build(hipStream_t s, const uint32_t* leaves, size_t n_leaves, uint32_t* nodes) {
    size_t P = 1;
    while (P < n_leaves) P <<= 1;
    const bool in_place = leaves == nodes + 8 * (P - 1);
    if (!in_place || n_leaves < P) {
        if (n_leaves < P) (void)hipMemsetAsync(nodes + 8 * (P - 1 + n_leaves), 0, 32 * (P - n_leaves), s);      // missing leaves = zero digests
        if (!in_place) (void)hipMemcpyAsync(nodes + 8 * (P - 1), leaves, 32 * n_leaves, hipMemcpyDeviceToDevice, s);
    }
    size_t width = P / 2;
    for (; width > 256; width /= 2) {
        const size_t first = width - 1;
        hipLaunchKernelGGL(k_merkle_level, dim3((uint32_t)((width + 63) / 64)), dim3(64), 0, s, nodes, first, width);
    }
    if (width >= 1) hipLaunchKernelGGL(k_merkle_top, dim3(1), dim3(256), 0, s, nodes, (uint32_t)width);
}

}  // namespace lig
