// fr29.hpp -- the arithmetic core of the hot kernels: BN254 Fr in 9 limbs of 29 bits (redundant / lazy).
//
// Why 29-bit limbs on gfx950 (measured with tools/ubench_valu.hip, profiles/r01_ubench_valu_issue_rates.txt):
// v_mad_u64_u32 (32x32+64 -> 64) issues at HALF rate, and so do the carry instructions v_add_co/v_addc and the
// 64-bit v_lshl_add_u64, while plain v_add_u32 / v_and / v_mov are full rate.  A 8 x 32-bit Montgomery product
// therefore spends more issue slots on carries and register moves (~490) than on its 136 multiplies.  With
// 29-bit limbs a whole column of 18 partial products (9 of a*b, 9 of m*p) accumulates in one 64-bit register
// pair with NO carry handling: the product is a chain of v_mad_u64_u32 whose addend is the previous result,
// plus one 64-bit shift per column.  Additions are 9 independent v_add_u32, subtractions add a precomputed
// borrow-proof multiple of p, and limbs are renormalised only once per radix-4 step.
//
//   value(x) = sum_i x.v[i] * 2^(29 i);  "normalised": v[0..7] < 2^29;  "lazy": v[i] < 2^32, value < 2^261
//   Montgomery radix R' = 2^261: mont(a, w*R') = a*w  (twiddles are stored as w*R' mod p, normalised)
//
// HBM / proof format stays the reference's canonical 8 x u32 (include/ligetron/webgpu/device_bignum.hpp:76-86):
// unpack29 / pack29 convert at kernel boundaries.  Replaces shader/bigint.wgsl.in + shader/bn254fr.wgsl.in.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fr.hpp"
#include "fr29_consts.hpp"

namespace lig {

struct f29 {
    uint32_t v[9];
};
// table entry: 9 limbs padded to 48 bytes so that it is fetched with three 16-byte loads
struct alignas(16) f29s {
    uint32_t v[12];
};

__device__ __forceinline__ f29 f29_load_tab(const f29s* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1];
    const uint32_t c = reinterpret_cast<const uint32_t*>(p)[8];
    f29 r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    r.v[8] = c;
    return r;
}

// canonical / < 2^256 value in 8 x u32  ->  normalised 29-bit limbs
__device__ __forceinline__ f29 unpack29(const fr& w) {
    f29 r;
    r.v[0] = w.v[0] & F29_MASK;
    r.v[1] = __builtin_amdgcn_alignbit(w.v[1], w.v[0], 29) & F29_MASK;
    r.v[2] = __builtin_amdgcn_alignbit(w.v[2], w.v[1], 26) & F29_MASK;
    r.v[3] = __builtin_amdgcn_alignbit(w.v[3], w.v[2], 23) & F29_MASK;
    r.v[4] = __builtin_amdgcn_alignbit(w.v[4], w.v[3], 20) & F29_MASK;
    r.v[5] = __builtin_amdgcn_alignbit(w.v[5], w.v[4], 17) & F29_MASK;
    r.v[6] = __builtin_amdgcn_alignbit(w.v[6], w.v[5], 14) & F29_MASK;
    r.v[7] = __builtin_amdgcn_alignbit(w.v[7], w.v[6], 11) & F29_MASK;
    r.v[8] = w.v[7] >> 8;
    return r;
}
// normalised limbs, value < 2^256  ->  8 x u32
__device__ __forceinline__ fr pack29(const f29& x) {
    fr w;
    w.v[0] = x.v[0] | (x.v[1] << 29);
    w.v[1] = (x.v[1] >> 3) | (x.v[2] << 26);
    w.v[2] = (x.v[2] >> 6) | (x.v[3] << 23);
    w.v[3] = (x.v[3] >> 9) | (x.v[4] << 20);
    w.v[4] = (x.v[4] >> 12) | (x.v[5] << 17);
    w.v[5] = (x.v[5] >> 15) | (x.v[6] << 14);
    w.v[6] = (x.v[6] >> 18) | (x.v[7] << 11);
    w.v[7] = (x.v[7] >> 21) | (x.v[8] << 8);
    return w;
}

__device__ __forceinline__ uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) { return (uint64_t)a * b + c; }
// p[0] = 2^28 + 1: hipcc strength-reduces m * p[0] + acc into two 64-bit shift-adds (2 half-rate ops + moves);
// keeping the constant opaque in an SGPR leaves it as one v_mad_u64_u32.
__device__ __forceinline__ uint32_t f29_p0_opaque() {
    uint32_t p0 = F29_P(0);
    asm volatile("" : "+s"(p0));
    return p0;
}

// Montgomery product a*b/2^261 mod p.  a: lazy, limbs < 1.25 * 2^31 (top limb < 2^31); b: normalised, < p.
// Result: normalised limbs, value < a*p/2^261 + p  (< 1.2 p for value(a) < 32 p).
// 81 + 81 v_mad_u64_u32 chained through one 64-bit accumulator, 9 v_mul_lo_u32, 17 v_lshrrev_b64.  The columns are
// written as inline-asm blocks: left to itself hipcc starts every column in a fresh accumulator and joins it to the
// carried sum with a v_lshl_add_u64 (17 extra quarter-rate instructions per product, to shorten a dependency chain that
// four waves per SIMD hide anyway).
// Column bound: 9 * (1.25*2^31 * 2^29) + 9 * 2^58 + carry < 2^64.
__device__ __forceinline__ f29 f29_montmul(const f29& a, const f29& b) {
    uint32_t m[9];
    f29 t;
    uint64_t acc = 0;
#ifdef LIG_MONTMUL_PLAIN             // experiment: the same 17 columns left to hipcc (profiles/r02_montmul_plain_ab.md)
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = 0; i < 9; i++) { const int j = k - i; if (j >= 0 && j < 9) acc = mad64(a.v[i], b.v[j], acc); }
#pragma unroll
        for (int i = 0; i < 9; i++) { const int j = k - i; if (j >= 1 && j < 9) acc = mad64(m[i], F29_P(j), acc); }
        if (k < 9) { m[k] = ((uint32_t)acc * F29_N0) & F29_MASK; acc = mad64(m[k], f29_p0_opaque(), acc); }
        if (k >= 9) t.v[k - 9] = (uint32_t)acc & F29_MASK;
        acc >>= 29;
    }
    t.v[8] = (uint32_t)acc;
#else
    const uint32_t p0 = f29_p0_opaque();
#include "fr29_montmul_gen.hpp"      // the 17 columns, one chained v_mad_u64_u32 block each (tools/gen_fr29_montmul.py)
#endif
    return t;
}

// ---- products with a TABLE operand in windowed form.  Every multiply of the encoder has a constant from a table on one
// side (stage twiddles, twists, seams).  A windowed entry for the constant w stores W_g = w * 2^(87 g) * 2^116 mod p, g = 0, 1, 2;
// with a = A_0 + A_1 2^87 + A_2 2^174 the sum A_0 W_0 + A_1 W_1 + A_2 W_2 = a * w * 2^116 (mod p) is below 2^91 p, so FOUR
// Montgomery steps bring it back to 9 limbs: 81 + 36 = 117 v_mad_u64_u32, 4 v_mul_lo_u32, 12 v_lshrrev_b64 against
// 162 / 9 / 17 for f29_montmul; the entry is 108 bytes instead of 36 (seven 16-byte loads).
// a: lazy, limbs <= 2.5 * 2^30 + 8.  Result: normalised limbs, value < p (1 + 2^-20) -- tighter than f29_montmul's 1.2 p.
// Column bound: 9 * (2.5*2^30 + 8) * 2^29 + 4 * 2^58 + carry < 2^64.
struct f29w {            // host-side entry
    uint32_t v[28];      // v[9 g + j] = limb j of W_g; v[27] unused
};
struct f29wv {
    uint32_t v[28];
};
// Device layout: SEVEN PLANES of 16-byte words, plane i holding words 4i .. 4i+3 of every entry (`stride` entries per plane).
// Adjacent lanes use adjacent entries in every table of the encoder, so each of the seven loads of a product touches 8
// consecutive 128-byte lines per wave instead of 64 scattered ones (array-of-entries measured 22 % SLOWER than the 36-byte
// Montgomery-form tables it replaced: the L1 tag rate, not the multiplier, was the limit).
// (struct f29wt {base, stride}: fr.hpp, shared with the host-side plan structs)
// Seven buffer loads that share ONE 32-bit lane offset: the plane offset travels in the scalar offset operand of
// buffer_load_dwordx4, so a table access costs no vector address arithmetic at all (with global loads the compiler formed a
// 64-bit address per plane: 6.6 v_lshl_add_u64 per product in the tile kernel).
__device__ __forceinline__ f29wv f29_load_w(const f29wt p) {
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    f29wv r;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(p.base), 0, -1, 0x00020000);   // raw buffer, gfx9 data format word
    const uint32_t off = p.idx << 4;
    const uint32_t plane = p.stride << 4;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const v4u x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, i * plane, 0);
        r.v[4 * i] = x.x; r.v[4 * i + 1] = x.y; r.v[4 * i + 2] = x.z; r.v[4 * i + 3] = x.w;
    }
    return r;
}
__device__ __forceinline__ f29 f29_mulw(const f29& a, const f29wv& W) {
    uint32_t m[4];
    f29 t;
    uint64_t acc = 0;
    const uint32_t p0 = f29_p0_opaque();
#include "fr29_mulw_gen.hpp"         // 12 columns, one chained v_mad_u64_u32 block each (tools/gen_fr29_mulw.py)
    return t;
}
// A windowed constant that is the SAME for every lane (the radix-8 constants): held in scalar registers.  The entry index must
// be uniform; the words come through scalar loads (uniform address, read-only table) and enter the multiplies as SGPR operands
// -- no vector registers, no vector memory instruction, no wait on the vector memory counter in front of the product.
struct f29ws {
    uint32_t v[28];
};
__device__ __forceinline__ f29ws f29_load_ws(const f29wt p) {
    // constant address space + uniform address = s_load_dwordx4 (the tables are never written while a kernel runs)
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    typedef const v4u __attribute__((address_space(4))) * cptr;
    f29ws r;
    const cptr b = (cptr)(uintptr_t)(p.base + p.idx);
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const v4u x = b[(size_t)i * p.stride];
        r.v[4 * i] = x.x; r.v[4 * i + 1] = x.y; r.v[4 * i + 2] = x.z; r.v[4 * i + 3] = x.w;
    }
    return r;
}
__device__ __forceinline__ f29 f29_mulw(const f29& a, const f29ws& W) {
    uint32_t m[4];
    f29 t;
    uint64_t acc = 0;
    const uint32_t p0 = f29_p0_opaque();
#include "fr29_mulw_s_gen.hpp"       // the same 12 columns with the constant's words as scalar operands (gen_fr29_mulw.py --uniform)
    return t;
}
// uniform access to both table formats (tile_dft.hpp is shared by kernels on either)
__device__ __forceinline__ f29 tab_get(const f29s* p) { return f29_load_tab(p); }
__device__ __forceinline__ f29wv tab_get(const f29wt p) { return f29_load_w(p); }
__device__ __forceinline__ f29 tab_mul(const f29& a, const f29& w) { return f29_montmul(a, w); }
__device__ __forceinline__ f29 tab_mul(const f29& a, const f29wv& w) { return f29_mulw(a, w); }
__device__ __forceinline__ f29 tab_mul(const f29& a, const f29ws& w) { return f29_mulw(a, w); }
// a table entry whose index is the same for every lane
__device__ __forceinline__ f29 tab_get_uniform(const f29s* p) { return f29_load_tab(p); }
#ifdef LIG_NO_UNIFORM_CONSTS      // A/B: the radix-8 constants through vector loads into vector registers, as before round 3
__device__ __forceinline__ f29wv tab_get_uniform(const f29wt p) { return f29_load_w(p); }
#else
__device__ __forceinline__ f29ws tab_get_uniform(const f29wt p) { return f29_load_ws(p); }
#endif

// limb-wise lazy add
__device__ __forceinline__ f29 f29_add(const f29& a, const f29& b) {
    f29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}
// a - b + K with K a borrow-proof multiple of p (every K limb >= the matching limb of b, value(K) >= value(b))
#define F29_SUBK(r, a, b, KMAC)                                              \
    do {                                                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < 9; i_++)(r).v[i_] = (a).v[i_] + (KMAC(i_) - (b).v[i_]); \
    } while (0)
__device__ __forceinline__ f29 f29_sub_k2(const f29& a, const f29& b) { f29 r; F29_SUBK(r, a, b, F29_K2P_C1); return r; }   // b normalised, < 2p
__device__ __forceinline__ f29 f29_sub_k4(const f29& a, const f29& b) { f29 r; F29_SUBK(r, a, b, F29_K4P_C2); return r; }   // b limbs < 2^30, < 4p
__device__ __forceinline__ f29 f29_sub_k8(const f29& a, const f29& b) { f29 r; F29_SUBK(r, a, b, F29_K8P_C4); return r; }   // b limbs < 2^31, < 8p
__device__ __forceinline__ f29 f29_sub_k16(const f29& a, const f29& b) { f29 r; F29_SUBK(r, a, b, F29_K16P_C2); return r; } // b limbs < 2^30, < 16p

// parallel carry pass: limbs < 2^32 -> limbs < 2^29 + 8 (top limb absorbs), value unchanged
__device__ __forceinline__ f29 f29_qnorm(const f29& a) {
    f29 r;
    r.v[0] = a.v[0] & F29_MASK;
#pragma unroll
    for (int i = 1; i < 8; i++) r.v[i] = (a.v[i] & F29_MASK) + (a.v[i - 1] >> 29);
    r.v[8] = a.v[8] + (a.v[7] >> 29);
    return r;
}

// V mod p approximately: lazy value < 2^261 (limbs < 2^32) -> normalised, value in [0, 2p).
// q = floor(V/p) or one less from the top ~30 bits; V - q*p is formed as (V + q*(2^261 - p)) mod 2^261 in one
// 64-bit carry chain (18 v_mad_u64_u32 + 9 shifts).
__device__ __forceinline__ f29 f29_reduce_2p(const f29& a) {
    const uint32_t hh = (a.v[8] << 3) + (a.v[7] >> 26);                       // ~ V / 2^229 (never above)
    const uint32_t q = (uint32_t)(((uint64_t)hh * F29_RECIP229) >> 56);       // <= floor(V / p)
    f29 r;
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        acc = mad64(a.v[i], 1u, acc);
        acc = mad64(q, F29_PBAR(i), acc);
        r.v[i] = (uint32_t)acc & F29_MASK;        // the top limb mask drops q * 2^261
        acc >>= 29;
    }
    return r;
}
// exact canonical residue: lazy value < 2^261 -> normalised limbs, value in [0, p)
__device__ __forceinline__ f29 f29_canon(const f29& a) {
    const f29 r = f29_reduce_2p(a);
    f29 d;
    int32_t br = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int32_t s = (int32_t)r.v[i] - (int32_t)F29_P(i) + br;
        d.v[i] = (uint32_t)s & F29_MASK;
        br = s >> 29;                              // arithmetic: 0 or -1
    }
    // top limb: a negative final value shows as br = -1 after limb 8 (limb 8 of r < 2^25)
    f29 o;
    const bool neg = br < 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o.v[i] = neg ? r.v[i] : d.v[i];
    return o;
}

__device__ __forceinline__ f29 f29_zero() {
    f29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = 0;
    return r;
}
__device__ __forceinline__ f29 f29_const_r2() {
    return f29{{F29_R2(0), F29_R2(1), F29_R2(2), F29_R2(3), F29_R2(4), F29_R2(5), F29_R2(6), F29_R2(7), F29_R2(8)}};
}

}  // namespace lig
