// fr.hpp -- BN254 scalar-field arithmetic for gfx950 (CDNA4), device side.
//
// Replaces shader/bigint.wgsl.in + shader/bn254fr.wgsl.in of the reference (WGSL, 2 x vec4u, 32x32
// multiplies emulated with four 16-bit products).  Here an element is 8 x u32 limbs in VGPRs and every
// partial product is one v_mad_u64_u32 (32x32+64 -> 64).  Values in HBM are always canonical residues
// in [0,p), little-endian limbs, exactly the reference's buffer format
// (include/ligetron/webgpu/device_bignum.hpp:76-86), so results are bit-identical to the reference's
// whatever reduction strategy is used internally.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lig {

// view of a windowed twiddle table on the device (fr29.hpp: f29_mulw): seven planes of 16-byte words, `stride` entries each
// (the entry index is kept apart from the uniform base so that all seven loads of an entry share one 32-bit lane offset)
struct f29wt {
    const uint4* base;
    uint32_t stride;
    uint32_t idx;
    __host__ __device__ f29wt operator+(size_t i) const { return f29wt{base, stride, idx + (uint32_t)i}; }
};


struct alignas(16) fr {
    uint32_t v[8];
};

// p, 2p (shader/bn254fr.wgsl.in:19-28), -p^-1 mod 2^32, R mod p (:36-39), R^2 mod p, mu = floor(2^508/p) (:41-45)
__device__ __constant__ static const uint32_t FR_P[8] = {0xF0000001u, 0x43E1F593u, 0x79B97091u, 0x2833E848u,
                                                         0x8181585Du, 0xB85045B6u, 0xE131A029u, 0x30644E72u};
__device__ __constant__ static const uint32_t FR_2P[8] = {0xE0000002u, 0x87C3EB27u, 0xF372E122u, 0x5067D090u,
                                                          0x0302B0BAu, 0x70A08B6Du, 0xC2634053u, 0x60C89CE5u};
__device__ __constant__ static const uint32_t FR_R[8] = {0x4FFFFFFBu, 0xAC96341Cu, 0x9F60CD29u, 0x36FC7695u,
                                                         0x7879462Eu, 0x666EA36Fu, 0x9A07DF2Fu, 0x0E0A77C1u};
__device__ __constant__ static const uint32_t FR_R2[8] = {0xAE216DA7u, 0x1BB8E645u, 0xE35C59E3u, 0x53FE3AB1u,
                                                          0x53BB8085u, 0x8C49833Du, 0x7F4E44A5u, 0x0216D0B1u};
static constexpr uint32_t FR_N0INV = 0xEFFFFFFFu;   // -p^-1 mod 2^32

// compile-time copies so that fully unrolled code uses literal operands instead of constant loads
#define LIG_P0 0xF0000001u
#define LIG_P1 0x43E1F593u
#define LIG_P2 0x79B97091u
#define LIG_P3 0x2833E848u
#define LIG_P4 0x8181585Du
#define LIG_P5 0xB85045B6u
#define LIG_P6 0xE131A029u
#define LIG_P7 0x30644E72u

__device__ __forceinline__ constexpr uint32_t fr_p_limb(int i) {
    return i == 0 ? LIG_P0 : i == 1 ? LIG_P1 : i == 2 ? LIG_P2 : i == 3 ? LIG_P3
         : i == 4 ? LIG_P4 : i == 5 ? LIG_P5 : i == 6 ? LIG_P6 : LIG_P7;
}

__device__ __forceinline__ fr fr_load(const fr* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    fr r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
__device__ __forceinline__ void fr_store(fr* p, const fr& x) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    q[1] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
}
__device__ __forceinline__ fr fr_zero() {
    fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
}
__device__ __forceinline__ fr fr_const(const uint32_t* c) {
    fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = c[i];
    return r;
}
__device__ __forceinline__ bool fr_is_zero(const fr& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0;
}

// r = a + b (no reduction), returns carry out of 256 bits
__device__ __forceinline__ uint32_t add256(fr& r, const fr& a, const fr& b) {
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.v[i] + b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    return (uint32_t)c;
}
// r = a - b, returns borrow (1 if a < b)
__device__ __forceinline__ uint32_t sub256(fr& r, const fr& a, const fr& b) {
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (int64_t)a.v[i] - (int64_t)b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;   // arithmetic shift keeps the borrow as -1
    }
    return (uint32_t)(c & 1);
}
__device__ __forceinline__ uint32_t sub256_p(fr& r, const fr& a) {   // r = a - p
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (int64_t)a.v[i] - (int64_t)fr_p_limb(i);
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    return (uint32_t)(c & 1);
}
__device__ __forceinline__ void select256(fr& r, bool take_b, const fr& a, const fr& b) {
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = take_b ? b.v[i] : a.v[i];
}
// conditional subtract of p: a in [0,2p) -> [0,p)   (bn254fr_reduce, shader/bn254fr.wgsl.in:50-58)
__device__ __forceinline__ fr fr_reduce_once(const fr& a) {
    fr t, r;
    uint32_t borrow = sub256_p(t, a);
    select256(r, borrow == 0, a, t);
    return r;
}
// a,b in [0,p) -> (a+b) mod p
__device__ __forceinline__ fr fr_add(const fr& a, const fr& b) {
    fr s;
    add256(s, a, b);            // < 2p < 2^256
    return fr_reduce_once(s);
}
// a,b in [0,p) -> (a-b) mod p
__device__ __forceinline__ fr fr_sub(const fr& a, const fr& b) {
    fr d, e, r;
    uint32_t borrow = sub256(d, a, b);
    fr pp = fr_const(FR_P);
    add256(e, d, pp);
    select256(r, borrow != 0, d, e);
    return r;
}
__device__ __forceinline__ fr fr_neg(const fr& a) {
    fr z = fr_zero();
    return fr_sub(z, a);
}

// ---------------------------------------------------------------------------------------------
// Montgomery product a*b*2^-256 mod p, result in [0,2p) when a < 4p and b < p (no final subtract).
// CIOS with the "no carry" fusion (top limb of p < 2^31 so the running value never needs a 9th limb):
// per outer step 8 v_mad_u64_u32 for a*b_i, one v_mul_lo_u32 for m, 8 v_mad_u64_u32 for m*p.
// Replaces montgomery_mul / montgomery_mul_2p (shader/bn254fr.wgsl.in:76-109).
__device__ __forceinline__ fr fr_montmul_lazy(const fr& a, const fr& b) {
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t bi = b.v[i];
        uint64_t c1 = (uint64_t)a.v[0] * bi + t[0];
        const uint32_t m = (uint32_t)c1 * FR_N0INV;
        uint64_t c2 = (uint64_t)m * fr_p_limb(0) + (uint32_t)c1;
        c1 >>= 32;
        c2 >>= 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            c1 += (uint64_t)a.v[j] * bi + t[j];
            c2 += (uint64_t)m * fr_p_limb(j) + (uint32_t)c1;
            t[j - 1] = (uint32_t)c2;
            c1 >>= 32;
            c2 >>= 32;
        }
        t[7] = (uint32_t)c1 + (uint32_t)c2;
    }
    fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    return r;
}
// canonical result
__device__ __forceinline__ fr fr_montmul(const fr& a, const fr& b) { return fr_reduce_once(fr_montmul_lazy(a, b)); }
// plain x plain -> plain:  montmul(montmul(a,b), R^2) = a*b   (same residue as the reference's Barrett path,
// barrett_reduce_wide shader/bn254fr.wgsl.in:113-124, since both return the canonical representative)
__device__ __forceinline__ fr fr_mul(const fr& a, const fr& b) {
    fr t = fr_montmul_lazy(a, b);            // a*b/R  in [0,2p)
    return fr_montmul(t, fr_const(FR_R2));   // (a*b/R)*R^2/R = a*b
}
__device__ __forceinline__ fr fr_to_mont(const fr& a) { return fr_montmul(a, fr_const(FR_R2)); }

}  // namespace lig
