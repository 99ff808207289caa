// host_field.hpp -- host-side BN254 Fr used by the product for one-time table generation (twiddles,
// twists, N^-1) and for scalars handed to kernels.  Counterpart of the GMP code the reference runs in
// webgpu_context::ntt_precompute_omegas (src/webgpu/engine.cpp:1382-1503) and bn254_gmp
// (src/bn254.cpp:21-64); written with 4 x u64 limbs and unsigned __int128 instead of GMP.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>

namespace lig {
namespace host {

using u128 = unsigned __int128;

struct Fr {
    uint64_t v[4];
    bool operator==(const Fr& o) const { return !std::memcmp(v, o.v, 32); }
};

static constexpr Fr P = {{0x43E1F593F0000001ull, 0x2833E84879B97091ull, 0xB85045B68181585Dull, 0x30644E72E131A029ull}};
static constexpr Fr R1 = {{0xAC96341C4FFFFFFBull, 0x36FC76959F60CD29ull, 0x666EA36F7879462Eull, 0x0E0A77C19A07DF2Full}};  // R mod p
static constexpr Fr R2 = {{0x1BB8E645AE216DA7ull, 0x53FE3AB1E35C59E3ull, 0x8C49833D53BB8085ull, 0x0216D0B17F4E44A5ull}};  // R^2 mod p
static constexpr uint64_t N0INV64 = 0xC2E1F593EFFFFFFFull;  // -p^-1 mod 2^64
// root1 = 7^((p-1)/2^28)  (src/bn254.cpp:36-37); root2 = root1^(2^61-1) is derived (src/bn254.cpp:38-39)
static constexpr Fr ROOT1 = {{0xd34f1ed960c37c9cull, 0x3215cf6dd39329c8ull, 0x98865ea93dd31f74ull, 0x03ddb9f5166d18b7ull}};

inline Fr from_u64(uint64_t x) { return Fr{{x, 0, 0, 0}}; }
inline bool geq(const Fr& a, const Fr& b) {
    for (int i = 3; i >= 0; i--) {
        if (a.v[i] != b.v[i]) return a.v[i] > b.v[i];
    }
    return true;
}
inline Fr sub_nored(const Fr& a, const Fr& b) {
    Fr r; uint64_t br = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)a.v[i] - b.v[i] - br; r.v[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
    return r;
}
inline Fr add(const Fr& a, const Fr& b) {
    Fr r; u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a.v[i] + b.v[i]; r.v[i] = (uint64_t)c; c >>= 64; }
    return geq(r, P) ? sub_nored(r, P) : r;
}
inline Fr sub(const Fr& a, const Fr& b) {
    if (geq(a, b)) return sub_nored(a, b);
    return add(a, sub_nored(P, b));   // a + (p - b) < p because b > a
}
inline Fr neg(const Fr& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) ? sub_nored(P, a) : a; }

// Montgomery product (CIOS, 64-bit limbs): a*b*R^-1 mod p, canonical
inline Fr montmul(const Fr& a, const Fr& b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a.v[j] * b.v[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * N0INV64;
        c = (u128)m * P.v[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * P.v[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    Fr r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || geq(r, P)) r = sub_nored(r, P);
    return r;
}
inline Fr to_mont(const Fr& a) { return montmul(a, R2); }
inline Fr from_mont(const Fr& a) { return montmul(a, from_u64(1)); }
// plain * plain -> plain
inline Fr mul(const Fr& a, const Fr& b) { return montmul(montmul(a, b), R2); }
inline Fr pow(const Fr& a, const Fr& e) {
    Fr acc = from_u64(1), base = a;
    for (int i = 0; i < 256; i++) {
        if ((e.v[i >> 6] >> (i & 63)) & 1) acc = mul(acc, base);
        base = mul(base, base);
    }
    return acc;
}
inline Fr pow_u64(const Fr& a, uint64_t e) { return pow(a, from_u64(e)); }
inline Fr inv(const Fr& a) { Fr e = P; e.v[0] -= 2; return pow(a, e); }

// bn254_gmp::generate_omegas (src/bn254.cpp:51-64)
inline void omegas(uint32_t k, Fr& wk, Fr& w2k, Fr& w4k) {
    const Fr root2 = pow_u64(ROOT1, (1ull << 61) - 1);
    const uint64_t top = 1ull << 28;
    wk = pow_u64(ROOT1, top / k);
    w2k = pow_u64(ROOT1, top / (2ull * k));
    w4k = pow_u64(root2, top / (4ull * k));
}

}  // namespace host
}  // namespace lig
