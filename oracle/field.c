/*
 * field.c -- BN254 scalar field, restating the reference's WGSL big-integer and
 * field routines with 4 x u64 limbs.  TEST INFRASTRUCTURE ONLY (see lig_oracle.h).
 *
 * Reference: shader/bigint.wgsl.in:29-340 (add/sub with carry, schoolbook
 * mul wide/lo/hi), shader/bn254fr.wgsl.in:17-168 (constants, reduce,
 * Montgomery, Barrett, powmod), src/bn254.cpp:21-64,110-121 (host twins,
 * roots of unity).
 */
#include "lig_oracle.h"
#include <string.h>

typedef unsigned __int128 u128;

/* shader/bn254fr.wgsl.in:19-45 */
const lo_fr LO_P  = {{0x43E1F593F0000001ull, 0x2833E84879B97091ull, 0xB85045B68181585Dull, 0x30644E72E131A029ull}};
const lo_fr LO_2P = {{0x87C3EB27E0000002ull, 0x5067D090F372E122ull, 0x70A08B6D0302B0BAull, 0x60C89CE5C2634053ull}};
const lo_fr LO_J  = {{0x3D1E0A6C10000001ull, 0x9A7979B4B396EE4Cull, 0x1C6567D766F9DC6Eull, 0x8C07D0E2F27CBE4Dull}};
const lo_fr LO_R  = {{0xAC96341C4FFFFFFBull, 0x36FC76959F60CD29ull, 0x666EA36F7879462Eull, 0x0E0A77C19A07DF2Full}};
const lo_fr LO_MU = {{0x620703A6BE1DE925ull, 0x7144852009E880AEull, 0xAB074A5868073014ull, 0x54A47462623A04A7ull}};
/* R^2 mod p (to enter Montgomery form with one montmul) */
static const lo_fr LO_R2 = {{0x1BB8E645AE216DA7ull, 0x53FE3AB1E35C59E3ull, 0x8C49833D53BB8085ull, 0x0216D0B17F4E44A5ull}};

/* src/bn254.cpp:36-43 */
static const lo_fr ROOT1 = {{0xd34f1ed960c37c9cull, 0x3215cf6dd39329c8ull, 0x98865ea93dd31f74ull, 0x03ddb9f5166d18b7ull}};
/* root2 = root1^(2^61-1) is computed in lo_omegas and checked in tests against src/bn254.cpp:39 */

void lo_fr_from_u64(lo_fr *o, uint64_t x) { o->v[0] = x; o->v[1] = o->v[2] = o->v[3] = 0; }

int lo_fr_cmp(const lo_fr *a, const lo_fr *b) {
    for (int i = 3; i >= 0; i--) {
        if (a->v[i] < b->v[i]) return -1;
        if (a->v[i] > b->v[i]) return 1;
    }
    return 0;
}
int lo_fr_is_zero(const lo_fr *a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }

/* bigint_add_cc / bigint_sub_cc (shader/bigint.wgsl.in:243-279) */
static uint64_t add_cc(lo_fr *o, const lo_fr *a, const lo_fr *b) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a->v[i] + b->v[i]; o->v[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static uint64_t sub_cc(lo_fr *o, const lo_fr *a, const lo_fr *b) {
    uint64_t br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->v[i] - b->v[i] - br;
        o->v[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1;
    }
    return br;
}
/* bn254fr_reduce (shader/bn254fr.wgsl.in:50-58) */
static void reduce_p(lo_fr *a) { lo_fr t; if (!sub_cc(&t, a, &LO_P)) *a = t; }

void lo_fr_add(lo_fr *o, const lo_fr *a, const lo_fr *b) { lo_fr t; add_cc(&t, a, b); reduce_p(&t); *o = t; }
void lo_fr_sub(lo_fr *o, const lo_fr *a, const lo_fr *b) {
    lo_fr t; if (sub_cc(&t, a, b)) add_cc(&t, &t, &LO_P); *o = t;
}
void lo_fr_neg(lo_fr *o, const lo_fr *a) { if (lo_fr_is_zero(a)) { *o = *a; return; } lo_fr t; sub_cc(&t, &LO_P, a); *o = t; }

/* bigint_mul_wide (shader/bigint.wgsl.in:307-340): schoolbook 256x256 -> 512 */
void lo_mul_wide(lo_wide *o, const lo_fr *a, const lo_fr *b) {
    uint64_t r[8] = {0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a->v[i] * b->v[j] + r[i + j];
            r[i + j] = (uint64_t)c; c >>= 64;
        }
        r[i + 4] = (uint64_t)c;
    }
    memcpy(o->v, r, sizeof r);
}
static void mul_lo(lo_fr *o, const lo_fr *a, const lo_fr *b) { lo_wide w; lo_mul_wide(&w, a, b); memcpy(o->v, w.v, 32); }
static void mul_hi(lo_fr *o, const lo_fr *a, const lo_fr *b) { lo_wide w; lo_mul_wide(&w, a, b); memcpy(o->v, w.v + 4, 32); }

/* montgomery_reduce_wide (shader/bn254fr.wgsl.in:76-91): subtractive form with J = p^-1 mod 2^256 */
static void mont_reduce(lo_fr *o, const lo_wide *w) {
    lo_fr lo, hi, q, h, t;
    memcpy(lo.v, w->v, 32); memcpy(hi.v, w->v + 4, 32);
    mul_lo(&q, &lo, &LO_J);
    mul_hi(&h, &q, &LO_P);
    if (sub_cc(&t, &hi, &h)) add_cc(&t, &t, &LO_P);
    *o = t;
}
void lo_fr_montmul(lo_fr *o, const lo_fr *a, const lo_fr *b) { lo_wide w; lo_mul_wide(&w, a, b); mont_reduce(o, &w); }

/* barrett_reduce_wide (shader/bn254fr.wgsl.in:113-124); host twin src/bn254.cpp:110-121 */
void lo_barrett_reduce(lo_fr *o, const lo_wide *x) {
    lo_fr xlo, xhi, xr_lo, sum_lo, sum_hi, z, q, r;
    lo_wide xr_hi;
    memcpy(xlo.v, x->v, 32); memcpy(xhi.v, x->v + 4, 32);
    lo_mul_wide(&xr_hi, &xhi, &LO_MU);
    mul_hi(&xr_lo, &xlo, &LO_MU);
    lo_fr hl, hh; memcpy(hl.v, xr_hi.v, 32); memcpy(hh.v, xr_hi.v + 4, 32);
    uint64_t c = add_cc(&sum_lo, &hl, &xr_lo);
    sum_hi = hh;
    for (int i = 0; i < 4 && c; i++) { sum_hi.v[i] += c; c = (sum_hi.v[i] == 0); }
    /* z = (sum_hi << 4) + (sum_lo >> 252) */
    for (int i = 3; i >= 0; i--) z.v[i] = (sum_hi.v[i] << 4) | (i ? (sum_hi.v[i - 1] >> 60) : 0);
    lo_fr zl = {{sum_lo.v[3] >> 60, 0, 0, 0}};
    add_cc(&z, &z, &zl);
    mul_lo(&q, &z, &LO_P);
    sub_cc(&r, &xlo, &q);
    reduce_p(&r);
    *o = r;
}
void lo_fr_mul(lo_fr *o, const lo_fr *a, const lo_fr *b) { lo_wide w; lo_mul_wide(&w, a, b); lo_barrett_reduce(o, &w); }
void lo_fr_to_mont(lo_fr *o, const lo_fr *a) { lo_fr_montmul(o, a, &LO_R2); }

void lo_fr_pow(lo_fr *o, const lo_fr *a, const lo_fr *e) {
    lo_fr acc, base = *a; lo_fr_from_u64(&acc, 1);
    for (int i = 0; i < 256; i++) {
        if ((e->v[i >> 6] >> (i & 63)) & 1) lo_fr_mul(&acc, &acc, &base);
        lo_fr_mul(&base, &base, &base);
    }
    *o = acc;
}
void lo_fr_pow_u64(lo_fr *o, const lo_fr *a, uint64_t e) { lo_fr ee = {{e, 0, 0, 0}}; lo_fr_pow(o, a, &ee); }
void lo_fr_inv(lo_fr *o, const lo_fr *a) {
    /* a^(p-2); equals the extended-Euclid inverse of bn254fr_invmod (shader/bn254fr.wgsl.in:128-153) */
    lo_fr e = LO_P; e.v[0] -= 2; lo_fr_pow(o, a, &e);
}

/* bn254_gmp::generate_omegas (src/bn254.cpp:51-64): w_k, w_2k from root1, w_4k from root2 = root1^(2^61-1).
 * NB powmod_ui takes a uint32 exponent; 2^28/k fits. */
void lo_omegas(uint32_t k, lo_fr *wk, lo_fr *w2k, lo_fr *w4k) {
    lo_fr root2;
    lo_fr_pow_u64(&root2, &ROOT1, (1ull << 61) - 1);
    uint64_t top = 1ull << 28;
    lo_fr_pow_u64(wk, &ROOT1, top / k);
    lo_fr_pow_u64(w2k, &ROOT1, top / (2ull * k));
    lo_fr_pow_u64(w4k, &root2, top / (4ull * k));
}
