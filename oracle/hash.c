/*
 * hash.c -- SHA-256, per-column codeword hashing, Merkle tree, AES-256-CTR
 * field sampler, Fiat-Shamir seeds and column sampling.
 * TEST INFRASTRUCTURE ONLY (see lig_oracle.h).
 *
 * Reference: shader/sha256.wgsl:65-230 (column hash), include/zkp/hash.hpp:44-118
 * (byte streams), include/zkp/merkle_tree.hpp:155-375, include/zkp/proof_serializer.hpp:82-117,
 * include/util/csprng.hpp:28-110, include/zkp/finite_field_gmp.hpp:66-78,
 * include/zkp/random.hpp:87-146, include/util/portable_sample.hpp:15-33,
 * src/webgpu_prover.cpp:162-168,281-282,337-351.
 */
#include "lig_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ SHA-256 (FIPS 180-4) */
static const uint32_t K256[64] = {
    0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,
    0xd807aa98,0x12835b01,0x243185be,0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,
    0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,
    0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,
    0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,
    0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,
    0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,
    0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2};
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))

static void sha256_block(uint32_t h[8], const uint8_t *p) {
    uint32_t m[64];
    for (int i = 0; i < 16; i++)
        m[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ROR(m[i - 15], 7) ^ ROR(m[i - 15], 18) ^ (m[i - 15] >> 3);
        uint32_t s1 = ROR(m[i - 2], 17) ^ ROR(m[i - 2], 19) ^ (m[i - 2] >> 10);
        m[i] = s1 + m[i - 7] + s0 + m[i - 16];
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t t1 = hh + (ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + m[i];
        uint32_t t2 = (ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
void lo_sha256_init(lo_sha256 *s) {
    static const uint32_t iv[8] = {0x6a09e667,0xbb67ae85,0x3c6ef372,0xa54ff53a,0x510e527f,0x9b05688c,0x1f83d9ab,0x5be0cd19};
    memcpy(s->h, iv, sizeof iv); s->len = 0; s->fill = 0;
}
void lo_sha256_update(lo_sha256 *s, const void *data, size_t n) {
    const uint8_t *p = data;
    s->len += n;
    while (n) {
        size_t take = 64 - s->fill; if (take > n) take = n;
        memcpy(s->buf + s->fill, p, take); s->fill += (uint32_t)take; p += take; n -= take;
        if (s->fill == 64) { sha256_block(s->h, s->buf); s->fill = 0; }
    }
}
static void sha256_finish_state(lo_sha256 *s) {
    uint64_t bits = s->len * 8;
    uint8_t pad = 0x80;
    lo_sha256_update(s, &pad, 1);
    pad = 0;
    while (s->fill != 56) lo_sha256_update(s, &pad, 1);
    uint8_t lenb[8];
    for (int i = 0; i < 8; i++) lenb[i] = (uint8_t)(bits >> (56 - 8 * i));
    lo_sha256_update(s, lenb, 8);
}
void lo_sha256_final(lo_sha256 *s, uint8_t out[32]) {
    sha256_finish_state(s);
    for (int i = 0; i < 8; i++) { out[4*i] = s->h[i] >> 24; out[4*i+1] = s->h[i] >> 16; out[4*i+2] = s->h[i] >> 8; out[4*i+3] = s->h[i]; }
}
void lo_sha256_buf(const void *data, size_t n, uint8_t out[32]) { lo_sha256 s; lo_sha256_init(&s); lo_sha256_update(&s, data, n); lo_sha256_final(&s, out); }
size_t lo_sizeof_sha256(void) { return sizeof(lo_sha256); }

/* ------------------------------------------------------------------ column hashing
 * sha256_update (shader/sha256.wgsl:148-177): each u32 limb of the element is fed most-significant
 * byte first, limbs in least-significant-first order.  sha256_final (:180-228) writes the eight
 * state words as native u32 => the stored leaf is the digest with every 4-byte word little-endian. */
void lo_colsha_init(lo_sha256 *st, size_t ncols) { for (size_t j = 0; j < ncols; j++) lo_sha256_init(&st[j]); }
extern int lo_omp_threads;
void lo_colsha_update(lo_sha256 *st, const lo_fr *row, size_t ncols) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (ncols >= 4096 && lo_omp_threads > 1) num_threads(lo_omp_threads)
#endif
    for (long j = 0; j < (long)ncols; j++) {
        const uint32_t *limb = (const uint32_t *)row[j].v;
        uint8_t b[32];
        for (int i = 0; i < 8; i++) { b[4*i] = limb[i] >> 24; b[4*i+1] = limb[i] >> 16; b[4*i+2] = limb[i] >> 8; b[4*i+3] = limb[i]; }
        lo_sha256_update(&st[j], b, 32);
    }
}
void lo_colsha_final(lo_sha256 *st, uint8_t *leaves, size_t ncols) {
    for (size_t j = 0; j < ncols; j++) {
        lo_sha256 s = st[j];
        sha256_finish_state(&s);
        memcpy(leaves + 32 * j, s.h, 32);      /* native (little-endian) u32 words */
    }
}

/* ------------------------------------------------------------------ Merkle tree
 * merkle_tree::initialize_from_digest / build_tree (include/zkp/merkle_tree.hpp:344-375):
 * heap layout, P = bit_ceil(n) leaves at nodes[P-1..2P-2] (missing leaves are zero digests),
 * node[i] = SHA256(node[2i+1] || node[2i+2]) with canonical digest bytes. */
static size_t bit_ceil(size_t n) { size_t p = 1; while (p < n) p <<= 1; return p; }
size_t lo_merkle_nodes(size_t nleaves) { return 2 * bit_ceil(nleaves) - 1; }
void lo_merkle_build(const uint8_t *leaves, size_t nleaves, uint8_t *nodes) {
    size_t P = bit_ceil(nleaves);
    memset(nodes, 0, 32 * (2 * P - 1));
    memcpy(nodes + 32 * (P - 1), leaves, 32 * nleaves);
    for (size_t i = P - 1; i-- > 0;) lo_sha256_buf(nodes + 32 * (2 * i + 1), 64, nodes + 32 * i);
}
/* decommit + canonical sibling order (merkle_tree.hpp:155-215; proof_serializer.hpp:82-117) */
size_t lo_merkle_decommit(const uint8_t *nodes, size_t nleaves, const uint32_t *idx, size_t nidx,
                          uint8_t *siblings, size_t cap) {
    size_t P = bit_ceil(nleaves), total = 2 * P - 1, count = 0;
    uint8_t *known = calloc(P, 1), *upper = calloc(P, 1);
    for (size_t i = 0; i < nidx; i++) known[idx[i]] = 1;
    size_t start = total / 2, end = total;
    while (start > 0) {
        size_t width = end - start;
        memset(upper, 0, P);
        for (size_t i = start; i < end; i += 2) {
            size_t ll = i - start, lr = ll + 1;
            int kl = known[ll], kr = (lr < width) ? known[lr] : 0;
            if (kl && kr) upper[ll / 2] = 1;
            else if (kr) { if (count < cap) memcpy(siblings + 32 * count, nodes + 32 * i, 32); count++; upper[ll / 2] = 1; }
            else if (kl) { if (count < cap) memcpy(siblings + 32 * count, nodes + 32 * (i + 1), 32); count++; upper[ll / 2] = 1; }
        }
        memcpy(known, upper, P);
        start = (start - 1) / 2; end = (end - 1) / 2;
    }
    free(known); free(upper);
    return count;
}
/* recommit (merkle_tree.hpp:232-318): rebuild the root from opened leaves + siblings */
int lo_merkle_recommit(size_t nleaves, const uint32_t *idx, size_t nidx, const uint8_t *leaf_digests,
                       const uint8_t *siblings, size_t nsib, uint8_t root[32]) {
    size_t P = bit_ceil(nleaves), total = 2 * P - 1, used = 0;
    uint8_t *cur = calloc(P, 32), *nxt = calloc(P, 32), *known = calloc(P, 1), *upper = calloc(P, 1);
    int ok = 1;
    for (size_t i = 0; i < nidx; i++) { known[idx[i]] = 1; memcpy(cur + 32 * idx[i], leaf_digests + 32 * i, 32); }
    size_t start = total / 2, end = total;
    while (start > 0 && ok) {
        memset(upper, 0, P);
        for (size_t i = start; i < end; i += 2) {
            size_t ll = i - start, lr = ll + 1;
            int kl = known[ll], kr = known[lr];
            if (!kl && !kr) continue;
            uint8_t pair[64];
            if (kl) memcpy(pair, cur + 32 * ll, 32);
            else { if (used >= nsib) { ok = 0; break; } memcpy(pair, siblings + 32 * used++, 32); }
            if (kr) memcpy(pair + 32, cur + 32 * lr, 32);
            else { if (used >= nsib) { ok = 0; break; } memcpy(pair + 32, siblings + 32 * used++, 32); }
            lo_sha256_buf(pair, 64, nxt + 32 * (ll / 2));
            upper[ll / 2] = 1;
        }
        uint8_t *t = cur; cur = nxt; nxt = t;
        memcpy(known, upper, P);
        start = (start - 1) / 2; end = (end - 1) / 2;
    }
    if (ok && used != nsib) ok = 0;
    if (ok) memcpy(root, cur, 32);
    free(cur); free(nxt); free(known); free(upper);
    return ok;
}

/* ------------------------------------------------------------------ AES-256 (FIPS 197) */
static uint8_t SBOX[256];
static int sbox_ready = 0;
static uint8_t xtime(uint8_t x) { return (uint8_t)((x << 1) ^ ((x >> 7) * 0x1b)); }
static void sbox_init(void) {
    /* multiplicative inverse in GF(2^8) followed by the affine map */
    uint8_t p = 1, q = 1;
    do {
        p = p ^ (uint8_t)(p << 1) ^ ((p & 0x80) ? 0x1b : 0);
        q ^= q << 1; q ^= q << 2; q ^= q << 4; if (q & 0x80) q ^= 0x09;
        uint8_t x = q ^ (uint8_t)((q << 1) | (q >> 7)) ^ (uint8_t)((q << 2) | (q >> 6)) ^ (uint8_t)((q << 3) | (q >> 5)) ^ (uint8_t)((q << 4) | (q >> 4));
        SBOX[p] = x ^ 0x63;
    } while (p != 1);
    SBOX[0] = 0x63;
    sbox_ready = 1;
}
void lo_aes256_expand(const uint8_t key[32], uint32_t rk[60]) {
    if (!sbox_ready) sbox_init();
    for (int i = 0; i < 8; i++) rk[i] = ((uint32_t)key[4*i] << 24) | ((uint32_t)key[4*i+1] << 16) | ((uint32_t)key[4*i+2] << 8) | key[4*i+3];
    uint32_t rcon = 1;
    for (int i = 8; i < 60; i++) {
        uint32_t t = rk[i - 1];
        if (i % 8 == 0) {
            t = (t << 8) | (t >> 24);
            t = ((uint32_t)SBOX[t >> 24] << 24) | ((uint32_t)SBOX[(t >> 16) & 255] << 16) | ((uint32_t)SBOX[(t >> 8) & 255] << 8) | SBOX[t & 255];
            t ^= rcon << 24; rcon = xtime((uint8_t)rcon);
        } else if (i % 8 == 4) {
            t = ((uint32_t)SBOX[t >> 24] << 24) | ((uint32_t)SBOX[(t >> 16) & 255] << 16) | ((uint32_t)SBOX[(t >> 8) & 255] << 8) | SBOX[t & 255];
        }
        rk[i] = rk[i - 8] ^ t;
    }
}
void lo_aes256_encrypt_block(const uint32_t rk[60], const uint8_t in[16], uint8_t out[16]) {
    uint8_t s[16];
    for (int i = 0; i < 16; i++) s[i] = in[i] ^ (uint8_t)(rk[i / 4] >> (24 - 8 * (i % 4)));
    for (int round = 1; round <= 14; round++) {
        uint8_t t[16];
        for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) t[4 * c + r] = SBOX[s[4 * ((c + r) % 4) + r]];  /* SubBytes+ShiftRows */
        if (round != 14) {
            for (int c = 0; c < 4; c++) {
                uint8_t a0 = t[4*c], a1 = t[4*c+1], a2 = t[4*c+2], a3 = t[4*c+3];
                s[4*c]   = xtime(a0) ^ (xtime(a1) ^ a1) ^ a2 ^ a3;
                s[4*c+1] = a0 ^ xtime(a1) ^ (xtime(a2) ^ a2) ^ a3;
                s[4*c+2] = a0 ^ a1 ^ xtime(a2) ^ (xtime(a3) ^ a3);
                s[4*c+3] = (xtime(a0) ^ a0) ^ a1 ^ a2 ^ xtime(a3);
            }
        } else memcpy(s, t, 16);
        for (int i = 0; i < 16; i++) s[i] ^= (uint8_t)(rk[4 * round + i / 4] >> (24 - 8 * (i % 4)));
    }
    memcpy(out, s, 16);
}
/* mpz_random_engine (include/util/csprng.hpp:54-107): EVP_aes_256_ctr, IV = 0^16 (params.hpp:42),
 * zero plaintext => keystream block b = AES_k(be128(b)).  16 KiB refills never split a 32-byte draw,
 * so the stream is continuous.  Element e uses blocks 2e, 2e+1. */
void lo_rng_init(lo_rng *r, const uint8_t key[32]) { lo_aes256_expand(key, r->rk); r->pos = 0; }
void lo_rng_keystream(const lo_rng *r, uint64_t first_block, uint8_t *out, size_t nblocks) {
    for (size_t b = 0; b < nblocks; b++) {
        uint8_t ctr[16] = {0};
        uint64_t v = first_block + b;
        for (int i = 0; i < 8; i++) ctr[15 - i] = (uint8_t)(v >> (8 * i));
        lo_aes256_encrypt_block(r->rk, ctr, out + 16 * b);
    }
}
/* bn254_gmp::generate_random (include/zkp/finite_field_gmp.hpp:66-78): 4 LE u64 -> >>2 -> -p if >= p */
void lo_rng_next(lo_rng *r, lo_fr *out) {
    uint8_t ks[32];
    lo_rng_keystream(r, 2 * r->pos, ks, 2);
    r->pos++;
    lo_fr v; memcpy(v.v, ks, 32);
    for (int i = 0; i < 4; i++) v.v[i] = (v.v[i] >> 2) | (i < 3 ? (v.v[i + 1] << 62) : 0);
    if (lo_fr_cmp(&v, &LO_P) >= 0) {
        uint64_t br = 0;
        for (int i = 0; i < 4; i++) { unsigned __int128 d = (unsigned __int128)v.v[i] - LO_P.v[i] - br; v.v[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
    }
    *out = v;
}
void lo_rng_fill(lo_rng *r, lo_fr *out, size_t count) { for (size_t i = 0; i < count; i++) lo_rng_next(r, &out[i]); }

/* ------------------------------------------------------------------ Fiat-Shamir seeds
 * zkp::hash<sha256>("LigetronStage1", root, instance_hash) (webgpu_prover.cpp:281-282): the string
 * literal goes through the array overload (hash.hpp:59-63) and is hashed WITH its trailing NUL. */
void lo_stage1_seed(const uint8_t root[32], const uint8_t instance_hash[32], uint8_t out[32]) {
    lo_sha256 s; lo_sha256_init(&s);
    lo_sha256_update(&s, "LigetronStage1", 15);
    lo_sha256_update(&s, root, 32);
    lo_sha256_update(&s, instance_hash, 32);
    lo_sha256_final(&s, out);
}
/* webgpu_prover.cpp:337-341: "LigetronStage2\0" || root || raw LE limb bytes of the three encoded rows */
void lo_stage2_seed(const uint8_t root[32], const lo_fr *code, const lo_fr *lin, const lo_fr *quad, size_t n, uint8_t out[32]) {
    lo_sha256 s; lo_sha256_init(&s);
    lo_sha256_update(&s, "LigetronStage2", 15);
    lo_sha256_update(&s, root, 32);
    lo_sha256_update(&s, code, 32 * n);
    lo_sha256_update(&s, lin, 32 * n);
    lo_sha256_update(&s, quad, 32 * n);
    lo_sha256_final(&s, out);
}
/* webgpu_prover.cpp:110-168 with no user args: h0 = 0^32, arg0 = "Ligero" + NUL hashed byte by byte:
 * instance_hash = SHA256(0^32 || "Ligero\0") */
void lo_instance_hash_default(uint8_t out[32]) {
    uint8_t z[32] = {0};
    lo_sha256 s; lo_sha256_init(&s);
    lo_sha256_update(&s, z, 32);
    lo_sha256_update(&s, "Ligero", 7);
    lo_sha256_final(&s, out);
}

/* webgpu_prover.cpp:110-168: instance_hash = hash(instance_hash, input_args[i]) over the public arguments, starting
 * from the zero digest; a vector<u8> argument is fed byte by byte (hash.hpp:75-80), i.e. as its bytes */
void lo_instance_hash(const uint8_t *args, const uint64_t *lens, size_t n_args, uint8_t out[32]) {
    lo_instance_hash_default(out);
    for (size_t i = 0; i < n_args; i++) {
        lo_sha256 s; lo_sha256_init(&s);
        lo_sha256_update(&s, out, 32);
        lo_sha256_update(&s, args, lens[i]);
        lo_sha256_final(&s, out);
        args += lens[i];
    }
}

/* ------------------------------------------------------------------ column sampling
 * hash_random_engine<sha256> (include/zkp/random.hpp:87-146): refill #0 = SHA256(le64(0)) (the seed is
 * absorbed only AFTER the first flush), refill #c = SHA256(seed || le64(c)); bytes are handed out from
 * digest[31] down to digest[0]. */
typedef struct { uint8_t seed[32], buf[32]; uint64_t state; int32_t offset; } hre_t;
static void hre_init(hre_t *e, const uint8_t seed[32]) { memcpy(e->seed, seed, 32); e->state = 0; e->offset = -1; }
static uint8_t hre_next(hre_t *e) {
    if (e->offset < 0 || e->offset >= 32) {
        lo_sha256 s; lo_sha256_init(&s);
        if (e->state != 0) lo_sha256_update(&s, e->seed, 32);
        uint8_t le[8]; for (int i = 0; i < 8; i++) le[i] = (uint8_t)(e->state >> (8 * i));
        lo_sha256_update(&s, le, 8);
        lo_sha256_final(&s, e->buf);
        e->state++;
        e->offset = 31;
    }
    return e->buf[e->offset--];
}
void lo_hash_engine_bytes(const uint8_t seed[32], size_t count, uint8_t *out) {
    hre_t e; hre_init(&e, seed);
    for (size_t i = 0; i < count; i++) out[i] = hre_next(&e);
}
/* Restatement of Boost.Random's integer-engine path for an engine with range [0,255] (hash_random_engine::result_type =
 * uint8_t, min 0, max 255: include/zkp/random.hpp:89-95) and Distance = ptrdiff_t (range_type = unsigned 64 bit).
 * Boost is NOT vendored in the reference (find_package(Boost COMPONENTS random), CMakeLists.txt:67-69, no version pin) and
 * not present in this image, so this is "parity unpinned" (SURVEY.md A.7).  To diff it against a real Boost in a minute:
 *   header   boost/random/uniform_int_distribution.hpp
 *   function boost::random::detail::generate_uniform_int(Engine& eng, T min_value, T max_value, boost::true_type)
 *            (the is_integral<Engine::result_type> overload; reached from uniform_int_distribution<T>::operator()(Engine&)
 *            through generate_uniform_int(eng, min, max)); the body has been the same in every release that has this header
 *            (it moved here from boost/random/uniform_int.hpp in Boost 1.47; checked from memory against 1.74 and 1.83).
 * The statements of that function, in order, and where each one is below (range = max_value - min_value, brange = eng.max() -
 * eng.min() = 255, bmin = 0):
 *   B1  if(range == 0) return min_value;                                                            -> [B1]
 *   B2  else if(brange == range) return add(subtract(eng(), bmin), min_value);                      -> [B2]
 *   B3  else if(brange < range) for(;;) {                                                           -> [B3]
 *   B3a   limit = (range == max(range_type)) ? range/(brange+1) (+1 if range%(brange+1) == brange) : (range+1)/(brange+1);
 *   B3b   result = 0; mult = 1;
 *   B3c   while(mult <= limit) { result += subtract(eng(), bmin) * mult;
 *   B3d                          if(mult * brange == range - mult + 1) return result;     // range+1 is a power of brange+1
 *   B3e                          mult *= brange + 1; }
 *   B3f   result_increment = generate_uniform_int(eng, 0, range/mult, true_type());       // recursion, consumes more bytes
 *   B3g   if(max(range_type) / mult < result_increment) continue;                         // multiplication would overflow
 *   B3h   result_increment *= mult; result += result_increment;
 *   B3i   if(result < result_increment) continue;                                          // addition overflowed
 *   B3j   if(result > range) continue;                                                     // too big
 *   B3k   return add(result, min_value); }
 *   B4  else {  // brange > range                                                                   -> [B4]
 *   B4a   bucket_size = (brange == max(base_unsigned)) ? brange/(range+1) (+1 if brange%(range+1) == range)
 *                                                      : (brange+1)/(range+1);
 *         // base_unsigned = uint8_t here, so Boost takes the FIRST form; both equal floor(256/(range+1)) for brange = 255
 *   B4b   for(;;) { result = subtract(eng(), bmin) / bucket_size; if(result <= range) return add(result, min_value); } }
 * (As remembered, B3d returns `result` without adding min_value.  Whether it does is immaterial here: min_value = 0 in the
 * recursion, and the outer call U(i, n-1) reaches B3d only when n - i is a power of 256, which for n = 4k a power of two and
 * i < 192 happens only at i = 0 = min_value.)  Returns a value in [0, range]; the caller adds lo. */
static uint64_t boost_uniform(hre_t *e, uint64_t range) {
    const uint64_t brange = 255;
    if (range == 0) return 0;                                                                   /* [B1] */
    if (range == brange) return hre_next(e);                                                    /* [B2] */
    if (range < brange) {                                                                       /* [B4] */
        uint64_t bucket = (brange + 1) / (range + 1);    /* B4a: = 255/(range+1) (+1 if 255%(range+1) == range) */
        for (;;) { uint64_t r = hre_next(e) / bucket; if (r <= range) return r; }               /* B4b */
    }
    for (;;) {                                                                                  /* [B3] */
        uint64_t limit = (range == UINT64_MAX) ? (range / (brange + 1) + ((range % (brange + 1) == brange) ? 1 : 0))
                                               : (range + 1) / (brange + 1);                    /* B3a */
        uint64_t result = 0, mult = 1;                                                          /* B3b */
        int done = 0;
        while (mult <= limit) {                                                                 /* B3c */
            result += (uint64_t)hre_next(e) * mult;
            if (mult * brange == range - mult + 1) { done = 1; break; }                         /* B3d */
            mult *= brange + 1;                                                                 /* B3e */
        }
        if (done) return result;
        uint64_t inc = boost_uniform(e, range / mult);                                          /* B3f */
        if (UINT64_MAX / mult < inc) continue;                                                  /* B3g */
        inc *= mult;                                                                            /* B3h */
        result += inc;
        if (result < inc) continue;                                                             /* B3i */
        if (result > range) continue;                                                           /* B3j */
        return result;                                                                          /* B3k */
    }
}
/* portable_sample (include/util/portable_sample.hpp:15-33) + sort (webgpu_prover.cpp:351) */
static int cmp_u32(const void *a, const void *b) { uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b; return (x > y) - (x < y); }
void lo_sample_indices(const uint8_t seed[32], uint32_t n, uint32_t t, uint32_t *out_sorted) {
    hre_t e; hre_init(&e, seed);
    uint32_t *a = malloc(sizeof(uint32_t) * n);
    for (uint32_t i = 0; i < n; i++) a[i] = i;
    if (t > n) t = n;
    for (uint32_t i = 0; i < t; i++) {
        uint64_t j = i + boost_uniform(&e, (uint64_t)(n - 1) - i);
        uint32_t tmp = a[i]; a[i] = a[j]; a[j] = tmp;
        out_sorted[i] = a[i];
    }
    qsort(out_sorted, t, sizeof(uint32_t), cmp_u32);
    free(a);
}
