/*
 * ntt.c -- radix-2 NTT / Reed-Solomon encode/decode and the eltwise kernels,
 * restating the reference executor's dispatch sequences on the CPU.
 * TEST INFRASTRUCTURE ONLY (see lig_oracle.h).
 *
 * Reference: src/webgpu/engine.cpp:755-796 (encode/decode), :844-882 (forward:
 * DIF stages M=N..2 then bit reversal), :932-968 (inverse: bit reversal, DIT
 * stages M=2..N, then x N^-1), :1382-1503 (twiddle tables w^i*R mod p, N^-1*R);
 * shader/kernels.wgsl.in:58-323 (butterflies, fold), :326-538 (eltwise).
 *
 * The reference keeps intermediate values lazily in [0,2p)/[0,4p) and reduces
 * at the end of each transform; every value it leaves in a buffer is the
 * canonical residue, so this restatement reduces after every butterfly and is
 * value-identical.
 */
#include "lig_oracle.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int lo_omp_threads = 1;     /* threads for the column-parallel loops (set by lo_prove from job.threads) */

typedef struct {
    uint32_t N, log2N;
    lo_fr *w;      /* w^i * R mod p, i < N/2      (engine.cpp:1391-1401) */
    lo_fr *winv;   /* w^-i * R mod p, i < N/2     (engine.cpp:1433-1447) */
    lo_fr ninv;    /* N^-1 * R mod p              (engine.cpp:1478-1484) */
} lo_plan;

struct lo_ctx {
    uint32_t l, k, n;
    lo_plan plan[3];
};

static uint32_t ilog2(uint32_t x) { uint32_t r = 0; while ((1u << r) < x) r++; return r; }

static void plan_init(lo_plan *pl, uint32_t N, const lo_fr *root) {
    pl->N = N; pl->log2N = ilog2(N);
    pl->w = malloc(sizeof(lo_fr) * (N / 2));
    pl->winv = malloc(sizeof(lo_fr) * (N / 2));
    lo_fr rinv, cur, curinv, one;
    lo_fr_inv(&rinv, root);
    lo_fr_from_u64(&one, 1);
    cur = one; curinv = one;
    for (uint32_t i = 0; i < N / 2; i++) {
        lo_fr_to_mont(&pl->w[i], &cur);
        lo_fr_to_mont(&pl->winv[i], &curinv);
        lo_fr_mul(&cur, &cur, root);
        lo_fr_mul(&curinv, &curinv, &rinv);
    }
    lo_fr nn, ni; lo_fr_from_u64(&nn, N); lo_fr_inv(&ni, &nn);
    lo_fr_to_mont(&pl->ninv, &ni);
}

lo_ctx *lo_ctx_new(uint32_t l, uint32_t k, uint32_t n) {
    if (n != 4 * k || (k & (k - 1)) || l > k) return NULL;
    lo_ctx *c = calloc(1, sizeof *c);
    c->l = l; c->k = k; c->n = n;
    lo_fr wk, w2k, w4k;
    lo_omegas(k, &wk, &w2k, &w4k);
    plan_init(&c->plan[LO_SIZE_K], k, &wk);
    plan_init(&c->plan[LO_SIZE_2K], 2 * k, &w2k);
    plan_init(&c->plan[LO_SIZE_N], n, &w4k);
    return c;
}
void lo_ctx_free(lo_ctx *c) {
    if (!c) return;
    for (int i = 0; i < 3; i++) { free(c->plan[i].w); free(c->plan[i].winv); }
    free(c);
}
uint32_t lo_ctx_k(const lo_ctx *c) { return c->k; }
uint32_t lo_ctx_l(const lo_ctx *c) { return c->l; }
uint32_t lo_ctx_n(const lo_ctx *c) { return c->n; }

/* ntt_bit_reverse (shader/kernels.wgsl.in:58-74) */
static void bit_reverse(lo_fr *buf, uint32_t N, uint32_t bits) {
    for (uint32_t id = 0; id < N; id++) {
        uint32_t r = 0;
        for (uint32_t b = 0; b < bits; b++) r |= ((id >> b) & 1u) << (bits - 1 - b);
        if (id < r) { lo_fr t = buf[id]; buf[id] = buf[r]; buf[r] = t; }
    }
}

/* ntt_forward_kernel (engine.cpp:844-882) with ntt_forward_radix2 (kernels.wgsl.in:125-153) */
static void ntt_forward(const lo_plan *pl, lo_fr *buf) {
    const uint32_t N = pl->N;
    for (uint32_t iter = pl->log2N; iter >= 1; iter--) {
        const uint32_t M = 1u << iter, M2 = M >> 1, stride = N / M;
        for (uint32_t inst = 0; inst < N / 2; inst++) {
            uint32_t group = inst / M2, index = inst % M2, k = group * M + index;
            lo_fr x = buf[k], y = buf[k + M2], d;
            lo_fr_add(&buf[k], &x, &y);
            lo_fr_sub(&d, &x, &y);
            lo_fr_montmul(&buf[k + M2], &d, &pl->w[index * stride]);
        }
    }
    bit_reverse(buf, N, pl->log2N);
}

/* ntt_inverse_kernel (engine.cpp:932-968) with ntt_inverse_radix2 (kernels.wgsl.in:230-262)
 * and ntt_adjust_inverse_reduce (:93-103) */
static void ntt_inverse(const lo_plan *pl, lo_fr *buf) {
    const uint32_t N = pl->N;
    bit_reverse(buf, N, pl->log2N);
    for (uint32_t iter = 1; iter <= pl->log2N; iter++) {
        const uint32_t M = 1u << iter, M2 = M >> 1, stride = N / M;
        for (uint32_t inst = 0; inst < N / 2; inst++) {
            uint32_t group = inst / M2, index = inst % M2, k = group * M + index;
            lo_fr x = buf[k], y;
            lo_fr_montmul(&y, &buf[k + M2], &pl->winv[index * stride]);
            lo_fr_add(&buf[k], &x, &y);
            lo_fr_sub(&buf[k + M2], &x, &y);
        }
    }
    for (uint32_t i = 0; i < N; i++) lo_fr_montmul(&buf[i], &buf[i], &pl->ninv);
}

void lo_ntt_forward(const lo_ctx *c, int which, lo_fr *buf) { ntt_forward(&c->plan[which], buf); }
void lo_ntt_inverse(const lo_ctx *c, int which, lo_fr *buf) { ntt_inverse(&c->plan[which], buf); }

/* encode_ntt_device (engine.cpp:755-770): INTT_k on buf[0..k) then NTT_n on buf[0..n) (buf[k..n) must be 0) */
void lo_encode(const lo_ctx *c, lo_fr *buf) {
    ntt_inverse(&c->plan[LO_SIZE_K], buf);
    ntt_forward(&c->plan[LO_SIZE_N], buf);
}
/* mask rows: ntt_inverse_2k + ntt_forward_n (include/zkp/nonbatch_context.hpp:485-486) */
void lo_encode_2k(const lo_ctx *c, lo_fr *buf) {
    ntt_inverse(&c->plan[LO_SIZE_2K], buf);
    ntt_forward(&c->plan[LO_SIZE_N], buf);
}
/* decode_ntt_device (engine.cpp:772-796) + ntt_fold (kernels.wgsl.in:105-116, N from the 2k config) */
void lo_decode(const lo_ctx *c, lo_fr *buf) {
    ntt_inverse(&c->plan[LO_SIZE_N], buf);
    const uint32_t half = c->k;
    for (uint32_t i = 0; i < half; i++) lo_fr_add(&buf[i], &buf[i], &buf[i + half]);
    ntt_forward(&c->plan[LO_SIZE_K], buf);
}

void lo_encode_rows(const lo_ctx *c, const lo_fr *msgs, lo_fr *codewords, size_t rows, int threads) {
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
#endif
    for (long r = 0; r < (long)rows; r++) {
        lo_fr *cw = codewords + (size_t)r * c->n;
        memcpy(cw, msgs + (size_t)r * c->k, sizeof(lo_fr) * c->k);
        memset(cw + c->k, 0, sizeof(lo_fr) * (c->n - c->k));
        lo_encode(c, cw);
    }
}

/* eltwise kernels (shader/kernels.wgsl.in:326-510) */
void lo_eltwise(int op, const lo_fr *x, const lo_fr *y, lo_fr *out, size_t count,
                const lo_fr *scalar, uint32_t bit) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (count >= 4096 && lo_omp_threads > 1) num_threads(lo_omp_threads)
#endif
    for (long i = 0; i < (long)count; i++) {
        lo_fr t;
        switch (op) {
        case LO_OP_ADD:        lo_fr_add(&out[i], &x[i], &y[i]); break;                 /* :326 */
        case LO_OP_SUB:        lo_fr_sub(&out[i], &x[i], &y[i]); break;                 /* :365 */
        case LO_OP_ADD_ASSIGN: lo_fr_add(&out[i], &out[i], &x[i]); break;               /* :339 */
        case LO_OP_ADD_CONST:  lo_fr_add(&out[i], &x[i], scalar); break;                /* :352 */
        case LO_OP_SUB_CONST:  lo_fr_sub(&out[i], &x[i], scalar); break;                /* :383 */
        case LO_OP_CONST_SUB:  lo_fr_sub(&out[i], scalar, &x[i]); break;                /* :400 */
        case LO_OP_MUL:        lo_fr_mul(&out[i], &x[i], &y[i]); break;                 /* :417 */
        case LO_OP_MUL_CONST:  lo_fr_mul(&out[i], &x[i], scalar); break;                /* :430 */
        case LO_OP_MONTMUL_CONST: lo_fr_montmul(&out[i], &x[i], scalar); break;         /* :442 */
        case LO_OP_FMA:        lo_fr_mul(&t, &x[i], &y[i]); lo_fr_add(&out[i], &out[i], &t); break;   /* :469 */
        case LO_OP_FMA_CONST:  lo_fr_mul(&t, &x[i], scalar); lo_fr_add(&out[i], &out[i], &t); break;  /* :486 */
        case LO_OP_DIV:        lo_fr_inv(&t, &y[i]); lo_fr_mul(&out[i], &x[i], &t); break;            /* :453 */
        case LO_OP_BIT_DECOMPOSE:                                                       /* :502 */
            lo_fr_from_u64(&out[i], (x[i].v[bit >> 6] >> (bit & 63)) & 1); break;
        }
    }
}

/* powmod_context (src/webgpu/powmod_context.cpp:178-268; shader/bn254fr.wgsl.in:157-168;
 * kernels.wgsl.in:513-538): out (=|+=) coeff * base^exp with a 32-entry table base^(2^i)*R */
void lo_powmod(const lo_fr *base, const uint32_t *exp, const lo_fr *coeff, lo_fr *out, size_t count, int add) {
    lo_fr table[32], cur = *base;
    for (int i = 0; i < 32; i++) { lo_fr_to_mont(&table[i], &cur); lo_fr_mul(&cur, &cur, &cur); }
    for (size_t e = 0; e < count; e++) {
        lo_fr acc = LO_R;
        for (int i = 0; i < 32; i++) if ((exp[e] >> i) & 1u) lo_fr_montmul(&acc, &acc, &table[i]);
        lo_fr r; lo_fr_montmul(&r, &coeff[e], &acc);   /* coeff plain x (base^exp * R) -> plain */
        if (add) lo_fr_add(&out[e], &out[e], &r); else out[e] = r;
    }
}
