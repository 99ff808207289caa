/*
 * prover.c -- synthetic row stream, reference-structured three-stage prover,
 * verifier restatement and proof envelope writer.
 * TEST INFRASTRUCTURE ONLY (see lig_oracle.h).
 *
 * Reference: include/zkp/backend/witness_manager.hpp:200-354,497-503 (row forming, masks,
 * commit order), include/zkp/nonbatch_context.hpp:445-553 (stage 1), :654-780 (stage 2),
 * :924-1000 (stage 3), :1219-1287 (verifier), src/webgpu_prover.cpp:226-494,
 * src/webgpu_verifier.cpp:263-452, include/zkp/proof_serializer.hpp:119-191,
 * proto/ligero_proof.proto:13-60, proto/common.proto:21-33.
 *
 * The WASM interpreter / constraint generator (out of scope) is replaced by a
 * synthetic constraint stream: n_linear witness slots and n_quad slots of x*y=z, every
 * witness carrying one dense linear-test coefficient (worst case for stage 2).
 */
#include "lig_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

extern int lo_omp_threads;
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

void lo_synth_key(uint64_t seed, uint8_t key[32]) {
    lo_sha256 s; lo_sha256_init(&s);
    lo_sha256_update(&s, "lig-synth", 9);
    uint8_t le[8]; for (int i = 0; i < 8; i++) le[i] = (uint8_t)(seed >> (8 * i));
    lo_sha256_update(&s, le, 8);
    lo_sha256_final(&s, key);
}

/* ------------------------------------------------------------------ row plan
 * Commit order (witness_manager.hpp:497-503 + callbacks firing as rows fill): full linear rows, full
 * quadratic triples, then at finalize the partial linear row, the partial quadratic triple, 3 masks. */
typedef struct { int kind; /* 0 linear, 1 quad x, 2 quad y, 3 quad z; batch rows: 4 init, 5 bit, 6/7 equal x/y, 8/9/10 quadratic x/y/z */ uint32_t data; } rowdesc;
enum { RK_INIT = 4, RK_BIT = 5, RK_EQX = 6, RK_EQY = 7, RK_BQX = 8, RK_BQY = 9, RK_BQZ = 10 };
static int has_code_check(int kind) { return kind != RK_EQX && kind != RK_EQY; }      /* nonbatch_context.hpp:811-825: no check_code */
static int mid_of_group(int kind) { return kind == 1 || kind == 2 || kind == RK_EQX || kind == RK_BQX || kind == RK_BQY; }

/* rows committed by the batch program, in program order (-1: malformed program) */
static long batch_plan(const lo_job *j, rowdesc *d) {
    long r = 0;
    for (uint64_t i = 0; i < j->n_batch_ops; i++) {
        const lo_batch_op *o = &j->batch_ops[i];
        if (o->op >= LO_BOP_COUNT || o->out >= 512 || o->x >= 512 || o->y >= 512) return -1;
        const uint64_t need = o->op == LO_BOP_SET ? 32ull * o->len : o->op == LO_BOP_BIT_DECOMPOSE ? 4ull * o->len :
                              (o->op == LO_BOP_SET_SCALAR || (o->op >= LO_BOP_ADD_CONST && o->op <= LO_BOP_MONTMUL_CONST)) ? 32 : 0;
        if (need && (o->data_off > j->batch_data_bytes || need > j->batch_data_bytes - o->data_off)) return -1;
        if (o->op == LO_BOP_SET && o->len > j->l) return -1;
        if (o->op == LO_BOP_BIT_DECOMPOSE) {
            if (o->len > 256) return -1;
            for (uint32_t b = 0; b < o->len; b++) { uint32_t slot; memcpy(&slot, j->batch_data + o->data_off + 4ull * b, 4); if (slot >= 512 || slot == o->x) return -1; }
        }
        switch (o->op) {
        case LO_BOP_SET: case LO_BOP_SET_SCALAR: if (d) d[r] = (rowdesc){RK_INIT, 0}; r += 1; break;
        case LO_BOP_COPY: case LO_BOP_ASSERT_EQUAL: if (d) { d[r] = (rowdesc){RK_EQX, 0}; d[r + 1] = (rowdesc){RK_EQY, 0}; } r += 2; break;
        case LO_BOP_MUL: case LO_BOP_DIV: if (d) { d[r] = (rowdesc){RK_BQX, 0}; d[r + 1] = (rowdesc){RK_BQY, 0}; d[r + 2] = (rowdesc){RK_BQZ, 0}; } r += 3; break;
        case LO_BOP_BIT_DECOMPOSE: for (uint32_t b = 0; b < o->len; b++) { if (d) d[r] = (rowdesc){RK_BIT, 0}; r++; } break;
        default: break;
        }
    }
    return r;
}

static size_t plan_rows(const lo_job *j, rowdesc **out) {
    size_t lf = j->n_linear / j->l, lp = j->n_linear % j->l;
    size_t qf = j->n_quad / j->l, qp = j->n_quad % j->l;
    long rb = batch_plan(j, NULL);
    if (rb < 0) rb = 0;                                   /* malformed programs are rejected by lo_prove / lo_verify */
    size_t total = (size_t)rb + lf + (lp ? 1 : 0) + 3 * (qf + (qp ? 1 : 0));
    rowdesc *d = malloc(sizeof(rowdesc) * (total ? total : 1));
    size_t r = (size_t)rb;
    if (rb) batch_plan(j, d);
    for (size_t i = 0; i < lf; i++) d[r++] = (rowdesc){0, j->l};
    for (size_t i = 0; i < qf; i++) for (int t = 1; t <= 3; t++) d[r++] = (rowdesc){t, j->l};
    if (lp) d[r++] = (rowdesc){0, (uint32_t)lp};
    if (qp) for (int t = 1; t <= 3; t++) d[r++] = (rowdesc){t, (uint32_t)qp};
    *out = d;
    return total;
}

/* The batch program on k-element variables; committed rows are appended to `rows` (vbn254fr.hpp:138-565 for the
 * operations, nonbatch_context.hpp:497-553 for what each hook commits).  `enc` = the encoding stream (pads). */
static void batch_run(const lo_job *j, lo_rng *enc, lo_fr *rows) {
    const uint32_t l = j->l, k = j->k;
    if (!j->n_batch_ops) return;
    lo_fr *vars = calloc((size_t)512 * k, sizeof(lo_fr)), *tmp = calloc(k, sizeof(lo_fr));
    size_t r = 0;
    int compat = 0;                       /* LO_BOP_UPSTREAM_COMPAT seen: buffer_view::slice_bytes as DEFINED (buffer_view.cpp:91-95) */
#define VAR(i) (vars + (size_t)(i) * k)
#define COMMIT(src) do { memcpy(rows + (r++) * (size_t)k, (src), sizeof(lo_fr) * k); } while (0)
    for (uint64_t i = 0; i < j->n_batch_ops; i++) {
        const lo_batch_op *o = &j->batch_ops[i];
        const uint8_t *data = j->batch_data + o->data_off;
        lo_fr c;
        switch (o->op) {
        case LO_BOP_UPSTREAM_COMPAT: compat = 1; break;
        case LO_BOP_SET: case LO_BOP_SET_SCALAR: {
            const int limbs = o->reserved & 1;                             /* write_limbs family: write_buffer only */
            const uint32_t len = o->op == LO_BOP_SET ? o->len : l;
            if (!limbs && !compat) memset(VAR(o->x), 0, sizeof(lo_fr) * k); /* write_buffer_clear, declared slice: the rest of x */
            if (o->op == LO_BOP_SET) memcpy(VAR(o->x), data, 32ull * o->len);
            else for (uint32_t e = 0; e < l; e++) memcpy(&VAR(o->x)[e], data, 32);
            /* write_buffer_clear upstream: clear_buffer(x.slice(len * 32)) AFTER the write, x.slice(B) = {offset B, size X + k*32 - B}
             * of the slab (device_context.hpp:95-98 over buffer_view.cpp:91-95): slab elements [len, x*k + k) */
            if (!limbs && compat) memset(vars + len, 0, sizeof(lo_fr) * ((size_t)o->x * k + k - len));
            /* on_batch_init: pad_encoding_random into x.slice(l * 32) (nonbatch_context.hpp:502-505): x's own pad slots as
             * declared, slab element l (variable 0's pad slots) as defined */
            lo_rng_fill(enc, compat ? vars + l : VAR(o->x) + l, k - l);
            COMMIT(VAR(o->x));
            break;
        }
        case LO_BOP_COPY:
            memmove(VAR(o->out), VAR(o->x), sizeof(lo_fr) * k);
            COMMIT(VAR(o->out)); COMMIT(VAR(o->x));                        /* on_batch_equal(out, in) */
            break;
        case LO_BOP_ADD: lo_eltwise(LO_OP_ADD, VAR(o->x), VAR(o->y), tmp, k, NULL, 0); memcpy(VAR(o->out), tmp, sizeof(lo_fr) * k); break;
        case LO_BOP_SUB: lo_eltwise(LO_OP_SUB, VAR(o->x), VAR(o->y), tmp, k, NULL, 0); memcpy(VAR(o->out), tmp, sizeof(lo_fr) * k); break;
        case LO_BOP_MUL:
            lo_eltwise(LO_OP_MUL, VAR(o->x), VAR(o->y), tmp, k, NULL, 0);
            COMMIT(VAR(o->x)); COMMIT(VAR(o->y)); COMMIT(tmp);             /* on_batch_quadratic(x, y, tmp) */
            memcpy(VAR(o->out), tmp, sizeof(lo_fr) * k);
            break;
        case LO_BOP_DIV:
            lo_eltwise(LO_OP_DIV, VAR(o->x), VAR(o->y), tmp, k, NULL, 0);
            COMMIT(tmp); COMMIT(VAR(o->y)); COMMIT(VAR(o->x));             /* on_batch_quadratic(tmp, y, x) */
            memcpy(VAR(o->out), tmp, sizeof(lo_fr) * k);
            break;
        case LO_BOP_ADD_CONST: case LO_BOP_SUB_CONST: case LO_BOP_CONST_SUB: case LO_BOP_MUL_CONST: case LO_BOP_MONTMUL_CONST: {
            static const int map[5] = {LO_OP_ADD_CONST, LO_OP_SUB_CONST, LO_OP_CONST_SUB, LO_OP_MUL_CONST, LO_OP_MONTMUL_CONST};
            memcpy(&c, data, 32);
            lo_eltwise(map[o->op - LO_BOP_ADD_CONST], VAR(o->x), NULL, tmp, k, &c, 0);
            memcpy(VAR(o->out), tmp, sizeof(lo_fr) * k);
            break;
        }
        case LO_BOP_ASSERT_EQUAL: COMMIT(VAR(o->x)); COMMIT(VAR(o->y)); break;
        case LO_BOP_BIT_DECOMPOSE:
            for (uint32_t b = 0; b < o->len; b++) {
                uint32_t slot; memcpy(&slot, data + 4ull * b, 4);
                lo_eltwise(LO_OP_BIT_DECOMPOSE, VAR(o->x), NULL, tmp, k, NULL, b);
                memcpy(VAR(slot & 511), tmp, sizeof(lo_fr) * k);
                COMMIT(VAR(slot & 511));                                   /* on_batch_bit */
            }
            break;
        case LO_BOP_FREE: memset(VAR(o->x), 0, sizeof(lo_fr) * k); break;
        default: break;
        }
    }
#undef VAR
#undef COMMIT
    free(vars); free(tmp);
}
void lo_row_kinds(const lo_job *j, uint8_t *kinds) { rowdesc *d; size_t R = plan_rows(j, &d); for (size_t r = 0; r < R; r++) kinds[r] = (uint8_t)d[r].kind; free(d); }
size_t lo_job_rows(const lo_job *j) { rowdesc *d; size_t r = plan_rows(j, &d); free(d); return r + 3; }

/* process_reset_linear_row / _quadratic_rows / process_masks (witness_manager.hpp:200-321).
 * Witness values come from the synthetic AES stream (one draw per data slot, x row then y row for a
 * quadratic triple; z = x*y).  Pads (k-l per row) and masks come from the encoding stream, in commit order. */
void lo_form_rows(const lo_job *j, lo_fr *rows, lo_fr *mask_code, lo_fr *mask_lin, lo_fr *mask_quad) {
    const uint32_t l = j->l, k = j->k;
    rowdesc *d; size_t R = plan_rows(j, &d);
    lo_rng wit, enc; lo_rng_init(&wit, j->witness_key); lo_rng_init(&enc, j->encoding_seed);
    batch_run(j, &enc, rows);                                               /* batch rows come first, with their own pads */
    /* both streams are counter-mode: a row's draws start at the prefix sum of the draws before it, so the rows are formed
     * independently (OpenMP over rows for the timing runs; the z rows of quadratic triples in a second pass) */
    uint64_t *wpos = malloc(sizeof(uint64_t) * (R + 1)), *epos = malloc(sizeof(uint64_t) * (R + 1));
    for (size_t r = 0; r < R; r++) {
        wpos[r] = wit.pos; epos[r] = enc.pos;
        if (d[r].kind >= RK_INIT) continue;
        if (d[r].kind != 3) wit.pos += d[r].data;
        enc.pos += k - l;
    }
    const int FT = j->threads > 0 ? j->threads : 1;
    (void)FT;
    for (int pass = 0; pass < 2; pass++) {
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(FT)
#endif
        for (long r = 0; r < (long)R; r++) {
            lo_fr *row = rows + (size_t)r * k;
            if (d[r].kind >= RK_INIT || (d[r].kind == 3) != (pass == 1)) continue;
            memset(row, 0, sizeof(lo_fr) * k);
            if (d[r].kind != 3) {                                              /* linear, x, y: fresh witnesses */
                lo_rng w = wit; w.pos = wpos[r]; lo_rng_fill(&w, row, d[r].data);
                if (j->witness_bits == 32 || j->witness_bits == 64)
                    for (uint32_t i = 0; i < d[r].data; i++) { row[i].v[0] &= j->witness_bits == 32 ? 0xffffffffull : ~0ull; row[i].v[1] = row[i].v[2] = row[i].v[3] = 0; }
            }
            else {                                                          /* z = x * y */
                const lo_fr *y = row - k, *x = row - 2 * (size_t)k;
                for (uint32_t i = 0; i < d[r].data; i++) lo_fr_mul(&row[i], &x[i], &y[i]);
            }
            lo_rng e = enc; e.pos = epos[r];
            lo_rng_fill(&e, row + l, k - l);                                /* pad_encoding_random */
        }
    }
    free(wpos); free(epos);
    /* masks (witness_manager.hpp:271-321) */
    lo_rng_fill(&enc, mask_code, l);
    memset(mask_code + l, 0, sizeof(lo_fr) * (k - l));
    lo_fr sum; lo_fr_from_u64(&sum, 0);
    memset(mask_lin, 0, sizeof(lo_fr) * 2 * k);
    for (uint32_t i = 0; i + 1 < l; i++) { lo_rng_next(&enc, &mask_lin[2 * i + 1]); lo_fr_add(&sum, &sum, &mask_lin[2 * i + 1]); }
    lo_fr_neg(&mask_lin[2 * (l - 1) + 1], &sum);
    lo_rng_fill(&enc, mask_lin + 2 * l, 2 * (k - l));
    memset(mask_quad, 0, sizeof(lo_fr) * 2 * k);
    for (uint32_t i = 0; i < l; i++) lo_rng_next(&enc, &mask_quad[2 * i + 1]);
    lo_rng_fill(&enc, mask_quad + 2 * l, 2 * (k - l));
    free(d);
}

/* ------------------------------------------------------------------ protobuf writer */
typedef struct { uint8_t *p; size_t len, cap; } wr;
static void wb(wr *w, const void *d, size_t n) { if (w->p && w->len + n <= w->cap) memcpy(w->p + w->len, d, n); w->len += n; }
static void wvar(wr *w, uint64_t v) { uint8_t b[10]; int n = 0; do { b[n] = (v & 0x7f) | (v > 0x7f ? 0x80 : 0); v >>= 7; n++; } while (v); wb(w, b, n); }
static size_t varlen(uint64_t v) { size_t n = 1; while (v > 0x7f) { v >>= 7; n++; } return n; }
static void wtag(wr *w, uint32_t field, uint32_t wt) { wvar(w, ((uint64_t)field << 3) | wt); }
static void wuint(wr *w, uint32_t field, uint64_t v) { if (v) { wtag(w, field, 0); wvar(w, v); } }
static void wbytes(wr *w, uint32_t field, const void *d, size_t n) { wtag(w, field, 2); wvar(w, n); wb(w, d, n); }
static size_t len_digest(void) { return 1 + 1 + 32; }                 /* HashDigest{1: bytes(32)} */
static void wdigest(wr *w, uint32_t field, const uint8_t d[32]) { wtag(w, field, 2); wvar(w, len_digest()); wbytes(w, 1, d, 32); }
static size_t len_fixedvec(size_t nbytes) { return nbytes ? 1 + varlen(nbytes) + nbytes : 0; }
static void wfixedvec(wr *w, uint32_t field, const void *d, size_t nbytes) {
    wtag(w, field, 2); wvar(w, len_fixedvec(nbytes));
    if (nbytes) wbytes(w, 1, d, nbytes);
}

size_t lo_serialize_proof(uint8_t *out, size_t cap, const char *version, const uint8_t program_hash[32],
                          int64_t generated_at, uint32_t k, uint32_t n, uint32_t t,
                          const uint8_t root[32], const uint8_t *siblings, size_t nsib,
                          const uint32_t *leaf_idx, size_t nidx,
                          const lo_fr *code, const lo_fr *lin, const lo_fr *quad,
                          const lo_fr *samples, size_t nsample_elems) {
    wr w = {out, 0, cap};
    /* --- ProofMetadata (webgpu_prover.cpp:410-427) */
    size_t vlen = strlen(version);
    size_t ts_len = generated_at ? 1 + varlen((uint64_t)generated_at) : 0;
    size_t meta_len = (vlen ? 1 + varlen(vlen) + vlen : 0) + 2 /*schema=1*/ + 2 /*type=1*/
                    + 1 + 1 + len_digest() + 1 + 1 + ts_len
                    + 1 + varlen(k) + 1 + varlen(n) + 1 + varlen(t) + 1 + varlen(128);
    wtag(&w, 1, 2); wvar(&w, meta_len);
    if (vlen) wbytes(&w, 1, version, vlen);
    wuint(&w, 2, 1);
    wuint(&w, 3, 1);                       /* PROOF_TYPE_CLASSIC */
    wdigest(&w, 4, program_hash);
    wtag(&w, 5, 2); wvar(&w, ts_len); if (generated_at) { wtag(&w, 1, 0); wvar(&w, (uint64_t)generated_at); }
    wuint(&w, 6, k); wuint(&w, 7, n); wuint(&w, 8, t); wuint(&w, 9, 128);
    /* --- LigeroProof */
    size_t idx_bytes = 0; for (size_t i = 0; i < nidx; i++) idx_bytes += varlen(leaf_idx[i]);
    size_t md_len = 2 /*alg=1*/ + 1 + 1 + len_digest() + nsib * (1 + 1 + len_digest())
                  + (nidx ? 1 + varlen(idx_bytes) + idx_bytes : 0);
    size_t vec_bytes = (size_t)n * 32, smp_bytes = nsample_elems * 32;
    size_t proof_len = 1 + varlen(md_len) + md_len
                     + 3 * (1 + varlen(len_fixedvec(vec_bytes)) + len_fixedvec(vec_bytes))
                     + 1 + varlen(len_fixedvec(smp_bytes)) + len_fixedvec(smp_bytes);
    wtag(&w, 2, 2); wvar(&w, proof_len);
    wtag(&w, 1, 2); wvar(&w, md_len);
    wuint(&w, 1, 1);                       /* HASH_ALGORITHM_SHA256 */
    wdigest(&w, 2, root);
    for (size_t i = 0; i < nsib; i++) wdigest(&w, 3, siblings + 32 * i);
    if (nidx) { wtag(&w, 4, 2); wvar(&w, idx_bytes); for (size_t i = 0; i < nidx; i++) wvar(&w, leaf_idx[i]); }
    wfixedvec(&w, 2, code, vec_bytes);
    wfixedvec(&w, 3, lin, vec_bytes);
    wfixedvec(&w, 4, quad, vec_bytes);
    wfixedvec(&w, 5, samples, smp_bytes);
    return w.len;
}

/* ------------------------------------------------------------------ prover */
static void encode_row(const lo_ctx *c, lo_fr *cw, const lo_fr *msg, uint32_t k, uint32_t n) {
    memcpy(cw, msg, sizeof(lo_fr) * k); memset(cw + k, 0, sizeof(lo_fr) * (n - k)); lo_encode(c, cw);
}
static void encode_mask2k(const lo_ctx *c, lo_fr *cw, const lo_fr *msg2k, uint32_t k, uint32_t n) {
    memcpy(cw, msg2k, sizeof(lo_fr) * 2 * k); memset(cw + 2 * k, 0, sizeof(lo_fr) * (n - 2 * k)); lo_encode_2k(c, cw);
}
static void gather(lo_fr *dst, const lo_fr *cw, const uint32_t *idx, uint32_t t) { for (uint32_t i = 0; i < t; i++) dst[i] = cw[idx[i]]; }

/* rows per batch, never splitting a quadratic triple (x,y,z must be encoded together) */
static size_t batch_len(const rowdesc *d, size_t b, size_t R, size_t B) {
    size_t nb = R - b < B ? R - b : B;
    while (nb > 0 && mid_of_group(d[b + nb - 1].kind)) nb--;
    return nb;
}

/* dense stage-2 randomness rows: one linear-stream draw per data slot, in commit order */
static void rand_row(lo_rng *lin, lo_fr *row, uint32_t data, uint32_t k) { memset(row, 0, sizeof(lo_fr) * k); lo_rng_fill(lin, row, data); }

void lo_rand_rows(const lo_job *j, const uint8_t stage1_seed[32], lo_fr *rands, lo_fr *const_sum) {
    const uint32_t k = j->k;
    rowdesc *d; size_t R = plan_rows(j, &d);
    lo_fr *rows = NULL, *mc = NULL, *ml = NULL, *mq = NULL, *rr = malloc(sizeof(lo_fr) * k);
    if (const_sum) {
        rows = malloc(sizeof(lo_fr) * (R ? R : 1) * k);
        mc = malloc(sizeof(lo_fr) * k); ml = malloc(sizeof(lo_fr) * 2 * k); mq = malloc(sizeof(lo_fr) * 2 * k);
        lo_form_rows(j, rows, mc, ml, mq);
    }
    lo_rng lin; lo_rng_init(&lin, stage1_seed);
    lo_fr csum; lo_fr_from_u64(&csum, 0);
    for (size_t r = 0; r < R; r++) {
        lo_fr *dst = rands ? rands + r * k : rr;
        rand_row(&lin, dst, d[r].data, k);
        if (const_sum) for (uint32_t i = 0; i < d[r].data; i++) { lo_fr pr; lo_fr_mul(&pr, &rows[r * k + i], &dst[i]); lo_fr_add(&csum, &csum, &pr); }
    }
    if (const_sum) lo_fr_neg(const_sum, &csum);
    free(rows); free(mc); free(ml); free(mq); free(rr); free(d);
}

/* The three stages over a formed row stream.  d / rows / masks: the committed rows in commit order.  given_rands == NULL: the dense
 * randomness rows of the synthetic stream are drawn here (d[r].data draws per row); otherwise given_rands (R x k) are the rows the
 * constraint generator produced (witness_manager's linear_random_ / quadratic_random_, nonbatch_context.hpp:654-700) and given_const its
 * constant sum (witness_manager.hpp:393-405 constsum) -- NULL: minus the sum of the inner products. */
static int prove_core(const lo_job *j, rowdesc *d, size_t R, lo_fr *rows, lo_fr *mc, lo_fr *ml, lo_fr *mq,
                      const lo_fr *given_rands, const lo_fr *given_const, lo_proof *P) {
    const uint32_t l = j->l, k = j->k, n = j->n, t = j->t;
    const int T = j->threads > 0 ? j->threads : 1;
    lo_omp_threads = T;
    lo_ctx *c = lo_ctx_new(l, k, n);
    if (!c) return -1;
    P->rows = R + 3;
    const size_t B = T > 32 ? 2 * (size_t)T : 64; /* rows encoded per parallel batch: at least two per thread */
    lo_fr *cws = malloc(sizeof(lo_fr) * B * n), *rws = malloc(sizeof(lo_fr) * B * n);
    lo_fr *m3 = malloc(sizeof(lo_fr) * 3 * n);
    /* column block of the hash / accumulator passes: at least 2 blocks per thread (n = 32768, 256 threads -> 64 columns) */
    const long CB = (long)n / (2 * T) >= 256 ? 256 : ((long)n / (2 * T) >= 16 ? (long)n / (2 * T) : 16);

    /* ---- stage 1 (nonbatch_context.hpp:445-486,555-558; webgpu_prover.cpp:255-282) */
    double t0 = now_s();
    lo_sha256 *st = malloc(sizeof(lo_sha256) * n);
    lo_colsha_init(st, n);
    for (size_t b = 0; b < R; b += B) {
        size_t nb = R - b < B ? R - b : B;
        lo_encode_rows(c, rows + b * k, cws, nb, T);
        /* per-row sha256_digest_update calls, run column-block parallel across the batch for the timing run */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(T)
#endif
        for (long jb = 0; jb < (long)n; jb += CB) {
            size_t blk = n - jb < CB ? n - jb : CB;
            for (size_t r = 0; r < nb; r++) lo_colsha_update(st + jb, cws + r * n + jb, blk);
        }
    }
    encode_row(c, m3, mc, k, n); encode_mask2k(c, m3 + n, ml, k, n); encode_mask2k(c, m3 + 2 * (size_t)n, mq, k, n);
    for (int r = 0; r < 3; r++) lo_colsha_update(st, m3 + (size_t)r * n, n);
    uint8_t *leaves = malloc(32 * (size_t)n), *nodes = malloc(32 * lo_merkle_nodes(n));
    lo_colsha_final(st, leaves, n);
    lo_merkle_build(leaves, n, nodes);
    memcpy(P->root, nodes, 32);
    uint8_t ih[32]; lo_instance_hash(j->public_args, j->public_arg_lens, j->n_public_args, ih);
    lo_stage1_seed(P->root, ih, P->stage1_seed);
    P->t_stage1 = now_s() - t0;

    /* ---- stage 2 (nonbatch_context.hpp:654-780; webgpu_prover.cpp:297-341): re-encodes every row */
    t0 = now_s();
    lo_rng code_rng, lin_rng, quad_rng;
    lo_rng_init(&code_rng, P->stage1_seed); lo_rng_init(&lin_rng, P->stage1_seed); lo_rng_init(&quad_rng, P->stage1_seed);
    P->code = calloc(n, sizeof(lo_fr)); P->lin = calloc(n, sizeof(lo_fr)); P->quad = calloc(n, sizeof(lo_fr));
    lo_fr *rrows = malloc(sizeof(lo_fr) * B * k), *tmp1 = malloc(sizeof(lo_fr) * n), *tmp2 = malloc(sizeof(lo_fr) * n);
    lo_fr csum; lo_fr_from_u64(&csum, 0);
    for (size_t b = 0, nb; b < R; b += nb) {
        nb = batch_len(d, b, R, B);
        /* dense randomness rows: positions in the linear stream are prefix sums, so rows are independent */
        uint64_t *lpos = malloc(sizeof(uint64_t) * nb);
        lo_fr *psum = malloc(sizeof(lo_fr) * nb), *rcs = malloc(sizeof(lo_fr) * nb), *rqs = malloc(sizeof(lo_fr) * nb);
        for (size_t r = 0; r < nb; r++) { lpos[r] = lin_rng.pos; lin_rng.pos += d[b + r].data; }
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(T)
#endif
        for (long r = 0; r < (long)nb; r++) {
            lo_rng lr = lin_rng; lr.pos = lpos[r];
            if (given_rands) memcpy(rrows + r * k, given_rands + (b + r) * k, sizeof(lo_fr) * k);
            else rand_row(&lr, rrows + r * k, d[b + r].data, k);
            const lo_fr *w = rows + (b + r) * k, *rr = rrows + r * k;
            lo_fr acc; lo_fr_from_u64(&acc, 0);
            for (uint32_t i = 0; i < (given_rands ? k : d[b + r].data); i++) { lo_fr pr; lo_fr_mul(&pr, &w[i], &rr[i]); lo_fr_add(&acc, &acc, &pr); }
            psum[r] = acc;
        }
        for (size_t r = 0; r < nb; r++) lo_fr_add(&csum, &csum, &psum[r]);
        lo_encode_rows(c, rows + b * k, cws, nb, T);
        lo_encode_rows(c, rrows, rws, nb, T);
        for (size_t r = 0; r < nb; r++) {                       /* coefficient draws in call order */
            const int kd = d[b + r].kind;
            if (has_code_check(kd)) lo_rng_next(&code_rng, &rcs[r]);
            if (kd == 3 || kd == RK_BIT || kd == RK_EQY || kd == RK_BQZ) lo_rng_next(&quad_rng, &rqs[r]);
        }
        /* the per-row executor calls of check_code / check_linear / check_quadratic, column-block parallel */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(T)
#endif
        for (long jb = 0; jb < (long)n; jb += CB) {
            size_t blk = n - jb < CB ? n - jb : CB;
            for (size_t r = 0; r < nb; r++) {
                const int kd = d[b + r].kind;
                if (has_code_check(kd)) lo_eltwise(LO_OP_FMA_CONST, cws + r * n + jb, NULL, P->code + jb, blk, &rcs[r], 0);
                if (d[b + r].data || given_rands) lo_eltwise(LO_OP_FMA, cws + r * n + jb, rws + r * n + jb, P->lin + jb, blk, NULL, 0);
                if (kd == 3 || kd == RK_BQZ || kd == RK_BIT) {           /* check_quadratic: x*y - z (bit: x*x - x) */
                    const size_t rx = kd == RK_BIT ? r : r - 2, ry = kd == RK_BIT ? r : r - 1;
                    lo_eltwise(LO_OP_MUL, cws + rx * n + jb, cws + ry * n + jb, tmp1 + jb, blk, NULL, 0);
                    lo_eltwise(LO_OP_SUB, tmp1 + jb, cws + r * n + jb, tmp2 + jb, blk, NULL, 0);
                    lo_eltwise(LO_OP_FMA_CONST, tmp2 + jb, NULL, P->quad + jb, blk, &rqs[r], 0);
                } else if (kd == RK_EQY) {                               /* on_batch_equal: quad += r * (x - y) */
                    lo_eltwise(LO_OP_SUB, cws + (r - 1) * n + jb, cws + r * n + jb, tmp2 + jb, blk, NULL, 0);
                    lo_eltwise(LO_OP_FMA_CONST, tmp2 + jb, NULL, P->quad + jb, blk, &rqs[r], 0);
                }
            }
        }
        free(lpos); free(psum); free(rcs); free(rqs);
    }
    lo_eltwise(LO_OP_ADD_ASSIGN, m3, NULL, P->code, n, NULL, 0);
    lo_eltwise(LO_OP_ADD_ASSIGN, m3 + n, NULL, P->lin, n, NULL, 0);
    lo_eltwise(LO_OP_ADD_ASSIGN, m3 + 2 * (size_t)n, NULL, P->quad, n, NULL, 0);
    if (given_const) P->const_sum = *given_const; else lo_fr_neg(&P->const_sum, &csum);
    lo_stage2_seed(P->root, P->code, P->lin, P->quad, n, P->stage2_seed);
    P->sample_idx = malloc(sizeof(uint32_t) * t);
    lo_sample_indices(P->stage2_seed, n, t, P->sample_idx);
    uint8_t *sib = malloc(32 * (size_t)t * 32);
    size_t nsib = lo_merkle_decommit(nodes, n, P->sample_idx, t, sib, (size_t)t * 32);
    /* self-check (webgpu_prover.cpp:355-386,465-469) */
    lo_fr *dec = malloc(sizeof(lo_fr) * n);
    memcpy(dec, P->code, sizeof(lo_fr) * n); lo_decode(c, dec);
    P->valid_code = 1; for (uint32_t i = k; i < n; i++) if (!lo_fr_is_zero(&dec[i])) P->valid_code = 0;
    memcpy(dec, P->lin, sizeof(lo_fr) * n); lo_decode(c, dec);
    lo_fr acc = P->const_sum; for (uint32_t i = 0; i < l; i++) lo_fr_add(&acc, &acc, &dec[i]);
    P->valid_linear = lo_fr_is_zero(&acc);
    memcpy(dec, P->quad, sizeof(lo_fr) * n); lo_decode(c, dec);
    P->valid_quad = 1; for (uint32_t i = 0; i < l; i++) if (!lo_fr_is_zero(&dec[i])) P->valid_quad = 0;
    P->t_stage2 = now_s() - t0;

    /* ---- stage 3 (nonbatch_context.hpp:924-1000): re-encodes every row again, gathers t columns */
    t0 = now_s();
    P->samples = malloc(sizeof(lo_fr) * (R + 3) * t);
    for (size_t b = 0; b < R; b += B) {
        size_t nb = R - b < B ? R - b : B;
        lo_encode_rows(c, rows + b * k, cws, nb, T);
        for (size_t r = 0; r < nb; r++) gather(P->samples + (b + r) * t, cws + r * n, P->sample_idx, t);
    }
    for (int r = 0; r < 3; r++) gather(P->samples + (R + r) * t, m3 + (size_t)r * n, P->sample_idx, t);
    uint8_t ph[32] = {0};
    size_t need = lo_serialize_proof(NULL, 0, "1.5.0", ph, j->generated_at, k, n, t, P->root, sib, nsib,
                                     P->sample_idx, t, P->code, P->lin, P->quad, P->samples, (R + 3) * t);
    P->proof = malloc(need); P->proof_len = need;
    lo_serialize_proof(P->proof, need, "1.5.0", ph, j->generated_at, k, n, t, P->root, sib, nsib,
                       P->sample_idx, t, P->code, P->lin, P->quad, P->samples, (R + 3) * t);
    P->t_stage3 = now_s() - t0;

    free(dec); free(sib); free(rrows); free(tmp1); free(tmp2); free(leaves); free(nodes); free(st);
    free(cws); free(rws); free(m3);
    lo_ctx_free(c);
    return 0;
}

int lo_prove(const lo_job *j, lo_proof *P) {
    memset(P, 0, sizeof *P);
    const uint32_t k = j->k;
    if (batch_plan(j, NULL) < 0) return -1;
    rowdesc *d; size_t R = plan_rows(j, &d);
    lo_fr *rows = malloc(sizeof(lo_fr) * (R ? R : 1) * k);
    lo_fr *mc = malloc(sizeof(lo_fr) * k), *ml = malloc(sizeof(lo_fr) * 2 * k), *mq = malloc(sizeof(lo_fr) * 2 * k);
    lo_form_rows(j, rows, mc, ml, mq);
    const int rc = prove_core(j, d, R, rows, mc, ml, mq, NULL, NULL, P);
    free(rows); free(mc); free(ml); free(mq); free(d);
    return rc;
}

/* masks as process_masks draws them (witness_manager.hpp:271-321) from the encoding stream at element position `pos` */
void lo_form_masks(const uint8_t encoding_seed[32], uint64_t pos, uint32_t l, uint32_t k, lo_fr *mask_code, lo_fr *mask_lin, lo_fr *mask_quad) {
    lo_rng enc; lo_rng_init(&enc, encoding_seed); enc.pos = pos;
    lo_rng_fill(&enc, mask_code, l);
    memset(mask_code + l, 0, sizeof(lo_fr) * (k - l));
    lo_fr sum; lo_fr_from_u64(&sum, 0);
    memset(mask_lin, 0, sizeof(lo_fr) * 2 * k);
    for (uint32_t i = 0; i + 1 < l; i++) { lo_rng_next(&enc, &mask_lin[2 * i + 1]); lo_fr_add(&sum, &sum, &mask_lin[2 * i + 1]); }
    lo_fr_neg(&mask_lin[2 * (l - 1) + 1], &sum);
    lo_rng_fill(&enc, mask_lin + 2 * l, 2 * (k - l));
    memset(mask_quad, 0, sizeof(lo_fr) * 2 * k);
    for (uint32_t i = 0; i < l; i++) lo_rng_next(&enc, &mask_quad[2 * i + 1]);
    lo_rng_fill(&enc, mask_quad + 2 * l, 2 * (k - l));
}

/* The prover over a row stream formed ELSEWHERE (a constraint generator's callbacks: tests feed it the stream recorded from the
 * reference's own witness_manager, tests/golden/ref_rows_*.npz): kinds[r] in {0 linear, 1 / 2 / 3 x / y / z of a triple}, rows R x k
 * with their pads in place, the three masks, the randomness rows R x k and the constant sum of the stage-2 replay.  Uses j->l, k, n, t,
 * generated_at, threads and the public arguments; the synthetic-stream members of j are ignored. */
int lo_prove_rows(const lo_job *j, const uint8_t *kinds, size_t R, const lo_fr *rows, const lo_fr *mask_code, const lo_fr *mask_lin,
                  const lo_fr *mask_quad, const lo_fr *rands, const lo_fr *const_sum, lo_proof *P) {
    memset(P, 0, sizeof *P);
    const uint32_t k = j->k;
    rowdesc *d = malloc(sizeof(rowdesc) * (R ? R : 1));
    for (size_t r = 0; r < R; r++) {
        if (kinds[r] > 3) { free(d); return -1; }
        if ((kinds[r] == 1 && !(r + 2 < R && kinds[r + 1] == 2 && kinds[r + 2] == 3)) || ((kinds[r] == 2 || kinds[r] == 3) && !(r > 0 && kinds[r - 1] == kinds[r] - 1))) { free(d); return -1; }
        d[r] = (rowdesc){kinds[r], 0};
    }
    lo_fr *rw = malloc(sizeof(lo_fr) * (R ? R : 1) * k), *mc = malloc(sizeof(lo_fr) * k), *ml = malloc(sizeof(lo_fr) * 2 * k), *mq = malloc(sizeof(lo_fr) * 2 * k);
    memcpy(rw, rows, sizeof(lo_fr) * R * k); memcpy(mc, mask_code, sizeof(lo_fr) * k);
    memcpy(ml, mask_lin, sizeof(lo_fr) * 2 * k); memcpy(mq, mask_quad, sizeof(lo_fr) * 2 * k);
    const int rc = prove_core(j, d, R, rw, mc, ml, mq, rands, const_sum, P);
    free(rw); free(mc); free(ml); free(mq); free(d);
    return rc;
}
void lo_proof_free(lo_proof *p) { free(p->sample_idx); free(p->code); free(p->lin); free(p->quad); free(p->samples); free(p->proof); memset(p, 0, sizeof *p); }

/* ------------------------------------------------------------------ verifier (webgpu_verifier.cpp:263-452,
 * nonbatch_context.hpp:1219-1287): minimal proto reader for the envelope written above. */
typedef struct { const uint8_t *p, *end; } rd;
static int rvar(rd *r, uint64_t *v) { *v = 0; for (int s = 0; r->p < r->end && s < 70; s += 7) { uint8_t b = *r->p++; *v |= (uint64_t)(b & 0x7f) << s; if (!(b & 0x80)) return 1; } return 0; }
static int rlen(rd *r, rd *sub) { uint64_t n; if (!rvar(r, &n) || n > (uint64_t)(r->end - r->p)) return 0; sub->p = r->p; sub->end = r->p + n; r->p += n; return 1; }
static int rskip(rd *r, uint32_t wt) { uint64_t v; rd s; if (wt == 0) return rvar(r, &v); if (wt == 2) return rlen(r, &s); if (wt == 5) { r->p += 4; return r->p <= r->end; } if (wt == 1) { r->p += 8; return r->p <= r->end; } return 0; }
static int rdigest(rd *s, uint8_t out[32]) { uint64_t tag; rd b; if (!rvar(s, &tag) || tag != 0x0a || !rlen(s, &b) || b.end - b.p != 32) return 0; memcpy(out, b.p, 32); return 1; }
static int rfixed(rd *s, const uint8_t **data, size_t *nbytes) { *data = NULL; *nbytes = 0; if (s->p == s->end) return 1; uint64_t tag; rd b; if (!rvar(s, &tag) || tag != 0x0a || !rlen(s, &b)) return 0; *data = b.p; *nbytes = b.end - b.p; return 1; }

int lo_verify(const lo_job *j, const lo_fr *const_sum, const uint8_t *proof, size_t len) {
    const uint32_t l = j->l, k = j->k, n = j->n, t = j->t;
    rd top = {proof, proof + len}, meta = {0, 0}, body = {0, 0};
    while (top.p < top.end) { uint64_t tag; if (!rvar(&top, &tag)) return 0; if (tag == 0x0a) { if (!rlen(&top, &meta)) return 0; } else if (tag == 0x12) { if (!rlen(&top, &body)) return 0; } else if (!rskip(&top, tag & 7)) return 0; }
    if (!body.p) return 0;
    uint32_t mk = 0, mn = 0, mt = 0;
    while (meta.p < meta.end) { uint64_t tag, v; if (!rvar(&meta, &tag)) return 0; if ((tag & 7) == 0) { if (!rvar(&meta, &v)) return 0; if ((tag >> 3) == 6) mk = (uint32_t)v; if ((tag >> 3) == 7) mn = (uint32_t)v; if ((tag >> 3) == 8) mt = (uint32_t)v; } else if (!rskip(&meta, tag & 7)) return 0; }
    if (mk != k || mn != n || mt != t) return 0;
    uint8_t root[32] = {0}; uint8_t *sib = malloc(32 * (size_t)t * 40); size_t nsib = 0;
    uint32_t *idx = malloc(sizeof(uint32_t) * (t + 1)); size_t nidx = 0;
    const uint8_t *code = 0, *lin = 0, *quad = 0, *smp = 0; size_t cb = 0, lb = 0, qb = 0, sb = 0;
    int ok = 1;
    while (ok && body.p < body.end) {
        uint64_t tag; rd s;
        if (!rvar(&body, &tag) || (tag & 7) != 2 || !rlen(&body, &s)) { ok = 0; break; }
        switch (tag >> 3) {
        case 1:
            while (ok && s.p < s.end) {
                uint64_t t2; if (!rvar(&s, &t2)) { ok = 0; break; }
                if (t2 == 0x08) { uint64_t v; ok = rvar(&s, &v); }
                else if (t2 == 0x12) { rd x; ok = rlen(&s, &x) && rdigest(&x, root); }
                else if (t2 == 0x1a) { rd x; ok = rlen(&s, &x) && nsib < (size_t)t * 40 && rdigest(&x, sib + 32 * nsib); nsib++; }
                else if (t2 == 0x22) { rd x; ok = rlen(&s, &x); while (ok && x.p < x.end) { uint64_t v; ok = rvar(&x, &v) && nidx < t; if (ok) idx[nidx++] = (uint32_t)v; } }
                else ok = rskip(&s, t2 & 7);
            }
            break;
        case 2: ok = rfixed(&s, &code, &cb); break;
        case 3: ok = rfixed(&s, &lin, &lb); break;
        case 4: ok = rfixed(&s, &quad, &qb); break;
        case 5: ok = rfixed(&s, &smp, &sb); break;
        default: break;
        }
    }
    if (batch_plan(j, NULL) < 0) ok = 0;
    rowdesc *d = NULL; size_t R = plan_rows(j, &d);
    if (ok && (cb != 32ull * n || lb != 32ull * n || qb != 32ull * n || sb != 32ull * (R + 3) * t || nidx != t)) ok = 0;
    lo_ctx *c = ok ? lo_ctx_new(l, k, n) : NULL;
    if (!c) ok = 0;
    if (ok) {
        lo_fr *pc = malloc(sizeof(lo_fr) * n), *pl = malloc(sizeof(lo_fr) * n), *pq = malloc(sizeof(lo_fr) * n);
        lo_fr *S = malloc(sizeof(lo_fr) * (R + 3) * t);
        memcpy(pc, code, 32ull * n); memcpy(pl, lin, 32ull * n); memcpy(pq, quad, 32ull * n); memcpy(S, smp, 32ull * (R + 3) * t);
        uint8_t ih[32], s1[32], s2[32];
        lo_instance_hash(j->public_args, j->public_arg_lens, j->n_public_args, ih); lo_stage1_seed(root, ih, s1); lo_stage2_seed(root, pc, pl, pq, n, s2);
        uint32_t *si = malloc(sizeof(uint32_t) * t);
        lo_sample_indices(s2, n, t, si);
        if (memcmp(si, idx, sizeof(uint32_t) * t)) ok = 0;
        /* re-run the (public) constraint stream on the opened columns */
        lo_rng code_rng, lin_rng, quad_rng; lo_rng_init(&code_rng, s1); lo_rng_init(&lin_rng, s1); lo_rng_init(&quad_rng, s1);
        lo_sha256 *st = malloc(sizeof(lo_sha256) * t); lo_colsha_init(st, t);
        lo_fr *vc = calloc(t, sizeof(lo_fr)), *vl = calloc(t, sizeof(lo_fr)), *vq = calloc(t, sizeof(lo_fr));
        lo_fr *rr = malloc(sizeof(lo_fr) * n), *rg = malloc(sizeof(lo_fr) * t), *t1 = malloc(sizeof(lo_fr) * t), *t2 = malloc(sizeof(lo_fr) * t);
        for (size_t r = 0; ok && r < R; r++) {
            const lo_fr *s = S + r * t;
            lo_colsha_update(st, s, t);
            const int kd = d[r].kind;
            if (d[r].data) {
                rand_row(&lin_rng, rr, d[r].data, k); memset(rr + k, 0, sizeof(lo_fr) * (n - k)); lo_encode(c, rr); gather(rg, rr, si, t);
                lo_eltwise(LO_OP_FMA, s, rg, vl, t, NULL, 0);
            }
            if (has_code_check(kd)) {
                lo_fr rc; lo_rng_next(&code_rng, &rc);
                lo_eltwise(LO_OP_FMA_CONST, s, NULL, vc, t, &rc, 0);
            }
            if (kd == 3 || kd == RK_BQZ || kd == RK_BIT) {
                lo_fr rq; lo_rng_next(&quad_rng, &rq);
                const lo_fr *sx = kd == RK_BIT ? s : s - 2 * (size_t)t, *sy = kd == RK_BIT ? s : s - t;
                lo_eltwise(LO_OP_MUL, sx, sy, t1, t, NULL, 0);
                lo_eltwise(LO_OP_SUB, t1, s, t2, t, NULL, 0);
                lo_eltwise(LO_OP_FMA_CONST, t2, NULL, vq, t, &rq, 0);
            } else if (kd == RK_EQY) {
                lo_fr rq; lo_rng_next(&quad_rng, &rq);
                lo_eltwise(LO_OP_SUB, s - t, s, t2, t, NULL, 0);
                lo_eltwise(LO_OP_FMA_CONST, t2, NULL, vq, t, &rq, 0);
            }
        }
        for (int m = 0; m < 3; m++) lo_colsha_update(st, S + (R + m) * t, t);
        lo_eltwise(LO_OP_ADD_ASSIGN, S + (R + 0) * t, NULL, vc, t, NULL, 0);
        lo_eltwise(LO_OP_ADD_ASSIGN, S + (R + 1) * t, NULL, vl, t, NULL, 0);
        lo_eltwise(LO_OP_ADD_ASSIGN, S + (R + 2) * t, NULL, vq, t, NULL, 0);
        uint8_t *lv = malloc(32 * (size_t)t), vroot[32];
        lo_colsha_final(st, lv, t);
        if (!lo_merkle_recommit(n, si, t, lv, sib, nsib, vroot) || memcmp(vroot, root, 32)) ok = 0;
        for (uint32_t i = 0; i < t; i++) {
            if (lo_fr_cmp(&pc[si[i]], &vc[i]) || lo_fr_cmp(&pl[si[i]], &vl[i]) || lo_fr_cmp(&pq[si[i]], &vq[i])) ok = 0;
        }
        /* the verifier recomputes the constant sum from the public constraint stream (linear_sums,
         * webgpu_verifier.cpp:318); in the synthetic stream it is a public input of the statement */
        lo_fr derived;
        if (!const_sum) { lo_rand_rows(j, s1, NULL, &derived); const_sum = &derived; }
        lo_decode(c, pl); { lo_fr acc = *const_sum; for (uint32_t i = 0; i < l; i++) lo_fr_add(&acc, &acc, &pl[i]); if (!lo_fr_is_zero(&acc)) ok = 0; }
        lo_decode(c, pc); for (uint32_t i = k; i < n; i++) if (!lo_fr_is_zero(&pc[i])) ok = 0;
        lo_decode(c, pq); for (uint32_t i = 0; i < l; i++) if (!lo_fr_is_zero(&pq[i])) ok = 0;
        free(pc); free(pl); free(pq); free(S); free(si); free(st); free(vc); free(vl); free(vq); free(rr); free(rg); free(t1); free(t2); free(lv);
    }
    if (c) lo_ctx_free(c);
    free(d); free(sib); free(idx);
    return ok;
}
