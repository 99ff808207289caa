// verifier.hip -- lig_synth_verify.
#include "prover_common.hpp"

// =====================================================================================================================
// Verifier (src/webgpu_verifier.cpp:263-452 with nonbatch_verifier_context, include/zkp/nonbatch_context.hpp:1081-1388)
// for the synthetic constraint stream: re-derives both seeds and the sampled columns, re-runs the public constraint
// stream on the 192 opened columns (column hash, code / linear / quadratic accumulators; the randomness rows are
// re-generated and encoded, then read at the sampled positions), recommits the Merkle root from the 192 leaves and the
// sibling hashes, decodes the prover's three polynomials, and evaluates the reference's seven acceptance predicates.
namespace {

struct PbReader {
    const uint8_t* p; const uint8_t* end;
    bool var(uint64_t& v) { v = 0; for (int sh = 0; p < end && sh < 70; sh += 7) { const uint8_t b = *p++; v |= (uint64_t)(b & 0x7f) << sh; if (!(b & 0x80)) return true; } return false; }
    bool len(PbReader& sub) { uint64_t n; if (!var(n) || n > (uint64_t)(end - p)) return false; sub.p = p; sub.end = p + n; p += n; return true; }
    bool skip(uint32_t wt) { uint64_t v; PbReader s; if (wt == 0) return var(v); if (wt == 2) return len(s); if (wt == 5) { p += 4; return p <= end; } if (wt == 1) { p += 8; return p <= end; } return false; }
};
bool read_digest(PbReader s, uint8_t out[32]) { uint64_t tag; PbReader b; if (!s.var(tag) || tag != 0x0a || !s.len(b) || b.end - b.p != 32) return false; std::memcpy(out, b.p, 32); return true; }
bool read_fixed(PbReader s, const uint8_t*& data, size_t& nbytes) { data = nullptr; nbytes = 0; if (s.p == s.end) return true; uint64_t tag; PbReader b; if (!s.var(tag) || tag != 0x0a || !s.len(b)) return false; data = b.p; nbytes = (size_t)(b.end - b.p); return true; }

// merkle_tree::recommit (include/zkp/merkle_tree.hpp:232-318) with the canonical sibling order
bool recommit(size_t P, const std::vector<uint32_t>& idx, const uint8_t* leaf_digests, const std::vector<uint8_t>& sib, uint8_t root[32]) {
    std::vector<uint8_t> cur(P * 32, 0), nxt(P * 32, 0), known(P, 0), upper(P, 0);
    for (size_t i = 0; i < idx.size(); i++) { if (idx[i] >= P) return false; known[idx[i]] = 1; std::memcpy(&cur[32 * (size_t)idx[i]], leaf_digests + 32 * i, 32); }
    size_t used = 0, width = P;
    while (width > 1) {
        std::fill(upper.begin(), upper.end(), 0);
        for (size_t ll = 0; ll < width; ll += 2) {
            const bool kl = known[ll], kr = known[ll + 1];
            if (!kl && !kr) continue;
            uint8_t pair[64];
            if (kl) std::memcpy(pair, &cur[32 * ll], 32); else { if (32 * (used + 1) > sib.size()) return false; std::memcpy(pair, &sib[32 * used++], 32); }
            if (kr) std::memcpy(pair + 32, &cur[32 * (ll + 1)], 32); else { if (32 * (used + 1) > sib.size()) return false; std::memcpy(pair + 32, &sib[32 * used++], 32); }
            Sha256().add(pair, 64).finish(&nxt[32 * (ll / 2)]);
            upper[ll / 2] = 1;
        }
        cur.swap(nxt); known.swap(upper);
        width /= 2;
    }
    if (32 * used != sib.size()) return false;
    std::memcpy(root, cur.data(), 32);
    return true;
}

}  // namespace

// where the verifier's randomness rows (and, for the synthetic stream, the public targets) come from
struct VSource {
    const uint8_t* witness_key = nullptr;      // synthetic stream: dense rows generated from rows[].data; the public targets for the derived constant
    const fr* rand_dev = nullptr;              // caller's rows (R x k) on the device, used in place
    const uint8_t* rand_host = nullptr;        // caller's rows in host memory, uploaded chunk by chunk
};

// The verifier proper.  seeds_only: stop after the envelope is parsed and both seeds / the sample indices are re-derived
// (lig_rows_verify_begin: the caller needs the stage-1 seed to produce its randomness rows).
static int verify_core(lig_ctx* c, const std::vector<RowDesc>& rows, const uint8_t ih[32], const VSource& src, const uint8_t* const_sum,
                       const uint8_t* proof, size_t proof_len, lig_verify_info* out, uint8_t* seed1_out, bool seeds_only) {
    std::memset(out, 0, sizeof *out);
    const auto t_begin = clk::now();
    struct Stamp { lig_verify_info* o; decltype(t_begin) t0; ~Stamp() { o->ms_total = ms_since(t0); } } stamp{out, t_begin};
    const uint32_t l = c->l, k = c->k, n = c->n, t = 192;
    hipStream_t s = c->stream;
    const size_t R = rows.size();
    // LIG_TRACE=1: synchronised phase marks on stderr (debug aid)
    const bool trace_on = lig::knobs().trace;
    auto t_mark = clk::now();
    auto mark = [&](const char* what) {
        if (!trace_on) return;
        (void)hipStreamSynchronize(s);
        std::fprintf(stderr, "[lig_verify] %-34s %8.3f ms\n", what, ms_since(t_mark));
        t_mark = clk::now();
    };
    // ---- parse the envelope (proto/ligero_proof.proto; deserialize_proof, proof_serializer.hpp:193-226)
    PbReader top{proof, proof + proof_len}, meta{nullptr, nullptr}, body{nullptr, nullptr};
    while (top.p < top.end) {
        uint64_t tag;
        if (!top.var(tag)) return LIG_OK;
        if (tag == 0x0a) { if (!top.len(meta)) return LIG_OK; }
        else if (tag == 0x12) { if (!top.len(body)) return LIG_OK; }
        else if (!top.skip(tag & 7)) return LIG_OK;
    }
    if (!body.p || !meta.p) return LIG_OK;
    uint32_t mk = 0, mn = 0, mt = 0;
    while (meta.p < meta.end) {
        uint64_t tag, v;
        if (!meta.var(tag)) return LIG_OK;
        if ((tag & 7) == 0) { if (!meta.var(v)) return LIG_OK; if ((tag >> 3) == 6) mk = (uint32_t)v; if ((tag >> 3) == 7) mn = (uint32_t)v; if ((tag >> 3) == 8) mt = (uint32_t)v; }
        else if (!meta.skip(tag & 7)) return LIG_OK;
    }
    if (mk != k || mn != n || mt != t) return LIG_OK;
    uint8_t root[32] = {0};
    std::vector<uint8_t> sib;
    std::vector<uint32_t> pidx;
    const uint8_t *pcode = nullptr, *plin = nullptr, *pquad = nullptr, *psmp = nullptr;
    size_t cb = 0, lb = 0, qb = 0, sb = 0;
    while (body.p < body.end) {
        uint64_t tag; PbReader f;
        if (!body.var(tag) || (tag & 7) != 2 || !body.len(f)) return LIG_OK;
        switch (tag >> 3) {
            case 1:
                while (f.p < f.end) {
                    uint64_t t2; PbReader x;
                    if (!f.var(t2)) return LIG_OK;
                    if (t2 == 0x08) { uint64_t v; if (!f.var(v)) return LIG_OK; }
                    else if (t2 == 0x12) { if (!f.len(x) || !read_digest(x, root)) return LIG_OK; }
                    else if (t2 == 0x1a) { uint8_t d[32]; if (!f.len(x) || !read_digest(x, d)) return LIG_OK; sib.insert(sib.end(), d, d + 32); if (sib.size() > 32u * t * 40) return LIG_OK; }
                    else if (t2 == 0x22) { if (!f.len(x)) return LIG_OK; while (x.p < x.end) { uint64_t v; if (!x.var(v) || pidx.size() > t) return LIG_OK; pidx.push_back((uint32_t)v); } }
                    else if (!f.skip(t2 & 7)) return LIG_OK;
                }
                break;
            case 2: if (!read_fixed(f, pcode, cb)) return LIG_OK; break;
            case 3: if (!read_fixed(f, plin, lb)) return LIG_OK; break;
            case 4: if (!read_fixed(f, pquad, qb)) return LIG_OK; break;
            case 5: if (!read_fixed(f, psmp, sb)) return LIG_OK; break;
            default: break;
        }
    }
    const size_t vec = (size_t)n * 32, smp_bytes = (R + 3) * (size_t)t * 32;
    if (cb != vec || lb != vec || qb != vec || sb != smp_bytes || pidx.size() != t) return LIG_OK;
    // every opened / accumulator element must be a canonical residue
    auto canonical_all = [](const uint8_t* p, size_t count) { for (size_t i = 0; i < count; i++) { H::Fr v; std::memcpy(v.v, p + 32 * i, 32); if (H::geq(v, H::P)) return false; } return true; };
    if (!canonical_all(pcode, n) || !canonical_all(plin, n) || !canonical_all(pquad, n) || !canonical_all(psmp, (R + 3) * (size_t)t)) return LIG_OK;
    out->parsed = 1;
    // ---- seeds and sample indices (src/webgpu_verifier.cpp:268-293)
    uint8_t seed1[32], seed2[32];
    {
        Sha256().add("LigetronStage1", 15).add(root, 32).add(ih, 32).finish(seed1);
        Sha256().add("LigetronStage2", 15).add(root, 32).add(pcode, vec).add(plin, vec).add(pquad, vec).finish(seed2);
    }
    const std::vector<uint32_t> idx = sample_columns(seed2, n, t);
    mark("parse, canonical checks, seeds, indices");
    out->indices_match = idx == pidx;
    if (seed1_out) std::memcpy(seed1_out, seed1, 32);
    if (!out->indices_match || seeds_only) return LIG_OK;
    // ---- device buffers
    const size_t CH = 512;
    fr *dS = nullptr, *drand = nullptr, *drcw = nullptr, *drg = nullptr, *dacc = nullptr, *dparts = nullptr, *dpoly = nullptr;
    uint32_t *dsha = nullptr, *dleaves = nullptr, *dtri = nullptr;
    lig::f29s* dcoef = nullptr;
    // device buffers come from the context's verifier workspace: slot i of this call reuses slot i of the previous one when it is
    // large enough (a verification allocates ~2.5 GB for 2^24 constraints; hipMalloc / hipFree of that per call cost milliseconds
    // and, right after another workload freed its memory, two orders of magnitude more)
    std::vector<void*> owned;                                   // (kept for the guard below: nothing is freed per call any more)
    size_t ws_slot = 0;
    auto dm0 = [&](void** p, size_t bytes, bool zero) -> int {
        const size_t need = bytes ? bytes : 16;
        // slot = the ordinal of the request in this call.  The request sequence is the same for every call mode up to the
        // derive-only buffers, which are requested LAST (below): a derive call after a non-derive call (or the other way round)
        // re-uses every common slot instead of shifting them all by one and re-allocating 2.5 GB
        if (ws_slot == c->vws.size()) c->vws.push_back({nullptr, 0});
        auto& w = c->vws[ws_slot++];
        if (w.second < need) {
            if (w.first) { HIP_TRY(c, hipStreamSynchronize(s)); HIP_TRY(c, hipFree(w.first)); w = {nullptr, 0}; }
            HIP_TRY(c, hipMalloc(&w.first, need));
            w.second = need;
        }
        *p = w.first;
        if (zero) HIP_TRY(c, hipMemsetAsync(*p, 0, bytes, s));
        return LIG_OK;
    };
    auto dm = [&](void** p, size_t bytes) -> int { return dm0(p, bytes, true); };
    struct Cleanup { std::vector<void*>& v; lig_ctx* c; uint32_t*& sha; ~Cleanup() { (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->stream2); if (c->stream3) (void)hipStreamSynchronize(c->stream3); c->sha.erase(sha); for (void* p : v) (void)hipFree(p); } };
    Cleanup cleanup{owned, c, dsha};            // constructed before the first allocation: a failing TRY below frees what exists
    const std::vector<uint32_t> triples = quad_terms(rows);
    const size_t NT = triples.size() / 3;
    TRY(dm((void**)&dS, smp_bytes));
    TRY(dm0((void**)&drand, 2 * CH * (size_t)k * 32, false));        // double-buffered, every element written by the sampler
    TRY(dm0((void**)&drcw, CH * (size_t)n * 32, false));
    TRY(dm((void**)&drg, (R ? R : 1) * (size_t)t * 32));
    TRY(dm((void**)&dacc, 3 * (size_t)t * 32));
    const uint32_t vgroup = 64;                                  // rows per lazily summed partial of the 192-wide accumulators
    const size_t vspan = 64 * (size_t)vgroup, vgroups = 64;      // rows per accumulate pass: 64 partials of < 2p each stay far below 2^261
    TRY(dm((void**)&dparts, 2 * vgroups * (size_t)t * 32));
    TRY(dm((void**)&dpoly, 3 * vec));
    TRY(dm((void**)&dsha, lig_sha_state_bytes(t)));
    TRY(dm((void**)&dleaves, (size_t)t * 32));
    TRY(dm((void**)&dtri, (triples.size() ? triples.size() : 1) * 4));
    TRY(dm((void**)&dcoef, (R + 2 * NT + 1) * sizeof(lig::f29s)));
    // The constant of the linear test is derived from public data (lig_hip.h): the public targets b of the synthetic
    // statement (= the witness_key stream, regenerated here like lig_synth_prepare does) against the same coefficient rows.
    const bool derive = const_sum == nullptr && src.witness_key != nullptr;
    if (!derive && !const_sum) return LIG_E_ARG;
    size_t RBv = 0;
    while (RBv < R && rows[RBv].kind >= RK_INIT) RBv++;
    fr *dW = nullptr, *dpl = nullptr, *dsum = nullptr;
    const size_t pgroups = (CH + lig_tune::GROUP / 4 - 1) / (lig_tune::GROUP / 4);
    if (derive) {
        TRY(dm((void**)&dW, (R ? R : 1) * (size_t)k * 32));
        TRY(dm((void**)&dpl, (pgroups + 1) * (size_t)k * 32));
        TRY(dm((void**)&dsum, 32));
    }
    if (derive) TRY(lig_internal_synth_witness(c, src.witness_key, rows, RBv, dW));    // before the stage-1 key is installed below
    HIP_TRY(c, hipMemcpyAsync(dS, psmp, smp_bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(dpoly, pcode, vec, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(dpoly + n, plin, vec, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(dpoly + 2 * (size_t)n, pquad, vec, hipMemcpyHostToDevice, s));
    if (!triples.empty()) TRY(lig_internal_upload_small(c, dtri, triples.data(), triples.size() * 4, s));
    {
        std::vector<H::Fr> rc, rq;
        FieldStream code(seed1), quad(seed1);
        size_t n_code = 0;
        for (size_t r = 0; r < R; r++) n_code += has_code_check(rows[r].kind);
        code.next(n_code, rc);
        quad.next(NT, rq);
        std::vector<lig::f29s> coef(R + 2 * NT + 1);
        std::memset(coef.data(), 0, coef.size() * sizeof(lig::f29s));
        const H::Fr R261sq = H::mul(R261, R261);
        for (size_t r = 0, ci = 0; r < R; r++) if (has_code_check(rows[r].kind)) coef[r] = to_f29s_host(rc[ci++], R261);
        for (size_t i = 0; i < NT; i++) { coef[R + i] = to_f29s_host(rq[i], R261sq); coef[R + NT + i] = to_f29s_host(rq[i], R261); }
        TRY(lig_internal_upload_small(c, dcoef, coef.data(), coef.size() * sizeof(lig::f29s), s));
        uint32_t rk[60];
        lig::aes256_expand_host(seed1, rk);
        TRY(lig_internal_upload_small(c, c->rk_dev, rk, sizeof rk, s));
    }
    // ---- column hash of the opened columns -> 192 leaves -> recommit (webgpu_verifier.cpp:309-310)
    mark("buffers, uploads, coefficients");
    // (a chain of (R+3)/2 compressions on only 192 lanes: latency-bound, 2.7 ms for 2^24 constraints.  It is cut into the same
    // row chunks as the randomness rows and rides on the side stream behind each sampler launch: 0.27 ms of sampling + 0.67 ms
    // of hashing per chunk stay under the 1.1 ms the main stream needs to encode a chunk.  A third stream was measured to
    // share a hardware queue with the main one -- the hash then simply serialised with the encodes.)
    std::vector<uint8_t> leaves((size_t)t * 32);
    HIP_TRY(c, hipEventRecord(c->ev_join, s));                  // dS uploaded
    // ---- randomness rows of the public stream, encoded, read at the sampled positions; accumulators on 192-vectors
    TRY(lig_sample_init(c, idx.data(), idx.size()));
    fr* vc = dacc; fr* vl = dacc + t; fr* vq = dacc + 2 * (size_t)t;
    // the sampler of chunk b+1 runs on the side stream under the encode of chunk b (double-buffered rows)
    uint64_t lpos = 0;
    hipStream_t s2 = c->stream2;
    hipEvent_t ev_ready[2] = {nullptr, nullptr}, ev_used[2] = {nullptr, nullptr};
    struct Events { hipEvent_t* a; hipEvent_t* b; ~Events() { for (int i = 0; i < 2; i++) { if (a[i]) (void)hipEventDestroy(a[i]); if (b[i]) (void)hipEventDestroy(b[i]); } } } events{ev_ready, ev_used};
    for (int i = 0; i < 2; i++) { HIP_TRY(c, hipEventCreateWithFlags(&ev_ready[i], hipEventDisableTiming)); HIP_TRY(c, hipEventCreateWithFlags(&ev_used[i], hipEventDisableTiming)); }
    HIP_TRY(c, hipEventRecord(c->ev_fork, s));                 // key upload done
    HIP_TRY(c, hipStreamWaitEvent(s2, c->ev_fork, 0));
    HIP_TRY(c, hipStreamWaitEvent(s2, c->ev_join, 0));
    lig::launch_sha_init(s2, dsha, t);
    const size_t n_chunks = (R + CH - 1) / CH;
    auto rand_buf = [&](size_t ci) -> fr* { return src.rand_dev ? const_cast<fr*>(src.rand_dev) + ci * CH * (size_t)k : drand + (ci & 1) * CH * (size_t)k; };
    auto sample_chunk = [&](size_t ci) -> int {
        const size_t b = ci * CH, nb = std::min(CH, R - b);
        fr* rb = rand_buf(ci);
        auto hash_chunk = [&]() { lig::launch_sha_update_rows(s2, dsha, t, dS + b * t, t, nb, b); };      // opened rows b .. b+nb of every sampled column
        if (src.rand_dev) { HIP_TRY(c, hipEventRecord(ev_ready[ci & 1], s2)); hash_chunk(); return LIG_OK; }
        if (ci >= 2) HIP_TRY(c, hipStreamWaitEvent(s2, ev_used[ci & 1], 0));
        if (src.rand_host) {
            HIP_TRY(c, hipMemcpyAsync(rb, src.rand_host + b * (size_t)k * 32, nb * (size_t)k * 32, hipMemcpyHostToDevice, s2));
            HIP_TRY(c, hipEventRecord(ev_ready[ci & 1], s2));
            hash_chunk();
            return LIG_OK;
        }
        for (size_t r = 0; r < nb;) {
            size_t run = 1;
            const uint32_t d = rows[b + r].data;
            while (r + run < nb && rows[b + r + run].data == d) run++;
            lig::launch_rng_fill_rows_dense(s2, c->rk_dev, lpos, rb + r * k, run, d, k);
            lpos += (uint64_t)run * d; r += run;
        }
        HIP_TRY(c, hipEventRecord(ev_ready[ci & 1], s2));
        hash_chunk();
        return LIG_OK;
    };
    if (n_chunks) TRY(sample_chunk(0));
    for (size_t ci = 0; ci < n_chunks; ci++) {
        const size_t b = ci * CH, nb = std::min(CH, R - b);
        fr* rb = rand_buf(ci);
        if (ci + 1 < n_chunks) TRY(sample_chunk(ci + 1));
        HIP_TRY(c, hipStreamWaitEvent(s, ev_ready[ci & 1], 0));
        TRY(lig_internal_encode_rows(c, rb, drcw, nb, lig::ENC_PLANAR));      // cosets 1..3 as planes; coset 0 is the row itself
        if (derive) lig::launch_rlc_accumulate29(s, dW + b * k, k, 1, rb, k, nb, k, nullptr, nullptr, dpl + k, lig_tune::GROUP / 4);   // sum_r b_r o rho_r
        lig::launch_gather_rows_planar(s, lig::CwView{rb, drcw, k}, nb, c->sample_idx, t, drg + b * t);
        HIP_TRY(c, hipEventRecord(ev_used[ci & 1], s));      // the gather still reads the rows themselves (coset 0)
    }
    // code / linear accumulators on the 192 opened positions: one pass per 4096 rows after the encodes (only 192 columns wide:
    // inside the loop it was a latency-bound 8-workgroup launch per chunk that the next encode had to wait for)
    for (size_t b = 0; b < R; b += vspan)
        lig::launch_rlc_rows29(s, dS + b * t, t, 1, drg + b * t, t, std::min(vspan, R - b), t, dcoef + b, vc, vl, dparts, dparts + vgroups * (size_t)t, vgroup);
    mark("randomness rows: sample, encode, gather, accumulate");
    lig::launch_quad_rows29(s, dS, t, 1, t, dtri, dcoef + R, dcoef + R + NT, NT, vq);
    lig::launch_eltwise(s, LIG_OP_ADD_ASSIGN, dS + R * (size_t)t, nullptr, vc, t, fr{}, 0);          // opened mask columns
    lig::launch_eltwise(s, LIG_OP_ADD_ASSIGN, dS + (R + 1) * (size_t)t, nullptr, vl, t, fr{}, 0);
    lig::launch_eltwise(s, LIG_OP_ADD_ASSIGN, dS + (R + 2) * (size_t)t, nullptr, vq, t, fr{}, 0);
    std::vector<H::Fr> vacc(3 * (size_t)t);
    HIP_TRY(c, hipMemcpyAsync(vacc.data(), dacc, vacc.size() * 32, hipMemcpyDeviceToHost, s));
    H::Fr derived = H::from_u64(0);
    if (derive) {
        lig::launch_rlc_combine(s, dpl, dpl + k, (uint32_t)pgroups, k);
        lig::launch_sum_elems(s, dpl, k, 1, dsum, nullptr);
        HIP_TRY(c, hipMemcpyAsync(&derived, dsum, 32, hipMemcpyDeviceToHost, s));
    }
    // ---- decode the prover's polynomials (webgpu_verifier.cpp:355-393)
    std::vector<H::Fr> dec(3 * (size_t)n);
    for (int a = 0; a < 3; a++) {
        TRY(lig_decode(c, dpoly + (size_t)a * n));
        HIP_TRY(c, hipMemcpyAsync(dec.data() + (size_t)a * n, dpoly + (size_t)a * n, vec, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(c, hipStreamSynchronize(s));
    mark("quad, masks, decodes");
    lig::launch_sha_update_rows(s2, dsha, t, dS + R * (size_t)t, t, 3, R);      // the three opened mask rows
    lig::launch_sha_final(s2, dsha, t, R + 3, dleaves);
    HIP_TRY(c, hipMemcpyAsync(leaves.data(), dleaves, leaves.size(), hipMemcpyDeviceToHost, s2));
    HIP_TRY(c, hipStreamSynchronize(s2));                       // the leaves of the opened columns
    mark("column hash join");
    // ---- the seven predicates (webgpu_verifier.cpp:412-442)
    uint8_t vroot[32];
    size_t P = 1;
    while (P < n) P <<= 1;
    out->valid_merkle = recommit(P, idx, leaves.data(), sib, vroot) && !std::memcmp(vroot, root, 32);
    auto is_zero = [](const H::Fr& v) { return !(v.v[0] | v.v[1] | v.v[2] | v.v[3]); };
    out->valid_code = 1;
    for (uint32_t i = k; i < n; i++) if (!is_zero(dec[i])) out->valid_code = 0;
    {
        H::Fr a;
        if (derive) a = H::neg(derived);
        else { std::memcpy(a.v, const_sum, 32); if (H::geq(a, H::P)) return LIG_OK; }
        for (uint32_t i = 0; i < l; i++) a = H::add(a, dec[(size_t)n + i]);
        out->valid_linear = is_zero(a);
    }
    out->valid_quad = 1;
    for (uint32_t i = 0; i < l; i++) if (!is_zero(dec[2 * (size_t)n + i])) out->valid_quad = 0;
    out->code_equal = out->linear_equal = out->quad_equal = 1;
    for (uint32_t i = 0; i < t; i++) {
        if (std::memcmp(pcode + 32 * (size_t)idx[i], &vacc[i], 32)) out->code_equal = 0;
        if (std::memcmp(plin + 32 * (size_t)idx[i], &vacc[t + i], 32)) out->linear_equal = 0;
        if (std::memcmp(pquad + 32 * (size_t)idx[i], &vacc[2 * (size_t)t + i], 32)) out->quad_equal = 0;
    }
    out->accept = out->valid_merkle && out->valid_code && out->valid_linear && out->valid_quad && out->code_equal && out->linear_equal && out->quad_equal;
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}

static bool instance_hash_v(const uint8_t* args, const uint64_t* lens, uint64_t n_args, uint8_t ih[32]) {
    if (n_args && (!args || !lens)) return false;
    std::memset(ih, 0, 32);
    Sha256().add(ih, 32).add("Ligero", 7).finish(ih);
    for (uint64_t i = 0; i < n_args; i++) {
        uint8_t prev[32];
        std::memcpy(prev, ih, 32);
        Sha256().add(prev, 32).add(args, lens[i]).finish(ih);
        args += lens[i];
    }
    return true;
}

// the verifier's side of a rows job (lig_rows_verify_*): kinds + public data, the proof, and between the two calls the
// caller's randomness rows
struct lig_vtrace {
    lig_ctx* c = nullptr;
    std::vector<RowDesc> rows;
    uint8_t ih[32] = {0};
    std::vector<uint8_t> proof;
};

extern "C" {

int lig_synth_verify(lig_ctx* c, const lig_synth_job* job, const uint8_t* const_sum, const uint8_t* proof, size_t proof_len,
                     lig_verify_info* out) {
    CHECK_CTX(c);
    if (!job || !proof || !out) return LIG_E_ARG;
    // ---- row plan of the public constraint stream
    std::vector<RowDesc> rows;
    size_t n_init = 0;
    uint8_t ih[32];
    if (!plan_rows(*job, c->l, rows, n_init)) return LIG_E_ARG;
    if (!instance_hash_v(job->public_args, job->public_arg_lens, job->n_public_args, ih)) return LIG_E_ARG;   // src/webgpu_verifier.cpp mirrors webgpu_prover.cpp:110-168
    VSource src;
    src.witness_key = job->witness_key;
    return verify_core(c, rows, ih, src, const_sum, proof, proof_len, out, nullptr, false);
}

int lig_rows_verify_begin(lig_ctx* c, const lig_rows_job* job, const uint8_t* proof, size_t proof_len, lig_vtrace** vt, uint8_t stage1_seed[32],
                          lig_verify_info* out) {
    CHECK_CTX(c);
    if (!job || !proof || !vt || !out || (job->rows && !job->kinds)) return LIG_E_ARG;
    *vt = nullptr;
    lig_vtrace* V = new lig_vtrace();
    V->c = c;
    V->rows.resize(job->rows);
    for (size_t r = 0; r < job->rows; r++) {
        const uint8_t kd = job->kinds[r] & 0x7f;
        if (kd > RK_BQZ) { delete V; FAIL(c, LIG_E_ARG, "rows job: unknown row kind"); }
        const bool first_of_3 = kd == 1 || kd == RK_BQX, first_of_2 = kd == RK_EQX, follower = kd == 2 || kd == 3 || kd == RK_EQY || kd == RK_BQY || kd == RK_BQZ;
        const bool ok = (!first_of_3 || (r + 2 < job->rows && (job->kinds[r + 1] & 0x7f) == kd + 1 && (job->kinds[r + 2] & 0x7f) == kd + 2)) &&
                        (!first_of_2 || (r + 1 < job->rows && (job->kinds[r + 1] & 0x7f) == RK_EQY)) && (!follower || (r > 0 && (job->kinds[r - 1] & 0x7f) == kd - 1));
        if (!ok) { delete V; FAIL(c, LIG_E_ARG, "rows job: incomplete row group"); }
        V->rows[r] = RowDesc{kd, 0};
    }
    if (!instance_hash_v(job->public_args, job->public_arg_lens, job->n_public_args, V->ih)) { delete V; return LIG_E_ARG; }
    V->proof.assign(proof, proof + proof_len);
    const int rc = verify_core(c, V->rows, V->ih, VSource{}, nullptr, V->proof.data(), V->proof.size(), out, stage1_seed, true);
    if (rc != LIG_OK || !out->parsed || !out->indices_match) { delete V; return rc; }      // malformed envelope: accept = 0, no trace
    *vt = V;
    return LIG_OK;
}

int lig_rows_verify_finish(lig_vtrace* V, const void* rands, int rands_on_device, const uint8_t const_sum[32], lig_verify_info* out) {
    if (!V || !out || !const_sum) return LIG_E_ARG;
    lig_ctx* c = V->c;
    CHECK_CTX(c);
    if (!V->rows.empty() && !rands) { delete V; FAIL(c, LIG_E_ARG, "lig_rows_verify_finish: null randomness rows"); }
    VSource src;
    if (rands_on_device) src.rand_dev = (const fr*)rands; else src.rand_host = (const uint8_t*)rands;
    const int rc = verify_core(c, V->rows, V->ih, src, const_sum, V->proof.data(), V->proof.size(), out, nullptr, false);
    delete V;
    return rc;
}

void lig_vtrace_destroy(lig_vtrace* V) { delete V; }

}  // extern "C"
