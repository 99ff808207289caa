// shard.hip -- one trace sharded over the GPUs of a node (lig_shard_*).
#include "prover_common.hpp"

// =====================================================================================================================
// One trace sharded over the GPUs of a node (BASELINE.json configs[3]).
//
// Rows are dealt to ranks in contiguous blocks (never splitting an x,y,z triple); every rank forms, encodes and keeps
// only its own rows.  Leaf_j hashes ALL rows in commit order, so the hash is column-partitioned: after one all-to-all of
// codeword column slices rank h owns columns [h*n/W, (h+1)*n/W) of every row, hashes them in rank (= row) order and the
// n/W leaves per rank are all-gathered; the Merkle tree is then built redundantly on every rank.  The stage-2 tests are
// sums over rows: every rank accumulates its rows on the low-degree domains (k + 2k + 2k values), the partial sums are
// all-gathered and added mod p locally (RCCL has no modular reduction).  Opened columns are all-gathered in row order.
// The collectives are supplied by the caller (lig_comm: torch.distributed over RCCL/xGMI in ligero-prover_amd/dist.py),
// this file only sees device pointers.  Every rank ends with the same envelope, byte-identical to lig_synth_prove.
struct lig_shard {
    lig_ctx* c = nullptr;
    lig_synth_job job;
    lig_comm comm;
    uint32_t rank = 0, world = 1;
    std::vector<RowDesc> rows;                 // global plan
    std::vector<size_t> bounds;                // world + 1 row boundaries
    std::vector<uint64_t> wit_pos, lin_pos;    // stream position of every global row (+1 entry)
    std::vector<uint64_t> code_ord;            // number of code-test draws before every global row (+1 entry)
    size_t RB = 0, n_init = 0;                 // leading rows committed by the batch program, of those: init rows
    size_t R = 0, r0 = 0, Rl = 0, rows_max = 0, ncol = 0;
    fr *msgs = nullptr, *cw = nullptr, *send = nullptr, *recv = nullptr, *randb = nullptr, *rhalf = nullptr, *acc = nullptr,
       *parts = nullptr, *accp = nullptr, *accg = nullptr, *dots = nullptr, *smp = nullptr, *smpg = nullptr;
    uint32_t *sha_state = nullptr, *leaves_slice = nullptr, *leaves = nullptr, *nodes = nullptr, *data_dev = nullptr, *tri_dev = nullptr;
    lig::f29s* coef_dev = nullptr;
    std::vector<uint32_t> triples;             // local row indices
    std::vector<size_t> triple_ord;            // global ordinal of each local triple
    uint8_t *h_proof = nullptr, *h_enc = nullptr, *h_nodes = nullptr, *h_small = nullptr;
    size_t h_proof_cap = 0;
    uint8_t ih[32] = {0};
};

extern "C" {

static int shard_prepare_impl(lig_ctx* c, const lig_synth_job* job, uint32_t rank, uint32_t world, lig_shard* S);
int lig_shard_prepare(lig_ctx* c, const lig_synth_job* job, uint32_t rank, uint32_t world, const lig_comm* comm, lig_shard** out) {
    CHECK_CTX(c);
    if (!job || !out || !comm || world == 0 || rank >= world) return LIG_E_ARG;
    *out = nullptr;
    lig_shard* S = new lig_shard();
    S->c = c; S->job = *job; S->comm = *comm; S->rank = rank; S->world = world;
    S->job.batch_ops = nullptr; S->job.batch_data = nullptr;
    {   // instance_hash over arg0 = "Ligero\0" and the public arguments (src/webgpu_prover.cpp:110-168)
        if (job->n_public_args && (!job->public_args || !job->public_arg_lens)) { delete S; return LIG_E_ARG; }
        std::memset(S->ih, 0, 32);
        Sha256().add(S->ih, 32).add("Ligero", 7).finish(S->ih);
        const uint8_t* a = job->public_args;
        for (uint64_t i = 0; i < job->n_public_args; i++) {
            uint8_t prev[32];
            std::memcpy(prev, S->ih, 32);
            Sha256().add(prev, 32).add(a, job->public_arg_lens[i]).finish(S->ih);
            a += job->public_arg_lens[i];
        }
        S->job.public_args = nullptr; S->job.public_arg_lens = nullptr;
    }
    const int rc = shard_prepare_impl(c, job, rank, world, S);
    if (rc != LIG_OK) { lig_shard_destroy(S); return rc; }
    *out = S;
    return LIG_OK;
}
static int shard_prepare_impl(lig_ctx* c, const lig_synth_job* job, uint32_t rank, uint32_t world, lig_shard* S) {
    const uint32_t l = c->l, k = c->k, n = c->n, t = 192;
    if (l >= k || l < 2 || t > n || k - l < t || n % world) FAIL(c, LIG_E_ARG, "sharded trace: need 2 <= l <= k - 192 and world | n");
    S->ncol = n / world;
    if (!plan_rows(*job, l, S->rows, S->n_init)) FAIL(c, LIG_E_ARG, "malformed batch program");
    const size_t R = S->R = S->rows.size();
    for (S->RB = 0; S->RB < R && S->rows[S->RB].kind >= RK_INIT; S->RB++) {}
    S->wit_pos.assign(R + 1, 0); S->lin_pos.assign(R + 1, 0); S->code_ord.assign(R + 1, 0);
    for (size_t r = 0; r < R; r++) {
        const uint8_t kd = S->rows[r].kind;
        S->wit_pos[r + 1] = S->wit_pos[r] + ((kd == 3 || kd >= RK_INIT) ? 0 : S->rows[r].data);   // z rows and batch rows draw nothing
        S->lin_pos[r + 1] = S->lin_pos[r] + S->rows[r].data;
        S->code_ord[r + 1] = S->code_ord[r] + has_code_check(kd);                                    // position in the code-test stream
    }
    S->bounds.assign(world + 1, R);
    S->bounds[0] = 0;
    for (uint32_t g = 1; g < world; g++) {
        size_t b = (size_t)(((unsigned __int128)R * g) / world);
        auto inside_group = [&](uint8_t kd) { return kd == 2 || kd == 3 || kd == RK_EQY || kd == RK_BQY || kd == RK_BQZ; };
        while (b < R && inside_group(S->rows[b].kind)) b++;                       // never split a triple / an equality pair
        S->bounds[g] = std::max(b, S->bounds[g - 1]);
    }
    for (uint32_t g = 0; g < world; g++) S->rows_max = std::max(S->rows_max, S->bounds[g + 1] - S->bounds[g]);
    if (!S->rows_max) S->rows_max = 1;
    S->r0 = S->bounds[rank]; S->Rl = S->bounds[rank + 1] - S->bounds[rank];
    const size_t Rl = S->Rl, r0 = S->r0, RM = S->rows_max;
    {   // quadratic-test terms whose rows are local, with local row indices; triple_ord = position in the quadratic stream
        const std::vector<uint32_t> all = quad_terms(S->rows);
        for (size_t i = 0; i < all.size() / 3; i++) {
            const size_t last = all[3 * i + 2];
            if (last < r0 || last >= r0 + Rl) continue;
            S->triples.push_back(all[3 * i] - (uint32_t)r0);
            S->triples.push_back(all[3 * i + 1] == 0xFFFFFFFFu ? 0xFFFFFFFFu : all[3 * i + 1] - (uint32_t)r0);
            S->triples.push_back(all[3 * i + 2] - (uint32_t)r0);
            S->triple_ord.push_back(i);
        }
    }
    const size_t chunk = lig_tune::CHUNK, groups = (chunk + lig_tune::GROUP - 1) / lig_tune::GROUP;
    auto dm = [&](void** p, size_t bytes) -> int { HIP_TRY(c, hipMalloc(p, bytes ? bytes : 16)); HIP_TRY(c, hipMemsetAsync(*p, 0, bytes, c->stream)); return LIG_OK; };
    TRY(dm((void**)&S->msgs, (Rl ? Rl : 1) * (size_t)k * 32));
    TRY(dm((void**)&S->cw, (Rl + 3) * (size_t)n * 32));
    TRY(dm((void**)&S->send, RM * (size_t)n * 32));
    TRY(dm((void**)&S->recv, RM * (size_t)n * 32));
    TRY(dm((void**)&S->randb, chunk * (size_t)k * 32));
    TRY(dm((void**)&S->rhalf, chunk * 2 * (size_t)k * 32));
    TRY(dm((void**)&S->acc, 4 * (size_t)n * 32));
    TRY(dm((void**)&S->parts, 2 * groups * (size_t)n * 32));
    TRY(dm((void**)&S->accp, 5 * (size_t)k * 32));
    TRY(dm((void**)&S->accg, (size_t)world * 5 * k * 32));
    TRY(dm((void**)&S->dots, (Rl ? Rl : 1) * 32));
    TRY(dm((void**)&S->smp, (RM + 3) * (size_t)t * 32));
    TRY(dm((void**)&S->smpg, (size_t)world * RM * t * 32));
    TRY(dm((void**)&S->sha_state, lig_sha_state_bytes(S->ncol)));
    TRY(dm((void**)&S->leaves_slice, S->ncol * 32));
    TRY(dm((void**)&S->leaves, (size_t)n * 32));
    TRY(dm((void**)&S->nodes, lig_merkle_nodes(n) * 32));
    TRY(dm((void**)&S->data_dev, (Rl ? Rl : 1) * sizeof(uint32_t)));
    TRY(dm((void**)&S->tri_dev, (S->triples.size() ? S->triples.size() : 1) * sizeof(uint32_t)));
    TRY(dm((void**)&S->coef_dev, (Rl + 2 * S->triple_ord.size() + 1) * sizeof(lig::f29s)));
    S->h_proof_cap = ((size_t)1 << 19) + 3 * (size_t)n * 32 + (R + 3) * (size_t)t * 32;
    HIP_TRY(c, hipHostMalloc((void**)&S->h_proof, S->h_proof_cap, hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc((void**)&S->h_enc, 3 * (size_t)n * 32, hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc((void**)&S->h_nodes, lig_merkle_nodes(n) * 32, hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc((void**)&S->h_small, ((Rl ? Rl : 1) + 2 * (size_t)l + 3 * (size_t)n + 2 * world) * 32, hipHostMallocDefault));
    {
        std::vector<uint32_t> d(Rl);
        for (size_t r = 0; r < Rl; r++) d[r] = S->rows[r0 + r].data;
        if (Rl) HIP_TRY(c, hipMemcpyAsync(S->data_dev, d.data(), Rl * 4, hipMemcpyHostToDevice, c->stream));
        if (!S->triples.empty()) HIP_TRY(c, hipMemcpyAsync(S->tri_dev, S->triples.data(), S->triples.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    // local witness rows: same stream positions as in the single-GPU trace
    uint32_t rk[60];
    lig::aes256_expand_host(job->witness_key, rk);
    HIP_TRY(c, hipMemcpyAsync(c->rk_dev, rk, sizeof rk, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    size_t r_first = 0;
    if (S->RB) {          // the batch program is small: every rank runs it and keeps the rows it owns
        fr* all = nullptr;
        HIP_TRY(c, hipMalloc((void**)&all, S->RB * (size_t)k * sizeof(fr)));
        const int rc = lig_run_batch_program(c, *job, all);
        const size_t lo = std::min(r0, S->RB), hi = std::min(r0 + Rl, S->RB);
        if (rc == LIG_OK && hi > lo) (void)hipMemcpyAsync(S->msgs + (lo - r0) * (size_t)k, all + lo * (size_t)k, (hi - lo) * (size_t)k * sizeof(fr), hipMemcpyDeviceToDevice, c->stream);
        (void)hipStreamSynchronize(c->stream);
        (void)hipFree(all);
        if (rc != LIG_OK) return rc;
        r_first = hi - lo;
        lig::aes256_expand_host(job->witness_key, rk);                    // the program used the encoding key
        HIP_TRY(c, hipMemcpyAsync(c->rk_dev, rk, sizeof rk, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    for (size_t r = r_first; r < Rl;) {
        const RowDesc d = S->rows[r0 + r];
        if (d.kind == 0) {
            lig::launch_rng_fill_rows(c->stream, c->rk_dev, S->wit_pos[r0 + r], S->msgs + r * k, 1, d.data, k, 0, 1, d.data);
            r += 1;
        } else {
            lig::launch_rng_fill_rows(c->stream, c->rk_dev, S->wit_pos[r0 + r], S->msgs + r * k, 2, d.data, k, 0, 1, d.data);
            lig::launch_eltwise(c->stream, LIG_OP_MUL, S->msgs + r * k, S->msgs + (r + 1) * k, S->msgs + (r + 2) * k, d.data, fr{}, 0);
            r += 3;
        }
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}

void lig_shard_destroy(lig_shard* S) {
    if (!S) return;
    (void)hipSetDevice(S->c->device);
    (void)hipStreamSynchronize(S->c->stream);
    S->c->sha.erase(S->sha_state);
    for (void* p : {(void*)S->msgs, (void*)S->cw, (void*)S->send, (void*)S->recv, (void*)S->randb, (void*)S->rhalf, (void*)S->acc,
                    (void*)S->parts, (void*)S->accp, (void*)S->accg, (void*)S->dots, (void*)S->smp, (void*)S->smpg, (void*)S->sha_state,
                    (void*)S->leaves_slice, (void*)S->leaves, (void*)S->nodes, (void*)S->data_dev, (void*)S->tri_dev, (void*)S->coef_dev})
        (void)hipFree(p);
    (void)hipHostFree(S->h_proof); (void)hipHostFree(S->h_enc); (void)hipHostFree(S->h_nodes); (void)hipHostFree(S->h_small);
    delete S;
}

int lig_shard_prove(lig_shard* S, const uint8_t** proof, size_t* proof_len, lig_proof_info* info) {
    if (!S || !proof || !proof_len || !info) return LIG_E_ARG;
    lig_ctx* c = S->c;
    CHECK_CTX(c);
    const uint32_t l = c->l, k = c->k, n = c->n, t = 192, pad = k - l, W = S->world;
    const size_t R = S->R, Rl = S->Rl, r0 = S->r0, RM = S->rows_max, ncol = S->ncol;
    hipStream_t s = c->stream;
    std::memset(info, 0, sizeof *info);
    info->rows = R + 3;
    const auto t_begin = clk::now();
    auto t0 = clk::now();
    auto comm_fail = [&](int rc, const char* what) { c->err = std::string("collective failed: ") + what; return rc ? LIG_E_STATE : LIG_OK; };

    // ---------------- stage 1
    uint32_t rk[60];
    lig::aes256_expand_host(S->job.encoding_seed, rk);
    HIP_TRY(c, hipMemcpyAsync(c->rk_dev, rk, sizeof rk, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    {   // pads of the local stream rows (batch rows carry theirs from the program); position = draws before the row
        const size_t first = std::max(r0, S->RB), last = r0 + Rl;
        if (last > first)
            lig::launch_rng_fill_rows(s, c->rk_dev, (uint64_t)(S->n_init + (first - S->RB)) * pad, S->msgs + (first - r0) * (size_t)k, last - first, pad, k, l, 1, pad);
    }
    uint64_t epos = (uint64_t)(S->n_init + (R - S->RB)) * pad;
    fr* mask = S->cw + Rl * (size_t)n; fr* mlin = mask + n; fr* mquad = mask + 2 * (size_t)n;           // masks: formed by every rank
    HIP_TRY(c, hipMemsetAsync(mask, 0, 3 * (size_t)n * 32, s));
    lig::launch_rng_fill_rows(s, c->rk_dev, epos, mask, 1, l, 0, 0, 1, 0); epos += l;
    lig::launch_rng_fill_rows(s, c->rk_dev, epos, mlin, 1, l - 1, 0, 1, 2, 0); epos += l - 1;
    lig::launch_sum_elems(s, mlin + 1, l - 1, 2, S->dots, mlin + 2 * (size_t)(l - 1) + 1);      // closing slot = -(sum of the others), on the device
    lig::launch_rng_fill_rows(s, c->rk_dev, epos, mlin, 1, 2 * pad, 0, 2 * l, 1, 0); epos += 2 * pad;
    lig::launch_rng_fill_rows(s, c->rk_dev, epos, mquad, 1, l, 0, 1, 2, 0); epos += l;
    lig::launch_rng_fill_rows(s, c->rk_dev, epos, mquad, 1, 2 * pad, 0, 2 * l, 1, 0); epos += 2 * pad;
    if (Rl) TRY(lig_internal_encode_rows(c, S->msgs, S->cw, Rl, false));
    TRY(lig_encode(c, mask));
    TRY(lig_internal_encode_2k_rows(c, mlin, 2));          // mlin and mquad are adjacent rows: one pass of 31 launches
    // column slices: block h of `send` = my rows restricted to rank h's columns
    for (uint32_t h = 0; h < W && Rl; h++)
        HIP_TRY(c, hipMemcpy2DAsync(S->send + (size_t)h * RM * ncol, ncol * 32, S->cw + (size_t)h * ncol, (size_t)n * 32, ncol * 32, Rl,
                                    hipMemcpyDeviceToDevice, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (int rc = S->comm.all_to_all(S->comm.user, S->send, S->recv, RM * ncol * 32)) return comm_fail(rc, "all_to_all(codeword column slices)");
    TRY(lig_sha_init(c, S->sha_state, ncol));
    uint64_t absorbed = 0;
    for (uint32_t g = 0; g < W; g++) {
        const size_t rg = S->bounds[g + 1] - S->bounds[g];
        lig::launch_sha_update_rows(s, S->sha_state, ncol, S->recv + (size_t)g * RM * ncol, ncol, rg, absorbed);
        absorbed += rg;
    }
    lig::launch_sha_update_rows(s, S->sha_state, ncol, mask + (size_t)S->rank * ncol, n, 3, absorbed);
    absorbed += 3;
    c->sha[S->sha_state].second = absorbed;
    TRY(lig_sha_final(c, S->sha_state, S->leaves_slice));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (int rc = S->comm.all_gather(S->comm.user, S->leaves_slice, S->leaves, ncol * 32)) return comm_fail(rc, "all_gather(leaves)");
    TRY(lig_merkle_build(c, S->leaves, n, S->nodes));
    HIP_TRY(c, hipMemcpyAsync(info->root, S->nodes, 32, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    Sha256().add("LigetronStage1", 15).add(info->root, 32).add(S->ih, 32).finish(info->stage1_seed);
    info->ms_stage1 = ms_since(t0);
    t0 = clk::now();

    // ---------------- stage 2
    const size_t NTl = S->triple_ord.size();
    {
        std::vector<H::Fr> rc, rq;
        const size_t NT = quad_terms(S->rows).size() / 3;
        FieldStream code(info->stage1_seed), quad(info->stage1_seed);
        code.next(S->code_ord[R], rc);
        quad.next(NT, rq);
        std::vector<lig::f29s> coef(Rl + 2 * NTl + 1);
        const H::Fr R261sq = H::mul(R261, R261);
        std::memset(coef.data(), 0, coef.size() * sizeof(lig::f29s));
        for (size_t r = 0; r < Rl; r++) if (has_code_check(S->rows[r0 + r].kind)) coef[r] = to_f29s_host(rc[S->code_ord[r0 + r]], R261);
        for (size_t i = 0; i < NTl; i++) { coef[Rl + i] = to_f29s_host(rq[S->triple_ord[i]], R261sq); coef[Rl + NTl + i] = to_f29s_host(rq[S->triple_ord[i]], R261); }
        HIP_TRY(c, hipMemcpyAsync(S->coef_dev, coef.data(), coef.size() * sizeof(lig::f29s), hipMemcpyHostToDevice, s));
        lig::aes256_expand_host(info->stage1_seed, rk);
        HIP_TRY(c, hipMemcpyAsync(c->rk_dev, rk, sizeof rk, hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipStreamSynchronize(s));
    }
    fr* code = S->acc; fr* lin = S->acc + n; fr* quad = S->acc + 2 * (size_t)n; fr* tmp = S->acc + 3 * (size_t)n;
    fr* linH = lin + 2 * (size_t)k; fr* linC = lin + 3 * (size_t)k;
    HIP_TRY(c, hipMemsetAsync(S->acc, 0, 3 * (size_t)n * 32, s));
    const size_t groups = (lig_tune::CHUNK + lig_tune::GROUP - 1) / lig_tune::GROUP;
    for (size_t b = 0; b < Rl; b += lig_tune::CHUNK) {
        const size_t nb = std::min(lig_tune::CHUNK, Rl - b);
        HIP_TRY(c, hipMemsetAsync(S->randb, 0, nb * (size_t)k * 32, s));
        for (size_t r = 0; r < nb;) {          // runs of rows with equal fill are contiguous in the linear stream
            size_t run = 1;
            const uint32_t d = S->rows[r0 + b + r].data;
            while (r + run < nb && S->rows[r0 + b + r + run].data == d) run++;
            lig::launch_rng_fill_rows(s, c->rk_dev, S->lin_pos[r0 + b + r], S->randb + r * k, run, d, k, 0, 1, d);
            r += run;
        }
        TRY(lig_internal_encode_rows(c, S->randb, S->rhalf, nb, true));
        lig::launch_rlc_rows29(s, S->cw + b * n + 2, n, 4, S->rhalf, k, nb, k, nullptr, nullptr, linC, S->parts,
                               S->parts + groups * (size_t)n, lig_tune::GROUP / 4);
        lig::launch_rlc_rows29(s, S->msgs + b * k, k, 1, S->randb, k, nb, k, S->coef_dev + b, code, linH, S->parts,
                               S->parts + groups * (size_t)n, lig_tune::GROUP / 4);
    }
    lig::launch_lin_interleave(s, lin, linH, linC, k);       // see lig_synth_prove: even points of <w_n^2> = message domain
    lig::launch_quad_rows29(s, S->cw, n, 2, 2 * k, S->tri_dev, S->coef_dev + Rl, S->coef_dev + Rl + NTl, NTl, quad);
    // partial sums [code (k) | lin (2k) | quad (2k)] -> every rank -> added mod p
    HIP_TRY(c, hipMemcpyAsync(S->accp, code, (size_t)k * 32, hipMemcpyDeviceToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(S->accp + k, lin, 2 * (size_t)k * 32, hipMemcpyDeviceToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(S->accp + 3 * (size_t)k, quad, 2 * (size_t)k * 32, hipMemcpyDeviceToDevice, s));
    H::Fr* dots = reinterpret_cast<H::Fr*>(S->h_small);
    HIP_TRY(c, hipStreamSynchronize(s));
    if (int rc = S->comm.all_gather(S->comm.user, S->accp, S->accg, 5 * (size_t)k * 32)) return comm_fail(rc, "all_gather(partial accumulators)");
    HIP_TRY(c, hipMemsetAsync(S->accp, 0, 5 * (size_t)k * 32, s));
    lig::launch_rlc_combine(s, S->accp, S->accg, W, 5 * k);
    HIP_TRY(c, hipMemsetAsync(S->acc, 0, 3 * (size_t)n * 32, s));
    HIP_TRY(c, hipMemcpyAsync(code, S->accp, (size_t)k * 32, hipMemcpyDeviceToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(lin, S->accp + k, 2 * (size_t)k * 32, hipMemcpyDeviceToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(quad, S->accp + 3 * (size_t)k, 2 * (size_t)k * 32, hipMemcpyDeviceToDevice, s));
    {   // linear-test constant = -(sum of the message-domain half of the combined accumulator: its even points)
        lig::launch_sum_elems(s, lin, k, 2, S->dots, nullptr);
        HIP_TRY(c, hipMemcpyAsync(dots, S->dots, 32, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        const H::Fr sum = H::neg(dots[0]);
        std::memcpy(info->const_sum, sum.v, 32);
    }
    TRY(lig_encode(c, code));
    TRY(lig_internal_extend_2k(c, lin));
    TRY(lig_internal_extend_2k(c, quad));
    lig::launch_eltwise(s, LIG_OP_ADD_ASSIGN, mask, nullptr, code, n, fr{}, 0);
    lig::launch_eltwise(s, LIG_OP_ADD_ASSIGN, mlin, nullptr, lin, n, fr{}, 0);
    lig::launch_eltwise(s, LIG_OP_ADD_ASSIGN, mquad, nullptr, quad, n, fr{}, 0);
    uint8_t* enc = S->h_enc;
    const size_t enc_bytes = 3 * (size_t)n * 32;
    HIP_TRY(c, hipMemcpyAsync(enc, S->acc, enc_bytes, hipMemcpyDeviceToHost, s));
    H::Fr* dec = reinterpret_cast<H::Fr*>(S->h_small + ((Rl ? Rl : 1) + 2 * (size_t)l) * 32);
    const fr* accs[3] = {code, lin, quad};
    for (int a3 = 0; a3 < 3; a3++) {
        HIP_TRY(c, hipMemcpyAsync(tmp, accs[a3], (size_t)n * 32, hipMemcpyDeviceToDevice, s));
        TRY(lig_decode(c, tmp));
        HIP_TRY(c, hipMemcpyAsync(dec + (size_t)a3 * n, tmp, (size_t)n * 32, hipMemcpyDeviceToHost, s));
    }
    const size_t n_nodes = lig_merkle_nodes(n);
    HIP_TRY(c, hipMemcpyAsync(S->h_nodes, S->nodes, n_nodes * 32, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    Sha256().add("LigetronStage2", 15).add(info->root, 32).add(enc, enc_bytes).finish(info->stage2_seed);
    const std::vector<uint32_t> idx = sample_columns(info->stage2_seed, n, t);
    TRY(lig_sample_init(c, idx.data(), idx.size()));
    auto is_zero = [](const H::Fr& v) { return !(v.v[0] | v.v[1] | v.v[2] | v.v[3]); };
    info->valid_code = 1;
    for (uint32_t i = k; i < n; i++) if (!is_zero(dec[i])) info->valid_code = 0;
    {
        H::Fr a;
        std::memcpy(a.v, info->const_sum, 32);
        for (uint32_t i = 0; i < l; i++) a = H::add(a, dec[(size_t)n + i]);
        info->valid_linear = is_zero(a);
    }
    info->valid_quad = 1;
    for (uint32_t i = 0; i < l; i++) if (!is_zero(dec[2 * (size_t)n + i])) info->valid_quad = 0;
    const std::vector<uint8_t> sib = decommit(S->h_nodes, (n_nodes + 1) / 2, idx);
    info->ms_stage2 = ms_since(t0);
    t0 = clk::now();

    // ---------------- stage 3
    TRY(lig_gather_rows(c, S->cw, Rl + 3, S->smp));                  // local rows, then the 3 masks
    HIP_TRY(c, hipStreamSynchronize(s));
    if (int rc = S->comm.all_gather(S->comm.user, S->smp, S->smpg, RM * (size_t)t * 32)) return comm_fail(rc, "all_gather(opened columns)");
    char ver[17] = {0};
    std::memcpy(ver, S->job.version, 16);
    const size_t smp_bytes = (R + 3) * (size_t)t * 32;
    const EnvelopeLayout lay = write_envelope(S->h_proof, S->h_proof_cap, ver, S->job.program_hash, S->job.generated_at, k, n, t,
                                              info->root, sib, idx, enc, smp_bytes);
    if (lay.total > S->h_proof_cap) FAIL(c, LIG_E_NOMEM, "proof buffer too small");
    uint8_t* dst = S->h_proof + lay.samples_off;
    for (uint32_t g = 0; g < W; g++) {
        const size_t rg = S->bounds[g + 1] - S->bounds[g];
        if (rg) HIP_TRY(c, hipMemcpyAsync(dst, S->smpg + (size_t)g * RM * t, rg * (size_t)t * 32, hipMemcpyDeviceToHost, s));
        dst += rg * (size_t)t * 32;
    }
    HIP_TRY(c, hipMemcpyAsync(dst, S->smp + Rl * (size_t)t, 3 * (size_t)t * 32, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    *proof = S->h_proof;
    *proof_len = lay.total;
    info->ms_stage3 = ms_since(t0);
    info->ms_total = ms_since(t_begin);
    HIP_TRY(c, hipGetLastError());
    return LIG_OK;
}


}  // extern "C"
