// row_batcher_bench.cpp -- what the row-batching shim (include/lig_hip_row_batcher.hpp) costs at the trace sizes of BASELINE.json:
// the oracle plays the guest + witness_manager (lo_form_rows / lo_rand_rows produce the rows a constraint generator would hand to
// linear_callback, include/zkp/backend/witness_manager.hpp:200-269), the shim takes them through its callbacks into page-locked
// staging and proves through lig_rows_*.  Reported per proof: the callback passes (the shim alone: one 256 KiB copy per row, or
// none with next_slot()), commit(), prove(), and the throughput of two batchers on two contexts.
//   row_batcher_bench <log2 constraints> [proofs = 5] [check = 0 | 1: compare the envelope with the oracle's prover]
// TEST / MEASUREMENT CODE: links oracle/liblig_oracle.so as the guest and the checker.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/lig_hip_row_batcher.hpp"
#include "../../oracle/lig_oracle.h"

using clk = std::chrono::steady_clock;
static double ms(clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); }

struct Guest {
    lo_job j;
    size_t R = 0;
    uint32_t k = 0;
    std::vector<lo_fr> rows, rands, mc, ml, mq;
    std::vector<uint8_t> kinds;
    lo_fr cs;
    const uint64_t* row(size_t r) const { return reinterpret_cast<const uint64_t*>(rows.data() + r * (size_t)k); }
    const uint64_t* rnd(size_t r) const { return reinterpret_cast<const uint64_t*>(rands.data() + r * (size_t)k); }
};

// one proof through the copying callbacks; times[0..3] = pass 1, commit, pass 2, prove
static const uint8_t* prove_once(ligero::hip_row_batcher& b, Guest& g, bool have_rands, size_t* len, double times[4], uint8_t seed1[32], lig_proof_info* info) {
    auto t = clk::now();
    for (size_t r = 0; r < g.R; r++) b.linear_callback(g.row(r));
    b.mask_callback(g.k, 2 * (size_t)g.k, 2 * (size_t)g.k);
    times[0] = ms(t); t = clk::now();
    uint8_t root[32];
    b.commit(root, seed1);
    times[1] = ms(t);
    if (!have_rands) {                                       // the guest's second run (same trace => same seed => same rows every proof)
        g.rands.resize((g.R ? g.R : 1) * (size_t)g.k);
        lo_rand_rows(&g.j, seed1, g.rands.data(), &g.cs);
    }
    t = clk::now();
    for (size_t r = 0; r < g.R; r++) b.linear_callback(g.row(r), g.rnd(r));
    b.mask_callback(g.k, 2 * (size_t)g.k, 2 * (size_t)g.k);
    times[2] = ms(t); t = clk::now();
    const uint8_t* proof = b.prove(reinterpret_cast<const uint8_t*>(g.cs.v), len, info);
    times[3] = ms(t);
    return proof;
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s log2_constraints [proofs] [check]\n", argv[0]); return 2; }
    const int lg = std::atoi(argv[1]), proofs = argc > 2 ? std::atoi(argv[2]) : 5, check = argc > 3 ? std::atoi(argv[3]) : 0;
    const uint32_t l = 8000, k = 8192, n = 32768;
    Guest g;
    std::memset(&g.j, 0, sizeof g.j);
    g.j.l = l; g.j.k = k; g.j.n = n; g.j.t = 192;
    g.j.n_linear = 1ull << lg;
    for (int i = 0; i < 32; i++) g.j.encoding_seed[i] = (uint8_t)i;
    lo_synth_key(1, g.j.witness_key);
    g.j.threads = (int)std::thread::hardware_concurrency();
    if (const char* e = std::getenv("OMP_NUM_THREADS")) g.j.threads = std::atoi(e);
    g.k = k;
    g.R = lo_job_rows(&g.j) - 3;
    g.rows.resize(g.R * (size_t)k); g.mc.resize(k); g.ml.resize(2 * (size_t)k); g.mq.resize(2 * (size_t)k);
    g.kinds.resize(g.R);
    lo_form_rows(&g.j, g.rows.data(), g.mc.data(), g.ml.data(), g.mq.data());
    lo_row_kinds(&g.j, g.kinds.data());

    lig_ctx* ctx[2] = {nullptr, nullptr};
    for (int i = 0; i < 2; i++)
        if (lig_ctx_create(&ctx[i], 0, l, k, n) != LIG_OK) { std::fprintf(stderr, "ctx: %s\n", ctx[i] ? lig_last_error(ctx[i]) : "?"); return 1; }
    int ok = 1;
    try {
        ligero::hip_proof_meta meta;
        std::memcpy(meta.encoding_seed, g.j.encoding_seed, 32);
        meta.expected_rows = g.R;
        ligero::hip_row_batcher b0(ctx[0], meta), b1(ctx[1], meta);
        size_t len = 0;
        double t[4], best[4] = {1e30, 1e30, 1e30, 1e30}, best_total = 1e30;
        uint8_t seed1[32];
        lig_proof_info info;
        const uint8_t* proof = prove_once(b0, g, false, &len, t, seed1, &info);         // warm-up: allocations, first touch of the staging
        ok = info.valid_code && info.valid_linear && info.valid_quad;
        if (check) {
            lo_proof P;
            if (lo_prove(&g.j, &P) != 0) throw std::runtime_error("oracle prover failed");
            ok = ok && len == P.proof_len && !std::memcmp(proof, P.proof, len);
            lo_proof_free(&P);
        }
        for (int p = 0; p < proofs; p++) {
            b0.reset();
            (void)prove_once(b0, g, true, &len, t, seed1, &info);
            ok = ok && info.valid_code && info.valid_linear && info.valid_quad;
            const double total = t[0] + t[1] + t[2] + t[3];
            if (total < best_total) { best_total = total; for (int i = 0; i < 4; i++) best[i] = t[i]; }
        }
        // zero-copy: the guest exports its rows straight into the slots (here: the row former writes the whole matrix in place)
        double slot_ms[4] = {0, 0, 0, 0};
        {
            b0.reset();
            uint64_t* first = b0.next_slot();                    // staging is contiguous: expected_rows reserved it
            auto t0 = clk::now();
            lo_form_rows(&g.j, reinterpret_cast<lo_fr*>(first), g.mc.data(), g.ml.data(), g.mq.data());
            const double guest_ms = ms(t0);
            t0 = clk::now();
            for (size_t r = 0; r < g.R; r++) { (void)b0.next_slot(); b0.commit_slot(LIG_ROW_LINEAR); }
            slot_ms[0] = ms(t0); t0 = clk::now();
            uint8_t root[32];
            b0.commit(root, seed1);
            slot_ms[1] = ms(t0); t0 = clk::now();
            for (size_t r = 0; r < g.R; r++) { uint64_t* s = b0.next_slot(); std::memcpy(s, g.rnd(r), (size_t)k * 32); b0.commit_slot(LIG_ROW_LINEAR); }
            slot_ms[2] = ms(t0); t0 = clk::now();
            size_t len2 = 0;
            const uint8_t* p2 = b0.prove(reinterpret_cast<const uint8_t*>(g.cs.v), &len2, &info);
            slot_ms[3] = ms(t0);
            ok = ok && len2 == len && info.valid_code && info.valid_linear && info.valid_quad;
            (void)p2;
            std::fprintf(stderr, "guest row former in place: %.1f ms\n", guest_ms);
        }
        // two batchers, two contexts, two host threads: a proving service's throughput through the shim
        double pair_ms = 0;
        {
            ligero::hip_row_batcher* bs[2] = {&b0, &b1};
            int okk[2] = {1, 1};
            auto run = [&](int i, int reps) {
                for (int p = 0; p < reps; p++) {
                    size_t ln; double tt[4]; uint8_t sd[32]; lig_proof_info inf;
                    bs[i]->reset();
                    (void)prove_once(*bs[i], g, true, &ln, tt, sd, &inf);
                    okk[i] = okk[i] && inf.valid_code && inf.valid_linear && inf.valid_quad;
                }
            };
            run(1, 1);                                            // warm-up of the second batcher
            const auto t0 = clk::now();
            std::thread th0(run, 0, proofs), th1(run, 1, proofs);
            th0.join(); th1.join();
            pair_ms = ms(t0) / (2.0 * proofs);
            ok = ok && okk[0] && okk[1];
        }
        const double cons = (double)(1ull << lg);
        std::printf("{\"log2_constraints\": %d, \"rows\": %zu, \"ok\": %d, \"checked_against_oracle\": %d, "
                    "\"callbacks_copying\": {\"pass1_ms\": %.2f, \"commit_ms\": %.2f, \"pass2_ms\": %.2f, \"prove_ms\": %.2f, \"total_ms\": %.2f, "
                    "\"shim_alone_rows_per_s\": %.0f, \"shim_alone_GBps\": %.2f, \"constraints_per_s\": %.3e}, "
                    "\"slots_zero_copy\": {\"pass1_ms\": %.3f, \"commit_ms\": %.2f, \"pass2_ms_incl_guest_copy\": %.2f, \"prove_ms\": %.2f, \"commit_plus_prove_ms\": %.2f}, "
                    "\"two_batchers\": {\"ms_per_proof\": %.2f, \"constraints_per_s\": %.3e}}\n",
                    lg, g.R + 3, ok, check, best[0], best[1], best[2], best[3], best_total,
                    2.0 * g.R / ((best[0] + best[2]) * 1e-3), 2.0 * g.R * k * 32.0 / ((best[0] + best[2]) * 1e-3) / 1e9, cons / (best_total * 1e-3),
                    slot_ms[0], slot_ms[1], slot_ms[2], slot_ms[3], slot_ms[1] + slot_ms[3], pair_ms, cons / (pair_ms * 1e-3));
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        ok = 0;
    }
    for (int i = 0; i < 2; i++) lig_ctx_destroy(ctx[i]);
    return ok ? 0 : 1;
}
