// row_batcher_prog.cpp -- drives include/lig_hip_row_batcher.hpp the way the reference's prover main drives its stage
// contexts (src/webgpu_prover.cpp:255-310): pass 1 fires linear_callback / quadratic_callback in witness_manager's commit
// order (full linear rows, full triples, partial row, partial triple, masks: witness_manager.hpp:497-503), commit, pass 2
// replays the callbacks with the per-witness randomness rows, prove.  The oracle plays the guest + witness_manager
// (lo_form_rows / lo_rand_rows) and its reference-structured prover supplies the expected envelope.
// TEST CODE: links oracle/liblig_oracle.so as the checker.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/lig_hip_row_batcher.hpp"
#include "../../oracle/lig_oracle.h"

int main(int argc, char** argv) {
    if (argc < 6) { std::fprintf(stderr, "usage: %s l k n n_linear n_quad [witness_bits: 64 = small witnesses, shipped as narrow rows]\n", argv[0]); return 2; }
    const uint32_t l = std::atoi(argv[1]), k = std::atoi(argv[2]), n = std::atoi(argv[3]);
    lo_job j;
    std::memset(&j, 0, sizeof j);
    j.l = l; j.k = k; j.n = n; j.t = 192;
    j.n_linear = std::strtoull(argv[4], nullptr, 10);
    j.n_quad = std::strtoull(argv[5], nullptr, 10);
    for (int i = 0; i < 32; i++) j.encoding_seed[i] = (uint8_t)i;
    lo_synth_key(1, j.witness_key);
    j.generated_at = 4242;
    j.threads = 8;
    j.witness_bits = argc > 6 ? (uint32_t)std::atoi(argv[6]) : 0;
    const int64_t arg_i64 = -7;                                  // one public i64 argument and one string (with its NUL)
    std::vector<uint8_t> args((const uint8_t*)&arg_i64, (const uint8_t*)&arg_i64 + 8);
    const char* str = "row-batcher";
    args.insert(args.end(), (const uint8_t*)str, (const uint8_t*)str + std::strlen(str) + 1);
    const uint64_t lens[2] = {8, std::strlen(str) + 1};
    j.public_args = args.data(); j.public_arg_lens = lens; j.n_public_args = 2;

    const size_t R = lo_job_rows(&j) - 3;
    std::vector<lo_fr> rows((R ? R : 1) * (size_t)k), mc(k), ml(2 * (size_t)k), mq(2 * (size_t)k);
    std::vector<uint8_t> kinds(R ? R : 1);
    lo_form_rows(&j, rows.data(), mc.data(), ml.data(), mq.data());
    lo_row_kinds(&j, kinds.data());

    lig_ctx* ctx = nullptr;
    if (lig_ctx_create(&ctx, 0, l, k, n) != LIG_OK) { std::fprintf(stderr, "ctx: %s\n", ctx ? lig_last_error(ctx) : "?"); return 1; }
    int ok = 0;
    try {
        ligero::hip_proof_meta meta;
        std::memcpy(meta.encoding_seed, j.encoding_seed, 32);
        meta.generated_at = j.generated_at;
        meta.public_args = {std::vector<uint8_t>(args.begin(), args.begin() + 8), std::vector<uint8_t>(args.begin() + 8, args.end())};
        meta.narrow_rows = j.witness_bits != 0;                   // x, y and linear rows fit 8 bytes per slot; the z rows (products) stay wide
        ligero::hip_row_batcher b(ctx, meta);
        auto at = [&](const std::vector<lo_fr>& v, size_t r) { return reinterpret_cast<const uint64_t*>(v.data() + r * (size_t)k); };
        auto replay = [&](const std::vector<lo_fr>* rands) {
            for (size_t r = 0; r < R;) {
                if (kinds[r] == 0) { b.linear_callback(at(rows, r), rands ? at(*rands, r) : nullptr); r += 1; }
                else {
                    b.quadratic_callback(at(rows, r), at(rows, r + 1), at(rows, r + 2), rands ? at(*rands, r) : nullptr,
                                         rands ? at(*rands, r + 1) : nullptr, rands ? at(*rands, r + 2) : nullptr);
                    r += 3;
                }
            }
            b.mask_callback(k, 2 * (size_t)k, 2 * (size_t)k);
        };
        replay(nullptr);                                         // pass 1
        uint8_t root[32], seed1[32];
        b.commit(root, seed1);
        std::vector<lo_fr> rands((R ? R : 1) * (size_t)k);
        lo_fr cs;
        lo_rand_rows(&j, seed1, rands.data(), &cs);              // the guest's second run: per-witness randomness + linear_sums
        replay(&rands);                                          // pass 2
        size_t len = 0;
        lig_proof_info info;
        const uint8_t* proof = b.prove(reinterpret_cast<const uint8_t*>(cs.v), &len, &info);
        lo_proof P;
        if (lo_prove(&j, &P) != 0) throw std::runtime_error("oracle prover failed");
        ok = len == P.proof_len && !std::memcmp(proof, P.proof, len) && !std::memcmp(root, P.root, 32) && !std::memcmp(seed1, P.stage1_seed, 32) &&
             info.valid_code && info.valid_linear && info.valid_quad;
        // the verifier's shim on the same public stream: accepts the proof, rejects it for another linear constant
        int vok = 0;
        {
            ligero::hip_row_verifier v(ctx, meta);
            v.expect_rows(std::vector<uint8_t>(kinds.begin(), kinds.begin() + R));
            auto vreplay = [&]() {
                for (size_t r = 0; r < R;) {
                    if (kinds[r] == 0) { v.linear_callback(at(rands, r)); r += 1; }
                    else { v.quadratic_callback(at(rands, r), at(rands, r + 1), at(rands, r + 2)); r += 3; }
                }
            };
            uint8_t vseed[32];
            lo_fr wrong = cs;
            wrong.v[0] ^= 1;
            const bool began = v.begin(proof, len, vseed) && !std::memcmp(vseed, seed1, 32);
            if (began) { vreplay(); vok = v.finish(reinterpret_cast<const uint8_t*>(cs.v)) ? 1 : 0; }
            if (vok && R) { vok = v.begin(proof, len, vseed) ? 1 : 0; if (vok) { vreplay(); vok = v.finish(reinterpret_cast<const uint8_t*>(wrong.v)) ? 0 : 1; } }
        }
        ok = ok && vok;
        std::printf("equal %d rows %zu proof_len %zu verifier_shim %d\n", ok, b.rows() + 3, len, vok);
        lo_proof_free(&P);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
    }
    lig_ctx_destroy(ctx);
    return ok ? 0 : 1;
}
