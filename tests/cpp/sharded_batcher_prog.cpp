// sharded_batcher_prog.cpp -- hip_row_batcher::shard_over: ONE guest trace proved by `world` processes (one per GPU on a node; here
// all on GPU 0 with the process-to-process communicator of csrc/comm_ipc.hip).  Every rank runs the same deterministic guest (the
// oracle plays guest + witness_manager), the batcher keeps the rows of its own chunks, and every rank must end with the oracle's
// envelope.   usage: sharded_batcher_prog rank world /shm_name n_linear n_quad
// TEST CODE: links oracle/liblig_oracle.so as the checker.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/lig_hip_row_batcher.hpp"
#include "../../oracle/lig_oracle.h"

int main(int argc, char** argv) {
    if (argc < 6) return 2;
    const uint32_t rank = std::atoi(argv[1]), world = std::atoi(argv[2]);
    const uint32_t l = 320, k = 512, n = 2048;
    lo_job j;
    std::memset(&j, 0, sizeof j);
    j.l = l; j.k = k; j.n = n; j.t = 192;
    j.n_linear = std::strtoull(argv[4], nullptr, 10);
    j.n_quad = std::strtoull(argv[5], nullptr, 10);
    for (int i = 0; i < 32; i++) j.encoding_seed[i] = (uint8_t)(7 * i + 1);
    lo_synth_key(5, j.witness_key);
    j.generated_at = 777;
    j.threads = 4;
    const size_t R = lo_job_rows(&j) - 3;
    std::vector<lo_fr> rows((R ? R : 1) * (size_t)k), mc(k), ml(2 * (size_t)k), mq(2 * (size_t)k);
    std::vector<uint8_t> kinds(R ? R : 1);
    lo_form_rows(&j, rows.data(), mc.data(), ml.data(), mq.data());
    lo_row_kinds(&j, kinds.data());
    lig_ctx* ctx = nullptr;
    if (lig_ctx_create(&ctx, 0, l, k, n) != LIG_OK) { std::fprintf(stderr, "ctx: %s\n", ctx ? lig_last_error(ctx) : "?"); return 1; }
    lig_comm comm;
    if (lig_ipc_comm_create(ctx, argv[3], rank, world, &comm) != LIG_OK) { std::fprintf(stderr, "comm: %s\n", lig_last_error(ctx)); return 1; }
    int ok = 0;
    try {
        ligero::hip_proof_meta meta;
        std::memcpy(meta.encoding_seed, j.encoding_seed, 32);
        meta.generated_at = j.generated_at;
        {
            ligero::hip_row_batcher b(ctx, meta);
            b.shard_over(rank, world, &comm);
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
            replay(nullptr);
            uint8_t root[32], seed1[32];
            b.commit(root, seed1);
            std::vector<lo_fr> rands((R ? R : 1) * (size_t)k);
            lo_fr cs;
            lo_rand_rows(&j, seed1, rands.data(), &cs);
            replay(&rands);
            size_t len = 0;
            lig_proof_info info;
            const uint8_t* proof = b.prove(reinterpret_cast<const uint8_t*>(cs.v), &len, &info);
            lo_proof P;
            if (lo_prove(&j, &P) != 0) throw std::runtime_error("oracle prover failed");
            ok = len == P.proof_len && !std::memcmp(proof, P.proof, len) && !std::memcmp(root, P.root, 32) && info.valid_code && info.valid_linear && info.valid_quad;
            std::printf("rank %u: equal %d rows %zu local %zu proof_len %zu\n", rank, ok, b.rows() + 3, b.local_rows(), len);
            lo_proof_free(&P);
        }       // the batcher (and its shard) goes before the communicator
    } catch (const std::exception& e) {
        std::fprintf(stderr, "rank %u error: %s\n", rank, e.what());
    }
    lig_ipc_comm_destroy(&comm);
    lig_ctx_destroy(ctx);
    return ok ? 0 : 1;
}
