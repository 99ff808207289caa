// row_batcher_batch_prog.cpp -- the batch hooks of include/lig_hip_row_batcher.hpp (on_batch_init / _bit / _equal /
// _quadratic) driven by the guest-visible batch layer itself (include/lig_hip_vbn254fr.hpp), the way vbn254fr_module drives
// a stage context upstream (include/host_modules/vbn254fr.hpp, include/zkp/nonbatch_context.hpp:497-553):
//   pass 1: run the batch program on a fresh slab -> every hook commits device rows; on_batch_init first writes the 192
//           pad elements of the encoding stream INTO the variable, so they flow into every row derived from it;
//           then the rows of the synthetic constraint stream (linear rows + quadratic triples, formed by the oracle)
//   commit; pass 2: run the program again on a fresh slab (the guest is re-run per stage), replay, prove.
// The expected envelope is the oracle's reference-structured prover on the same job: batch program (recorded by the layer
// as lig_batch_op, same layout as lo_batch_op) + synthetic stream.
// TEST CODE: links oracle/liblig_oracle.so as the checker.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lig_hip_row_batcher.hpp"
#include "../../include/lig_hip_vbn254fr.hpp"
#include "../../oracle/lig_oracle.h"

using ligero::hip_context;
using buffer_t = hip_context::buffer_type;

struct stage_context {                    // what vbn254fr_module sees of a stage context
    stage_context(hip_context& e, ligero::hip_row_batcher& b) : exe(e), batcher(b) {}
    hip_context& executor() { return exe; }
    // nonbatch_context.hpp:502-505: the pad is written to x.slice(message_size * 32) -- through the view's slice(), so that the
    // upstream slicing semantics (hip_context::set_upstream_slice_compat) decide where it lands
    void on_batch_init(buffer_t& x) { batcher.on_batch_init(x.data(), x.slice(exe.message_size() * 32).data()); }
    void on_batch_bit(buffer_t& x) { batcher.on_batch_bit(x.data()); }
    void on_batch_equal(buffer_t& x, buffer_t& y) { batcher.on_batch_equal(x.data(), y.data()); }
    void on_batch_quadratic(buffer_t& x, buffer_t& y, buffer_t& z) { batcher.on_batch_quadratic(x.data(), y.data(), z.data()); }
    hip_context& exe;
    ligero::hip_row_batcher& batcher;
};
using layer_t = ligero::hip_vbn254fr<stage_context>;

static ligero::hip::scalar scalar_of(uint64_t lo, uint64_t hi) {
    ligero::hip::scalar s{};
    std::memcpy(s.data(), &lo, 8);
    std::memcpy(s.data() + 16, &hi, 8);
    return s;
}

// the guest: inits, a product, a quotient constrained back, an equality that holds on all k slots, constants, copies, a
// square with aliasing, a bit decomposition
static void guest(layer_t& v, bool with_bits) {
    using H = layer_t::handle_t;
    const H a = v.vbn254fr_alloc(), b = v.vbn254fr_alloc(), c = v.vbn254fr_alloc(), d = v.vbn254fr_alloc(), e = v.vbn254fr_alloc();
    uint32_t ui[10];
    for (int i = 0; i < 10; i++) ui[i] = 3 + 2 * i;
    v.vbn254fr_set_ui(a, ui, 10);                                  // I
    v.vbn254fr_set_ui_scalar(b, 9);                                // I
    v.vbn254fr_mulmod(c, a, b);                                    // Q (a, b, a*b): pads of a and b multiply as well
    v.vbn254fr_divmod(d, c, b);                                    // Q (c/b, b, c): d == a on ALL k slots (b's pads are nonzero)
    v.vbn254fr_assert_equal(d, a);                                 // E (d, a)
    const ligero::hip::scalar K = scalar_of(0x123456789abcdef0ull, 0x0fedcba987654321ull);
    v.vbn254fr_addmod(e, a, b);
    v.vbn254fr_addmod_constant(e, e, K);
    v.vbn254fr_submod_constant(e, e, scalar_of(77, 0));
    v.vbn254fr_mulmod_constant(e, e, K);
    v.vbn254fr_copy(d, e);                                         // E (d, e)
    v.vbn254fr_mulmod(e, e, e);                                    // Q with x == y, out aliasing both
    if (with_bits) {
        std::vector<ligero::hip::scalar> vals(3);
        vals[0] = scalar_of(0xffffffffffffffffull, 0x1fffffffffffffffull);
        vals[1] = scalar_of(1, 0);
        vals[2] = scalar_of(0, 0);
        v.vbn254fr_set(c, vals);                                   // I
        std::vector<H> bits(layer_t::num_bits);
        for (auto& h : bits) h = v.vbn254fr_alloc();
        v.vbn254fr_bit_decompose(bits.data(), c);                  // 254 x B: also the bits of the 192 pad elements
    }
    v.finalize();
}

// a guest that is a true statement under BOTH slicing semantics of buffer_view: variable 0 through
// write_buffer_clear, the others through the write_limbs family, one write_buffer_clear on a later variable (upstream: wipes
// the slab up to the end of that variable; products of zeros still hold)
static void guest_slicing(layer_t& v) {
    using H = layer_t::handle_t;
    H x[8];
    for (auto& h : x) h = v.vbn254fr_alloc();
    uint32_t ui[10];
    for (int i = 0; i < 10; i++) ui[i] = 3 + 2 * i;
    v.vbn254fr_set_ui(x[0], ui, 10);
    v.vbn254fr_set_scalar(x[1], scalar_of(9, 0));
    v.vbn254fr_mulmod(x[2], x[0], x[1]);
    v.vbn254fr_set(x[3], {scalar_of(5, 0), scalar_of(6, 0), scalar_of(7, 0)});
    v.vbn254fr_addmod(x[4], x[2], x[3]);
    v.vbn254fr_copy(x[5], x[4]);
    v.vbn254fr_mulmod(x[4], x[4], x[4]);
    v.vbn254fr_set(x[2], {scalar_of(11, 0), scalar_of(12, 0)});
    v.vbn254fr_mulmod(x[7], x[2], x[5]);
    v.vbn254fr_set_ui_scalar(x[6], 77);
    v.vbn254fr_mulmod(x[7], x[6], x[6]);
    v.vbn254fr_mulmod(x[7], x[7], x[2]);
    v.finalize();
}

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: %s n_linear n_quad with_bits k [slicing: declared | upstream]\n", argv[0]); return 2; }
    const uint32_t k = std::atoi(argv[4]), l = k - 192, n = 4 * k;
    const bool with_bits = std::atoi(argv[3]) != 0;
    const bool slicing = argc > 5, upstream = slicing && std::string(argv[5]) == "upstream";
    hip_context executor;
    executor.webgpu_init(k, "");
    executor.ntt_init(l, k, n, 0, 0, 0, 0, 0);
    executor.set_upstream_slice_compat(upstream);         // before the layer makes its slab
    lig_ctx* ctx = executor.native();

    lo_job j;
    std::memset(&j, 0, sizeof j);
    j.l = l; j.k = k; j.n = n; j.t = 192;
    j.n_linear = std::strtoull(argv[1], nullptr, 10);
    j.n_quad = std::strtoull(argv[2], nullptr, 10);
    for (int i = 0; i < 32; i++) j.encoding_seed[i] = (uint8_t)(200 - i);
    lo_synth_key(3, j.witness_key);
    j.generated_at = 99;
    j.threads = 8;

    int ok = 0;
    try {
        ligero::hip_proof_meta meta;
        std::memcpy(meta.encoding_seed, j.encoding_seed, 32);
        meta.generated_at = j.generated_at;
        ligero::hip_row_batcher b(ctx, meta);
        stage_context sc(executor, b);

        // ---- pass 1: batch program, recorded for the oracle
        std::vector<lig_batch_op> ops;
        std::vector<uint8_t> data;
        {
            layer_t v(&sc);
            v.record(true);
            if (slicing) guest_slicing(v); else guest(v, with_bits);
            ops = v.recorded_ops(); data = v.recorded_data();
            if (upstream != (!ops.empty() && ops[0].op == LIG_BOP_UPSTREAM_COMPAT)) throw std::runtime_error("the recorded program does not say which slicing it ran under");
        }
        static_assert(sizeof(lig_batch_op) == sizeof(lo_batch_op), "same program layout");
        j.batch_ops = reinterpret_cast<const lo_batch_op*>(ops.data()); j.n_batch_ops = ops.size();
        j.batch_data = data.data(); j.batch_data_bytes = data.size();

        // the synthetic stream's rows, in commit order behind the batch rows (the oracle plays guest + witness_manager)
        const size_t R = lo_job_rows(&j) - 3;
        std::vector<lo_fr> rows((R ? R : 1) * (size_t)k), mc(k), ml(2 * (size_t)k), mq(2 * (size_t)k);
        std::vector<uint8_t> kinds(R ? R : 1);
        lo_form_rows(&j, rows.data(), mc.data(), ml.data(), mq.data());
        lo_row_kinds(&j, kinds.data());
        size_t RB = 0;
        while (RB < R && kinds[RB] >= 4) RB++;
        if (RB != b.rows()) throw std::runtime_error("the hooks committed " + std::to_string(b.rows()) + " batch rows, the oracle plans " + std::to_string(RB));
        auto at = [&](const std::vector<lo_fr>& v, size_t r) { return reinterpret_cast<const uint64_t*>(v.data() + r * (size_t)k); };
        auto stream = [&](const std::vector<lo_fr>* rands) {
            for (size_t r = RB; r < R;) {
                if (kinds[r] == 0) { b.linear_callback(at(rows, r), rands ? at(*rands, r) : nullptr); r += 1; }
                else {
                    b.quadratic_callback(at(rows, r), at(rows, r + 1), at(rows, r + 2), rands ? at(*rands, r) : nullptr,
                                         rands ? at(*rands, r + 1) : nullptr, rands ? at(*rands, r + 2) : nullptr);
                    r += 3;
                }
            }
            b.mask_callback(k, 2 * (size_t)k, 2 * (size_t)k);
        };
        stream(nullptr);
        uint8_t root[32], seed1[32];
        b.commit(root, seed1);

        // ---- pass 2: the guest runs again (fresh slab), then the stream rows with their randomness rows
        std::vector<lo_fr> rands((R ? R : 1) * (size_t)k);
        lo_fr cs;
        lo_rand_rows(&j, seed1, rands.data(), &cs);
        {
            layer_t v(&sc);
            if (slicing) guest_slicing(v); else guest(v, with_bits);
        }
        stream(&rands);
        size_t len = 0;
        lig_proof_info info;
        const uint8_t* proof = b.prove(reinterpret_cast<const uint8_t*>(cs.v), &len, &info);

        lo_proof P;
        if (lo_prove(&j, &P) != 0) throw std::runtime_error("oracle prover failed");
        const bool same = len == P.proof_len && !std::memcmp(proof, P.proof, len);
        ok = same && !std::memcmp(root, P.root, 32) && !std::memcmp(seed1, P.stage1_seed, 32) && info.valid_code && info.valid_linear && info.valid_quad;
        std::printf("equal %d rows %zu (batch %zu) proof_len %zu root_equal %d valid %d%d%d\n", ok, b.rows() + 3, RB, len,
                    !std::memcmp(root, P.root, 32), info.valid_code, info.valid_linear, info.valid_quad);
        lo_proof_free(&P);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
    }
    return ok ? 0 : 1;
}
