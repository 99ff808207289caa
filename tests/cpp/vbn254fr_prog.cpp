// Runs a small guest-style batch program through ligero::hip_vbn254fr (include/lig_hip_vbn254fr.hpp) with a context
// that records every constraint hook the way a stage context sees it (include/zkp/nonbatch_context.hpp:497-556):
// on_batch_init writes the k - l padding slots of the variable first, then the row(s) are committed.  The log
// (kind, row count, raw k x 32-byte rows) goes to argv[1]; tests/test_vbn254fr.py replays the same program with Python
// integers and compares every committed row.  Build: g++ -std=c++17 -Iinclude ... -llig_hip
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lig_hip_vbn254fr.hpp"

using ligero::hip_context;
using buffer_t = hip_context::buffer_type;

struct recording_context {
    explicit recording_context(hip_context& e, FILE* f) : exe(e), out(f) {}
    hip_context& executor() { return exe; }
    void record(char kind, std::initializer_list<buffer_t*> rows) {
        std::fputc(kind, out);
        std::fputc((int)rows.size(), out);
        for (buffer_t* b : rows) {
            std::vector<uint8_t> h = exe.copy_to_host<uint8_t>(*b);
            std::fwrite(h.data(), 1, h.size(), out);
        }
    }
    // pad_encoding_random stand-in: pad j of the c-th initialised row = 1000 * c + j (the real stream is the AES sampler)
    void on_batch_init(buffer_t& x) {
        const size_t l = exe.message_size(), k = exe.padding_size();
        std::vector<uint64_t> limbs(4 * (k - l), 0);
        for (size_t j = 0; j < k - l; j++) limbs[4 * j] = 1000ull * inits + j;
        inits++;
        exe.write_buffer(x.slice(l * 32), limbs.data(), limbs.size());
        record('I', {&x});
    }
    void on_batch_bit(buffer_t& x) { record('B', {&x}); }
    void on_batch_equal(buffer_t& x, buffer_t& y) { record('E', {&x, &y}); }
    void on_batch_quadratic(buffer_t& x, buffer_t& y, buffer_t& z) { record('Q', {&x, &y, &z}); }
    hip_context& exe;
    FILE* out;
    uint64_t inits = 0;
};

static ligero::hip::scalar scalar_of(uint64_t lo, uint64_t hi) {
    ligero::hip::scalar s{};
    std::memcpy(s.data(), &lo, 8);
    std::memcpy(s.data() + 16, &hi, 8);
    return s;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = std::fopen(argv[1], "wb");
    if (!f) return 3;
    const size_t k = 512, l = 320, n = 2048;
    hip_context executor;
    executor.webgpu_init(k, "");
    executor.ntt_init(l, k, n, 0, 0, 0, 0, 0);
    recording_context ctx(executor, f);
    ligero::hip_vbn254fr<recording_context> v(&ctx);
    v.record(argc > 2);
    using H = ligero::hip_vbn254fr<recording_context>::handle_t;

    const H a = v.vbn254fr_alloc(), b = v.vbn254fr_alloc(), c = v.vbn254fr_alloc(), d = v.vbn254fr_alloc(), e = v.vbn254fr_alloc();
    std::printf("handles %u %u %u %u %u size %zu\n", a, b, c, d, e, v.vbn254fr_get_size());
    uint32_t ui[10];
    for (int i = 0; i < 10; i++) ui[i] = 3 + 2 * i;
    v.vbn254fr_set_ui(a, ui, 10);                                  // I
    v.vbn254fr_set_ui_scalar(b, 9);                                // I
    v.vbn254fr_mulmod(c, a, b);                                    // Q (a, b, a*b)
    v.vbn254fr_divmod(d, c, b);                                    // Q (c/b, b, c)
    v.vbn254fr_assert_equal(d, a);                                 // E (d, a) -- differ in the padding slots only
    const ligero::hip::scalar K = scalar_of(0x123456789abcdef0ull, 0x0fedcba987654321ull);
    v.vbn254fr_addmod(e, a, b);
    v.vbn254fr_addmod_constant(e, e, K);
    v.vbn254fr_submod(e, e, a);
    v.vbn254fr_submod_constant(e, e, scalar_of(77, 0));
    v.vbn254fr_constant_submod(e, K, e);
    v.vbn254fr_mulmod_constant(e, e, K);
    v.vbn254fr_mont_mul_constant(e, e, K);
    v.vbn254fr_copy(d, e);                                         // E (d, e)
    v.vbn254fr_copy(e, e);                                         // E (e, e), through the temporary
    v.vbn254fr_mulmod(e, e, e);                                    // Q with x == y, out aliasing both
    std::vector<ligero::hip::scalar> big(3);
    big[0] = scalar_of(0xffffffffffffffffull, 0x1fffffffffffffffull);
    big[1] = scalar_of(1, 0);
    big[2] = scalar_of(0, 0);
    v.vbn254fr_set(c, big);                                        // I
    v.vbn254fr_set_scalar(d, K);                                   // I
    v.vbn254fr_addmod(c, c, d);                                    // no zero denominators: q * y = x must hold to be provable
    v.vbn254fr_divmod(c, d, c);                                    // Q (d/c, c, d), out aliasing y
    v.vbn254fr_free(a);                                            // cleared, goes to the BACK of the FIFO free list
    const H g = v.vbn254fr_alloc();                                // -> the 6th slot, not a's
    std::printf("realloc %u free %zu\n", g, v.free_variables());
    std::vector<H> bits(ligero::hip_vbn254fr<recording_context>::num_bits);
    for (auto& h : bits) h = v.vbn254fr_alloc();
    v.vbn254fr_bit_decompose(bits.data(), d);                      // 254 x B
    v.finalize();
    std::fclose(f);
    if (argc > 2) {                                                // the recorded lig_batch_op program: n_ops, ops, n_bytes, data
        FILE* g = std::fopen(argv[2], "wb");
        if (!g) return 4;
        const uint64_t n_ops = v.recorded_ops().size(), n_bytes = v.recorded_data().size();
        std::fwrite(&n_ops, 8, 1, g);
        std::fwrite(v.recorded_ops().data(), sizeof(lig_batch_op), n_ops, g);
        std::fwrite(&n_bytes, 8, 1, g);
        std::fwrite(v.recorded_data().data(), 1, n_bytes, g);
        std::fclose(g);
    }
    std::printf("inits %llu\n", (unsigned long long)ctx.inits);
    return 0;
}
