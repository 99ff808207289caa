// ref_backend.cpp -- C entry points around the REFERENCE's own GMP-side code, compiled from where it lies under /root/reference
// (oracle/Makefile target `_ref`; output oracle/_ref/libref_backend.so, git-ignored, never shipped as source).
//
// TEST INFRASTRUCTURE ONLY.  Round 5: the image does have GMP (headers in /opt/conda/include, the same
// 6.2.1 as the system's libgmp.so.10), so the reference's field, sampler, limb export, constraint backend and row former run in
// the build container.  Everything in sections 1 and 2 below instantiates reference code and forwards to it -- no algorithm of
// this repository is involved:
//
//   src/bn254.cpp (compiled as a translation unit of this library)      constants, generate_omegas, mulmod / invmod / powmod / ...
//   include/zkp/finite_field_gmp.hpp:66-78                              bn254_gmp::generate_random  (>> 2, - p)
//   include/util/csprng.hpp:28-110                                      mpz_random_engine (AES-256-CTR, 16 KiB refills)
//   include/util/mpz_vector.hpp:108-160                                 export_limbs / import_limbs
//   include/zkp/backend/witness_manager.hpp:200-354,497-507             row forming, pads, masks, finalize order
//   include/zkp/backend/core.hpp:277-857                                ligetron_backend: expression evaluation, bit_decompose, ...
//   include/zkp/backend/lazy_witness.hpp                                commit_notify / quadratic slots
//
// Section 3 is the one part that is NOT reference code: the guest.  The WASM interpreter cannot be built here (wabt is absent;
// include/stack_value.hpp needs <format>, which g++ 11 lacks; nonbatch_context.hpp needs Dawn's webgpu.h), so the constraint
// calls that tests/i32_add.wat makes are REPLAYED BY HAND on the reference's backend: a value stack whose entries have the move
// semantics of include/stack_value.hpp:84-110, the two host functions of include/host_modules/env.hpp:64-77,166-176, the
// conversions of include/zkp/nonbatch_context.hpp:275-299 and exec_inn_add of include/interpreter_impl.hpp:265-298, with the same
// locals in the same order -- the row stream is the destruction order of those temporaries (core.hpp:283-291: the shared_ptr
// deleter commits a witness).  Read-checked, not compiled from the interpreter: the fixture says so.
//
// Build (oracle/Makefile): -std=c++20 -D__EMSCRIPTEN__ (the reference's own no-Boost.Log switch) -I$(REF)/include
// -idirafter /opt/conda/include (gmp.h / gmpxx.h only: system OpenSSL headers keep precedence), linked against the system's
// libgmp.so.10 and a copy of the image's libgmpxx.so.4 placed next to the library (rpath $ORIGIN).  No stand-in headers.
#include <types.hpp>
#include <util/mpz_vector.hpp>
#include <zkp/finite_field_gmp.hpp>
#include <zkp/backend/core.hpp>
#include <util/csprng.hpp>

#include <cstdint>
#include <cstring>
#include <functional>
#include <sstream>
#include <iostream>
#include <variant>
#include <vector>

namespace vm = ligero::vm;
namespace zkp = ligero::vm::zkp;
using field = zkp::bn254_gmp;

namespace {

// one mpz_class <-> 32 bytes, through the reference's own mpz_vector (4 x u64, least significant first)
void put(const mpz_class& v, uint8_t out[32]) {
    vm::mpz_vector vec;
    vec.push_back(v);
    vec.export_limbs(out, 4, sizeof(uint64_t), 4);
}
mpz_class get(const uint8_t in[32]) {
    vm::mpz_vector vec;
    vec.import_limbs(in, 4, sizeof(uint64_t), 4);
    return vec[0];
}

}  // namespace

extern "C" {

// ------------------------------------------------------------------------------------------------------------------
// 1. field, sampler, limbs
// `count` elements of bn254_gmp::generate_random over mpz_random_engine(key, iv = 0), exported with mpz_vector::export_limbs
void ref_field_random(const uint8_t key[32], size_t count, uint8_t* out) {
    unsigned char k[32], iv[16] = {0};
    std::memcpy(k, key, 32);
    vm::mpz_random_engine eng(k, iv);
    vm::mpz_vector vec;
    mpz_class v;
    for (size_t i = 0; i < count; i++) { field::generate_random(v, eng); vec.push_back(v); }
    vec.export_limbs(out, count * 4, sizeof(uint64_t), 4);
}
// the raw draws (before >> 2 and - p), to find out which of them took the subtraction
void ref_engine_raw(const uint8_t key[32], size_t count, uint8_t* out) {
    unsigned char k[32], iv[16] = {0};
    std::memcpy(k, key, 32);
    vm::mpz_random_engine eng(k, iv);
    vm::mpz_vector vec;
    mpz_class v;
    for (size_t i = 0; i < count; i++) { eng(v, 32); vec.push_back(v); }
    vec.export_limbs(out, count * 4, sizeof(uint64_t), 4);
}
void ref_omegas(uint64_t k, uint8_t out[96]) {
    auto [wk, w2k, w4k] = field::generate_omegas(k, 4 * k);
    put(wk, out); put(w2k, out + 32); put(w4k, out + 64);
}
// constants of src/bn254.cpp:22-49 as the reference holds them: modulus, 2x, 4x, middle, root1, root2, montgomery_factor, barrett_factor
void ref_constants(uint8_t out[8 * 32]) {
    const mpz_class* c[8] = {&field::modulus, &field::modulus_2x, &field::modulus_4x, &field::modulus_middle, &field::root1, &field::root2,
                             &field::montgomery_factor, &field::barrett_factor};
    for (int i = 0; i < 8; i++) put(*c[i], out + 32 * i);
}
// op: 0 mulmod, 1 invmod(a), 2 powmod(a, b), 3 divmod, 4 mont_mulmod, 5 addmod, 6 submod, 7 negate(a), 8 reduce(a), 9 powmod_ui(a, low 32 bits of b),
// 10 reduce_u256(a)
int ref_field_op(int op, const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) {
    const mpz_class x = get(a), y = get(b);
    mpz_class z;
    switch (op) {
    case 0: field::mulmod(z, x, y); break;
    case 1: field::invmod(z, x); break;
    case 2: field::powmod(z, x, y); break;
    case 3: field::divmod(z, x, y); break;
    case 4: field::mont_mulmod(z, x, y); break;
    case 5: field::addmod(z, x, y); break;
    case 6: field::submod(z, x, y); break;
    case 7: field::negate(z, x); break;
    case 8: field::reduce(z, x); break;
    case 9: field::powmod_ui(z, x, (uint32_t)mpz_get_ui(y.get_mpz_t())); break;
    case 10: field::reduce_u256(z, x); break;
    default: return -1;
    }
    put(z, out);
    return 0;
}
// import_limbs(in: count integers of limb_count limbs of limb_size bytes) -> export_limbs(out, out_limb_size, out_limb_count); returns the words exported
size_t ref_limbs_roundtrip(const void* in, size_t count, size_t limb_size, size_t limb_count, void* out, size_t out_limb_size, size_t out_limb_count) {
    vm::mpz_vector vec;
    vec.import_limbs(in, count * limb_count, limb_size, limb_count);
    return vec.export_limbs(out, count * out_limb_count, out_limb_size, out_limb_count);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// 2. recording the row stream of the reference's witness_manager
namespace {

struct stage1_policy { static constexpr bool pad_encoding_random = true, enable_code_check = false, enable_linear_check = false, enable_quadratic_check = false; };   // nonbatch_context.hpp:39-44
struct stage2_policy { static constexpr bool pad_encoding_random = true, enable_code_check = true, enable_linear_check = true, enable_quadratic_check = true; };      // :46-51
struct verifier_policy { static constexpr bool pad_encoding_random = false, enable_code_check = true, enable_linear_check = true, enable_quadratic_check = true; };   // :60-65

struct Recording {
    size_t k = 0;
    std::vector<uint8_t> kinds;            // 0 linear, 1 / 2 / 3 = x / y / z of a quadratic triple, in callback order
    std::vector<uint8_t> vals, rands;      // rows x k x 32 bytes (rands: zero rows under the stage-1 policy)
    std::vector<uint8_t> masks;            // code (k) | linear (2k) | quadratic (2k) elements
    uint8_t constsum[32] = {0};
    void row(uint8_t kind, vm::mpz_vector& v, vm::mpz_vector& r) {
        kinds.push_back(kind);
        const size_t at = vals.size();
        vals.resize(at + k * 32); rands.resize(at + k * 32);
        v.export_limbs(vals.data() + at, k * 4, sizeof(uint64_t), 4);                 // the call of nonbatch_context.hpp:450 (write_limbs of the row)
        if (r.size()) r.export_limbs(rands.data() + at, r.size() * 4, sizeof(uint64_t), 4);
    }
};

template <class Policy>
struct Guest {
    zkp::ligetron_backend<field, Policy> backend;
    Recording rec;
    // where the rows go: recorded (default), or handed to a consumer (section 4: the HIP row-batching shim)
    using row_t = std::pair<vm::mpz_vector&, vm::mpz_vector&>;
    std::function<void(row_t)> on_linear;
    std::function<void(row_t, row_t, row_t)> on_quadratic;
    std::function<void(vm::mpz_vector&, vm::mpz_vector&, vm::mpz_vector&)> on_masks;
    Guest(size_t l, size_t k, const uint8_t enc_key[32], const uint8_t wit_key[32]) : backend(l, k) {
        rec.k = k;
        unsigned char key[32], iv[16] = {0};                                              // params::any_iv (include/params.hpp:42)
        std::memcpy(key, enc_key, 32);
        backend.manager().encoding_random_engine().init(key, iv);                          // nonbatch_context.hpp:99-103
        if (wit_key) {
            std::memcpy(key, wit_key, 32);
            backend.manager().code_random_engine().init(key, iv);                          // :105-112: one seed, three engines
            backend.manager().linear_random_engine().init(key, iv);
            backend.manager().quadratic_random_engine().init(key, iv);
        }
        backend.manager()
            .register_linear_callback([this](auto row) { if (on_linear) on_linear(row); else rec.row(0, row.first, row.second); })
            .register_quadratic_callback([this](auto x, auto y, auto z) {
                if (on_quadratic) { on_quadratic(x, y, z); return; }
                rec.row(1, x.first, x.second); rec.row(2, y.first, y.second); rec.row(3, z.first, z.second);
            })
            .register_mask_callback([this](vm::mpz_vector& c, vm::mpz_vector& lin, vm::mpz_vector& q) {
                if (on_masks) { on_masks(c, lin, q); return; }
                const size_t k = rec.k;
                rec.masks.resize(5 * k * 32);
                c.export_limbs(rec.masks.data(), k * 4, sizeof(uint64_t), 4);
                lin.export_limbs(rec.masks.data() + k * 32, 2 * k * 4, sizeof(uint64_t), 4);
                q.export_limbs(rec.masks.data() + 3 * k * 32, 2 * k * 4, sizeof(uint64_t), 4);
            });
    }
    void finish() {
        std::streambuf* old = std::cout.rdbuf(nullptr);        // witness_manager::finalize prints its counters
        backend.finalize();
        std::cout.rdbuf(old);
        std::cout.clear();
        put(backend.manager().constsum(), rec.constsum);
    }
};

// ------------------------------------------------------------------------------------------------------------------
// 3. the guests (hand replay of the interpreter's glue, see the header of this file)

// a stack entry: the alternatives a guest of this file can produce, with the copy / move behaviour of stack_value (stack_value.hpp:84-110:
// copy deleted, move defaulted -- and zkp::decomposed_bits declares a destructor, so its "move" is a copy that shares the witnesses)
struct Value {
    std::variant<uint32_t, zkp::managed_witness, zkp::decomposed_bits> data;
    Value(uint32_t v) : data(v) {}
    Value(zkp::managed_witness w) : data(std::move(w)) {}
    Value(zkp::decomposed_bits b) : data(std::move(b)) {}
    Value(const Value&) = delete;
    Value(Value&&) = default;
    Value& operator=(const Value&) = delete;
    Value& operator=(Value&&) = default;
};

template <class Policy>
struct Machine : Guest<Policy> {
    using Guest<Policy>::Guest;
    using Guest<Policy>::backend;
    std::vector<Value> stack;

    void push(Value v) { stack.push_back(std::move(v)); }                                  // nonbatch_context.hpp:114-116
    Value pop() { Value top = std::move(stack.back()); stack.pop_back(); return top; }     // :118-123
    zkp::managed_witness make_witness(Value s) {                                           // :275-299
        if (auto* v = std::get_if<uint32_t>(&s.data)) { auto x = backend.acquire_witness(); x.val(*v); return x; }
        if (auto* w = std::get_if<zkp::managed_witness>(&s.data)) return std::move(*w);
        return backend.bit_compose(std::get<zkp::decomposed_bits>(s.data));
    }
    // host_modules/env.hpp:166-176
    void i32_private_const() {
        uint32_t v = std::get<uint32_t>(pop().data);
        auto x = backend.acquire_witness();
        x.val(v);
        auto range_checked_x = backend.bit_decompose(x, 32);
        push(std::move(range_checked_x));
    }
    // host_modules/env.hpp:64-77
    void assert_equal() {
        auto sy = pop();
        auto sx = pop();
        auto wx = make_witness(std::move(sx));
        auto wy = make_witness(std::move(sy));
        if (wx.val() != wy.val()) throw std::runtime_error("guest assertion failed");
        backend.assert_equal(wx, wy);
    }
    // interpreter_impl.hpp:265-298, i32, both operands witnesses
    void i32_add() {
        auto sy = pop();
        auto sx = pop();
        const size_t num_bits = 32, num_overflowed_bits = num_bits + 1;
        auto x = make_witness(std::move(sx));
        auto y = make_witness(std::move(sy));
        auto overflowed = backend.eval(x + y);
        auto bits = backend.bit_decompose(overflowed, num_overflowed_bits);
        bits.drop_msb(1);
        push(std::move(bits));
    }
    // one line of tests/i32_add.wat:6-13: (call $assert_equal (i32.add (pc a) (pc b)) (pc c)), operands evaluated left to right
    void line(uint32_t a, uint32_t b, uint32_t c) {
        push(a); i32_private_const();
        push(b); i32_private_const();
        i32_add();
        push(c); i32_private_const();
        assert_equal();
    }
};

// guest 0: tests/i32_add.wat:6-13
template <class Policy>
void guest_i32_add(Machine<Policy>& m) {
    m.line(1, 1, 2);
    m.line(1, 0, 1);
    m.line(0xffffffffu, 0xffffffffu, 0xfffffffeu);
    m.line(0xffffffffu, 1, 0);
    m.line(0x7fffffffu, 1, 0x80000000u);
    m.line(0x80000000u, 0xffffffffu, 0x7fffffffu);
    m.line(0x80000000u, 0x80000000u, 0);
    m.line(0x3fffffffu, 1, 0x40000000u);
}
// guest 1: a stream that interleaves full linear rows and full quadratic triples (`reps` times: w = a * b + a - 3, checked against a constant;
// values from a small LCG).  Exercises eval() on nested expressions with constants and the emission order of full rows (a full row is flushed
// by the NEXT commit of its kind, witness_manager.hpp:137-139,154-156).
template <class Policy>
void guest_mul_add(Machine<Policy>& m, size_t reps) {
    uint64_t s = 0x9e3779b97f4a7c15ull;
    for (size_t i = 0; i < reps; i++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const uint32_t av = (uint32_t)(s >> 33), bv = (uint32_t)(s >> 13) & 0xffffu;
        auto a = m.backend.acquire_witness(); a.val(av);
        auto b = m.backend.acquire_witness(); b.val(bv);
        auto w = m.backend.eval(a * b + a - 3u);
        const mpz_class expect = mpz_class(av) * bv + av - 3;
        if (w.val() != expect) throw std::runtime_error("guest_mul_add: unexpected value");
        m.backend.assert_const(w, expect);
    }
}

template <class Policy>
Recording* run_guest(int which, size_t l, size_t k, const uint8_t enc_key[32], const uint8_t wit_key[32], size_t reps) {
    Machine<Policy> m(l, k, enc_key, wit_key);
    if (which == 0) guest_i32_add(m);
    else if (which == 1) guest_mul_add(m, reps);
    else return nullptr;
    if (!m.stack.empty()) throw std::runtime_error("guest left values on the stack");
    m.finish();
    return new Recording(std::move(m.rec));
}

}  // namespace

extern "C" {

// stage2 == 0: the stage-1 policy (rows, pads, masks).  stage2 == 1: the stage-2 policy with the three witness engines keyed by `wit_key`
// (= the stage-1 seed): the same rows plus every row's randomness row and the constant sum.  stage2 == 2: the VERIFIER's policy
// (nonbatch_context.hpp:60-65: no pads drawn, all checks on) -- the public data the verifier derives by running the guest itself.
// Returns NULL on a guest error.
void* ref_guest_run(int which, uint64_t l, uint64_t k, const uint8_t enc_key[32], const uint8_t* wit_key, int stage2, uint64_t reps) {
    try {
        if (stage2 == 2) return run_guest<verifier_policy>(which, l, k, enc_key, wit_key, reps);
        return stage2 ? run_guest<stage2_policy>(which, l, k, enc_key, wit_key, reps) : run_guest<stage1_policy>(which, l, k, enc_key, nullptr, reps);
    } catch (const std::exception& e) {
        std::cerr << "ref_guest_run: " << e.what() << std::endl;
        return nullptr;
    }
}
size_t ref_guest_rows(const void* h) { return static_cast<const Recording*>(h)->kinds.size(); }
// kinds: rows bytes; vals / rands: rows x k x 32; masks: 5k x 32; constsum: 32
void ref_guest_read(const void* h, uint8_t* kinds, uint8_t* vals, uint8_t* rands, uint8_t* masks, uint8_t constsum[32]) {
    const Recording* r = static_cast<const Recording*>(h);
    if (kinds) std::memcpy(kinds, r->kinds.data(), r->kinds.size());
    if (vals) std::memcpy(vals, r->vals.data(), r->vals.size());
    if (rands) std::memcpy(rands, r->rands.data(), r->rands.size());
    if (masks) std::memcpy(masks, r->masks.data(), r->masks.size());
    if (constsum) std::memcpy(constsum, r->constsum, 32);
}
void ref_guest_free(void* h) { delete static_cast<Recording*>(h); }

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// 4. (libref_hip_guest.so, -DREF_WITH_HIP_SHIM) the reference's constraint backend IN FRONT OF THE HIP BACKEND, through the shim a maintainer
// would add (include/lig_hip_row_batcher.hpp): witness_manager's callbacks export their rows' limbs straight into the
// batcher's page-locked slots (mpz_vector::export_limbs, as nonbatch_context.hpp:447 does into its own vector), two runs of the guest
// (stage-1 policy -> commit -> stage-2 policy keyed by the seed of the commitment -> prove), constsum() as the public constant.
// Runs on the GPU box (the library travels prebuilt); tests/test_gpu_ref_backend.py compares the envelope with the oracle's over the same stream.
#ifdef REF_WITH_HIP_SHIM
#include "../include/lig_hip_row_batcher.hpp"

namespace {
template <class Policy>
void feed_batcher(Machine<Policy>& m, ligero::hip_row_batcher& b, size_t k, bool with_rands) {
    using row_t = typename Guest<Policy>::row_t;
    auto slot_row = [&b, k, with_rands](uint8_t kind, row_t row) {
        uint64_t* slot = b.next_slot();                 // pass 1: the next message row; pass 2: the next row's randomness row (packed)
        vm::mpz_vector& src = with_rands ? row.second : row.first;
        const bool has = !with_rands || src.size() != 0;
        if (has && slot) src.export_limbs(slot, k * 4, sizeof(uint64_t), 4);
        b.commit_slot(kind, has);
    };
    m.on_linear = [slot_row](row_t row) { slot_row(LIG_ROW_LINEAR, row); };
    m.on_quadratic = [slot_row](row_t x, row_t y, row_t z) { slot_row(LIG_ROW_QX, x); slot_row(LIG_ROW_QY, y); slot_row(LIG_ROW_QZ, z); };
    m.on_masks = [&b](vm::mpz_vector& c, vm::mpz_vector& lin, vm::mpz_vector& q) { b.mask_callback(c.size(), lin.size(), q.size()); };
}
template <class Policy>
void run_into(Machine<Policy>& m, int which, size_t reps) {
    if (which == 0) guest_i32_add(m);
    else if (which == 1) guest_mul_add(m, reps);
    else throw std::invalid_argument("unknown guest");
    m.finish();
}
}  // namespace

extern "C" int ref_guest_prove_hip(int which, uint64_t l, uint64_t k, const uint8_t enc_key[32], int64_t generated_at, uint64_t reps,
                                   uint8_t root[32], uint8_t seed1[32], uint8_t constsum[32], uint8_t* proof_out, size_t cap, size_t* proof_len,
                                   uint64_t* rows_out, int valid[3], char* err, size_t errcap) {
    auto fail = [&](const std::string& why) { if (err && errcap) { std::strncpy(err, why.c_str(), errcap - 1); err[errcap - 1] = 0; } return 1; };
    lig_ctx* ctx = nullptr;
    if (lig_ctx_create(&ctx, 0, (uint32_t)l, (uint32_t)k, (uint32_t)(4 * k)) != LIG_OK) { const std::string w = ctx ? lig_last_error(ctx) : "lig_ctx_create"; if (ctx) lig_ctx_destroy(ctx); return fail(w); }
    int rc = 0;
    try {
        ligero::hip_proof_meta meta;
        std::memcpy(meta.encoding_seed, enc_key, 32);
        meta.generated_at = generated_at;
        ligero::hip_row_batcher b(ctx, meta);
        {   // run 1 of the guest: stage-1 policy (rows with their pads; the masks are formed by the library from the same stream)
            Machine<stage1_policy> m(l, k, enc_key, nullptr);
            feed_batcher(m, b, k, false);
            run_into(m, which, reps);
        }
        b.commit(root, seed1);
        uint8_t cs[32];
        {   // run 2: stage-2 policy, the three witness engines keyed by the seed the HIP backend derived from its commitment
            Machine<stage2_policy> m(l, k, enc_key, seed1);
            feed_batcher(m, b, k, true);
            run_into(m, which, reps);
            std::memcpy(cs, m.rec.constsum, 32);
        }
        std::memcpy(constsum, cs, 32);
        size_t len = 0;
        lig_proof_info info;
        const uint8_t* proof = b.prove(cs, &len, &info);
        *proof_len = len;
        *rows_out = b.rows();
        valid[0] = info.valid_code; valid[1] = info.valid_linear; valid[2] = info.valid_quad;
        if (len > cap) rc = fail("proof buffer too small");
        else std::memcpy(proof_out, proof, len);
    } catch (const std::exception& e) {
        rc = fail(e.what());
    }
    lig_ctx_destroy(ctx);
    return rc;
}
#endif
