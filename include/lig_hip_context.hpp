// lig_hip_context.hpp -- C++ mirror of the reference's executor `ligero::webgpu_context`
// (include/wgpu.hpp:50-183, include/ligetron/webgpu/{device_context,buffer_view,buffer_binding,device_bignum}.hpp)
// on top of the C ABI of lig_hip.h.  Same member names, argument meaning and ordering, so that the stage
// drivers of include/zkp/nonbatch_context.hpp and the vbn254fr host module -- which take the executor as a
// template parameter (`using executor_t = webgpu_context`, src/webgpu_prover.cpp:54) -- compile against it:
//
//     using executor_t = ligero::hip_context;          // instead of ligero::webgpu_context
//
// Differences that a caller can observe are listed in INTEGRATION.md (no shader path, sha256_context is 64
// bytes, bindings are plain structs of buffer views instead of WGPUBindGroup handles).
// Scalars: the reference passes `const mpz_class&`; when <gmpxx.h> is available the same overloads exist here,
// otherwise scalars are 32-byte little-endian arrays (lig::scalar).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "lig_hip.h"
#if __has_include(<gmpxx.h>)
#include <gmpxx.h>
#define LIG_HAVE_GMP 1
#endif

namespace ligero {

namespace hip {

using scalar = std::array<uint8_t, 32>;   // canonical field element, little endian

inline void check(lig_ctx* c, int rc, const char* what) {
    // the reference aborts on device errors (src/webgpu/device_context.cpp:121-127); here they surface as exceptions
    if (rc != LIG_OK) throw std::runtime_error(std::string(what) + ": " + lig_last_error(c));
}

// buffer_view (include/ligetron/webgpu/buffer_view.hpp:27-69): ref-counted view of a device allocation
class buffer_view {
public:
    buffer_view() = default;
    buffer_view(lig_ctx* c, size_t bytes, bool upstream_slices = false) : size_(bytes), storage_size_(bytes), upstream_(upstream_slices) {
        void* p = nullptr;
        check(c, lig_malloc(c, bytes, &p), "make_device_buffer");
        base_ = std::shared_ptr<void>(p, [c](void* q) { lig_free(c, q); });
    }
    size_t size() const { return size_; }
    size_t offset() const { return offset_; }
    void* data() const { return static_cast<char*>(base_.get()) + offset_; }
    // slice by bytes / by element count of T (buffer_view.hpp:52-60).  Two semantics:
    //   declared (default)  slice_bytes(from, n_bytes) as buffer_view.hpp:52 declares it: {offset + from, n_bytes};
    //   upstream            as src/webgpu/buffer_view.cpp:91-95 DEFINES it -- the parameter names are swapped there, so the call
    //                       slice_bytes(A, B) returns buffer_view(storage, /*offset*/ A, /*size*/ offset + B): correct for a view
    //                       of a whole buffer (offset 0), an absolute offset into the storage for a nested slice.
    buffer_view slice_bytes(size_t begin, size_t len) const {
        if (begin + len > size_) throw std::out_of_range("buffer_view::slice");      // upstream: assert(from + n_bytes <= size_bytes_), same sum
        buffer_view v(*this);
        if (upstream_) {
            v.offset_ = begin; v.size_ = offset_ + len;
            if (v.offset_ + v.size_ > storage_size_) throw std::out_of_range("buffer_view::slice (upstream semantics): beyond the storage");
        } else { v.offset_ += begin; v.size_ = len; }
        return v;
    }
    bool upstream_slices() const { return upstream_; }
    buffer_view slice(size_t begin_bytes) const { return slice_bytes(begin_bytes, size_ - begin_bytes); }
    buffer_view slice(size_t begin_bytes, size_t len_bytes) const { return slice_bytes(begin_bytes, len_bytes); }
    template <typename T> buffer_view slice_n(size_t begin, size_t n) const { return slice_bytes(begin * sizeof(T), n * sizeof(T)); }
    bool operator==(const buffer_view& o) const { return base_ == o.base_ && offset_ == o.offset_ && size_ == o.size_; }

private:
    std::shared_ptr<void> base_;
    size_t offset_ = 0, size_ = 0, storage_size_ = 0;
    bool upstream_ = false;
};

// device_bignum (include/ligetron/webgpu/device_bignum.hpp:30-90): 8 x u32 little-endian limbs
struct device_bignum {
    static constexpr size_t num_limbs = 8, num_bytes = 32, num_bits = 256;
    uint32_t limbs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    device_bignum() = default;
    device_bignum(uint32_t v) { limbs[0] = v; }
    explicit device_bignum(const scalar& s) { std::memcpy(limbs, s.data(), 32); }
#ifdef LIG_HAVE_GMP
    device_bignum(const mpz_class& v) { size_t cnt = 0; mpz_export(limbs, &cnt, -1, 4, 0, 0, v.get_mpz_t()); }
    mpz_class to_mpz() const { mpz_class r; mpz_import(r.get_mpz_t(), 8, -1, 4, 0, 0, limbs); return r; }
#endif
    uint32_t operator[](size_t i) const { return limbs[i]; }
    scalar to_scalar() const { scalar s; std::memcpy(s.data(), limbs, 32); return s; }
};

// buffer_binding (include/ligetron/webgpu/buffer_binding.hpp:31-48): the buffers an op works on, captured once
struct buffer_binding {
    std::vector<buffer_view> bufs;
    const std::vector<buffer_view>& buffers() const { return bufs; }
};
struct eltwise_offset { size_t x = 0, y = 0, z = 0; };   // element offsets (buffer_binding.hpp:27-29)

}  // namespace hip

class hip_context {
public:
    using buffer_type = hip::buffer_view;
    using device_bignum_type = hip::device_bignum;
    struct sha256_context { uint32_t words[16]; };       // only sizeof() is used (nonbatch_context.hpp:424)

    hip_context() = default;
    hip_context(const hip_context&) = delete;
    hip_context& operator=(const hip_context&) = delete;
    ~hip_context() { if (ctx_) { lig_sync(ctx_); lig_ctx_destroy(ctx_); } }

    // ---- lifecycle (wgpu.hpp:71-82).  shader_path is accepted and ignored: kernels are compiled into the library.
    void webgpu_init(size_t gpu_threads = 0, const std::string& shader_path = "") { (void)gpu_threads; (void)shader_path; }
    // p, barrett factor and the roots are fixed BN254 constants inside the library (the reference's WGSL hard-codes
    // p / mu / J as well, shader/bn254fr.wgsl.in:19-45); the arguments are accepted for source compatibility.
    template <typename... Ignored>
    void ntt_init(size_t l, size_t k, size_t n, const Ignored&...) {
        if (ctx_) { lig_ctx_destroy(ctx_); ctx_ = nullptr; }
        const int rc = lig_ctx_create(&ctx_, device_, (uint32_t)l, (uint32_t)k, (uint32_t)n);
        if (rc != LIG_OK) {
            std::string msg = ctx_ ? lig_last_error(ctx_) : "invalid (l, k, n)";
            if (ctx_) { lig_ctx_destroy(ctx_); ctx_ = nullptr; }
            throw std::runtime_error("ntt_init: " + msg);
        }
    }
    void set_device(int d) { device_ = d; }
    // Buffers made from now on slice as upstream's buffer_view::slice_bytes is DEFINED (parameters swapped against its
    // declaration, src/webgpu/buffer_view.cpp:91-95 vs include/ligetron/webgpu/buffer_view.hpp:52) instead of as declared.
    // Needed, and only needed, to produce byte-identical proofs of vbn254fr programs with an UNMODIFIED v1.5.0 build: every
    // other caller slices views of whole buffers, where the two agree.  INTEGRATION.md section 3 lists what it changes.
    void set_upstream_slice_compat(bool on) { upstream_slices_ = on; }
    bool upstream_slice_compat() const { return upstream_slices_; }
    void device_synchronize() { hip::check(ctx_, lig_sync(ctx_), "device_synchronize"); }
    size_t message_size() const { return lig_message_size(ctx_); }
    size_t padding_size() const { return lig_padding_size(ctx_); }
    size_t encoding_size() const { return lig_encoding_size(ctx_); }
    lig_ctx* native() const { return ctx_; }

    // ---- buffers (wgpu.hpp:159-169, device_context.hpp:79-98)
    buffer_type make_device_buffer(size_t bytes) { return buffer_type(ctx_, bytes, upstream_slices_); }
    buffer_type make_codeword_buffer() { return make_device_buffer(encoding_size() * 32); }
    buffer_type make_message_buffer() { return make_device_buffer(message_size() * 32); }
    buffer_type make_sample_buffer() { return make_device_buffer(192 * 32); }
    template <typename T> void write_buffer(buffer_type buf, const T* data, size_t len) {
        hip::check(ctx_, lig_write(ctx_, buf.data(), data, len * sizeof(T)), "write_buffer");
    }
    // device_context.hpp:95-98: write_buffer(buf, data, len); clear_buffer(buf.slice(len * sizeof(T))) -- through slice(), so that
    // the upstream slicing semantics (set_upstream_slice_compat) reach it
    template <typename T> void write_buffer_clear(buffer_type buf, const T* data, size_t len) {
        if (!buf.upstream_slices()) { hip::check(ctx_, lig_write_clear(ctx_, buf.data(), buf.size(), data, len * sizeof(T)), "write_buffer_clear"); return; }
        write_buffer(buf, data, len);
        clear_buffer(buf.slice(len * sizeof(T)));
    }
    // write_limbs (include/wgpu.hpp:171-183): `size` copies of one element / a vector of elements, as device_bignum limbs
    void write_limbs(buffer_type buf, const hip::scalar& val, size_t size) {
        std::vector<hip::device_bignum> host_buf(size, hip::device_bignum(val));
        write_buffer(buf, host_buf.data(), host_buf.size());
    }
    void write_limbs(buffer_type buf, const std::vector<hip::scalar>& vals) {
        std::vector<hip::device_bignum> host_buf;
        host_buf.reserve(vals.size());
        for (const auto& v : vals) host_buf.emplace_back(v);
        write_buffer(buf, host_buf.data(), host_buf.size());
    }
    void clear_buffer(buffer_type buf) { hip::check(ctx_, lig_clear(ctx_, buf.data(), buf.size()), "clear_buffer"); }
    void copy_buffer_to_buffer(buffer_type from, buffer_type to) {
        hip::check(ctx_, lig_copy(ctx_, to.data(), from.data(), from.size() < to.size() ? from.size() : to.size()), "copy_buffer_to_buffer");
    }
    void copy_buffer_clear(buffer_type from, buffer_type to) {      // copy, then zero the rest of `to`
        copy_buffer_to_buffer(from, to);
        if (to.size() > from.size())
            hip::check(ctx_, lig_clear(ctx_, static_cast<char*>(to.data()) + from.size(), to.size() - from.size()), "copy_buffer_clear");
    }
    template <typename T> std::vector<T> copy_to_host(buffer_type buf) {   // blocking
        std::vector<T> out(buf.size() / sizeof(T));
        hip::check(ctx_, lig_read(ctx_, out.data(), buf.data(), out.size() * sizeof(T)), "copy_to_host");
        return out;
    }

    // ---- binding factories (wgpu.hpp:87-96)
    hip::buffer_binding bind_ntt(buffer_type buf) { return {{buf}}; }
    hip::buffer_binding bind_eltwise2(buffer_type x, buffer_type out) { return {{x, out}}; }
    hip::buffer_binding bind_eltwise3(buffer_type x, buffer_type y, buffer_type out) { return {{x, y, out}}; }
    hip::buffer_binding bind_sha256_context(buffer_type ctx, buffer_type digest) { return {{ctx, digest}}; }
    hip::buffer_binding bind_sha256_buffer(buffer_type in) { return {{in}}; }
    hip::buffer_binding bind_sampling(buffer_type from, buffer_type to) { return {{from, to}}; }

    hip::buffer_binding bind_scalar(buffer_type s) { return {{s}}; }
    hip::buffer_binding bind_powmod(buffer_type exp, buffer_type coeff, buffer_type out) { return {{exp, coeff, out}}; }

    // ---- powmod (wgpu.hpp:84-85,104-107; src/webgpu/engine.cpp:214-223,674-680; powmod_context.cpp:178-268).  Exponents
    // are one u32 per element; the per-base table of squarings is rebuilt by set_base inside the library.
    void powmod_init(size_t num_exponent_bits = 32) {
        if (num_exponent_bits == 0 || num_exponent_bits > 32) throw std::invalid_argument("powmod_init: 1..32 exponent bits");
        powmod_bits_ = num_exponent_bits;
    }
    void powmod_set_base(const hip::scalar& base) { need_powmod(); powmod_base_ = base; powmod_has_base_ = true; }
    void EltwisePowMod(const hip::buffer_binding& b) { pow(b, 0); }        // out = coeff * base^exp
    void EltwisePowAddMod(const hip::buffer_binding& b) { pow(b, 1); }     // out += coeff * base^exp

    // ---- transforms (wgpu.hpp:98-110)
    void encode_ntt_device(const hip::buffer_binding& b) { hip::check(ctx_, lig_encode(ctx_, b.bufs[0].data()), "encode_ntt_device"); }
    void decode_ntt_device(const hip::buffer_binding& b) { hip::check(ctx_, lig_decode(ctx_, b.bufs[0].data()), "decode_ntt_device"); }
    void ntt_forward_k(const hip::buffer_binding& b) { ntt(b, LIG_SIZE_K, 0); }
    void ntt_forward_2k(const hip::buffer_binding& b) { ntt(b, LIG_SIZE_2K, 0); }
    void ntt_forward_n(const hip::buffer_binding& b) { ntt(b, LIG_SIZE_N, 0); }
    void ntt_inverse_k(const hip::buffer_binding& b) { ntt(b, LIG_SIZE_K, 1); }
    void ntt_inverse_2k(const hip::buffer_binding& b) { ntt(b, LIG_SIZE_2K, 1); }
    void ntt_inverse_n(const hip::buffer_binding& b) { ntt(b, LIG_SIZE_N, 1); }

    // ---- eltwise (wgpu.hpp:112-139).  bind_eltwise3 = (x, y, out), bind_eltwise2 = (x, out)
    void EltwiseAddMod(const hip::buffer_binding& b, hip::eltwise_offset o = {}) { el3(LIG_OP_ADD, b, o); }
    void EltwiseSubMod(const hip::buffer_binding& b, hip::eltwise_offset o = {}) { el3(LIG_OP_SUB, b, o); }
    void EltwiseMultMod(const hip::buffer_binding& b, hip::eltwise_offset o = {}) { el3(LIG_OP_MUL, b, o); }
    void EltwiseDivMod(const hip::buffer_binding& b, hip::eltwise_offset o = {}) { el3(LIG_OP_DIV, b, o); }
    void EltwiseFMAMod(const hip::buffer_binding& b, hip::eltwise_offset o = {}) { el3(LIG_OP_FMA, b, o); }
    void EltwiseAddAssignMod(const hip::buffer_binding& b, hip::eltwise_offset o = {}) { el2(LIG_OP_ADD_ASSIGN, b, o, nullptr); }
    void EltwiseAddMod(const hip::buffer_binding& b, const hip::scalar& c, hip::eltwise_offset o = {}) { el2(LIG_OP_ADD_CONST, b, o, &c); }
    void EltwiseSubConstMod(const hip::buffer_binding& b, const hip::scalar& c, hip::eltwise_offset o = {}) { el2(LIG_OP_SUB_CONST, b, o, &c); }
    void EltwiseConstSubMod(const hip::buffer_binding& b, const hip::scalar& c, hip::eltwise_offset o = {}) { el2(LIG_OP_CONST_SUB, b, o, &c); }
    void EltwiseMultMod(const hip::buffer_binding& b, const hip::scalar& c, hip::eltwise_offset o = {}) { el2(LIG_OP_MUL_CONST, b, o, &c); }
    void EltwiseMontMultMod(const hip::buffer_binding& b, const hip::scalar& c, hip::eltwise_offset o = {}) { el2(LIG_OP_MONTMUL_CONST, b, o, &c); }
    void EltwiseFMAMod(const hip::buffer_binding& b, const hip::scalar& c, hip::eltwise_offset o = {}) { el2(LIG_OP_FMA_CONST, b, o, &c); }
    void EltwiseBitDecompose(const hip::buffer_binding& b, uint32_t bit, hip::eltwise_offset o = {}) {
        const auto& x = b.bufs[0]; const auto& out = b.bufs[1];
        hip::check(ctx_, lig_eltwise(ctx_, LIG_OP_BIT_DECOMPOSE, at(x, o.x), nullptr, at(out, o.z), count(out, o.z), nullptr, bit), "EltwiseBitDecompose");
    }
#ifdef LIG_HAVE_GMP
    static hip::scalar to_scalar(const mpz_class& v) { return hip::device_bignum(v).to_scalar(); }
    void write_limbs(buffer_type buf, const mpz_class& val, size_t size) { write_limbs(buf, to_scalar(val), size); }
    void write_limbs(buffer_type buf, const std::vector<mpz_class>& vals) {
        std::vector<hip::scalar> s;
        for (const auto& v : vals) s.push_back(to_scalar(v));
        write_limbs(buf, s);
    }
    void powmod_set_base(const mpz_class& base, const mpz_class& /*p: fixed BN254 modulus*/) { powmod_set_base(to_scalar(base)); }
    void EltwiseAddMod(const hip::buffer_binding& b, const mpz_class& c, hip::eltwise_offset o = {}) { EltwiseAddMod(b, to_scalar(c), o); }
    void EltwiseSubConstMod(const hip::buffer_binding& b, const mpz_class& c, hip::eltwise_offset o = {}) { EltwiseSubConstMod(b, to_scalar(c), o); }
    void EltwiseConstSubMod(const hip::buffer_binding& b, const mpz_class& c, hip::eltwise_offset o = {}) { EltwiseConstSubMod(b, to_scalar(c), o); }
    void EltwiseMultMod(const hip::buffer_binding& b, const mpz_class& c, hip::eltwise_offset o = {}) { EltwiseMultMod(b, to_scalar(c), o); }
    void EltwiseMontMultMod(const hip::buffer_binding& b, const mpz_class& c, hip::eltwise_offset o = {}) { EltwiseMontMultMod(b, to_scalar(c), o); }
    void EltwiseFMAMod(const hip::buffer_binding& b, const mpz_class& c, hip::eltwise_offset o = {}) { EltwiseFMAMod(b, to_scalar(c), o); }
#endif

    // ---- column hash (wgpu.hpp:141-146)
    void sha256_init(size_t instances) { sha_instances_ = instances; }
    void sha256_digest_init(const hip::buffer_binding& b) { hip::check(ctx_, lig_sha_init(ctx_, b.bufs[0].data(), sha_instances_), "sha256_digest_init"); }
    void sha256_digest_update(const hip::buffer_binding& ctxb, const hip::buffer_binding& in) {
        hip::check(ctx_, lig_sha_update(ctx_, ctxb.bufs[0].data(), in.bufs[0].data()), "sha256_digest_update");
    }
    void sha256_digest_final(const hip::buffer_binding& b) { hip::check(ctx_, lig_sha_final(ctx_, b.bufs[0].data(), b.bufs[1].data()), "sha256_digest_final"); }

    // ---- sampling (wgpu.hpp:148-151)
    void sampling_init(const std::vector<size_t>& idx) {
        std::vector<uint32_t> v(idx.begin(), idx.end());
        hip::check(ctx_, lig_sample_init(ctx_, v.data(), v.size()), "sampling_init");
    }
    void sample_gather(const hip::buffer_binding& b, size_t slot) {
        hip::check(ctx_, lig_sample_gather(ctx_, b.bufs[0].data(), b.bufs[1].data(), slot), "sample_gather");
    }

private:
    void need_powmod() const {       // the reference throws std::logic_error when powmod is used before powmod_init (engine.cpp:1505-1512)
        if (!powmod_bits_) throw std::logic_error("powmod context is not initialised: call powmod_init first");
    }
    void pow(const hip::buffer_binding& b, int add) {
        need_powmod();
        if (!powmod_has_base_) throw std::logic_error("powmod: call powmod_set_base first");
        const auto& exp = b.bufs[0]; const auto& coeff = b.bufs[1]; const auto& out = b.bufs[2];
        size_t n = out.size() / 32;
        if (coeff.size() / 32 < n) n = coeff.size() / 32;
        if (exp.size() / 4 < n) n = exp.size() / 4;
        hip::check(ctx_, lig_powmod(ctx_, powmod_base_.data(), exp.data(), coeff.data(), out.data(), n, add), "EltwisePowMod");
    }
    static void* at(const buffer_type& b, size_t elem_off) { return static_cast<char*>(b.data()) + elem_off * 32; }
    // element offsets are WebGPU dynamic offsets (buffer_binding.hpp:27-29, engine.cpp eltwise dispatch): they move the bound
    // window of size() bytes inside the underlying allocation, they do not shrink it (vbn254fr binds variable 0 of its
    // 512-variable slab once and addresses the others by offset, host_modules/vbn254fr.hpp:64-69)
    static size_t count(const buffer_type& b, size_t) { return b.size() / 32; }
    void ntt(const hip::buffer_binding& b, int which, int inverse) { hip::check(ctx_, lig_ntt(ctx_, b.bufs[0].data(), which, inverse), "ntt"); }
    // the reference runs eltwise kernels over arrayLength(x) elements (kernels.wgsl.in:326-): the shortest operand bounds the op
    void el3(int op, const hip::buffer_binding& b, hip::eltwise_offset o) {
        const auto& x = b.bufs[0]; const auto& y = b.bufs[1]; const auto& out = b.bufs[2];
        size_t n = count(x, o.x);
        if (count(y, o.y) < n) n = count(y, o.y);
        if (count(out, o.z) < n) n = count(out, o.z);
        hip::check(ctx_, lig_eltwise(ctx_, op, at(x, o.x), at(y, o.y), at(out, o.z), n, nullptr, 0), "eltwise");
    }
    void el2(int op, const hip::buffer_binding& b, hip::eltwise_offset o, const hip::scalar* c) {
        const auto& x = b.bufs[0]; const auto& out = b.bufs[1];
        size_t n = count(x, o.x);
        if (count(out, o.z) < n) n = count(out, o.z);
        hip::check(ctx_, lig_eltwise(ctx_, op, at(x, o.x), nullptr, at(out, o.z), n, c ? c->data() : nullptr, 0), "eltwise");
    }

    lig_ctx* ctx_ = nullptr;
    int device_ = 0;
    bool upstream_slices_ = false;
    size_t sha_instances_ = 0;
    size_t powmod_bits_ = 0;
    bool powmod_has_base_ = false;
    hip::scalar powmod_base_{};
};

}  // namespace ligero
