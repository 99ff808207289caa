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
#if defined(__x86_64__) && defined(__SSE2__)
#include <emmintrin.h>
#endif
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
    ~hip_context() {
        if (ctx_) {
            try { flush(); } catch (...) {}
            release_rings();
            lig_sync(ctx_); lig_ctx_destroy(ctx_);
        }
    }

    // ---- deferred row mode (no upstream counterpart).  The stage contexts call the executor ROW BY ROW
    // (write_buffer_clear -> encode_ntt_device -> sha256_digest_update | EltwiseFMAMod x 2 | sample_gather: nonbatch_context.hpp:445-471,
    // 654-780, 924-970), which on this device is launch- and transfer-latency bound.  With set_deferred_rows(cap) the SAME call sequence
    // is recorded instead of executed: write_buffer_clear snapshots the host row into page-locked staging and gives the bound buffer a new
    // "version" (a slot of a device ring); encode / column-hash update / the two FMA forms / sample_gather on versioned buffers are logged;
    // every `cap` rows -- and before ANY other call of this class (copy_to_host, device_synchronize, clear_buffer, the single transforms,
    // every other eltwise op, ...) -- the log is flushed through the batched entry points lig_encode_rows / lig_sha_update_rows /
    // lig_rlc_rows / lig_gather_rows, and the real buffers receive the contents the eager execution would have left in them.  Results
    // are identical (exact field arithmetic: the accumulations commute; hash rows keep their order); the flush is asynchronous (the
    // upload and the kernels of one batch run under the caller's next rows; two staging halves).
    void set_deferred_rows(size_t ring_rows = 512) {
        flush();
        release_rings();
        ring_cap_ = ring_rows;
    }
    size_t deferred_rows() const { return ring_cap_; }
    void flush() { if (ring_cap_ && dirty_) flush_impl(); }

    // ---- lifecycle (wgpu.hpp:71-82).  shader_path is accepted and ignored: kernels are compiled into the library.
    void webgpu_init(size_t gpu_threads = 0, const std::string& shader_path = "") { (void)gpu_threads; (void)shader_path; }
    // p, barrett factor and the roots are fixed BN254 constants inside the library (the reference's WGSL hard-codes
    // p / mu / J as well, shader/bn254fr.wgsl.in:19-45); the arguments are accepted for source compatibility.
    template <typename... Ignored>
    void ntt_init(size_t l, size_t k, size_t n, const Ignored&...) {
        if (ctx_) { try { flush(); } catch (...) {} release_rings(); lig_ctx_destroy(ctx_); ctx_ = nullptr; }
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
    void device_synchronize() { flush(); hip::check(ctx_, lig_sync(ctx_), "device_synchronize"); }
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
        flush();
        hip::check(ctx_, lig_write(ctx_, buf.data(), data, len * sizeof(T)), "write_buffer");
    }
    // device_context.hpp:95-98: write_buffer(buf, data, len); clear_buffer(buf.slice(len * sizeof(T))) -- through slice(), so that
    // the upstream slicing semantics (set_upstream_slice_compat) reach it
    template <typename T> void write_buffer_clear(buffer_type buf, const T* data, size_t len) {
        if (ring_cap_ && defer_write(buf, reinterpret_cast<const uint8_t*>(data), len * sizeof(T))) return;
        flush();
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
    void clear_buffer(buffer_type buf) { flush(); hip::check(ctx_, lig_clear(ctx_, buf.data(), buf.size()), "clear_buffer"); }
    void copy_buffer_to_buffer(buffer_type from, buffer_type to) {
        flush();
        hip::check(ctx_, lig_copy(ctx_, to.data(), from.data(), from.size() < to.size() ? from.size() : to.size()), "copy_buffer_to_buffer");
    }
    void copy_buffer_clear(buffer_type from, buffer_type to) {      // copy, then zero the rest of `to`
        copy_buffer_to_buffer(from, to);
        if (to.size() > from.size())
            hip::check(ctx_, lig_clear(ctx_, static_cast<char*>(to.data()) + from.size(), to.size() - from.size()), "copy_buffer_clear");
    }
    template <typename T> std::vector<T> copy_to_host(buffer_type buf) {   // blocking
        flush();
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
    void encode_ntt_device(const hip::buffer_binding& b) {
        if (ring_cap_ && defer_encode(b.bufs[0])) return;
        flush();
        hip::check(ctx_, lig_encode(ctx_, b.bufs[0].data()), "encode_ntt_device");
    }
    void decode_ntt_device(const hip::buffer_binding& b) { flush(); hip::check(ctx_, lig_decode(ctx_, b.bufs[0].data()), "decode_ntt_device"); }
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
    void EltwiseFMAMod(const hip::buffer_binding& b, hip::eltwise_offset o = {}) {
        if (ring_cap_ && !o.x && !o.y && !o.z && defer_fma(b.bufs[0], b.bufs[1], b.bufs[2])) return;
        el3(LIG_OP_FMA, b, o);
    }
    void EltwiseAddAssignMod(const hip::buffer_binding& b, hip::eltwise_offset o = {}) { el2(LIG_OP_ADD_ASSIGN, b, o, nullptr); }
    void EltwiseAddMod(const hip::buffer_binding& b, const hip::scalar& c, hip::eltwise_offset o = {}) { el2(LIG_OP_ADD_CONST, b, o, &c); }
    void EltwiseSubConstMod(const hip::buffer_binding& b, const hip::scalar& c, hip::eltwise_offset o = {}) { el2(LIG_OP_SUB_CONST, b, o, &c); }
    void EltwiseConstSubMod(const hip::buffer_binding& b, const hip::scalar& c, hip::eltwise_offset o = {}) { el2(LIG_OP_CONST_SUB, b, o, &c); }
    void EltwiseMultMod(const hip::buffer_binding& b, const hip::scalar& c, hip::eltwise_offset o = {}) { el2(LIG_OP_MUL_CONST, b, o, &c); }
    void EltwiseMontMultMod(const hip::buffer_binding& b, const hip::scalar& c, hip::eltwise_offset o = {}) { el2(LIG_OP_MONTMUL_CONST, b, o, &c); }
    void EltwiseFMAMod(const hip::buffer_binding& b, const hip::scalar& c, hip::eltwise_offset o = {}) {
        if (ring_cap_ && !o.x && !o.z && defer_fma_const(b.bufs[0], b.bufs[1], c)) return;
        el2(LIG_OP_FMA_CONST, b, o, &c);
    }
    void EltwiseBitDecompose(const hip::buffer_binding& b, uint32_t bit, hip::eltwise_offset o = {}) {
        const auto& x = b.bufs[0]; const auto& out = b.bufs[1];
        flush();
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
    void sha256_digest_init(const hip::buffer_binding& b) { flush(); hip::check(ctx_, lig_sha_init(ctx_, b.bufs[0].data(), sha_instances_), "sha256_digest_init"); }
    void sha256_digest_update(const hip::buffer_binding& ctxb, const hip::buffer_binding& in) {
        if (ring_cap_ && defer_hash(ctxb.bufs[0], in.bufs[0])) return;
        flush();
        hip::check(ctx_, lig_sha_update(ctx_, ctxb.bufs[0].data(), in.bufs[0].data()), "sha256_digest_update");
    }
    void sha256_digest_final(const hip::buffer_binding& b) { flush(); hip::check(ctx_, lig_sha_final(ctx_, b.bufs[0].data(), b.bufs[1].data()), "sha256_digest_final"); }

    // ---- sampling (wgpu.hpp:148-151)
    void sampling_init(const std::vector<size_t>& idx) {
        std::vector<uint32_t> v(idx.begin(), idx.end());
        flush();
        sample_count_ = v.size();
        hip::check(ctx_, lig_sample_init(ctx_, v.data(), v.size()), "sampling_init");
    }
    void sample_gather(const hip::buffer_binding& b, size_t slot) {
        if (ring_cap_ && defer_gather(b.bufs[0], b.bufs[1], slot)) return;
        flush();
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
        flush();
        hip::check(ctx_, lig_powmod(ctx_, powmod_base_.data(), exp.data(), coeff.data(), out.data(), n, add), "EltwisePowMod");
    }
    static void* at(const buffer_type& b, size_t elem_off) { return static_cast<char*>(b.data()) + elem_off * 32; }
    // element offsets are WebGPU dynamic offsets (buffer_binding.hpp:27-29, engine.cpp eltwise dispatch): they move the bound
    // window of size() bytes inside the underlying allocation, they do not shrink it (vbn254fr binds variable 0 of its
    // 512-variable slab once and addresses the others by offset, host_modules/vbn254fr.hpp:64-69)
    static size_t count(const buffer_type& b, size_t) { return b.size() / 32; }
    void ntt(const hip::buffer_binding& b, int which, int inverse) { flush(); hip::check(ctx_, lig_ntt(ctx_, b.bufs[0].data(), which, inverse), "ntt"); }
    // the reference runs eltwise kernels over arrayLength(x) elements (kernels.wgsl.in:326-): the shortest operand bounds the op
    void el3(int op, const hip::buffer_binding& b, hip::eltwise_offset o) {
        const auto& x = b.bufs[0]; const auto& y = b.bufs[1]; const auto& out = b.bufs[2];
        size_t n = count(x, o.x);
        if (count(y, o.y) < n) n = count(y, o.y);
        if (count(out, o.z) < n) n = count(out, o.z);
        flush();
        hip::check(ctx_, lig_eltwise(ctx_, op, at(x, o.x), at(y, o.y), at(out, o.z), n, nullptr, 0), "eltwise");
    }
    void el2(int op, const hip::buffer_binding& b, hip::eltwise_offset o, const hip::scalar* c) {
        const auto& x = b.bufs[0]; const auto& out = b.bufs[1];
        size_t n = count(x, o.x);
        if (count(out, o.z) < n) n = count(out, o.z);
        flush();
        hip::check(ctx_, lig_eltwise(ctx_, op, at(x, o.x), nullptr, at(out, o.z), n, c ? c->data() : nullptr, 0), "eltwise");
    }

    // ------------------------------------------------------------------------------------------------ deferred row mode
    // A versioned buffer = a codeword buffer (whole allocation, n elements) that write_buffer_clear has written since the last flush.
    struct dring {
        void* key = nullptr;                  // the real buffer's device address
        buffer_type buf;                      // held while versions are pending (keeps the allocation alive)
        void* d_msgs = nullptr;               // cap x k elements: the rows as written
        void* d_cw = nullptr;                 // cap x n elements: their codewords
        uint8_t* h_stage[2] = {nullptr, nullptr};   // page-locked, cap x k elements each
        size_t count = 0;                     // versions since the last flush
        bool last_encoded = false;
    };
    enum { DOP_HASH = 0, DOP_FMAC = 1, DOP_FMA = 2, DOP_GATHER = 3 };
    struct dop { int kind; void* dst; int rx; size_t sx; int ry; size_t sy; size_t aux; hip::scalar c; };

    static bool whole(const buffer_type& b, size_t bytes) { return b.offset() == 0 && b.size() == bytes; }
    int ring_of(const buffer_type& b) const {
        for (size_t i = 0; i < rings_.size(); i++) if (rings_[i].key == b.data()) return (int)i;
        return -1;
    }
    // the latest version of `b`, if it has one and it has been encoded
    int encoded_version(const buffer_type& b, size_t* slot) const {
        if (!whole(b, (size_t)encoding_size() * 32)) return -1;
        const int r = ring_of(b);
        if (r < 0 || !rings_[r].count || !rings_[r].last_encoded) return -1;
        *slot = rings_[r].count - 1;
        return r;
    }
    bool pending_version(const buffer_type& b) const { const int r = ring_of(b); return r >= 0 && rings_[r].count; }

    // The snapshot of a row: 256 KiB into a page-locked staging area of hundreds of MB that is read next by the DMA engine, never by this
    // core.  A plain memcpy pulls every destination line into the cache first (read-for-ownership) and writes it back later: three
    // memory transfers per byte; streaming stores write the line once (x86-64 baseline SSE2; elsewhere memcpy).
    static void stream_copy(uint8_t* dst, const uint8_t* src, size_t bytes) {
#if defined(__x86_64__) && defined(__SSE2__)
        if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (bytes & 63) == 0) {
            for (size_t i = 0; i < bytes; i += 64) {
                const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i)), b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 16));
                const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 32)), d = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 48));
                _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i), a); _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i + 16), b);
                _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i + 32), c); _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i + 48), d);
            }
            _mm_sfence();
            return;
        }
#endif
        std::memcpy(dst, src, bytes);
    }
    bool defer_write(const buffer_type& buf, const uint8_t* data, size_t bytes) {
        const size_t k = padding_size(), n = encoding_size(), row = k * 32;
        if (!whole(buf, n * 32) || buf.upstream_slices() || bytes < row || bytes > n * 32 || (bytes & 7)) return false;
        // the reference writes 2k elements per row, the upper k of them zero (limbs_ of 2 * padding_size elements,
        // nonbatch_context.hpp:447); a row with anything beyond k (the two degree-2k mask rows) is not a row of the ring
        const uint64_t* tail = reinterpret_cast<const uint64_t*>(data + row);
        uint64_t any = 0;
        for (size_t i = 0, m = (bytes - row) / 8; i < m; i++) any |= tail[i];
        if (any) return false;
        int r = ring_of(buf);
        if (r < 0) {                          // a ring nobody is bound to (rings are a pool: a flush unbinds them; the stage contexts come and go)
            for (size_t i = 0; i < rings_.size() && r < 0; i++) if (!rings_[i].key) r = (int)i;
            if (r >= 0) rings_[r].key = buf.data();
        }
        if (r < 0) {
            dring g;
            g.key = buf.data();
            hip::check(ctx_, lig_malloc(ctx_, ring_cap_ * row, &g.d_msgs), "deferred rows: ring");
            hip::check(ctx_, lig_malloc(ctx_, ring_cap_ * n * 32, &g.d_cw), "deferred rows: ring");
            for (int h = 0; h < 2; h++) { void* p = nullptr; hip::check(ctx_, lig_host_alloc(ctx_, ring_cap_ * row, &p), "deferred rows: staging"); g.h_stage[h] = (uint8_t*)p; }
            rings_.push_back(g);
            r = (int)rings_.size() - 1;
        }
        if (rings_[r].count && !rings_[r].last_encoded) rings_[r].count--;       // overwritten before anything used it: the slot is reused
        if (rings_[r].count == ring_cap_) { flush_impl(); rings_[r].key = buf.data(); }      // (the flush unbound the ring: it stays this buffer's)
        if (!dirty_) {                                                              // first row of a batch: its staging half must be free
            hip::check(ctx_, lig_fence_wait(ctx_, fence_[half_]), "deferred rows: fence");
        }
        dring& g = rings_[r];
        stream_copy(g.h_stage[half_] + g.count * row, data, row);
        if (!g.count) g.buf = buf;
        g.count++;
        g.last_encoded = false;
        dirty_ = true;
        return true;
    }
    bool defer_encode(const buffer_type& buf) {
        const int r = ring_of(buf);
        if (r < 0 || !rings_[r].count || rings_[r].last_encoded) return false;       // (encoding a codeword again: eager)
        rings_[r].last_encoded = true;
        return true;
    }
    bool defer_hash(const buffer_type& state, const buffer_type& in) {
        size_t s;
        const int r = encoded_version(in, &s);
        if (r < 0 || sha_instances_ != encoding_size() || pending_version(state)) return false;
        ops_.push_back(dop{DOP_HASH, state.data(), r, s, -1, 0, 0, {}});
        return true;
    }
    bool defer_fma_const(const buffer_type& x, const buffer_type& out, const hip::scalar& c) {
        size_t s;
        const int r = encoded_version(x, &s);
        if (r < 0 || !whole(out, (size_t)encoding_size() * 32) || pending_version(out)) return false;
        ops_.push_back(dop{DOP_FMAC, out.data(), r, s, -1, 0, 0, c});
        return true;
    }
    bool defer_fma(const buffer_type& x, const buffer_type& y, const buffer_type& out) {
        size_t sx, sy;
        const int rx = encoded_version(x, &sx), ry = encoded_version(y, &sy);
        if (rx < 0 || ry < 0 || !whole(out, (size_t)encoding_size() * 32) || pending_version(out)) return false;
        ops_.push_back(dop{DOP_FMA, out.data(), rx, sx, ry, sy, 0, {}});
        return true;
    }
    bool defer_gather(const buffer_type& from, const buffer_type& to, size_t slot) {
        size_t s;
        const int r = encoded_version(from, &s);
        if (r < 0 || !sample_count_ || pending_version(to) || (slot + 1) * sample_count_ * 32 > to.size()) return false;
        ops_.push_back(dop{DOP_GATHER, to.data(), r, s, -1, 0, slot, {}});
        return true;
    }
    void flush_impl() {
        const size_t k = padding_size(), n = encoding_size(), row = k * 32, cwb = n * 32;
        for (dring& g : rings_) {
            if (!g.count) continue;
            hip::check(ctx_, lig_write_async(ctx_, g.d_msgs, g.h_stage[half_], g.count * row), "deferred rows: upload");
            const size_t enc = g.last_encoded ? g.count : g.count - 1;          // (only the latest version can be unencoded)
            if (enc) hip::check(ctx_, lig_encode_rows(ctx_, g.d_msgs, g.d_cw, enc), "deferred rows: encode");
        }
        hip::check(ctx_, lig_fence_record(ctx_, &fence_[half_]), "deferred rows: fence");      // the staging half is free once this is reached
        auto cw = [&](int r, size_t s) { return static_cast<char*>(rings_[r].d_cw) + s * cwb; };
        std::vector<char> done(ops_.size(), 0);
        std::vector<uint8_t> rc;
        for (size_t i = 0; i < ops_.size(); i++) {
            if (done[i]) continue;
            const dop& o = ops_[i];
            // the maximal run of ops of this kind and destination over CONSECUTIVE versions (and consecutive sample slots); ops of other
            // kinds / destinations in between are independent of it (accumulations commute, every hash state and gather target has
            // its own order), so they are skipped over, not waited for
            std::vector<size_t> run{i};
            for (size_t j = i + 1; j < ops_.size(); j++) {
                const dop& q = ops_[j];
                if (done[j] || q.kind != o.kind || q.dst != o.dst) continue;
                const dop& last = ops_[run.back()];
                if (q.rx != o.rx || q.sx != last.sx + 1 || q.ry != o.ry || (o.ry >= 0 && q.sy != last.sy + 1) || (o.kind == DOP_GATHER && q.aux != last.aux + 1)) break;
                run.push_back(j);
            }
            for (size_t j : run) done[j] = 1;
            const size_t cnt = run.size();
            switch (o.kind) {
                case DOP_HASH: hip::check(ctx_, lig_sha_update_rows(ctx_, o.dst, cw(o.rx, o.sx), cnt), "deferred rows: sha256_digest_update"); break;
                case DOP_FMAC: {
                    // an FMA run over the same versions of the same buffer rides along in the same pass (check_code + check_linear of a row)
                    void* lin = nullptr; const char* rn = nullptr;
                    for (size_t j = i + 1; j < ops_.size() && !lin; j++) {
                        const dop& q = ops_[j];
                        if (done[j] || q.kind != DOP_FMA || q.rx != o.rx || q.sx != o.sx) continue;
                        size_t m = 0, jj = j;
                        std::vector<size_t> frun;
                        for (; jj < ops_.size() && m < cnt; jj++) {
                            const dop& f = ops_[jj];
                            if (done[jj] || f.kind != DOP_FMA || f.dst != q.dst) continue;
                            if (f.rx != o.rx || f.sx != o.sx + m || f.ry != q.ry || f.sy != q.sy + m) break;
                            frun.push_back(jj); m++;
                        }
                        if (m == cnt) { lin = q.dst; rn = cw(q.ry, q.sy); for (size_t f : frun) done[f] = 1; }
                        break;
                    }
                    rc.resize(cnt * 32);
                    for (size_t m = 0; m < cnt; m++) std::memcpy(rc.data() + 32 * m, ops_[run[m]].c.data(), 32);
                    hip::check(ctx_, lig_rlc_rows(ctx_, cw(o.rx, o.sx), rn, cnt, rc.data(), o.dst, lin, nullptr, nullptr, 0, nullptr), "deferred rows: EltwiseFMAMod");
                    break;
                }
                case DOP_FMA: hip::check(ctx_, lig_rlc_rows(ctx_, cw(o.rx, o.sx), cw(o.ry, o.sy), cnt, nullptr, nullptr, o.dst, nullptr, nullptr, 0, nullptr), "deferred rows: EltwiseFMAMod"); break;
                case DOP_GATHER: hip::check(ctx_, lig_gather_rows(ctx_, cw(o.rx, o.sx), cnt, static_cast<char*>(o.dst) + o.aux * sample_count_ * 32), "deferred rows: sample_gather"); break;
            }
        }
        ops_.clear();
        // what the eager execution would have left in the real buffers: the latest version
        for (dring& g : rings_) {
            if (!g.count) continue;
            const size_t last = g.count - 1;
            if (g.last_encoded) hip::check(ctx_, lig_copy(ctx_, g.key, static_cast<char*>(g.d_cw) + last * cwb, cwb), "deferred rows: materialise");
            else {
                hip::check(ctx_, lig_copy(ctx_, g.key, static_cast<char*>(g.d_msgs) + last * row, row), "deferred rows: materialise");
                hip::check(ctx_, lig_clear(ctx_, static_cast<char*>(g.key) + row, cwb - row), "deferred rows: materialise");
            }
            g.count = 0; g.last_encoded = false; g.buf = buffer_type(); g.key = nullptr;
        }
        dirty_ = false;
        half_ ^= 1;
    }
    void release_rings() {
        if (!ctx_) { rings_.clear(); return; }
        if (!rings_.empty() || fence_[0] || fence_[1]) lig_sync(ctx_);
        for (dring& g : rings_) {
            lig_free(ctx_, g.d_msgs); lig_free(ctx_, g.d_cw);
            for (int h = 0; h < 2; h++) lig_host_free(ctx_, g.h_stage[h]);
        }
        rings_.clear(); ops_.clear(); dirty_ = false;
        for (int h = 0; h < 2; h++) { lig_fence_destroy(ctx_, fence_[h]); fence_[h] = nullptr; }
    }

    size_t ring_cap_ = 0, sample_count_ = 0;
    bool dirty_ = false;
    int half_ = 0;
    void* fence_[2] = {nullptr, nullptr};
    std::vector<dring> rings_;
    std::vector<dop> ops_;

    lig_ctx* ctx_ = nullptr;
    int device_ = 0;
    bool upstream_slices_ = false;
    size_t sha_instances_ = 0;
    size_t powmod_bits_ = 0;
    bool powmod_has_base_ = false;
    hip::scalar powmod_base_{};
};

}  // namespace ligero
