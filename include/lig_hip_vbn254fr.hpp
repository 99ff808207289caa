/* lig_hip_vbn254fr.hpp -- the guest-visible batch ("vbn254fr") operations on top of ligero::hip_context.
 *
 * Mirrors vbn254fr_module (include/host_modules/vbn254fr.hpp:33-600): a slab of 512 batch variables of k = padding_size()
 * elements each (l = message_size() data slots + k - l slots of zero-knowledge padding), a FIFO free list of element
 * offsets (:84-99,:123-127), and one executor call sequence per guest operation followed by the constraint hook the
 * reference raises for it (Context::on_batch_init / _bit / _equal / _quadratic, include/zkp/nonbatch_context.hpp:497-556).
 * Method names and the executor calls they make are the reference's; what differs is only where the arguments come from:
 * the reference pops them from the WASM value stack and guest memory (interpreter, out of scope here), this class takes
 * them as C++ values (variable handles = element offsets, exactly what the reference stores in guest memory).
 *
 * Slicing semantics: the layer slices through hip_context's buffer views, so executor.set_upstream_slice_compat(true) (before
 * the first variable is allocated) reproduces what an unmodified v1.5.0 build does to vbn254fr variables -- the pad of
 * on_batch_init lands in variable 0, write_buffer_clear of vbn254fr_set_ui / _set_ui_scalar wipes the slab up to the end of the
 * variable (include/lig_hip.h, LIG_BOP_UPSTREAM_COMPAT).  A recorded program then starts with that op.
 *
 * Reference defects that are NOT reproduced: vbn254fr_set_ui writes through an empty vector
 * (:150-153) -- here the values are written as intended; vbn254fr_set_bytes reads every element from bytes + len * count
 * (:266) -- callers of set() pass the decoded elements.
 *
 * Recording: with record(true) every guest operation is also appended to a lig_batch_op program (include/lig_hip.h) --
 * the form in which the batched prover / verifier (lig_synth_prepare, lig_synth_verify) and the oracle take a batch
 * computation: slots = handle / k, constants and values in the data blob.
 *
 * Context concept: executor() -> hip_context&, on_batch_init(buffer&), on_batch_bit(buffer&), on_batch_equal(buffer&,
 * buffer&), on_batch_quadratic(buffer&, buffer&, buffer&).
 */
#pragma once
#include <deque>
#include <stdexcept>
#include <vector>

#include "lig_hip_context.hpp"

namespace ligero {

template <typename Context>
class hip_vbn254fr {
public:
    using executor_t = hip_context;
    using buffer_t = hip_context::buffer_type;
    using bignum_t = hip_context::device_bignum_type;
    using handle_t = uint32_t;                                   // element offset of the variable inside the slab

    static constexpr auto module_name = "vbn254fr";
    static constexpr size_t max_variables = 512;                 // vbn254fr.hpp:43
    static constexpr size_t num_bits = 254;                      // Context::field_type::num_bits for BN254 Fr

    explicit hip_vbn254fr(Context* ctx) : ctx_(ctx), executor_(ctx->executor()) {}

    // ---- program recording (no upstream counterpart: the reference re-runs the guest instead of keeping a program)
    void record(bool on) { record_ = on; }
    const std::vector<lig_batch_op>& recorded_ops() const { return ops_; }
    const std::vector<uint8_t>& recorded_data() const { return data_; }

    // vbn254fr.hpp:54-78 (allocation is delayed until the first batch variable is requested)
    void initialize_buffer() {
        num_buf_elements_ = executor_.message_size();
        num_buf_bytes_ = executor_.padding_size() * bignum_t::num_bytes;
        tmp_buf_ = executor_.make_device_buffer(num_buf_bytes_);
        buffer_base_ = executor_.make_device_buffer(max_variables * num_buf_bytes_);
        buffer_t base_view = buffer_base_.slice_bytes(0, num_buf_bytes_);
        bind_compute2_ = executor_.bind_eltwise2(base_view, tmp_buf_);
        bind_compute3_ = executor_.bind_eltwise3(base_view, base_view, tmp_buf_);
        for (size_t i = 0; i < max_variables; i++) free_list_.push_back(i * executor_.padding_size());
        initialized_ = true;
        if (record_ && buffer_base_.upstream_slices()) ops_.push_back(lig_batch_op{LIG_BOP_UPSTREAM_COMPAT, 0, 0, 0, 0, 0, 0});
    }
    buffer_t get_buffer_from_offset(size_t element_offset) {
        return buffer_base_.slice_bytes(element_offset * bignum_t::num_bytes, num_buf_bytes_);
    }

    size_t vbn254fr_get_size() const { return executor_.message_size(); }

    // vbn254fr_alloc / vbn254fr_allocate (:84-99,:136-141): front of the FIFO free list; exhaustion aborts upstream, throws here
    handle_t vbn254fr_alloc() {
        if (!initialized_) initialize_buffer();
        if (free_list_.empty()) throw std::runtime_error("vbn254fr: bad alloc, 0/512 free buffers available");
        const handle_t h = (handle_t)free_list_.front();
        free_list_.pop_front();
        return h;
    }
    // vbn254fr_free / vbn254fr_deallocate (:123-127,:143-149): the buffer is cleared and goes to the BACK of the list
    void vbn254fr_free(handle_t h) {
        log(LIG_BOP_FREE, 0, h, 0);
        executor_.clear_buffer(get_buffer_from_offset(h));
        free_list_.emplace_back(h);
    }

    // vbn254fr_set_ui (:151-168): len 32-bit values, rest of the variable cleared, then on_batch_init
    void vbn254fr_set_ui(handle_t fp, const uint32_t* ui, size_t len) {
        std::vector<bignum_t> vals(len);
        for (size_t i = 0; i < len; i++) vals[i] = bignum_t(ui[i]);
        log_set(fp, vals);
        set_elements(fp, vals);
    }
    // vbn254fr_set_ui_scalar (:170-183): the same value in all l data slots
    void vbn254fr_set_ui_scalar(handle_t fp, uint32_t ui) {
        const bignum_t v(ui);
        if (record_) log(LIG_BOP_SET_SCALAR, 0, fp, 0, 0, blob(v.limbs, 32));
        set_elements(fp, std::vector<bignum_t>(num_buf_elements_, v));
    }
    // vbn254fr_set_str / _set_bytes (:185-275) after parsing: one canonical element per slot -- write_limbs (wgpu.hpp:177-183:
    // write_buffer, NOTHING cleared: slots beyond elems.size() keep their content) + on_batch_init
    void vbn254fr_set(handle_t fp, const std::vector<hip::scalar>& elems) {
        if (elems.size() > num_buf_elements_) throw std::invalid_argument("vbn254fr_set: more than message_size() elements");
        if (record_) {
            std::vector<bignum_t> vals;
            for (const auto& e : elems) vals.emplace_back(e);
            log(LIG_BOP_SET, 0, fp, 0, (uint32_t)vals.size(), blob(vals.data(), 32 * vals.size()), LIG_BOP_F_WRITE_LIMBS);
        }
        buffer_t x = get_buffer_from_offset(fp);
        executor_.write_limbs(x, elems);
        ctx_->on_batch_init(x);
    }
    // vbn254fr_set_str_scalar / _set_bytes_scalar (:219-243,:277-296): write_limbs(x, value, message_size()) + on_batch_init
    void vbn254fr_set_scalar(handle_t fp, const hip::scalar& e) {
        if (record_) log(LIG_BOP_SET_SCALAR, 0, fp, 0, 0, blob(e.data(), 32), LIG_BOP_F_WRITE_LIMBS);
        buffer_t x = get_buffer_from_offset(fp);
        executor_.write_limbs(x, e, num_buf_elements_);
        ctx_->on_batch_init(x);
    }

    // vbn254fr_copy (:298-317)
    void vbn254fr_copy(handle_t out_h, handle_t in_h) {
        log(LIG_BOP_COPY, out_h, in_h, 0);
        buffer_t in = get_buffer_from_offset(in_h), out = get_buffer_from_offset(out_h);
        if (in == out) {
            executor_.copy_buffer_to_buffer(in, tmp_buf_);
            executor_.copy_buffer_to_buffer(tmp_buf_, out);
        } else {
            executor_.copy_buffer_to_buffer(in, out);
        }
        ctx_->on_batch_equal(out, in);
    }
    // vbn254fr_print reads the variable back (:319-349); here: all k elements as host bignums
    std::vector<bignum_t> vbn254fr_read(handle_t h) { return executor_.template copy_to_host<bignum_t>(get_buffer_from_offset(h)); }

    // ---- arithmetic (:353-560).  Every op computes into the temporary and copies to `out`, so out may alias x or y.
    void vbn254fr_addmod(handle_t out, handle_t x, handle_t y) {
        log(LIG_BOP_ADD, out, x, y);
        executor_.EltwiseAddMod(bind_compute3_, {.x = x, .y = y});
        executor_.copy_buffer_to_buffer(tmp_buf_, get_buffer_from_offset(out));
    }
    void vbn254fr_addmod_constant(handle_t out, handle_t x, const hip::scalar& k) {
        log_const(LIG_BOP_ADD_CONST, out, x, k);
        executor_.EltwiseAddMod(bind_compute2_, k, {.x = x});
        executor_.copy_buffer_to_buffer(tmp_buf_, get_buffer_from_offset(out));
    }
    void vbn254fr_submod(handle_t out, handle_t x, handle_t y) {
        log(LIG_BOP_SUB, out, x, y);
        executor_.EltwiseSubMod(bind_compute3_, {.x = x, .y = y});
        executor_.copy_buffer_to_buffer(tmp_buf_, get_buffer_from_offset(out));
    }
    void vbn254fr_submod_constant(handle_t out, handle_t x, const hip::scalar& k) {
        log_const(LIG_BOP_SUB_CONST, out, x, k);
        executor_.EltwiseSubConstMod(bind_compute2_, k, {.x = x});
        executor_.copy_buffer_to_buffer(tmp_buf_, get_buffer_from_offset(out));
    }
    void vbn254fr_constant_submod(handle_t out, const hip::scalar& k, handle_t x) {
        log_const(LIG_BOP_CONST_SUB, out, x, k);
        executor_.EltwiseConstSubMod(bind_compute2_, k, {.x = x});
        executor_.copy_buffer_to_buffer(tmp_buf_, get_buffer_from_offset(out));
    }
    // vbn254fr_mulmod (:446-470): z = x * y is a quadratic row triple (x, y, z) raised BEFORE the copy to out
    void vbn254fr_mulmod(handle_t out, handle_t x, handle_t y) {
        log(LIG_BOP_MUL, out, x, y);
        buffer_t bx = get_buffer_from_offset(x), by = get_buffer_from_offset(y);
        executor_.EltwiseMultMod(bind_compute3_, {.x = x, .y = y});
        ctx_->on_batch_quadratic(bx, by, tmp_buf_);
        executor_.copy_buffer_to_buffer(tmp_buf_, get_buffer_from_offset(out));
    }
    void vbn254fr_mulmod_constant(handle_t out, handle_t x, const hip::scalar& k) {
        log_const(LIG_BOP_MUL_CONST, out, x, k);
        executor_.EltwiseMultMod(bind_compute2_, k, {.x = x});
        executor_.copy_buffer_to_buffer(tmp_buf_, get_buffer_from_offset(out));
    }
    void vbn254fr_mont_mul_constant(handle_t out, handle_t x, const hip::scalar& k) {
        log_const(LIG_BOP_MONTMUL_CONST, out, x, k);
        executor_.EltwiseMontMultMod(bind_compute2_, k, {.x = x});
        executor_.copy_buffer_to_buffer(tmp_buf_, get_buffer_from_offset(out));
    }
    // vbn254fr_divmod (:507-525): q = x / y is constrained as q * y = x, i.e. the triple (q, y, x)
    void vbn254fr_divmod(handle_t out, handle_t x, handle_t y) {
        log(LIG_BOP_DIV, out, x, y);
        buffer_t bx = get_buffer_from_offset(x), by = get_buffer_from_offset(y);
        executor_.EltwiseDivMod(bind_compute3_, {.x = x, .y = y});
        ctx_->on_batch_quadratic(tmp_buf_, by, bx);
        executor_.copy_buffer_to_buffer(tmp_buf_, get_buffer_from_offset(out));
    }
    // vbn254fr_assert_equal (:527-547)
    void vbn254fr_assert_equal(handle_t x, handle_t y) {
        log(LIG_BOP_ASSERT_EQUAL, 0, x, y);
        buffer_t bx = get_buffer_from_offset(x), by = get_buffer_from_offset(y);
        ctx_->on_batch_equal(bx, by);
    }
    // vbn254fr_bit_decompose (:549-565): bit i of every element of x into out[i], each a committed "bit" row
    void vbn254fr_bit_decompose(const handle_t* out, handle_t x) {
        if (record_) {
            std::vector<uint32_t> slots(num_bits);
            for (uint32_t i = 0; i < num_bits; i++) slots[i] = slot(out[i]);
            log(LIG_BOP_BIT_DECOMPOSE, 0, x, 0, (uint32_t)num_bits, blob(slots.data(), 4 * slots.size()));
        }
        for (uint32_t i = 0; i < num_bits; i++) {
            buffer_t bit = get_buffer_from_offset(out[i]);
            executor_.EltwiseBitDecompose(bind_compute2_, i, {.x = x});
            executor_.copy_buffer_to_buffer(tmp_buf_, bit);
            ctx_->on_batch_bit(bit);
        }
    }
    void finalize() { executor_.device_synchronize(); }          // :592-594

    size_t free_variables() const { return free_list_.size(); }

private:
    void set_elements(handle_t fp, const std::vector<bignum_t>& vals) {
        buffer_t x = get_buffer_from_offset(fp);
        executor_.write_buffer_clear(x, vals.data(), vals.size());
        ctx_->on_batch_init(x);
    }
    void log_set(handle_t fp, const std::vector<bignum_t>& vals) {
        if (record_) log(LIG_BOP_SET, 0, fp, 0, (uint32_t)vals.size(), blob(vals.data(), 32 * vals.size()));
    }
    uint32_t slot(handle_t h) const { return (uint32_t)(h / executor_.padding_size()); }
    uint64_t blob(const void* p, size_t n) {
        const uint64_t off = data_.size();
        data_.insert(data_.end(), static_cast<const uint8_t*>(p), static_cast<const uint8_t*>(p) + n);
        return off;
    }
    void log(uint32_t op, handle_t out, handle_t x, handle_t y, uint32_t len = 0, uint64_t off = 0, uint32_t flags = 0) {
        if (record_) ops_.push_back(lig_batch_op{op, slot(out), slot(x), slot(y), len, flags, off});
    }
    void log_const(uint32_t op, handle_t out, handle_t x, const hip::scalar& k) { if (record_) log(op, out, x, 0, 0, blob(k.data(), 32)); }

    Context* ctx_;
    executor_t& executor_;
    bool initialized_ = false;
    size_t num_buf_elements_ = 0, num_buf_bytes_ = 0;
    std::deque<size_t> free_list_;
    buffer_t buffer_base_, tmp_buf_;
    hip::buffer_binding bind_compute2_, bind_compute3_;
    bool record_ = false;
    std::vector<lig_batch_op> ops_;
    std::vector<uint8_t> data_;
};

}  // namespace ligero
