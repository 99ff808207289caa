// lig_hip_row_batcher.hpp -- the row-batching shim between the reference's per-row stage callbacks and the batched
// caller-rows prover entry of liblig_hip.so (lig_rows_*, include/lig_hip.h).
//
// The reference's stage contexts push ONE row through the executor per callback and run the guest three times
// (include/zkp/nonbatch_context.hpp: stage 1 :445-553, stage 2 :654-850, stage 3 :924-1047; src/webgpu_prover.cpp:266,305,408).
// `ligero::hip_row_batcher` offers the same callbacks -- linear_callback / quadratic_callback / mask_callback /
// on_batch_init / on_batch_bit / on_batch_equal / on_batch_quadratic, called by witness_manager
// (include/zkp/backend/witness_manager.hpp:200-321) and vbn254fr_module (include/host_modules/vbn254fr.hpp) -- but only
// RECORDS the rows; the GPU sees them in one batch per stage:
//
//     hip_row_batcher b(ctx, meta);                       // meta: encoding seed, program hash, timestamp, public arguments
//     run_program(..., b)            pass 1               // callbacks record the message rows (pads included, as the
//     root, seed1 = b.commit();                           //   witness_manager forms them) -> stage 1 on the GPU
//     init_witness_random(seed1); run_program(..., b)     // pass 2: the same callbacks now also carry the per-witness
//     proof = b.prove(linear_sums);                       //   randomness rows -> stages 2 + 3 on the GPU
//
// A third run of the guest (the reference's stage 3) is not needed: the codewords stayed resident, the opened columns
// are gathered from them.  Rows are `k * 4` little-endian u64 limbs, exactly what mpz_vector::export_limbs produces
// (include/util/mpz_vector.hpp:108-127; nonbatch_context.hpp:447); batch rows are device rows of the vbn254fr slab.
// The three mask rows arrive through mask_callback upstream; the library forms the identical rows itself from the
// encoding seed (same stream, same position: after the pads of all rows), so mask_callback only checks sizes.
//
// Data path (round 4): rows are written ONCE, into page-locked staging (lig_host_alloc; `next_slot()` lets a driver export a
// row's limbs straight into it), the staging is kept for the next pass and the next proof (`reset()`), `commit()` uploads
// from it chunk by chunk under the encodes, and in pass 2 every 256 completed randomness rows are handed to the library at
// once (lig_rows_push_rands) -- their upload runs while the guest is still producing the next ones, `prove()` only waits for
// the tail.  tests/cpp/row_batcher_bench.cpp measures it (profiles/r04_row_batcher_bench.md).
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "lig_hip.h"

namespace ligero {

struct hip_proof_meta {
    uint8_t encoding_seed[32] = {0};          // src/webgpu_prover.cpp:239-245
    uint8_t program_hash[32] = {0};           // :222-223
    int64_t generated_at = 0;                 // :422-427
    std::string version = "1.5.0";
    std::vector<std::vector<uint8_t>> public_args;   // input_args entries of the PUBLIC arguments after arg0 (:110-168)
    // Ship narrow rows (lig_rows_job.elem_bytes): a linear / x / y / z row all of whose data slots fit 8 bytes is uploaded as
    // l x 8 bytes instead of k x 32, and its k - l pad slots are drawn by the library.  Sound only because those pads ARE the
    // encoding stream at the row's position (witness_manager::pad_encoding_random, witness_manager.hpp:323-336, keyed by
    // encoding_seed): the library draws the same elements the row arrived with.
    bool narrow_rows = false;
    // Rows the guest is expected to commit (0: unknown).  The staging is page-locked memory that grows geometrically when the
    // guest outruns it; a driver that proves the same program again (or knows its size) saves the re-allocations.
    size_t expected_rows = 0;
};

// Page-locked host staging for rows of k x 4 u64 (lig_host_alloc): the callbacks write rows straight into it -- no pageable
// std::vector, no second copy at commit() -- and it is kept for the next pass and the next proof (reset()).
class hip_row_staging {
public:
    hip_row_staging(lig_ctx* ctx, size_t row_words) : ctx_(ctx), words_(row_words) {}
    hip_row_staging(const hip_row_staging&) = delete;
    hip_row_staging& operator=(const hip_row_staging&) = delete;
    ~hip_row_staging() { if (base_) (void)lig_host_free(ctx_, base_); }
    uint64_t* row(size_t r) { return base_ + r * words_; }
    const uint64_t* data() const { return base_; }
    size_t capacity() const { return cap_; }
    // room for `rows` rows; the first `keep` rows survive a re-allocation
    void reserve(size_t rows, size_t keep) {
        if (rows <= cap_) return;
        size_t want = cap_ ? cap_ : 64;
        while (want < rows) want += want < 4096 ? want : 4096;         // double up to 4096 rows (1 GiB at k = 8192), then 1 GiB steps
        void* p = nullptr;
        if (lig_host_alloc(ctx_, want * words_ * 8, &p) != LIG_OK) throw std::runtime_error(std::string("lig_host_alloc: ") + lig_last_error(ctx_));
        if (keep) std::memcpy(p, base_, keep * words_ * 8);
        if (base_) (void)lig_host_free(ctx_, base_);
        base_ = static_cast<uint64_t*>(p);
        cap_ = want;
    }
private:
    lig_ctx* ctx_;
    size_t words_, cap_ = 0;
    uint64_t* base_ = nullptr;
};

class hip_row_batcher {
public:
    hip_row_batcher(lig_ctx* ctx, hip_proof_meta meta)
        : ctx_(ctx), meta_(std::move(meta)), k_(ctx ? lig_padding_size(ctx) : 0), l_(ctx ? lig_message_size(ctx) : 0),
          rows_(ctx, (size_t)k_ * 4), rands_(ctx, (size_t)k_ * 4) {
        if (!ctx_) throw std::invalid_argument("hip_row_batcher: null context");
        if (meta_.expected_rows) rows_.reserve(meta_.expected_rows, 0);
    }
    hip_row_batcher(const hip_row_batcher&) = delete;
    hip_row_batcher& operator=(const hip_row_batcher&) = delete;
    ~hip_row_batcher() { if (trace_) lig_trace_destroy(trace_); if (shard_) lig_shard_destroy(shard_); }

    // ONE trace over the GPUs of a node (configs[4]: the guest's rows on 8 GPUs): every rank runs the same guest, so every rank's
    // callbacks see all rows; after pass 1 the batcher keeps the rows of its own chunks (lig_shard_rows_plan), in pass 2 the
    // randomness rows of those.  Call before commit(); `comm` from lig_rccl_comm_create (or lig_ipc_comm_create) must outlive
    // the batcher.  Every rank's prove() returns the same envelope as an unsharded batcher would.
    void shard_over(uint32_t rank, uint32_t world, const lig_comm* comm) {
        if (pass_ != 1 || !comm || !world || rank >= world) throw std::invalid_argument("hip_row_batcher::shard_over");
        sharded_ = true; rank_ = rank; world_ = world; comm_ = *comm;
    }

    // ---- the callbacks of nonbatch_context_base (nonbatch_context.hpp:78-86).  `rand` rows are null in pass 1.
    void linear_callback(const uint64_t* val, const uint64_t* rand = nullptr) { row(LIG_ROW_LINEAR, val, rand); }
    void quadratic_callback(const uint64_t* x, const uint64_t* y, const uint64_t* z, const uint64_t* x_rand = nullptr,
                            const uint64_t* y_rand = nullptr, const uint64_t* z_rand = nullptr) {
        row(LIG_ROW_QX, x, x_rand); row(LIG_ROW_QY, y, y_rand); row(LIG_ROW_QZ, z, z_rand);
    }
    // The same callbacks without the copy: where the reference exports a row's limbs into its `limbs_` vector
    // (mpz_vector::export_limbs, nonbatch_context.hpp:447 / :657-663) an integrated driver exports them straight into the slot --
    // k x 4 u64 of page-locked staging -- and then commits the slot.  Pass 1: the slot of the next message row; pass 2: the slot
    // of the next row's randomness row (nullptr for a row this rank does not own when sharded: nothing to export).
    uint64_t* next_slot() {
        if (pass_ == 1) { rows_.reserve(kinds_.size() + 1, kinds_.size()); return rows_.row(kinds_.size()); }
        if (pass_ != 2 || next_ >= kinds_.size()) throw std::logic_error("hip_row_batcher::next_slot: no row expected");
        if (!sharded_) return rands_.row(n_present_);            // randomness rows are kept packed: only rows that have one take a slot
        const size_t slot = local_of_[next_];
        return slot == (size_t)-1 ? nullptr : rands_.row(slot);
    }
    // has_rand = false in pass 2: the row has no randomness row (nothing was exported into the slot; it stays free for the next row)
    void commit_slot(uint8_t kind, bool has_rand = true) { row(kind, pass_ == 1 ? rows_.row(kinds_.size()) : nullptr, nullptr, true, has_rand); }
    void mask_callback(size_t code_size, size_t linear_size, size_t quad_size) const {
        if (code_size != k_ || linear_size != 2 * (size_t)k_ || quad_size != 2 * (size_t)k_) throw std::invalid_argument("mask_callback: unexpected mask sizes");
    }
    // vbn254fr hooks (nonbatch_context.hpp:497-553, :782-850): device rows of k elements; they carry no linear-test randomness.
    // on_batch_init (:497-510) first draws params::sample_size = 192 elements from the encoding stream
    // (pad_encoding_random) and writes them INTO the variable at slot message_size, then commits the row -- the pad is part of
    // the variable from then on and flows into every batch row derived from it.  Every stage re-runs the guest with the
    // encoding engine re-seeded, so pass 2 writes the same 192 elements again.
    void on_batch_init(void* dev_x) { on_batch_init(dev_x, static_cast<uint8_t*>(dev_x) + (size_t)l_ * 32); }
    // `pad_dst`: where the 192 pad elements go.  The declared semantics of buffer_view::slice put them into x's own pad slots
    // (the overload above); upstream's buffer_view::slice_bytes as DEFINED (src/webgpu/buffer_view.cpp:91-95 swaps its parameters)
    // puts them at byte l * 32 of the whole variable slab, i.e. into variable 0 -- hip_context's upstream_slice_compat mode hands
    // that address in.
    void on_batch_init(void* dev_x, void* pad_dst) {
        if (pass_ != 1 && pass_ != 2) throw std::logic_error("hip_row_batcher: callback after prove");
        if (k_ - l_ != init_pad) throw std::invalid_argument("hip_row_batcher::on_batch_init: k - l must be params::sample_size (192)");
        check(lig_rng_fill(ctx_, meta_.encoding_seed, enc_pos_, pad_dst, init_pad), "lig_rng_fill(init pad)");
        enc_pos_ += init_pad;
        dev_row(LIG_ROW_INIT, dev_x);
    }
    void on_batch_bit(const void* dev_x) { dev_row(LIG_ROW_BIT, dev_x); }
    void on_batch_equal(const void* dev_x, const void* dev_y) { dev_row(LIG_ROW_EQX, dev_x); dev_row(LIG_ROW_EQY, dev_y); }
    void on_batch_quadratic(const void* dev_x, const void* dev_y, const void* dev_z) {
        dev_row(LIG_ROW_BQX, dev_x); dev_row(LIG_ROW_BQY, dev_y); dev_row(LIG_ROW_BQZ, dev_z);
    }

    // ---- end of pass 1: stage 1 on the GPU.  Returns the Merkle root and the stage-1 seed (= the key of the code /
    // linear / quadratic random engines of pass 2, nonbatch_context.hpp:105-112).
    void commit(uint8_t root[32], uint8_t stage1_seed[32]) {
        if (pass_ != 1) throw std::logic_error("hip_row_batcher::commit called twice");
        std::vector<uint8_t> args;
        std::vector<uint64_t> lens;
        for (const auto& a : meta_.public_args) { args.insert(args.end(), a.begin(), a.end()); lens.push_back(a.size()); }
        lig_rows_job job;
        std::memset(&job, 0, sizeof job);
        job.rows = kinds_.size();
        job.kinds = kinds_.data();
        job.msgs = rows_.data();
        job.msgs_on_device = 0;
        std::memcpy(job.encoding_seed, meta_.encoding_seed, 32);
        std::memcpy(job.program_hash, meta_.program_hash, 32);
        job.generated_at = meta_.generated_at;
        std::strncpy(job.version, meta_.version.c_str(), sizeof job.version - 1);
        job.public_args = args.empty() ? nullptr : args.data();
        job.public_arg_lens = lens.empty() ? nullptr : lens.data();
        job.n_public_args = lens.size();
        if (sharded_) { commit_sharded(job, root, stage1_seed); return; }
        std::vector<uint8_t> widths;
        if (meta_.narrow_rows) {
            // packed IN PLACE: a narrow row shrinks to l x 8 bytes, rows only ever move towards the front of the staging
            const size_t R = kinds_.size(), words = (size_t)k_ * 4;
            widths.assign(R ? R : 1, 32);
            bool any = false;
            for (size_t r = 0; r < R; r++) {
                if (kinds_[r] > LIG_ROW_QZ) continue;                       // batch rows are device rows of full width
                const uint64_t* rw = rows_.row(r);
                bool fits = true;
                for (uint32_t i = 0; i < l_ && fits; i++) fits = !(rw[4 * i + 1] | rw[4 * i + 2] | rw[4 * i + 3]);
                if (fits) { widths[r] = 8; any = true; }
            }
            if (any) {
                uint64_t* out = rows_.row(0);
                for (size_t r = 0; r < R; r++) {
                    const uint64_t* rw = rows_.row(r);
                    if (widths[r] == 8) {
                        kinds_[r] |= LIG_ROW_DRAW_PAD;
                        for (uint32_t i = 0; i < l_; i++) out[i] = rw[4 * i];
                        out += l_;
                    } else {
                        if (out != rw) std::memmove(out, rw, words * 8);
                        out += words;
                    }
                }
                job.elem_bytes = widths.data();
            }
        }
        if (trace_) {                                      // the next proof of the same program: every device buffer is reused
            if (same_shape(job)) check(lig_rows_restart(trace_, job.msgs, 0), "lig_rows_restart");
            else { lig_trace_destroy(trace_); trace_ = nullptr; }
        }
        if (!trace_) check(lig_rows_begin(ctx_, &job, &trace_), "lig_rows_begin");
        shape_kinds_ = kinds_; shape_widths_ = widths; shape_meta_ = meta_;
        check(lig_rows_commit(trace_, root, stage1_seed), "lig_rows_commit");
        for (auto& kd : kinds_) kd &= 0x7f;               // (pass 2 compares plain kinds)
        begin_pass2(kinds_.size());
    }

    // ---- end of pass 2: stages 2 + 3 on the GPU.  const_sum = the public constant of the linear test, 32 bytes little
    // endian (linear_sums(), src/webgpu_prover.cpp:307).  The returned bytes are the serialized LigeroProofEnvelope
    // (owned by the batcher, valid until it is destroyed or reset); `info` (optional) receives the prover's self-check.
    const uint8_t* prove(const uint8_t const_sum[32], size_t* proof_len, lig_proof_info* info = nullptr) {
        if (pass_ != 2) throw std::logic_error("hip_row_batcher::prove before commit");
        if (next_ != kinds_.size()) throw std::logic_error("hip_row_batcher::prove: pass 2 replayed " + std::to_string(next_) + " of " + std::to_string(kinds_.size()) + " rows");
        const uint8_t* proof = nullptr;
        lig_proof_info local;
        if (sharded_) check(lig_shard_rows_prove(shard_, rands_.data(), 0, const_sum, &proof, proof_len, info ? info : &local), "lig_shard_rows_prove");
        else {
            push_rands(kinds_.size());                     // the tail; everything else went out while the guest was running
            check(lig_rows_prove(trace_, nullptr, 0, const_sum, &proof, proof_len, info ? info : &local), "lig_rows_prove");
        }
        pass_ = 3;
        return proof;
    }
    // the next proof with this batcher: staging and (for the same row kinds) every device buffer of the trace are kept
    void reset(const hip_proof_meta* meta = nullptr) {
        if (pass_ == 2) throw std::logic_error("hip_row_batcher::reset between commit and prove");
        if (meta) meta_ = *meta;
        kinds_.clear();
        pass_ = 1; next_ = 0; enc_pos_ = 0; pushed_ = 0;
    }
    size_t rows() const { return kinds_.size(); }
    size_t local_rows() const { return sharded_ ? n_local_ : kinds_.size(); }

private:
    static constexpr size_t push_rows = 256;             // randomness rows per lig_rows_push_rands (64 MiB at k = 8192)

    bool same_shape(const lig_rows_job& job) const {           // lig_rows_restart keeps kinds, seeds and metadata of the trace
        if (kinds_ != shape_kinds_) return false;
        if (std::memcmp(meta_.encoding_seed, shape_meta_.encoding_seed, 32) || std::memcmp(meta_.program_hash, shape_meta_.program_hash, 32) ||
            meta_.generated_at != shape_meta_.generated_at || meta_.version != shape_meta_.version || meta_.public_args != shape_meta_.public_args) return false;
        const std::vector<uint8_t> w = job.elem_bytes ? std::vector<uint8_t>(job.elem_bytes, job.elem_bytes + kinds_.size()) : std::vector<uint8_t>();
        return w == shape_widths_ || (w.empty() && shape_widths_.empty());
    }
    void begin_pass2(size_t local_rows) {
        rands_.reserve(local_rows ? local_rows : 1, 0);
        present_.assign(kinds_.size(), 0);
        pass_ = 2; next_ = 0; enc_pos_ = 0; pushed_ = 0; n_present_ = 0; pushed_present_ = 0;  // the guest's second run starts the encoding stream over
    }
    // randomness rows [pushed_, upto) are complete: hand them to the library while the guest goes on.  Only the rows that HAVE a
    // randomness row are in the staging (packed) and go over the link; the library zero-fills the others on the device
    // (lig_rows_push_rands_sparse) -- batch rows never have one, quadratic rows often do not.
    void push_rands(size_t upto) {
        if (sharded_ || upto <= pushed_) return;
        check(lig_rows_push_rands_sparse(trace_, pushed_, upto - pushed_, present_.data() + pushed_, rands_.row(pushed_present_)), "lig_rows_push_rands_sparse");
        pushed_ = upto; pushed_present_ = n_present_;
    }
    void commit_sharded(lig_rows_job& job, uint8_t root[32], uint8_t stage1_seed[32]) {
        const size_t R = kinds_.size(), words = (size_t)k_ * 4;
        uint64_t rounds = 0;
        std::vector<uint64_t> b((size_t)world_ * ((R + 511) / 512 + 2) + 2);
        if (lig_shard_rows_plan(kinds_.data(), R, world_, &rounds, b.data(), b.size()) != LIG_OK) throw std::runtime_error("lig_shard_rows_plan failed");
        local_of_.assign(R, (size_t)-1);
        n_local_ = 0;
        // this rank's rows, compacted IN PLACE to the front of the staging (commit order is kept, rows only move forward)
        for (uint64_t g = rank_; g < rounds * world_; g += world_)
            for (uint64_t r = b[g]; r < b[g + 1]; r++) {
                if (n_local_ != r) std::memmove(rows_.row(n_local_), rows_.row(r), words * 8);
                local_of_[r] = n_local_++;
            }
        job.msgs = n_local_ ? rows_.data() : nullptr;
        if (shard_) { lig_shard_destroy(shard_); shard_ = nullptr; }
        check(lig_shard_rows_begin(ctx_, &job, rank_, world_, &comm_, &shard_), "lig_shard_rows_begin");
        check(lig_shard_rows_commit(shard_, root, stage1_seed), "lig_shard_rows_commit");
        begin_pass2(n_local_);
        std::memset(rands_.row(0), 0, (n_local_ ? n_local_ : 1) * words * 8);       // batch rows and rows without a callback keep zero rows
    }
    void check(int rc, const char* what) const {
        if (rc != LIG_OK) throw std::runtime_error(std::string(what) + ": " + lig_last_error(ctx_));
    }
    void row(uint8_t kind, const uint64_t* val, const uint64_t* rand, bool in_slot = false, bool slot_has_rand = true) {
        const size_t words = (size_t)k_ * 4;
        // witness_manager pads every linear row and every row of a quadratic triple with k - l stream elements when it forms
        // it (the rows arrive with their pads): the position of the next on_batch_init pad moves past them
        if (kind <= LIG_ROW_QZ) enc_pos_ += k_ - l_;
        if (pass_ == 1) {
            if (!val) throw std::invalid_argument("hip_row_batcher: null row");
            const size_t r = kinds_.size();
            if (!in_slot) { rows_.reserve(r + 1, r); std::memcpy(rows_.row(r), val, words * 8); }
            kinds_.push_back(kind);
        } else if (pass_ == 2) {
            // the guest is deterministic: pass 2 must replay the callbacks of pass 1 in the same order
            if (next_ >= kinds_.size() || kinds_[next_] != kind) throw std::logic_error("hip_row_batcher: pass 2 diverges from pass 1");
            if (sharded_) {                                                          // dense local matrix: only the rows of this rank's chunks are kept
                const size_t slot = local_of_[next_];
                if (slot != (size_t)-1 && !in_slot && rand) std::memcpy(rands_.row(slot), rand, words * 8);
            } else if (in_slot ? slot_has_rand : rand != nullptr) {                   // packed: this row takes the next slot
                if (!in_slot) std::memcpy(rands_.row(n_present_), rand, words * 8);
                present_[next_] = 1;
                n_present_++;
            }
            next_++;
            if (!sharded_ && next_ - pushed_ >= push_rows) push_rands(next_);
        } else throw std::logic_error("hip_row_batcher: callback after prove");
    }
    void dev_row(uint8_t kind, const void* dev) {
        if (pass_ == 1) {
            check(lig_sync(ctx_), "lig_sync");           // (lig_read is ordered on the context stream; kept explicit: the row must be final)
            const size_t r = kinds_.size();
            rows_.reserve(r + 1, r);
            check(lig_read(ctx_, rows_.row(r), dev, (size_t)k_ * 32), "lig_read(batch row)");
            row(kind, rows_.row(r), nullptr, true);
        } else row(kind, nullptr, nullptr);
    }

    static constexpr uint32_t init_pad = 192;            // params::sample_size (include/params.hpp:27)
    lig_ctx* ctx_;
    hip_proof_meta meta_, shape_meta_;
    uint32_t k_, l_;
    int pass_ = 1;
    size_t next_ = 0, pushed_ = 0, n_present_ = 0, pushed_present_ = 0;   // pass 2: rows replayed / pushed, rows with a randomness row seen / pushed
    std::vector<uint8_t> present_;
    uint64_t enc_pos_ = 0;                                // encoding-stream position (elements) of the next row's pad
    std::vector<uint8_t> kinds_, shape_kinds_, shape_widths_;
    hip_row_staging rows_, rands_;
    lig_trace* trace_ = nullptr;
    bool sharded_ = false;
    uint32_t rank_ = 0, world_ = 1;
    lig_comm comm_{};
    lig_shard* shard_ = nullptr;
    std::vector<size_t> local_of_;
    size_t n_local_ = 0;
};

// The verifier's counterpart (src/webgpu_verifier.cpp:263-452 with nonbatch_verifier_context, nonbatch_context.hpp:1081-1388):
// the verifier runs the guest as well, its callbacks deliver the public randomness rows (the "value" rows it sees are the
// 192 opened elements it pops from the proof, which the library reads from the envelope itself).  One pass suffices when the
// caller derives stage1_seed first: begin(proof) -> seed -> run_program with the callbacks below -> finish(linear_sums).
class hip_row_verifier {
public:
    hip_row_verifier(lig_ctx* ctx, hip_proof_meta meta) : ctx_(ctx), meta_(std::move(meta)), k_(lig_padding_size(ctx)) {
        if (!ctx_) throw std::invalid_argument("hip_row_verifier: null context");
    }
    hip_row_verifier(const hip_row_verifier&) = delete;
    hip_row_verifier& operator=(const hip_row_verifier&) = delete;
    ~hip_row_verifier() { lig_vtrace_destroy(vt_); }          // begin without finish (the guest threw)

    // the row kinds of the public constraint stream, in commit order (a dry run of the guest, or the prover's kinds)
    void expect_rows(const std::vector<uint8_t>& kinds) { kinds_ = kinds; }
    // parse the envelope, re-derive both seeds and the sample indices; false: malformed envelope / wrong indices (reject)
    bool begin(const uint8_t* proof, size_t len, uint8_t stage1_seed[32], lig_verify_info* info = nullptr) {
        std::vector<uint8_t> args;
        std::vector<uint64_t> lens;
        for (const auto& a : meta_.public_args) { args.insert(args.end(), a.begin(), a.end()); lens.push_back(a.size()); }
        lig_rows_job job;
        std::memset(&job, 0, sizeof job);
        job.rows = kinds_.size();
        job.kinds = kinds_.data();
        job.public_args = args.empty() ? nullptr : args.data();
        job.public_arg_lens = lens.empty() ? nullptr : lens.data();
        job.n_public_args = lens.size();
        lig_verify_info local;
        const int rc = lig_rows_verify_begin(ctx_, &job, proof, len, &vt_, stage1_seed, info ? info : &local);
        if (rc != LIG_OK) throw std::runtime_error(std::string("lig_rows_verify_begin: ") + lig_last_error(ctx_));
        rands_.assign(kinds_.size() * (size_t)k_ * 4, 0);
        next_ = 0;
        return vt_ != nullptr;
    }
    void linear_callback(const uint64_t* rand) { row(LIG_ROW_LINEAR, rand); }
    void quadratic_callback(const uint64_t* x_rand, const uint64_t* y_rand, const uint64_t* z_rand) { row(LIG_ROW_QX, x_rand); row(LIG_ROW_QY, y_rand); row(LIG_ROW_QZ, z_rand); }
    void on_batch_init() { row(LIG_ROW_INIT, nullptr); }
    void on_batch_bit() { row(LIG_ROW_BIT, nullptr); }
    void on_batch_equal() { row(LIG_ROW_EQX, nullptr); row(LIG_ROW_EQY, nullptr); }
    void on_batch_quadratic() { row(LIG_ROW_BQX, nullptr); row(LIG_ROW_BQY, nullptr); row(LIG_ROW_BQZ, nullptr); }
    // the seven predicates of webgpu_verifier.cpp:412-442; returns accept
    bool finish(const uint8_t const_sum[32], lig_verify_info* info = nullptr) {
        if (!vt_) throw std::logic_error("hip_row_verifier::finish without a successful begin");
        if (next_ != kinds_.size()) throw std::logic_error("hip_row_verifier::finish: fewer rows replayed than expected");
        lig_verify_info local;
        lig_verify_info* o = info ? info : &local;
        lig_vtrace* vt = vt_;
        vt_ = nullptr;                                        // finish frees the trace
        if (lig_rows_verify_finish(vt, rands_.data(), 0, const_sum, o) != LIG_OK) throw std::runtime_error(std::string("lig_rows_verify_finish: ") + lig_last_error(ctx_));
        return o->accept != 0;
    }

private:
    void row(uint8_t kind, const uint64_t* rand) {
        if (next_ >= kinds_.size() || (kinds_[next_] & 0x7f) != kind) throw std::logic_error("hip_row_verifier: the guest diverges from the expected row kinds");
        if (rand) std::memcpy(rands_.data() + next_ * (size_t)k_ * 4, rand, (size_t)k_ * 32);
        next_++;
    }
    lig_ctx* ctx_;
    hip_proof_meta meta_;
    uint32_t k_;
    size_t next_ = 0;
    std::vector<uint8_t> kinds_;
    std::vector<uint64_t> rands_;
    lig_vtrace* vt_ = nullptr;
};

}  // namespace ligero
