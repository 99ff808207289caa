// prover_common.hpp -- host-side helpers shared by the prover, the sharded prover and the verifier: transcript hashing
// and samplers (OpenSSL), column sampling, Merkle decommitment, the protobuf envelope writer, the row plan of a job and
// the batch-program interpreter.  Everything here is internal to liblig_hip.so.
#pragma once
#include <openssl/evp.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "ctx_internal.hpp"
#include "fr29.hpp"
#include "host_field.hpp"

namespace H = lig::host;

namespace lig {
void launch_rng_fill_rows(hipStream_t s, const uint32_t* rk60_dev, uint64_t first, fr* out, size_t rows, uint32_t per_row,
                          size_t row_stride, uint32_t col_off, uint32_t elem_stride, uint64_t stream_stride);
void launch_rlc_rows29(hipStream_t s, const fr* U, size_t urs, uint32_t ues, const fr* Rn, size_t rrs, size_t rows, uint32_t count,
                       const f29s* rc_dev, fr* code, fr* lin, fr* part_code, fr* part_lin, uint32_t group_rows);
void launch_quad_rows29(hipStream_t s, const fr* U, size_t urs, uint32_t ues, uint32_t count, const uint32_t* triples_dev,
                        const f29s* rq2, const f29s* rq1, size_t n_triples, fr* quad, fr* part = nullptr, size_t part_elems = 0);
void launch_quad_rows29_view(hipStream_t s, CwView cw, uint32_t count, const uint32_t* triples_dev, const f29s* rq2, const f29s* rq1,
                             size_t n_triples, fr* quad, fr* part = nullptr, size_t part_elems = 0);
void launch_sum_elems(hipStream_t s, const fr* in, uint32_t count, uint32_t stride, fr* out, fr* neg_out);
void launch_rlc_accumulate29(hipStream_t s, const fr* U, size_t urs, uint32_t ues, const fr* Rn, size_t rrs, size_t rows, uint32_t count,
                             const f29s* rc_dev, fr* part_code, fr* part_lin, uint32_t group_rows);
void launch_rng_fill_rows_dense(hipStream_t s, const uint32_t* rk60_dev, uint64_t first, fr* out, size_t rows, uint32_t per_row, uint32_t k);
void launch_rand_rlc(hipStream_t s, const uint32_t* rk60_dev, uint64_t first, fr* out, const fr* msgs, size_t rows, uint32_t per_row, uint32_t k,
                     const f29s* rc_dev, uint32_t group_rows, fr* code_part, fr* lin_part);
void launch_lin_interleave(hipStream_t s, fr* out, const fr* accH, const fr* accC, uint32_t k);
void launch_rlc_combine(hipStream_t s, fr* acc, const fr* part, uint32_t groups, uint32_t count);
void launch_copy_from_host(hipStream_t s, uint8_t* dst_dev, const uint8_t* src_mapped, size_t bytes);
}  // namespace lig

// ---- host rows -> device through the library's uploader thread (prover.hip; one thread per device, shared by every trace and
// shard): no copy, event or barrier packet of such a transfer sits in a HIP queue of a proof.  A job copies `bytes`, waits for
// the copy ON THE HOST, then publishes `seq` in *flag (pinned host memory; streams wait for it with hipStreamWaitValue32).
// `wait` (optional): the copy may only start once *wait >= wait_val -- a word in pinned host memory that a stream of the proof writes
// (hipStreamWriteValue32) when it is done with the destination buffer (the double-buffered randomness rows of stage 2)
// `segs` (optional, instead of dst / src / bytes): several pieces under one arrival word; a piece without a source is zero-filled on
// the device (rows a sparse randomness matrix does not ship)
struct UploadSeg { uint8_t* dst; const uint8_t* src; size_t bytes; };
struct UploadJob { uint8_t* dst; const uint8_t* src; size_t bytes; volatile uint32_t* flag; uint32_t seq; std::atomic<int>* failed;
                   const volatile uint32_t* wait = nullptr; uint32_t wait_val = 0; const std::atomic<int>* abort = nullptr;
                   int prio = 0;        // 1: a proof is waiting for it NOW (randomness rows) -- ahead of the prefetch of a next trace's witness rows
                   std::shared_ptr<std::vector<UploadSeg>> segs; };
extern "C" bool lig_internal_uploader_available(lig_ctx* c);          // false: no stream memory operations on this device (callers fall back to stream copies)
extern "C" void lig_internal_uploader_submit(int device, const std::vector<UploadJob>& jobs, std::atomic<int>* pending);   // *pending += jobs, -1 per finished job
std::string lig_internal_uploader_state(int device);                  // diagnostics: queue length, the copy in progress and for how long

// (outside the anonymous namespace: these types appear in functions shared between translation units)
// kind: 0 linear, 1 x, 2 y, 3 z of the synthetic stream; >= 4: rows committed by the batch program (RK_* below)
struct RowDesc { uint8_t kind; uint32_t data; };
// `count` consecutive rows from `first` whose k-l pads are element pos, pos + (k-l), ... of the encoding stream
struct PadRun { size_t first, count; uint64_t pos; };

namespace {

using clk = std::chrono::steady_clock;
double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

// ---------------------------------------------------------------- host crypto (OpenSSL, as the reference: hash.hpp:153-214, csprng.hpp)
struct Sha256 {
    EVP_MD_CTX* c;
    Sha256() : c(EVP_MD_CTX_new()) { EVP_DigestInit_ex(c, EVP_sha256(), nullptr); }
    ~Sha256() { EVP_MD_CTX_free(c); }
    Sha256& add(const void* p, size_t n) { EVP_DigestUpdate(c, p, n); return *this; }
    void finish(uint8_t out[32]) { unsigned int l = 32; EVP_DigestFinal_ex(c, out, &l); }
};
// keystream element e of the AES-256-CTR stream (IV = 0): blocks 2e, 2e+1  -> field element (finite_field_gmp.hpp:66-78)
struct FieldStream {
    EVP_CIPHER_CTX* c;
    explicit FieldStream(const uint8_t key[32]) : c(EVP_CIPHER_CTX_new()) {
        const uint8_t iv[16] = {0};
        EVP_EncryptInit_ex(c, EVP_aes_256_ctr(), nullptr, key, iv);
    }
    ~FieldStream() { EVP_CIPHER_CTX_free(c); }
    // sequential draws (the engine is only ever read front to back on the host)
    void next(size_t count, std::vector<H::Fr>& out) {
        std::vector<uint8_t> zero(32 * count, 0), ks(32 * count);
        int len = 0;
        EVP_EncryptUpdate(c, ks.data(), &len, zero.data(), (int)zero.size());
        out.resize(count);
        for (size_t i = 0; i < count; i++) {
            H::Fr v;
            std::memcpy(v.v, ks.data() + 32 * i, 32);
            for (int w = 0; w < 4; w++) v.v[w] = (v.v[w] >> 2) | (w < 3 ? (v.v[w + 1] << 62) : 0);
            if (H::geq(v, H::P)) v = H::sub_nored(v, H::P);
            out[i] = v;
        }
    }
};

// hash_random_engine<sha256> (include/zkp/random.hpp:87-146)
struct HashRandomEngine {
    uint8_t seed[32], buf[32];
    uint64_t state = 0;
    int off = -1;
    explicit HashRandomEngine(const uint8_t s[32]) { std::memcpy(seed, s, 32); }
    uint8_t operator()() {
        if (off < 0) {
            Sha256 h;
            if (state) h.add(seed, 32);             // the seed is absorbed only after the first flush
            uint8_t le[8];
            for (int i = 0; i < 8; i++) le[i] = (uint8_t)(state >> (8 * i));
            h.add(le, 8).finish(buf);
            state++;
            off = 31;
        }
        return buf[off--];
    }
};
// boost::random::detail::generate_uniform_int(Engine&, T min, T max, boost::true_type) of
// boost/random/uniform_int_distribution.hpp (the integer-engine overload; unchanged since the header appeared in Boost 1.47,
// read from memory against 1.74 / 1.83 -- Boost is neither vendored upstream, CMakeLists.txt:67-69, nor in this image, so the
// sample indices stay "parity unpinned") for the 8-bit engine above (brange = 255, bmin = 0) and range_type = uint64_t
// (Distance = ptrdiff_t, include/util/portable_sample.hpp:26).  Tags B1..B4 name the statements of the Boost function in
// order; oracle/hash.c (boost_uniform) quotes them one by one.  Returns a value in [0, range].
uint64_t uniform_u64(HashRandomEngine& e, uint64_t range) {
    if (range == 0) return 0;                         // B1  if(range == 0) return min_value;
    if (range == 255) return e();                     // B2  else if(brange == range) return eng() - bmin + min_value;
    if (range < 255) {                                // B4  else (brange > range):
        const uint64_t bucket = 256 / (range + 1);    // B4a bucket_size: base_unsigned = uint8_t, brange == max -> 255/(range+1), +1 if 255%(range+1) == range: the same number
        for (;;) { const uint64_t r = e() / bucket; if (r <= range) return r; }      // B4b
    }
    for (;;) {                                        // B3  else if(brange < range) for(;;)
        const uint64_t limit = (range + 1) / 256;     // B3a (range < 2^64 - 1 always here: the max(range_type) special case cannot occur)
        uint64_t result = 0, mult = 1;                // B3b
        bool exact = false;
        while (mult <= limit) {                       // B3c
            result += (uint64_t)e() * mult;
            if (mult * 255 == range - mult + 1) { exact = true; break; }     // B3d: range+1 is a power of 256 -> return result
            mult *= 256;                              // B3e
        }
        if (exact) return result;
        uint64_t inc = uniform_u64(e, range / mult);  // B3f  generate_uniform_int(eng, 0, range/mult, true_type())
        if (UINT64_MAX / mult < inc) continue;        // B3g
        inc *= mult;                                  // B3h
        result += inc;
        if (result < inc || result > range) continue; // B3i, B3j
        return result;                                // B3k
    }
}
// portable_sample + sort (include/util/portable_sample.hpp:15-33, src/webgpu_prover.cpp:343-351)
std::vector<uint32_t> sample_columns(const uint8_t seed[32], uint32_t n, uint32_t t) {
    HashRandomEngine e(seed);
    std::vector<uint32_t> a(n), out;
    for (uint32_t i = 0; i < n; i++) a[i] = i;
    if (t > n) t = n;
    for (uint32_t i = 0; i < t; i++) {
        const uint64_t j = i + uniform_u64(e, (uint64_t)(n - 1) - i);
        std::swap(a[i], a[j]);
        out.push_back(a[i]);
    }
    std::sort(out.begin(), out.end());
    return out;
}
// merkle_tree::decommit + canonical sibling order (merkle_tree.hpp:155-215, proof_serializer.hpp:82-117)
std::vector<uint8_t> decommit(const uint8_t* nodes, size_t P, const std::vector<uint32_t>& idx) {
    std::vector<uint8_t> sib;
    std::vector<uint8_t> known(P, 0), upper(P, 0);
    for (uint32_t i : idx) known[i] = 1;
    size_t start = P - 1, end = 2 * P - 1;
    while (start > 0) {
        std::fill(upper.begin(), upper.end(), 0);
        for (size_t i = start; i < end; i += 2) {
            const size_t ll = i - start;
            const bool kl = known[ll], kr = known[ll + 1];
            if (kl && kr) upper[ll / 2] = 1;
            else if (kr) { sib.insert(sib.end(), nodes + 32 * i, nodes + 32 * i + 32); upper[ll / 2] = 1; }
            else if (kl) { sib.insert(sib.end(), nodes + 32 * (i + 1), nodes + 32 * (i + 1) + 32); upper[ll / 2] = 1; }
        }
        known.swap(upper);
        start = (start - 1) / 2; end = (end - 1) / 2;
    }
    return sib;
}

// ---------------------------------------------------------------- protobuf wire writer (proto/ligero_proof.proto, proto/common.proto)
struct Pb {
    std::vector<uint8_t> b;
    void var(uint64_t v) { do { uint8_t c = v & 0x7f; v >>= 7; if (v) c |= 0x80; b.push_back(c); } while (v); }
    void tag(uint32_t f, uint32_t wt) { var(((uint64_t)f << 3) | wt); }
    void u(uint32_t f, uint64_t v) { if (v) { tag(f, 0); var(v); } }
    void bytes(uint32_t f, const void* p, size_t n) { tag(f, 2); var(n); const uint8_t* q = (const uint8_t*)p; b.insert(b.end(), q, q + n); }
    void msg(uint32_t f, const Pb& m) { bytes(f, m.b.data(), m.b.size()); }
};
size_t varlen(uint64_t v) { size_t n = 1; while (v > 0x7f) { v >>= 7; n++; } return n; }

// serialize_proof (include/zkp/proof_serializer.hpp:166-191) + metadata (src/webgpu_prover.cpp:410-427), written
// straight into a caller-provided (pinned) buffer.  The four FixedU32Vector payloads are raw little-endian limb
// bytes: the three accumulators are copied in, the position of the sample payload is returned so that the
// device->host copy of the opened columns lands directly inside the envelope.
// (enc3 == nullptr: the framing is written, the three accumulators are NOT copied -- vec_off says where they belong, so that
// the caller can start the download of the opened columns first and copy them while it runs)
struct EnvelopeLayout { size_t total = 0, samples_off = 0, vec_off[3] = {0, 0, 0}; };
EnvelopeLayout write_envelope(uint8_t* dst, size_t cap, const char* version, const uint8_t program_hash[32], int64_t generated_at,
                              uint32_t k, uint32_t n, uint32_t t, const uint8_t root[32], const std::vector<uint8_t>& siblings,
                              const std::vector<uint32_t>& idx, const uint8_t* enc3, size_t sample_bytes) {
    auto digest = [](const uint8_t d[32]) { Pb m; m.bytes(1, d, 32); return m; };
    Pb meta;
    if (version[0]) meta.bytes(1, version, std::strlen(version));
    meta.u(2, 1); meta.u(3, 1);
    meta.msg(4, digest(program_hash));
    { Pb ts; ts.u(1, (uint64_t)generated_at); meta.msg(5, ts); }
    meta.u(6, k); meta.u(7, n); meta.u(8, t); meta.u(9, 128);
    Pb md;
    md.u(1, 1);
    md.msg(2, digest(root));
    for (size_t i = 0; i < siblings.size() / 32; i++) md.msg(3, digest(siblings.data() + 32 * i));
    if (!idx.empty()) { Pb pk; for (uint32_t v : idx) pk.var(v); md.bytes(4, pk.b.data(), pk.b.size()); }
    const size_t vec = (size_t)n * 32;
    auto fixed_len = [](size_t nb) { return nb ? 1 + varlen(nb) + nb : 0; };
    const size_t body_len = 1 + varlen(md.b.size()) + md.b.size() + 3 * (1 + varlen(fixed_len(vec)) + fixed_len(vec)) + 1 +
                            varlen(fixed_len(sample_bytes)) + fixed_len(sample_bytes);
    Pb head;
    head.msg(1, meta);
    head.tag(2, 2); head.var(body_len);
    head.msg(1, md);
    EnvelopeLayout L;
    size_t pos = 0;
    auto put = [&](const void* p, size_t nb) { if (pos + nb <= cap) std::memcpy(dst + pos, p, nb); pos += nb; };
    put(head.b.data(), head.b.size());
    for (uint32_t f = 2; f <= 5; f++) {
        const size_t nb = f < 5 ? vec : sample_bytes;
        Pb h;
        h.tag(f, 2); h.var(fixed_len(nb));
        if (nb) { h.tag(1, 2); h.var(nb); }
        put(h.b.data(), h.b.size());
        if (f < 5) { L.vec_off[f - 2] = pos; if (enc3) put(enc3 + (size_t)(f - 2) * vec, nb); else pos += nb; }
        else { L.samples_off = pos; pos += nb; }
    }
    L.total = pos;
    return L;
}

// kind: 0 linear, 1 x, 2 y, 3 z of the synthetic stream; rows committed by the batch program (lig_hip.h, lig_batch_op):
// 4 init, 5 bit, 6 / 7 the two rows of an equality, 8 / 9 / 10 the x, y, z of a batch product or quotient

enum : uint8_t { RK_INIT = 4, RK_BIT = 5, RK_EQX = 6, RK_EQY = 7, RK_BQX = 8, RK_BQY = 9, RK_BQZ = 10 };
inline bool has_code_check(uint8_t kind) { return kind != RK_EQX && kind != RK_EQY; }      // nonbatch_context.hpp:811-825

// Commit order: rows of the batch program in program order, then witness_manager's order for the synthetic stream
// (witness_manager.hpp:497-503): full linear rows, full quadratic triples, partial linear row, partial quadratic triple.
// Returns false for a malformed batch program.  n_init = rows that draw padding from the encoding stream at init time.
bool plan_rows(const lig_synth_job& job, uint32_t l, std::vector<RowDesc>& rows, size_t& n_init) {
    rows.clear();
    n_init = 0;
    if (job.n_batch_ops && !job.batch_ops) return false;
    for (uint64_t i = 0; i < job.n_batch_ops; i++) {
        const lig_batch_op& o = job.batch_ops[i];
        if (o.op >= LIG_BOP_COUNT || o.out >= 512 || o.x >= 512 || o.y >= 512) return false;
        const uint64_t need = o.op == LIG_BOP_SET ? 32ull * o.len : o.op == LIG_BOP_BIT_DECOMPOSE ? 4ull * o.len :
                              (o.op == LIG_BOP_SET_SCALAR || (o.op >= LIG_BOP_ADD_CONST && o.op <= LIG_BOP_MONTMUL_CONST)) ? 32 : 0;
        if (need && (!job.batch_data || o.data_off > job.batch_data_bytes || need > job.batch_data_bytes - o.data_off)) return false;
        if (o.op == LIG_BOP_SET && o.len > l) return false;
        if (o.op == LIG_BOP_BIT_DECOMPOSE) {
            if (o.len > 256) return false;
            for (uint32_t b = 0; b < o.len; b++) {           // every output slot inside the slab and different from the source
                uint32_t slot;
                std::memcpy(&slot, job.batch_data + o.data_off + 4ull * b, 4);
                if (slot >= 512 || slot == o.x) return false;
            }
        }
        switch (o.op) {
            case LIG_BOP_SET: case LIG_BOP_SET_SCALAR: rows.push_back({RK_INIT, 0}); n_init++; break;
            case LIG_BOP_COPY: case LIG_BOP_ASSERT_EQUAL: rows.push_back({RK_EQX, 0}); rows.push_back({RK_EQY, 0}); break;
            case LIG_BOP_MUL: case LIG_BOP_DIV: rows.push_back({RK_BQX, 0}); rows.push_back({RK_BQY, 0}); rows.push_back({RK_BQZ, 0}); break;
            case LIG_BOP_BIT_DECOMPOSE: for (uint32_t b = 0; b < o.len; b++) rows.push_back({RK_BIT, 0}); break;
            default: break;
        }
    }
    const size_t lf = job.n_linear / l, lp = job.n_linear % l, qf = job.n_quad / l, qp = job.n_quad % l;
    for (size_t i = 0; i < lf; i++) rows.push_back({0, l});
    for (size_t i = 0; i < qf; i++) for (uint8_t q = 1; q <= 3; q++) rows.push_back({q, l});
    if (lp) rows.push_back({0, (uint32_t)lp});
    if (qp) for (uint8_t q = 1; q <= 3; q++) rows.push_back({q, (uint32_t)qp});
    return true;
}
// quadratic-test terms in hook order (one quadratic-stream draw each): (x, y, z) row indices; y = 0xFFFFFFFF marks the
// equality term r * (x - z) (prover_kernels.hip k_quad_rows)
std::vector<uint32_t> quad_terms(const std::vector<RowDesc>& rows) {
    std::vector<uint32_t> t;
    for (size_t r = 0; r < rows.size(); r++) {
        const uint8_t kd = rows[r].kind;
        if (kd == 3 || kd == RK_BQZ) { t.push_back((uint32_t)r - 2); t.push_back((uint32_t)r - 1); t.push_back((uint32_t)r); }
        else if (kd == RK_BIT) { t.push_back((uint32_t)r); t.push_back((uint32_t)r); t.push_back((uint32_t)r); }
        else if (kd == RK_EQY) { t.push_back((uint32_t)r - 1); t.push_back(0xFFFFFFFFu); t.push_back((uint32_t)r); }
    }
    return t;
}

// Row-chunk schedule [begin, end) pairs.  Chunks are `big` rows except that the exposed end of a two-stream pipeline
// is kept short: `head` rows first (stage 2: the encode stream waits for the first randomness rows) and/or a short
// last chunk of `tail` rows (stage 1: the column hash of the last chunk runs after the last encode).
std::vector<std::pair<size_t, size_t>> chunk_schedule(size_t R, size_t big, size_t head, size_t tail) {
    std::vector<std::pair<size_t, size_t>> out;
    size_t b = 0;
    if (head && R > head + tail) { out.push_back({0, head}); b = head; }
    const size_t stop = (tail && R > b + tail) ? R - tail : R;
    while (b < stop) { const size_t e = std::min(stop, b + big); out.push_back({b, e}); b = e; }
    if (b < R) out.push_back({b, R});
    return out;
}

lig::f29s to_f29s_host(const H::Fr& plain, const H::Fr& scale) {
    const H::Fr m = H::mul(plain, scale);
    lig::f29s o;
    std::memset(&o, 0, sizeof o);
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, w = bit >> 6, sh = bit & 63;
        uint64_t v = m.v[w] >> sh;
        if (sh > 35 && w < 3) v |= m.v[w + 1] << (64 - sh);
        o.v[i] = (uint32_t)(i < 8 ? (v & 0x1FFFFFFFull) : v);
    }
    return o;
}
const H::Fr R261 = {{0x2fd4e1568fffff57ull, 0x75bba827a494b01aull, 0x5301fa84819caa80ull, 0x0dc83629563d4475ull}};   // 2^261 mod p

}  // namespace

#define TRY(x) do { int rc__ = (x); if (rc__ != LIG_OK) return rc__; } while (0)


// launch granularity shared by the prover, the sharded prover and the verifier
struct lig_tune {
#ifndef LIG_CHUNK
#define LIG_CHUNK 512
#endif
    static constexpr size_t CHUNK = LIG_CHUNK;       // rows per encode / hash / accumulate launch group
    static constexpr uint32_t GROUP = 64;      // rows per lazily accumulated group (n-column passes; k-column passes use GROUP / 4)
#ifndef LIG_DOT_GROUP
#define LIG_DOT_GROUP 8
#endif
    static constexpr uint32_t DOT_GROUP = LIG_DOT_GROUP;   // rows per group of the fused coset-2 encode + dot (lig_internal_encode_dot): 64 groups x 1024 threads per chunk
};

// the batch program of a job on the device: committed rows are written to rows_out in program order (prover.hip)
int lig_run_batch_program(lig_ctx* c, const lig_synth_job& job, fr* rows_out);
// witness values of the synthetic stream rows [first, rows.size()) (one draw of the witness_key stream per data slot of every
// linear / x / y row in commit order, z = x*y); row r is written to msgs + r*k.  Enqueued on the context stream.
int lig_internal_synth_witness(lig_ctx* c, const uint8_t witness_key[32], const std::vector<RowDesc>& rows, size_t first, fr* msgs);
