// Drives ligero::hip_context exactly like nonbatch_stage1_context does for linear rows
// (include/zkp/nonbatch_context.hpp:409-471,555-558): make_codeword_buffer, bind_ntt, bind_sha256_*,
// write_buffer_clear -> encode_ntt_device -> sha256_digest_update per row, then sha256_digest_final +
// copy_to_host.  Prints the digests' SHA-independent checksum; tests/test_gpu_context_cpp.py compares it with the
// oracle.  Build: g++ -std=c++17 -Iinclude tests/cpp/stage1_rows.cpp -L ligero-prover_amd -llig_hip
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lig_hip_context.hpp"

template <typename Executor>
struct mini_stage1 {
    using buffer_t = typename Executor::buffer_type;
    explicit mini_stage1(Executor& exe) : executor_(exe) {
        executor_.sha256_init(executor_.encoding_size());
        device_x_ = exe.make_codeword_buffer();
        sha256_context_ = exe.make_device_buffer(executor_.encoding_size() * sizeof(typename Executor::sha256_context));
        sha256_digest_ = exe.make_device_buffer(executor_.encoding_size() * 32);
        bind_ntt_x_ = exe.bind_ntt(device_x_);
        bind_sha256_ctx_ = exe.bind_sha256_context(sha256_context_, sha256_digest_);
        bind_sha256_x_ = exe.bind_sha256_buffer(device_x_);
        executor_.sha256_digest_init(bind_sha256_ctx_);
    }
    void linear_callback(const std::vector<uint64_t>& limbs) {
        executor_.write_buffer_clear(device_x_, limbs.data(), limbs.size());
        executor_.encode_ntt_device(bind_ntt_x_);
        executor_.sha256_digest_update(bind_sha256_ctx_, bind_sha256_x_);
    }
    std::vector<uint8_t> flush_digests() {
        executor_.sha256_digest_final(bind_sha256_ctx_);
        return executor_.template copy_to_host<uint8_t>(sha256_digest_);
    }
    Executor& executor_;
    buffer_t device_x_, sha256_context_, sha256_digest_;
    ligero::hip::buffer_binding bind_ntt_x_, bind_sha256_ctx_, bind_sha256_x_;
};

int main(int argc, char** argv) {
    const size_t k = 512, l = 320, n = 2048, rows = argc > 1 ? (size_t)std::atoi(argv[1]) : 3;
    ligero::hip_context executor;
    executor.webgpu_init(k, "unused-shader-path");
    executor.ntt_init(l, k, n, 0, 0, 0, 0, 0);
    mini_stage1<ligero::hip_context> ctx(executor);
    for (size_t r = 0; r < rows; r++) {
        std::vector<uint64_t> limbs(k * 4, 0);
        for (size_t i = 0; i < k; i++) limbs[4 * i] = 1000003ull * (r + 1) + i;     // small canonical values
        ctx.linear_callback(limbs);
    }
    std::vector<uint8_t> d = ctx.flush_digests();
    executor.device_synchronize();
    for (size_t i = 0; i < d.size(); i++) std::printf("%02x", d[i]);
    std::printf("\n");
    return 0;
}
