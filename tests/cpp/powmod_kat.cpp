// The executor-level powmod surface of ligero::hip_context driven like the reference's own fixture
// (tests/webgpu/test_powmod.cpp:21-48: bind_buffer(exp, coeff, out), set_base, powmod_kernel / powmod_add_kernel) for its
// generator case: base 7, exp[i] = i, coeff = 1, N = 8192, then powmod_add on top.  Prints out[i] as hex words; the
// Python test compares with pow(7, i, p).  Also checks the misuse behaviour: powmod before powmod_init throws logic_error.
#include <cstdio>
#include <cstring>
#include <vector>

#include "lig_hip_context.hpp"

int main() {
    const size_t N = 8192;
    ligero::hip_context ex;
    ex.webgpu_init(0, "");
    ex.ntt_init(320, 512, 2048, 0, 0, 0, 0, 0);
    auto exp = ex.make_device_buffer(N * 4), coeff = ex.make_device_buffer(N * 32), out = ex.make_device_buffer(N * 32);
    auto bind = ex.bind_powmod(exp, coeff, out);
    bool threw = false;
    try { ex.EltwisePowMod(bind); } catch (const std::logic_error&) { threw = true; }
    std::printf("misuse_throws %d\n", (int)threw);
    ex.powmod_init(32);
    std::vector<uint32_t> e(N);
    for (size_t i = 0; i < N; i++) e[i] = (uint32_t)i;
    std::vector<uint32_t> one(N * 8, 0);
    for (size_t i = 0; i < N; i++) one[8 * i] = 1;
    ex.write_buffer(exp, e.data(), e.size());
    ex.write_buffer(coeff, one.data(), one.size());
    ligero::hip::scalar seven{};
    seven[0] = 7;
    ex.powmod_set_base(seven);
    ex.EltwisePowMod(bind);
    ex.EltwisePowAddMod(bind);                       // out = 2 * 7^i
    auto host = ex.copy_to_host<uint32_t>(out);
    for (size_t i : {size_t(0), size_t(1), size_t(2), size_t(255), size_t(4099), N - 1}) {
        std::printf("%zu", i);
        for (int w = 7; w >= 0; w--) std::printf(" %08x", host[8 * i + w]);
        std::printf("\n");
    }
    return 0;
}
