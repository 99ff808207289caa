// make_boost_pin.cpp -- INGEST HOOK (round 6): pins the 192 sample indices to the REAL Boost.
// Runs the reference's own call sequence (src/webgpu_prover.cpp:343-351: hash_random_engine -> portable_sample -> sort) with the
// reference's own headers and boost::random::uniform_int_distribution<ptrdiff_t> (include/util/portable_sample.hpp:26-27).
// Boost is neither vendored upstream nor in the build image, so this file cannot be compiled there: on ANY machine with Boost headers and
// a checkout of ligero-prover v1.5.0 ($REF):
//   g++ -std=c++20 -O2 -D__EMSCRIPTEN__ -include memory -include limits -include bit -include stdexcept -include string -include cassert \
//       -include array -include tuple -include algorithm -I$REF/include tools/make_boost_pin.cpp -lcrypto -o /tmp/make_boost_pin
//   /tmp/make_boost_pin > tests/golden/boost_sampling.json
// tests/test_ref_pins.py::test_sample_indices_equal_real_boost then compares the oracle (lo_sample_indices) and the product
// (lig_sample_columns) with it; until the file exists that test skips and DESIGN.md section 5 keeps saying "parity unpinned".
#include <params.hpp>
#include <zkp/hash.hpp>
#include <zkp/random.hpp>
#include <util/portable_sample.hpp>
#include <boost/version.hpp>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <vector>
using namespace ligero::vm;
int main() {
    // n below / at / above the engine's 8-bit range (Boost's three branches), powers of 256 (the exact-multiple exit), the geometries in use
    const size_t ns[] = {1, 2, 192, 193, 255, 256, 257, 300, 2048, 65536, 65537, 32768, 131072, 16777216, 16777217};
    std::printf("{\"generator\": \"tools/make_boost_pin.cpp, Boost %d\", \"cases\": [", BOOST_VERSION);
    bool first = true;
    for (int s = 0; s < 3; s++)
        for (size_t n : ns) {
            params::hasher::digest seed;
            for (int i = 0; i < 32; i++) seed.data[i] = (uint8_t)(s == 0 ? 0 : s == 1 ? i : 255 - 7 * i);
            zkp::hash_random_engine<params::hasher> engine(seed);
            std::vector<size_t> indexes(n), sample_index;
            std::iota(indexes.begin(), indexes.end(), 0);
            portable_sample(indexes.begin(), indexes.end(), std::back_inserter(sample_index), params::sample_size, engine);
            std::sort(sample_index.begin(), sample_index.end());
            std::printf("%s\n{\"seed\": \"", first ? "" : ",");
            for (int i = 0; i < 32; i++) std::printf("%02x", seed.data[i]);
            std::printf("\", \"n\": %zu, \"t\": %zu, \"indices\": [", n, (size_t)params::sample_size);
            for (size_t i = 0; i < sample_index.size(); i++) std::printf("%s%zu", i ? ", " : "", sample_index[i]);
            std::printf("]}");
            first = false;
        }
    std::printf("\n]}\n");
    return 0;
}
