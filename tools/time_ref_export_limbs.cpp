// time_ref_export_limbs.cpp -- how long the REFERENCE's own mpz_vector::export_limbs (include/util/mpz_vector.hpp:108-127) takes for one row of k = 8192 elements:
// the host cost every write_buffer_clear of the stage contexts is preceded by (nonbatch_context.hpp:447).  Built against /root/reference + the image's GMP (same flags as
// oracle/Makefile's _ref targets);  profiles/r06_export_limbs_cost.txt holds the result.  Measurement only.
#include <gmpxx.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <cstring>
#include <util/mpz_vector.hpp>
int main(){
  using namespace ligero::vm;
  const size_t k=8192;
  mpz_vector v(k);
  gmp_randclass rng(gmp_randinit_default);
  mpz_class p("21888242871839275222246405745257275088548364400416034343698204186575808495617");
  for(size_t i=0;i<k;i++) v[i]=rng.get_z_range(p);
  std::vector<uint64_t> limbs(2*k*4);
  auto t0=std::chrono::steady_clock::now();
  const int reps=200;
  for(int r=0;r<reps;r++) v.export_limbs(limbs.data(), limbs.size(), sizeof(uint64_t), 4);
  double us=std::chrono::duration<double,std::micro>(std::chrono::steady_clock::now()-t0).count()/reps;
  std::printf("mpz_vector::export_limbs of %zu elements: %.1f us per row\n", k, us);
}
