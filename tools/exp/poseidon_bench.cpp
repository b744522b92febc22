// Host Poseidon permutation rate (one core).   clang++ -O3 -march=native -I zk-fhe_amd/host tools/exp/poseidon_bench.cpp
#include <chrono>
#include <cstdio>
#include "poseidon.hpp"
using namespace zkhost;
int main() {
  pos::F s[3] = {pos::ONE, pos::ONE, pos::ONE};
  pos::permute(s);
  const int n = 200000;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) pos::permute(s);
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  printf("%.3f us per permutation (%016llx)\n", us / n, (unsigned long long)s[1].l[0]);
  return 0;
}
