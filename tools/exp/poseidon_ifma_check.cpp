#include <cstdio>
#include <chrono>
#include "poseidon.hpp"
using namespace zkhost; using namespace zkhost::pos;
int main() {
  F a[3] = {ONE, from_canon(U256{{5,0,0,0}}), from_canon(U256{{7,1,2,3}})}, b[3];
  for (int it = 0; it < 2000; ++it) {
    b[0]=a[0]; b[1]=a[1]; b[2]=a[2];
    permute_scalar(a); permute(b);
    for (int i = 0; i < 3; ++i) { U256 u = to_canon(a[i]), v = to_canon(b[i]); if (memcmp(u.l, v.l, 32)) { printf("MISMATCH it %d lane %d\n", it, i); return 1; } }
  }
  printf("ifma matches scalar on 2000 chained permutations\n");
  const int n = 100000;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) permute_scalar(a);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  printf("scalar %.3f us per permutation\n", us / n);
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) permute(a);
  us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  printf("ifma   %.3f us per permutation (%llx)\n", us / n, (unsigned long long)a[0].l[0]);
}
