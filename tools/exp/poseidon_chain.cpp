#include <cstdio>
#include <chrono>
#include "poseidon.hpp"
using namespace zkhost; using namespace zkhost::pos;
int main() {
  const Constants &c = constants();
  F s0 = ONE, x, A = from_canon(U256{{7,1,2,3}});
  const int n = 100000 * 57;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) {   // the scalar chain of one partial round
    x = pow5w(addw(s0, c.pc[i % 57][0]));
    const F m = mulw(c.s_row[i % 57][0], x);
    s0 = addw(m, A);
  }
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  printf("scalar chain: %.1f ns per round (%llx)\n", us * 1000 / n, (unsigned long long)s0.l[0]);
  F st[3] = {ONE, ONE, A};
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 100000 * 8; ++i) full_round(st, c.rc[i % 8], c.mds);
  us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  printf("full round: %.1f ns (%llx)\n", us * 1000 / (100000 * 8), (unsigned long long)st[0].l[0]);
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) s0 = mulw(s0, A);
  us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  printf("mulw dependent chain: %.1f ns (%llx)\n", us * 1000 / n, (unsigned long long)s0.l[0]);
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) s0 = sqrw(s0);
  us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  printf("sqrw dependent chain: %.1f ns (%llx)\n", us * 1000 / n, (unsigned long long)s0.l[0]);
}
