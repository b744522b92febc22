#include <cstdio>
#include <random>
#include "poseidon.hpp"
using namespace zkhost; using namespace zkhost::pos;
int main() {
  std::mt19937_64 g(12345);
  for (int it = 0; it < 300000; ++it) {
    F a[3], b[3];
    for (int i = 0; i < 3; ++i) {
      U256 v; for (int k = 0; k < 4; ++k) v.l[k] = g();
      if (it % 7 == 0) v.l[3] = 0; if (it % 11 == 0) { v.l[0] = v.l[1] = v.l[2] = ~0ull; }
      v.l[3] &= 0x3fffffffffffffffULL;   // below 2^254: may exceed r, the weak arithmetic takes values below 2 r
      memcpy(a[i].l, v.l, 32);
      if (!(v < fe::MOD) ) { /* between r and 2^254 < 2r: still a legal weak input */ }
      b[i] = a[i];
    }
    permute_scalar(a); permute(b);
    for (int i = 0; i < 3; ++i) { U256 u = to_canon(a[i]), v = to_canon(b[i]); if (memcmp(u.l, v.l, 32)) { printf("MISMATCH it %d lane %d\n", it, i); return 1; } }
  }
  printf("300000 random states: ifma == scalar\n");
}
