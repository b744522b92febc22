#include "bn254.hip.hpp"
#include <cstdio>
#include <random>
using namespace zk;
template <class P> int run(const char *name) {
  std::mt19937_64 rng(12345);
  int bad = 0;
  for (int it = 0; it < 200000; ++it) {
    Fp<P> a;
    for (int i = 0; i < 8; ++i) a.l[i] = (u32)rng();
    a.l[7] &= 0x0fffffff;  // < p
    if (it < 64) { for (int i = 0; i < 8; ++i) a.l[i] = 0; a.l[it / 8] = 1u << (it % 8 * 4); }
    if (it == 64) { for (int i = 0; i < 8; ++i) a.l[i] = P::MOD[i]; a.l[0] -= 1; }
    if (it == 65) { for (int i = 0; i < 8; ++i) a.l[i] = P::MOD[i]; a.l[0] -= 2; }
    if (it > 65 && it < 1000) { for (int i = 1 + it % 7; i < 8; ++i) a.l[i] = 0; }
    Fp<P> x = fp_inv<P>(a), y = fp_inv_euclid<P>(a);
    bool eq = true;
    for (int i = 0; i < 8; ++i) eq &= x.l[i] == y.l[i];
    Fp<P> one = fp_mul<P>(a, x);
    bool isone = true;
    for (int i = 0; i < 8; ++i) isone &= one.l[i] == P::ONE[i];
    if (!eq || !isone) { if (bad < 5) printf("%s mismatch at %d eq=%d one=%d\n", name, it, eq, isone); ++bad; }
  }
  printf("%s: %d bad\n", name, bad);
  return bad;
}
int main() { return run<FrP>("Fr") + run<FqP>("Fq"); }
