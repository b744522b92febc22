// Host-side check of the Fr nine-limb arithmetic the element-wise prover kernels use since round 4 (csrc/fr29.hip.hpp):
// fr29_to_mont against fp_to_mont, and the lazy sums of k_eval_jobs / k_lincomb_ptrs / k_quotient_combine -- constants in the
// 2^261 form (zk_fr_to_29), two terms per Montgomery reduction (fr29_mul2), the accumulator weakly reduced after every pair,
// one canonical reduction at the end -- against sum_i c_i x_i in the standard 8 x 32-bit arithmetic of bn254.hip.hpp.
#include "fr29.hip.hpp"
#include <cstdio>
#include <random>
#include <vector>
using namespace zk;

static std::mt19937_64 rng(414);
static Fr rand_fr(int kind) {
  Fr a;
  for (;;) {
    for (int i = 0; i < 8; ++i) a.l[i] = (u32)rng();
    a.l[7] &= 0x3fffffff;
    if (kind == 1) { for (int i = 1; i < 8; ++i) a.l[i] = 0; a.l[0] &= 0xff; }
    if (kind == 2) { for (int i = 0; i < 8; ++i) a.l[i] = FrP::MOD[i]; a.l[0] -= 1 + (u32)(rng() % 1000); }   // r - small
    bool lt = false;
    for (int i = 7; i >= 0; --i) {
      if (a.l[i] != FrP::MOD[i]) { lt = a.l[i] < FrP::MOD[i]; break; }
    }
    if (lt) return a;
  }
}
static bool eq(const Fr &a, const Fr &b) {
  for (int i = 0; i < 8; ++i) if (a.l[i] != b.l[i]) return false;
  return true;
}

int main() {
  long bad = 0;
  // canonical integer -> standard form
  for (int t = 0; t < 200000; ++t) {
    const Fr x = rand_fr(t % 3);
    if (!eq(fr29_to_mont(x), fp_to_mont<FrP>(x))) ++bad;
  }
  Fr z = Fr::zero(), top;
  for (int i = 0; i < 8; ++i) top.l[i] = FrP::MOD[i];
  top.l[0] -= 1;
  if (!eq(fr29_to_mont(z), fp_to_mont<FrP>(z)) || !eq(fr29_to_mont(top), fp_to_mont<FrP>(top))) ++bad;
  // lazy sums of m terms, m = 1 .. 97 (odd and even: the unpaired last term), constants and data of every kind
  for (int m = 1; m <= 97; ++m) {
    for (int rep = 0; rep < 40; ++rep) {
      std::vector<Fr> c(m), x(m), c29(m);
      for (int i = 0; i < m; ++i) {
        c[i] = fp_to_mont<FrP>(rand_fr((i + rep) % 3));   // standard-form operands (what the kernels hold)
        x[i] = fp_to_mont<FrP>(rand_fr((i + 2 * rep) % 3));
        c29[i] = zk_fr_to_29(c[i]);
      }
      Fr want = Fr::zero();
      for (int i = 0; i < m; ++i) want = want + c[i] * x[i];
      F29 acc;
      for (int l = 0; l < 9; ++l) acc.l[l] = 0;
      int k = 0;
      for (; k + 1 < m; k += 2)
        acc = fr29_weak_reduce(f29_add(acc, fr29_mul2(fr29_unpack(x[k]), fr29_unpack(c29[k]), fr29_unpack(x[k + 1]), fr29_unpack(c29[k + 1]))));
      if (k < m) acc = fr29_weak_reduce(f29_add(acc, fr29_mul(fr29_unpack(x[k]), fr29_unpack(c29[k]))));
      if (!eq(fr29_pack(fr29_canonical(acc)), want)) ++bad;
      // ... and the final product by one more 2^261-form constant (k_quotient_combine's zinv)
      const Fr zc = fp_to_mont<FrP>(rand_fr(rep % 3));
      if (!eq(fr29_pack(fr29_canonical(fr29_mul(acc, fr29_unpack(zk_fr_to_29(zc))))), want * zc)) ++bad;
    }
  }
  printf("fr29 sums: %ld bad\n", bad);
  return bad ? 1 : 0;
}
