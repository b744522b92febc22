// Host-side check of csrc/lz29.hip.hpp (signed lazy radix-2^29 values of the 2^13 NTT tile) against the standard 8 x 32-bit Fr
// arithmetic of bn254.hip.hpp: products with loose signed operands, the fused two-product form, carry propagation, the weak
// reduction of values of either sign up to 16 r, and the canonical store -- on the radix-8 butterfly network of ntt13.hip.
#include "lz29.hip.hpp"
#include <cstdio>
#include <random>
using namespace zk;

static std::mt19937_64 rng(77);
static Fr rand_fr() {
  Fr a;
  for (;;) {
    for (int i = 0; i < 8; ++i) a.l[i] = (u32)rng();
    a.l[7] &= 0x3fffffff;
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
// the integer a stored value stands for, as a standard field element: a column value is x 2^256 as is; a product against a
// twiddle w 2^261 is x w 2^256 again
static Lw tw_of(const Fr &w) { return lw_unpack(zk_fr_to_29(w)); }

int main() {
  int bad = 0;
  for (int it = 0; it < 100000; ++it) {
    const Fr a = rand_fr(), b = rand_fr(), c = rand_fr(), d = rand_fr(), w = rand_fr(), v = rand_fr();
    const auto A = lz_load(a), B = lz_load(b), C = lz_load(c), D = lz_load(d);
    // store of loose values of either sign: a + b - c - d, a - b - c - d (|value| < 4 r), and a sum of sixteen terms
    const auto s1 = lz_sub(lz_add(A, B), lz_add(C, D));
    if (!eq(lz_store(s1), fp_sub<FrP>(fp_add<FrP>(a, b), fp_add<FrP>(c, d)))) { if (bad++ < 5) printf("store(a+b-c-d) mismatch at %d\n", it); }
    const auto s2 = lz_sub(lz_sub(A, B), lz_add(C, D));
    if (!eq(lz_store(s2), fp_sub<FrP>(fp_sub<FrP>(a, b), fp_add<FrP>(c, d)))) { if (bad++ < 5) printf("store(a-b-c-d) mismatch at %d\n", it); }
    const auto n4 = lz_norm(lz_add(lz_add(A, B), lz_add(C, D)));            // (0,1), < 4 r
    const auto s16 = lz_add(lz_add(n4, n4), lz_add(n4, n4));                   // (0,4), < 16 r
    Fr want16 = fp_add<FrP>(fp_add<FrP>(a, b), fp_add<FrP>(c, d));
    want16 = fp_dbl<FrP>(fp_dbl<FrP>(want16));
    if (!eq(lz_store(s16), want16)) { if (bad++ < 5) printf("store(16 terms) mismatch at %d\n", it); }
    const auto m16 = lz_sub(lz_sub(lz_zero(), lz_add(n4, n4)), n4);           // (3,1): -12 terms
    const Fr want4 = fp_add<FrP>(fp_add<FrP>(a, b), fp_add<FrP>(c, d));
    const Fr want12 = fp_add<FrP>(fp_dbl<FrP>(want4), want4);   // three times the four-term sum
    if (!eq(lz_store(m16), fp_neg<FrP>(want12))) { if (bad++ < 5) printf("store(-12 terms) mismatch at %d\n", it); }
    // products: tight, loose-positive (0,2), loose-signed (2,2) and (1,2), negative values
    const Lw W = tw_of(w), V = tw_of(v);
    if (!eq(lz_store(lz_mul(A, W)), fp_mul<FrP>(a, w))) { if (bad++ < 5) printf("mul mismatch at %d\n", it); }
    if (!eq(lz_store(lz_mul(lz_add(A, B), W)), fp_mul<FrP>(fp_add<FrP>(a, b), w))) { if (bad++ < 5) printf("mul(a+b) mismatch at %d\n", it); }
    const auto x22 = lz_sub(lz_add(A, B), lz_add(C, D));   // (2,2)
    if (!eq(lz_store(lz_mul(x22, W)), fp_mul<FrP>(fp_sub<FrP>(fp_add<FrP>(a, b), fp_add<FrP>(c, d)), w))) { if (bad++ < 5) printf("mul(2,2) mismatch at %d\n", it); }
    const auto x21 = lz_sub(lz_sub(A, B), C);              // (2,1)
    if (!eq(lz_store(lz_mul(x21, W)), fp_mul<FrP>(fp_sub<FrP>(fp_sub<FrP>(a, b), c), w))) { if (bad++ < 5) printf("mul(2,1) mismatch at %d\n", it); }
    // a product of a product (signed top limb) and of a weakly reduced value
    const LzT p1 = lz_mul(x22, W);
    if (!eq(lz_store(lz_mul(p1, V)), fp_mul<FrP>(fp_mul<FrP>(fp_sub<FrP>(fp_add<FrP>(a, b), fp_add<FrP>(c, d)), w), v))) { if (bad++ < 5) printf("mul(mul) mismatch at %d\n", it); }
    if (!eq(lz_store(lz_mul(lz_weak(s16), V)), fp_mul<FrP>(want16, v))) { if (bad++ < 5) printf("mul(weak) mismatch at %d\n", it); }
    if (!eq(lz_store(lz_mul2(A, W, B, V)), fp_add<FrP>(fp_mul<FrP>(a, w), fp_mul<FrP>(b, v)))) { if (bad++ < 5) printf("mul2 mismatch at %d\n", it); }
    // the two-product form on signed operands (limbs of either sign below 2^29: differences of column values) and the unsigned
    // four-product form on canonical data -- the radix-4 first stage of the quarter-column 2^13 tile
    if (!eq(lz_store(lz_mul2(lz_sub(A, C), W, lz_sub(B, D), V)), fp_add<FrP>(fp_mul<FrP>(fp_sub<FrP>(a, c), w), fp_mul<FrP>(fp_sub<FrP>(b, d), v)))) { if (bad++ < 5) printf("mul2(signed) mismatch at %d\n", it); }
    {
      const Lw WC = tw_of(c), WD = tw_of(d);   // any canonical constants will do
      const Fr want = fp_add<FrP>(fp_add<FrP>(fp_mul<FrP>(a, w), fp_mul<FrP>(b, v)), fp_add<FrP>(fp_mul<FrP>(c, c), fp_mul<FrP>(d, d)));
      const LzT got4 = lz_mul4u(A, W, B, V, C, WC, D, WD);
      if (!eq(lz_store(got4), want)) { if (bad++ < 5) printf("mul4u mismatch at %d\n", it); }
      bool ok4 = got4.l[8] >= 0;
      for (int i = 0; i < 8; ++i) ok4 = ok4 && got4.l[i] >= 0 && got4.l[i] < (1 << 29);
      if (!ok4) { if (bad++ < 5) printf("mul4u result not tight at %d\n", it); }
    }
    // weak: result limbs tight, value in [0, 2 r)
    const LzT wk = lz_weak(m16);
    bool tight = wk.l[8] >= 0;
    for (int i = 0; i < 8; ++i) tight = tight && wk.l[i] >= 0 && wk.l[i] < (1 << 29);
    if (!tight) { if (bad++ < 5) printf("weak result not tight at %d\n", it); }
  }
  printf("lz29: %d bad\n", bad);
  return bad != 0;
}
