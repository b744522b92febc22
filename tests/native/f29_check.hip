// Host-side check of csrc/fq29.hip.hpp (the radix-2^29 field and point arithmetic of the MSM kernels) against the standard
// 8 x 32-bit arithmetic of bn254.hip.hpp, which tests/test_oracle_vectors.py / test_gpu_parity.py pin to the big-integer oracle.
#include "fq29.hip.hpp"
#include <cstdio>
#include <random>
using namespace zk;

static std::mt19937_64 rng(2024);
static Fq rand_fq() {
  Fq a;
  for (;;) {
    for (int i = 0; i < 8; ++i) a.l[i] = (u32)rng();
    a.l[7] &= 0x3fffffff;
    bool lt = false;
    for (int i = 7; i >= 0; --i) {
      if (a.l[i] != FqP::MOD[i]) { lt = a.l[i] < FqP::MOD[i]; break; }
    }
    if (lt) return a;
  }
}
static bool eq(const Fq &a, const Fq &b) {
  for (int i = 0; i < 8; ++i) if (a.l[i] != b.l[i]) return false;
  return true;
}
static Fq times32(Fq a) { for (int i = 0; i < 5; ++i) a = fp_dbl<FqP>(a); return a; }
// random point: k * G by the standard arithmetic
static G1Affine rand_point() {
  G1Affine g;
  Fq one = Fq::zero(), two = Fq::zero();
  one.l[0] = 1; two.l[0] = 2;
  g.x = fp_to_mont<FqP>(one); g.y = fp_to_mont<FqP>(two);
  G1X acc = G1X::identity();
  const unsigned k = 2 + (unsigned)(rng() % 5000);
  for (int bit = 13; bit >= 0; --bit) { acc = g1x_dbl(acc); if ((k >> bit) & 1) g1x_add_affine(acc, g, false); }
  return g1x_to_affine(acc);
}
static bool same_point(const G1X29 &p29, const G1X &pstd) {
  const G1Affine a = g1x_to_affine(g1x29_to_std(p29)), b = g1x_to_affine(pstd);
  return eq(a.x, b.x) && eq(a.y, b.y);
}

int main() {
  int bad = 0;
  const u32 P1[9] = ZK_Q29_P, P2[9] = ZK_Q29_2P, P4[9] = ZK_Q29_4P;
  // pack / unpack round trip, products, fused products, lazy operands
  for (int it = 0; it < 200000; ++it) {
    const Fq a = rand_fq(), b = rand_fq(), c = rand_fq(), d = rand_fq();
    const F29 A = f29_unpack(a), B = f29_unpack(b), C = f29_unpack(c), D = f29_unpack(d);
    if (!eq(f29_pack(A), a)) { if (bad++ < 5) printf("pack/unpack mismatch at %d\n", it); }
    // a b / 2^261 * 32 == a b / 2^256
    const Fq want = fp_mul<FqP>(a, b);
    if (!eq(times32(f29_pack(f29_canonical(f29_mul(A, B)))), want)) { if (bad++ < 5) printf("mul mismatch at %d\n", it); }
    const Fq want2 = fp_add<FqP>(fp_mul<FqP>(a, b), fp_mul<FqP>(c, d));
    if (!eq(times32(f29_pack(f29_canonical(f29_mul2(A, B, C, D)))), want2)) { if (bad++ < 5) printf("mul2 mismatch at %d\n", it); }
    // lazy operands: (a + b + 2p - c) and (4p - d + a) are below 11 p
    const F29 X = f29_sub(f29_add(A, B), C, P2), Y = f29_add(f29_neg(D, P4), A);
    const Fq xs = fp_sub<FqP>(fp_add<FqP>(a, b), c), ys = fp_sub<FqP>(a, d);
    if (!eq(times32(f29_pack(f29_canonical(f29_mul(X, Y)))), fp_mul<FqP>(xs, ys))) { if (bad++ < 5) printf("lazy mul mismatch at %d\n", it); }
    if (!eq(times32(f29_pack(f29_canonical(f29_sqr(A)))), fp_mul<FqP>(a, a)) || !eq(times32(f29_pack(f29_canonical(f29_sqr(X)))), fp_mul<FqP>(xs, xs)) ||
        !eq(times32(f29_pack(f29_canonical(f29_sqr(Y)))), fp_mul<FqP>(ys, ys))) { if (bad++ < 5) printf("sqr mismatch at %d\n", it); }
    if (!eq(f29_pack(f29_canonical(X)), xs) || !eq(f29_pack(f29_canonical(Y)), ys)) { if (bad++ < 5) printf("add/sub mismatch at %d\n", it); }
    const F29 W = f29_weak_reduce(f29_add(f29_add(X, Y), f29_add(X, Y)));   // < 16 p in
    if (!eq(f29_pack(f29_canonical(W)), fp_dbl<FqP>(fp_add<FqP>(xs, ys)))) { if (bad++ < 5) printf("weak reduce mismatch at %d\n", it); }
    // zero test on differences of representatives
    const F29 Ap = f29_add(A, f29_const(P1));   // a + p: another representative of a, < 2p
    if (!f29_is_zero_mod_p(f29_sub(A, Ap, P2)) || !f29_is_zero_mod_p(f29_sub(Ap, A, P2)) || !f29_is_zero_mod_p(f29_sub(A, A, P2)) ||
        f29_is_zero_mod_p(f29_sub(A, B, P2))) { if (bad++ < 5) printf("zero test mismatch at %d\n", it); }
  }
  // point arithmetic: chains of mixed / full additions and doublings, with the special cases
  for (int it = 0; it < 3000; ++it) {
    const G1Affine p = rand_point(), q = rand_point();
    const G1A29 p29 = g1a29_load(g1_affine_to_29(p)), q29 = g1a29_load(g1_affine_to_29(q));
    G1X s = G1X::identity();
    G1X29 s29 = G1X29::identity();
    g1x_add_affine(s, p, false); g1x29_add_affine(s29, p29, false);
    g1x_add_affine(s, q, it & 1); g1x29_add_affine(s29, q29, it & 1);
    if (!same_point(s29, s)) { if (bad++ < 5) printf("mixed add mismatch at %d\n", it); }
    G1X t = s; G1X29 t29 = s29;
    g1x_add_affine(t, p, false); g1x29_add_affine(t29, p29, false);   // p + q + p
    g1x_add(t, s); g1x29_add(t29, s29);
    t = g1x_dbl(t); t29 = g1x29_dbl(t29);
    if (!same_point(t29, t)) { if (bad++ < 5) printf("add / dbl chain mismatch at %d\n", it); }
    // doubling through the addition formulas, and cancellation
    G1X d = G1X::identity(); G1X29 d29 = G1X29::identity();
    g1x_add_affine(d, p, false); g1x29_add_affine(d29, p29, false);
    g1x_add_affine(d, p, false); g1x29_add_affine(d29, p29, false);
    if (!same_point(d29, d)) { if (bad++ < 5) printf("mixed doubling mismatch at %d\n", it); }
    G1X e = t; G1X29 e29 = t29;
    g1x_add(e, t); g1x29_add(e29, t29);
    if (!same_point(e29, e)) { if (bad++ < 5) printf("full doubling mismatch at %d\n", it); }
    g1x29_add_affine(d29, p29, true); g1x29_add_affine(d29, p29, true);
    if (!d29.is_identity()) { if (bad++ < 5) printf("cancellation mismatch at %d\n", it); }
    // store / load round trip keeps the point
    if (!same_point(g1x29_load(g1x29_store(t29)), t)) { if (bad++ < 5) printf("store/load mismatch at %d\n", it); }
  }
  printf("fq29: %d bad\n", bad);
  return bad != 0;
}
