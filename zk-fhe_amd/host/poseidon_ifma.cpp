// The Poseidon permutation (poseidon.hpp) with the multiplications by constants on AVX-512 IFMA lanes -- all of them in the 57
// partial rounds, eight of the nine matrix products in the full rounds.  Why: a Fiat-Shamir sponge is one sequential chain on one host core (3 750 permutations per k = 13 proof, 2 561 of
// them before the first challenge), and a partial round in sparse form is 8 Montgomery products of which only 4 are on the
// dependent chain.  Rewritten one round ahead,
//     Y_t = col0_{t-1} x_{t-1} + W_t,        W_t = Y_{t-1} + pc1_t            (same for Z with col1, pc2)
//     s0_{t+1} = row0_t x_t + A_t,            A_t = K_t x_{t-1} + row1_t W^y_t + row2_t W^z_t,   K_t = row1_t col0_{t-1} + row2_t col1_{t-1}
// everything except the S-box x_t = (s0_t + pc0_t)^5 and the product row0_t x_t depends on x_{t-1} only: five products by
// constants (plus two by one, which keep W bounded) that have a whole round of slack.  They run as ONE eight-lane
// Montgomery product in radix 2^52 (vpmadd52luq / vpmadd52huq, 115 of them) beside the scalar chain -- the integer multiplier
// and the vector unit work in parallel.  Data lanes hold the same x * 2^256 representatives as the scalar code, re-limbed;
// the constants are held as c * 2^260, so a lane product (x 2^256)(c 2^260) / 2^260 is again a data value.
// Checked against the scalar rounds on every permutation of tests/test_poseidon.py (ZKFHE_POSEIDON_SCALAR=1 runs without).
#include <immintrin.h>

#include <cstdlib>

#include "poseidon.hpp"

#define ZK_IFMA __attribute__((target("avx512f,avx512ifma,avx512vl,avx512dq,avx512bw")))

namespace zkhost {
namespace pos {

namespace {

const uint64_t M52 = ((uint64_t)1 << 52) - 1;

struct L5 {   // one value, five 52-bit limbs
  uint64_t l[5];
};
inline L5 to_l5(const F &a) {   // any value below 2^256
  L5 r;
  r.l[0] = a.l[0] & M52;
  r.l[1] = ((a.l[0] >> 52) | (a.l[1] << 12)) & M52;
  r.l[2] = ((a.l[1] >> 40) | (a.l[2] << 24)) & M52;
  r.l[3] = ((a.l[2] >> 28) | (a.l[3] << 36)) & M52;
  r.l[4] = a.l[3] >> 16;
  return r;
}
inline F from_l5(const uint64_t l[5]) {   // normalised limbs, value below 2^256
  F r;
  r.l[0] = l[0] | (l[1] << 52);
  r.l[1] = (l[1] >> 12) | (l[2] << 40);
  r.l[2] = (l[2] >> 24) | (l[3] << 28);
  r.l[3] = (l[3] >> 36) | (l[4] << 16);
  return r;
}

struct alignas(64) V8 {   // eight values, limb j of all of them in l[j]
  uint64_t l[5][8];
};

struct Tables {
  V8 B[R_P];        // job t: lanes {col0_{t-1}, col1_{t-1}, K_t, row1_t, row2_t, 1, 1, 0} as c * 2^260
  V8 PC[R_P + 1];   // PC[t]: pc1_t in lanes 3 and 5, pc2_t in lanes 4 and 6 (data form); PC[R_P] = 0
  V8 M[2];          // full rounds: lanes {m00, m01, m02, m10, m11, m12, m20, m21} as c * 2^260 of mds ([0]) and pre ([1]); m22 stays scalar
  uint64_t p[5], inv;
};

const Tables &tables() {
  static const Tables T = [] {
    Tables t;
    memset(&t, 0, sizeof(t));
    const Constants &c = constants();
    const F sixteen = from_canon(U256{{16, 0, 0, 0}});   // x * 2^256 -> x * 2^260 as an integer: times 16
    auto put = [&](V8 &v, int lane, const F &val) {
      const L5 q = to_l5(val);
      for (int j = 0; j < 5; ++j) v.l[j][lane] = q.l[j];
    };
    auto c260 = [&](const F &v) { return mul(v, sixteen); };
    for (int r = 0; r < R_P; ++r) {
      if (r > 0) {
        put(t.B[r], 0, c260(c.s_col[r - 1][0]));
        put(t.B[r], 1, c260(c.s_col[r - 1][1]));
        put(t.B[r], 2, c260(add(mul(c.s_row[r][1], c.s_col[r - 1][0]), mul(c.s_row[r][2], c.s_col[r - 1][1]))));
      }
      put(t.B[r], 3, c260(c.s_row[r][1]));
      put(t.B[r], 4, c260(c.s_row[r][2]));
      put(t.B[r], 5, c260(ONE));
      put(t.B[r], 6, c260(ONE));
      put(t.PC[r], 3, c.pc[r][1]);
      put(t.PC[r], 5, c.pc[r][1]);
      put(t.PC[r], 4, c.pc[r][2]);
      put(t.PC[r], 6, c.pc[r][2]);
    }
    for (int w = 0; w < 2; ++w) {
      const F(*m)[3] = w ? c.pre : c.mds;
      for (int lane = 0; lane < 8; ++lane) put(t.M[w], lane, c260(m[lane / 3][lane % 3]));
    }
    F pf;
    memcpy(pf.l, P, 32);
    const L5 pl = to_l5(pf);
    for (int j = 0; j < 5; ++j) t.p[j] = pl.l[j];
    uint64_t x = 1;   // -p^-1 mod 2^52 by Newton iteration on p^-1 mod 2^64
    for (int i = 0; i < 6; ++i) x *= 2 - P[0] * x;
    t.inv = (0 - x) & M52;
    return t;
  }();
  return T;
}

struct R5 {
  __m512i l[5];
};

// carry propagation: limbs below 2^63 in, below 2^52 out (the value must be below 2^260)
ZK_IFMA inline void normalise(R5 &v) {
  const __m512i mask = _mm512_set1_epi64((long long)M52);
  __m512i c = _mm512_srli_epi64(v.l[0], 52);
  v.l[0] = _mm512_and_si512(v.l[0], mask);
  v.l[1] = _mm512_add_epi64(v.l[1], c);
  c = _mm512_srli_epi64(v.l[1], 52);
  v.l[1] = _mm512_and_si512(v.l[1], mask);
  v.l[2] = _mm512_add_epi64(v.l[2], c);
  c = _mm512_srli_epi64(v.l[2], 52);
  v.l[2] = _mm512_and_si512(v.l[2], mask);
  v.l[3] = _mm512_add_epi64(v.l[3], c);
  c = _mm512_srli_epi64(v.l[3], 52);
  v.l[3] = _mm512_and_si512(v.l[3], mask);
  v.l[4] = _mm512_add_epi64(v.l[4], c);
}

// One iteration of the eight-lane product (operand scanning over the limbs of b, reduction interleaved)
#define ZK_MUL8_ITER(i)                                               \
  do {                                                                \
    const __m512i bi = bv.l[i];                                       \
    t0 = _mm512_madd52lo_epu64(t0, av.l[0], bi);                      \
    t1 = _mm512_madd52lo_epu64(t1, av.l[1], bi);                      \
    t2 = _mm512_madd52lo_epu64(t2, av.l[2], bi);                      \
    t3 = _mm512_madd52lo_epu64(t3, av.l[3], bi);                      \
    t4 = _mm512_madd52lo_epu64(t4, av.l[4], bi);                      \
    t1 = _mm512_madd52hi_epu64(t1, av.l[0], bi);                      \
    t2 = _mm512_madd52hi_epu64(t2, av.l[1], bi);                      \
    t3 = _mm512_madd52hi_epu64(t3, av.l[2], bi);                      \
    t4 = _mm512_madd52hi_epu64(t4, av.l[3], bi);                      \
    t5 = _mm512_madd52hi_epu64(t5, av.l[4], bi);                      \
    const __m512i m = _mm512_madd52lo_epu64(zero, t0, inv);           \
    t0 = _mm512_madd52lo_epu64(t0, m, p0);                            \
    t1 = _mm512_madd52lo_epu64(t1, m, p1);                            \
    t2 = _mm512_madd52lo_epu64(t2, m, p2);                            \
    t3 = _mm512_madd52lo_epu64(t3, m, p3);                            \
    t4 = _mm512_madd52lo_epu64(t4, m, p4);                            \
    t1 = _mm512_madd52hi_epu64(t1, m, p0);                            \
    t2 = _mm512_madd52hi_epu64(t2, m, p1);                            \
    t3 = _mm512_madd52hi_epu64(t3, m, p2);                            \
    t4 = _mm512_madd52hi_epu64(t4, m, p3);                            \
    t5 = _mm512_madd52hi_epu64(t5, m, p4);                            \
    t0 = _mm512_add_epi64(t1, _mm512_srli_epi64(t0, 52));             \
    t1 = t2, t2 = t3, t3 = t4, t4 = t5, t5 = zero;                    \
  } while (0)

// A full round: the three S-boxes on the integer multiplier, the matrix (eight of its nine products) as one eight-lane product
ZK_IFMA inline void full_round_v(F s[T], const F rc[T], int which, const Tables &Tb, const Constants &c) {
  const __m512i zero = _mm512_setzero_si512();
  const __m512i inv = _mm512_set1_epi64((long long)Tb.inv);
  const __m512i p0 = _mm512_set1_epi64((long long)Tb.p[0]), p1 = _mm512_set1_epi64((long long)Tb.p[1]), p2 = _mm512_set1_epi64((long long)Tb.p[2]),
                p3 = _mm512_set1_epi64((long long)Tb.p[3]), p4 = _mm512_set1_epi64((long long)Tb.p[4]);
  const F v0 = pow5w(addw(s[0], rc[0])), v1 = pow5w(addw(s[1], rc[1])), v2 = pow5w(addw(s[2], rc[2]));
  const L5 a0 = to_l5(v0), a1 = to_l5(v1), a2 = to_l5(v2);
  R5 av, bv;
  for (int j = 0; j < 5; ++j) {
    av.l[j] = _mm512_setr_epi64((long long)a0.l[j], (long long)a1.l[j], (long long)a2.l[j], (long long)a0.l[j], (long long)a1.l[j], (long long)a2.l[j],
                                (long long)a0.l[j], (long long)a1.l[j]);
    bv.l[j] = _mm512_load_si512((const void *)Tb.M[which].l[j]);
  }
  __m512i t0 = zero, t1 = zero, t2 = zero, t3 = zero, t4 = zero, t5 = zero;
  ZK_MUL8_ITER(0);
  ZK_MUL8_ITER(1);
  ZK_MUL8_ITER(2);
  const F last = mulw((which ? c.pre : c.mds)[2][2], v2);   // the ninth product
  ZK_MUL8_ITER(3);
  ZK_MUL8_ITER(4);
  alignas(64) uint64_t buf[5][8];
  _mm512_store_si512((void *)buf[0], t0);
  _mm512_store_si512((void *)buf[1], t1);
  _mm512_store_si512((void *)buf[2], t2);
  _mm512_store_si512((void *)buf[3], t3);
  _mm512_store_si512((void *)buf[4], t4);
  // row sums in limb form (each lane below 2 r, limbs below 2^58 before normalisation), carried and packed
  auto row = [&](int l0, int l1, int l2) {
    uint64_t q[5];
    for (int j = 0; j < 5; ++j) q[j] = buf[j][l0] + buf[j][l1] + (l2 >= 0 ? buf[j][l2] : 0);
    for (int j = 0; j < 4; ++j) {
      q[j + 1] += q[j] >> 52;
      q[j] &= M52;
    }
    return from_l5(q);   // below 6 r < 2^256 ... folded by the caller
  };
  auto fold2 = [](const F &f) {
    const F g = fold_2r(f.l[0], f.l[1], f.l[2], f.l[3]);
    return fold_2r(g.l[0], g.l[1], g.l[2], g.l[3]);
  };
  // three lanes below 2 r each: below 6 r = 1.13 * 2^256 would not fit -- the lanes are below r (a / 2^260 + 1) with a below 2 r:
  // below 1.03 r each, sums below 3.1 r (and 2.1 r + 1.4 r for the last row)
  s[0] = fold2(row(0, 1, 2));
  s[1] = fold2(row(3, 4, 5));
  const F r2 = row(6, 7, -1);
  unsigned long long cy;
  const uint64_t w0 = __builtin_addcll(r2.l[0], last.l[0], 0, &cy);
  const uint64_t w1 = __builtin_addcll(r2.l[1], last.l[1], cy, &cy);
  const uint64_t w2 = __builtin_addcll(r2.l[2], last.l[2], cy, &cy);
  const uint64_t w3 = __builtin_addcll(r2.l[3], last.l[3], cy, &cy);
  s[2] = fold2(F{{w0, w1, w2, w3}});
}

ZK_IFMA void rounds(F s[T]) {
  const Constants &c = constants();
  const Tables &Tb = tables();
  const __m512i zero = _mm512_setzero_si512();
  const __m512i inv = _mm512_set1_epi64((long long)Tb.inv);
  const __m512i p0 = _mm512_set1_epi64((long long)Tb.p[0]), p1 = _mm512_set1_epi64((long long)Tb.p[1]), p2 = _mm512_set1_epi64((long long)Tb.p[2]),
                p3 = _mm512_set1_epi64((long long)Tb.p[3]), p4 = _mm512_set1_epi64((long long)Tb.p[4]);
  const __m512i take01 = _mm512_setr_epi64(7, 7, 7, 0, 1, 0, 1, 7), take56 = _mm512_setr_epi64(7, 7, 7, 5, 6, 5, 6, 7);
  const __m512i take3 = _mm512_setr_epi64(7, 7, 3, 7, 7, 7, 7, 7), take4 = _mm512_setr_epi64(7, 7, 4, 7, 7, 7, 7, 7);   // lane 7 is zero
  // W_0 = (s1 + pc1_0, s2 + pc2_0) in lanes 3..6
  const L5 y0 = to_l5(addw(s[1], c.pc[0][1])), z0 = to_l5(addw(s[2], c.pc[0][2]));
  R5 W;
  for (int j = 0; j < 5; ++j) W.l[j] = _mm512_setr_epi64(0, 0, 0, (long long)y0.l[j], (long long)z0.l[j], (long long)y0.l[j], (long long)z0.l[j], 0);
  F s0 = s[0], x = {{0, 0, 0, 0}};   // x_{-1} = 0: job 0 has no x terms
  alignas(64) uint64_t buf[5][8];
  for (int t = 0; t < R_P; ++t) {
    // Iteration t: the S-box x_t = (s0_t + pc0_t)^5 on the integer multiplier and, interleaved in program order so that both
    // are in the scheduler's window together, job t on the vector unit: lanes {x_{t-1}, x_{t-1}, x_{t-1}, W^y_t, W^z_t, W^y_t, W^z_t}
    // times {col0_{t-1}, col1_{t-1}, K_t, row1_t, row2_t, 1, 1}.  They meet in s0_{t+1} = row0_t x_t + A_t.
    const L5 xl = to_l5(x);
    R5 av, bv;
    for (int j = 0; j < 5; ++j) {
      av.l[j] = _mm512_mask_set1_epi64(W.l[j], 0x07, (long long)xl.l[j]);
      bv.l[j] = _mm512_load_si512((const void *)Tb.B[t].l[j]);
    }
    __m512i t0 = zero, t1 = zero, t2 = zero, t3 = zero, t4 = zero, t5 = zero;
    const F u = addw(s0, c.pc[t][0]);
    ZK_MUL8_ITER(0);
    ZK_MUL8_ITER(1);
    const F u2 = sqrw(u);
    ZK_MUL8_ITER(2);
    ZK_MUL8_ITER(3);
    const F u4 = sqrw(u2);
    ZK_MUL8_ITER(4);
    R5 o;
    o.l[0] = t0, o.l[1] = t1, o.l[2] = t2, o.l[3] = t3, o.l[4] = t4;
    normalise(o);
    x = mulw(u4, u);
    // Y_t = col0 x + W (lanes 0 + 5 -> lanes 3 and 5), Z likewise, plus the next round's constants: W_{t+1}; A_t = lanes 2 + 3 + 4
    R5 sum;
    for (int j = 0; j < 5; ++j) {
      W.l[j] = _mm512_add_epi64(_mm512_add_epi64(_mm512_permutexvar_epi64(take01, o.l[j]), _mm512_permutexvar_epi64(take56, o.l[j])),
                                _mm512_load_si512((const void *)Tb.PC[t + 1].l[j]));
      sum.l[j] = _mm512_add_epi64(_mm512_mask_blend_epi64(0x04, zero, o.l[j]),
                                  _mm512_add_epi64(_mm512_permutexvar_epi64(take3, o.l[j]), _mm512_permutexvar_epi64(take4, o.l[j])));
    }
    normalise(W);
    normalise(sum);
    for (int j = 0; j < 5; ++j) _mm512_store_si512((void *)buf[j], sum.l[j]);
    const uint64_t al[5] = {buf[0][2], buf[1][2], buf[2][2], buf[3][2], buf[4][2]};
    const F A = from_l5(al);   // below 3.2 r
    // s0_{t+1} = row0_t x_t + A_t: below 1.4 r + 3.2 r, folded below 2.6 r; the next S-box's addw folds again
    const F m = mulw(c.s_row[t][0], x);
    unsigned long long cy;
    const uint64_t w0 = __builtin_addcll(m.l[0], A.l[0], 0, &cy);
    const uint64_t w1 = __builtin_addcll(m.l[1], A.l[1], cy, &cy);
    const uint64_t w2 = __builtin_addcll(m.l[2], A.l[2], cy, &cy);
    const uint64_t w3 = __builtin_addcll(m.l[3], A.l[3], cy, &cy);
    s0 = fold_2r(w0, w1, w2, w3);
  }
  // W is W_{R_P} = Y_{R_P-1}, Z_{R_P-1} (PC[R_P] = 0); the last round's s1 = col0 x + Y, s2 = col1 x + Z
  for (int j = 0; j < 5; ++j) _mm512_store_si512((void *)buf[j], W.l[j]);
  const uint64_t yl[5] = {buf[0][3], buf[1][3], buf[2][3], buf[3][3], buf[4][3]}, zl[5] = {buf[0][4], buf[1][4], buf[2][4], buf[3][4], buf[4][4]};
  const F Y = from_l5(yl), Z = from_l5(zl);   // below 3.1 r
  auto add_fold = [](const F &a, const F &b) {   // below 1.4 r + 3.1 r, folded twice: below 2 r
    unsigned long long cy;
    const uint64_t q0 = __builtin_addcll(a.l[0], b.l[0], 0, &cy);
    const uint64_t q1 = __builtin_addcll(a.l[1], b.l[1], cy, &cy);
    const uint64_t q2 = __builtin_addcll(a.l[2], b.l[2], cy, &cy);
    const uint64_t q3 = __builtin_addcll(a.l[3], b.l[3], cy, &cy);
    const F f = fold_2r(q0, q1, q2, q3);
    return fold_2r(f.l[0], f.l[1], f.l[2], f.l[3]);
  };
  s[0] = fold_2r(s0.l[0], s0.l[1], s0.l[2], s0.l[3]);
  s[1] = add_fold(mulw(c.s_col[R_P - 1][0], x), Y);
  s[2] = add_fold(mulw(c.s_col[R_P - 1][1], x), Z);
}
#undef ZK_MUL8_ITER

ZK_IFMA void permute_v(F s[T]) {
  const Constants &c = constants();
  const Tables &Tb = tables();
  const int half = R_F / 2;
  for (int r = 0; r < half; ++r) full_round_v(s, c.rc[r], r == half - 1 ? 1 : 0, Tb, c);
  rounds(s);
  for (int r = half + R_P; r < ROUNDS; ++r) full_round_v(s, c.rc[r], 0, Tb, c);
}

}  // namespace

bool permute_ifma(F s[T]) {
  static const bool ok = [] {
    if (getenv("ZKFHE_POSEIDON_SCALAR")) return false;
    __builtin_cpu_init();
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512ifma") && __builtin_cpu_supports("avx512vl") &&
           __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512bw");
  }();
  if (!ok) return false;
  permute_v(s);
  return true;
}

}  // namespace pos
}  // namespace zkhost
