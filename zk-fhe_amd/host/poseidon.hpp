// Poseidon over BN254 Fr as snark-verifier's PoseidonTranscript<NativeLoader> uses it (T = 3, RATE = 2, R_F = 8, R_P = 57, x^5),
// the Fiat-Shamir hash of the reference's `gen_snark_shplonk` / `verify` (examples/bfv.rs:311 through halo2-scaffold;
// third-party, restated in oracle/poseidon_ref.py, pinned by the public vector poseidonperm_x5_254_3).
//
//  * constants: Grain LFSR of the Poseidon paper (the `poseidon` crate's Spec::new), generated once at first use;
//  * permutation: partial rounds in sparse-matrix form (an equivalent rewriting of the plain Hades permutation, as the
//    crate's own pre-sparse-MDS form is); the oracle computes the plain form and tests/test_poseidon.py compares the two;
//  * sponge: state [2^64, 0, 0]; a chunk shorter than RATE is followed by a 1; an exact multiple gets one more
//    permutation of the empty chunk; squeeze returns state[1].
// A Fiat-Shamir sponge is one sequential chain (about 3600 permutations per k = 13 proof, 2561 of them for the 5121 public
// inputs), so it runs on a host core, in 64-bit Montgomery arithmetic.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "fe.hpp"

namespace zkhost {
namespace pos {

typedef unsigned __int128 u128;

struct F {  // Montgomery form, R = 2^256, value < r
  uint64_t l[4];
};

static const uint64_t P[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const uint64_t INV = 0xc2e1f593efffffffULL;  // -r^-1 mod 2^64
static const F R2 = {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}};  // 2^512 mod r
static const F ONE = {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}};  // 2^256 mod r

// r = t - P if t >= P else t, branch-free (a reduced value is above or below r with no pattern a predictor could learn)
inline F reduce_once(uint64_t t0, uint64_t t1, uint64_t t2, uint64_t t3) {
  unsigned long long br;
  const uint64_t r0 = __builtin_subcll(t0, P[0], 0, &br);
  const uint64_t r1 = __builtin_subcll(t1, P[1], br, &br);
  const uint64_t r2 = __builtin_subcll(t2, P[2], br, &br);
  const uint64_t r3 = __builtin_subcll(t3, P[3], br, &br);
  const uint64_t keep = (uint64_t)0 - (uint64_t)br;  // all ones when t < P
  return F{{(t0 & keep) | (r0 & ~keep), (t1 & keep) | (r1 & ~keep), (t2 & keep) | (r2 & ~keep), (t3 & keep) | (r3 & ~keep)}};
}
inline F add(const F &a, const F &b) {
  unsigned long long c;
  const uint64_t t0 = __builtin_addcll(a.l[0], b.l[0], 0, &c);
  const uint64_t t1 = __builtin_addcll(a.l[1], b.l[1], c, &c);
  const uint64_t t2 = __builtin_addcll(a.l[2], b.l[2], c, &c);
  const uint64_t t3 = __builtin_addcll(a.l[3], b.l[3], c, &c);  // a + b < 2r < 2^255: no carry out
  return reduce_once(t0, t1, t2, t3);
}

// Montgomery product, operand scanning with the reduction interleaved (CIOS).  The top limb of r is < 2^62, so the running
// value never needs a fifth limb ("no-carry" variant): t < 2 r throughout, one conditional subtraction at the end.
inline F mul(const F &a, const F &b) {
  uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint64_t bi = b.l[i];
    u128 A = (u128)a.l[0] * bi + t0;
    const uint64_t m = (uint64_t)A * INV;
    u128 C = (u128)m * P[0] + (uint64_t)A;
    A = (u128)a.l[1] * bi + t1 + (uint64_t)(A >> 64);
    C = (u128)m * P[1] + (uint64_t)A + (uint64_t)(C >> 64);
    t0 = (uint64_t)C;
    A = (u128)a.l[2] * bi + t2 + (uint64_t)(A >> 64);
    C = (u128)m * P[2] + (uint64_t)A + (uint64_t)(C >> 64);
    t1 = (uint64_t)C;
    A = (u128)a.l[3] * bi + t3 + (uint64_t)(A >> 64);
    C = (u128)m * P[3] + (uint64_t)A + (uint64_t)(C >> 64);
    t2 = (uint64_t)C;
    t3 = (uint64_t)(C >> 64) + (uint64_t)(A >> 64);
  }
  return reduce_once(t0, t1, t2, t3);
}
inline F dot3(const F m[3], const F s[3]) { return add(add(mul(m[0], s[0]), mul(m[1], s[1])), mul(m[2], s[2])); }
inline F from_canon(const U256 &v) {
  F a;
  memcpy(a.l, v.l, 32);
  return mul(a, R2);
}
inline U256 to_canon(const F &a) {   // a below 2 r
  const F r = mul(a, F{{1, 0, 0, 0}});
  U256 v;
  memcpy(v.l, r.l, 32);
  return v;
}
inline F pow5(const F &x) {
  const F x2 = mul(x, x);
  return mul(mul(x2, x2), x);
}
inline F inv(const F &a) {  // a^(r-2); used nine times, for the Cauchy matrix
  uint64_t e[4] = {P[0] - 2, P[1], P[2], P[3]};
  F r = ONE, b = a;
  for (int i = 0; i < 256; ++i) {
    if ((e[i >> 6] >> (i & 63)) & 1) r = mul(r, b);
    b = mul(b, b);
  }
  return r;
}

// ---- "weak" arithmetic of the permutation: values below 2 r, reduced only when they leave the sponge -------------------
// r < 2^254, so with R = 2^256 a Montgomery product of two values below 2 r is again below 2 r (4 r^2 / R + r < 1.76 r): the
// conditional subtraction after every product goes.  A three-term inner product is reduced ONCE (48 + 20 word products
// instead of 108), a square takes 10 + 20 instead of 36.
static const uint64_t P2[4] = {0x87c3eb27e0000002ULL, 0x5067d090f372e122ULL, 0x70a08b6d0302b0baULL, 0x60c89ce5c2634053ULL};  // 2 r

// t < 4 r  ->  t - 2 r if t >= 2 r else t
inline F fold_2r(uint64_t t0, uint64_t t1, uint64_t t2, uint64_t t3) {
  unsigned long long br;
  const uint64_t r0 = __builtin_subcll(t0, P2[0], 0, &br);
  const uint64_t r1 = __builtin_subcll(t1, P2[1], br, &br);
  const uint64_t r2 = __builtin_subcll(t2, P2[2], br, &br);
  const uint64_t r3 = __builtin_subcll(t3, P2[3], br, &br);
  const uint64_t keep = (uint64_t)0 - (uint64_t)br;
  return F{{(t0 & keep) | (r0 & ~keep), (t1 & keep) | (r1 & ~keep), (t2 & keep) | (r2 & ~keep), (t3 & keep) | (r3 & ~keep)}};
}
inline F addw(const F &a, const F &b) {   // a, b < 2 r -> < 2 r
  unsigned long long c;
  const uint64_t t0 = __builtin_addcll(a.l[0], b.l[0], 0, &c);
  const uint64_t t1 = __builtin_addcll(a.l[1], b.l[1], c, &c);
  const uint64_t t2 = __builtin_addcll(a.l[2], b.l[2], c, &c);
  const uint64_t t3 = __builtin_addcll(a.l[3], b.l[3], c, &c);  // < 4 r < 2^256
  return fold_2r(t0, t1, t2, t3);
}
inline F mulw(const F &a, const F &b) {   // a, b < 2 r -> < 2 r, no final subtraction
  uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint64_t bi = b.l[i];
    u128 A = (u128)a.l[0] * bi + t0;
    const uint64_t m = (uint64_t)A * INV;
    u128 C = (u128)m * P[0] + (uint64_t)A;
    A = (u128)a.l[1] * bi + t1 + (uint64_t)(A >> 64);
    C = (u128)m * P[1] + (uint64_t)A + (uint64_t)(C >> 64);
    t0 = (uint64_t)C;
    A = (u128)a.l[2] * bi + t2 + (uint64_t)(A >> 64);
    C = (u128)m * P[2] + (uint64_t)A + (uint64_t)(C >> 64);
    t1 = (uint64_t)C;
    A = (u128)a.l[3] * bi + t3 + (uint64_t)(A >> 64);
    C = (u128)m * P[3] + (uint64_t)A + (uint64_t)(C >> 64);
    t2 = (uint64_t)C;
    t3 = (uint64_t)(C >> 64) + (uint64_t)(A >> 64);
  }
  return F{{t0, t1, t2, t3}};
}
// T (eight words, below 0.8 * 2^512 - R r) -> T / R mod r, below T / R + r
inline void mont_reduce_wide(uint64_t T[8], uint64_t out[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint64_t m = T[i] * INV;
    u128 C = (u128)m * P[0] + T[i];
    C = (u128)m * P[1] + T[i + 1] + (uint64_t)(C >> 64);
    T[i + 1] = (uint64_t)C;
    C = (u128)m * P[2] + T[i + 2] + (uint64_t)(C >> 64);
    T[i + 2] = (uint64_t)C;
    C = (u128)m * P[3] + T[i + 3] + (uint64_t)(C >> 64);
    T[i + 3] = (uint64_t)C;
    unsigned long long c;
    T[i + 4] = __builtin_addcll(T[i + 4], (uint64_t)(C >> 64), 0, &c);
    for (int k = i + 5; k < 8; ++k) T[k] = __builtin_addcll(T[k], 0, c, &c);
  }
  out[0] = T[4], out[1] = T[5], out[2] = T[6], out[3] = T[7];
}
// T += a * b (eight-word accumulator, no carry out by the callers' bounds)
inline void mac_wide(uint64_t T[8], const F &a, const F &b) {
  uint64_t p[8];
  u128 c = (u128)a.l[0] * b.l[0];
  p[0] = (uint64_t)c;
  c = (u128)a.l[1] * b.l[0] + (uint64_t)(c >> 64);
  p[1] = (uint64_t)c;
  c = (u128)a.l[2] * b.l[0] + (uint64_t)(c >> 64);
  p[2] = (uint64_t)c;
  c = (u128)a.l[3] * b.l[0] + (uint64_t)(c >> 64);
  p[3] = (uint64_t)c;
  p[4] = (uint64_t)(c >> 64);
#pragma unroll
  for (int i = 1; i < 4; ++i) {
    c = (u128)a.l[0] * b.l[i] + p[i];
    p[i] = (uint64_t)c;
    c = (u128)a.l[1] * b.l[i] + p[i + 1] + (uint64_t)(c >> 64);
    p[i + 1] = (uint64_t)c;
    c = (u128)a.l[2] * b.l[i] + p[i + 2] + (uint64_t)(c >> 64);
    p[i + 2] = (uint64_t)c;
    c = (u128)a.l[3] * b.l[i] + p[i + 3] + (uint64_t)(c >> 64);
    p[i + 3] = (uint64_t)c;
    p[i + 4] = (uint64_t)(c >> 64);
  }
  unsigned long long cy = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) T[k] = __builtin_addcll(T[k], p[k], cy, &cy);
}
// m . s, constants m below r, s below 2 r: sum below 6 r^2, reduced once to below 2.2 r, folded below 2 r
inline F dot3w(const F m[3], const F s[3]) {
  uint64_t T[8] = {0, 0, 0, 0, 0, 0, 0, 0}, o[4];
  mac_wide(T, m[0], s[0]);
  mac_wide(T, m[1], s[1]);
  mac_wide(T, m[2], s[2]);
  mont_reduce_wide(T, o);
  return fold_2r(o[0], o[1], o[2], o[3]);
}
// a^2, a below 2 r -> below 2 r: the six cross products once, doubled
inline F sqrw(const F &a) {
  uint64_t T[8], o[4];
  u128 c = (u128)a.l[0] * a.l[1];
  T[1] = (uint64_t)c;
  c = (u128)a.l[0] * a.l[2] + (uint64_t)(c >> 64);
  T[2] = (uint64_t)c;
  c = (u128)a.l[0] * a.l[3] + (uint64_t)(c >> 64);
  T[3] = (uint64_t)c;
  T[4] = (uint64_t)(c >> 64);
  c = (u128)a.l[1] * a.l[2] + T[3];
  T[3] = (uint64_t)c;
  c = (u128)a.l[1] * a.l[3] + T[4] + (uint64_t)(c >> 64);
  T[4] = (uint64_t)c;
  T[5] = (uint64_t)(c >> 64);
  c = (u128)a.l[2] * a.l[3] + T[5];
  T[5] = (uint64_t)c;
  T[6] = (uint64_t)(c >> 64);
  // double (the cross sum is below 2^509: the shifted-out bit is zero), then add the squares
  T[7] = T[6] >> 63;
  T[6] = (T[6] << 1) | (T[5] >> 63);
  T[5] = (T[5] << 1) | (T[4] >> 63);
  T[4] = (T[4] << 1) | (T[3] >> 63);
  T[3] = (T[3] << 1) | (T[2] >> 63);
  T[2] = (T[2] << 1) | (T[1] >> 63);
  T[1] = T[1] << 1;
  unsigned long long cy;
  c = (u128)a.l[0] * a.l[0];
  T[0] = (uint64_t)c;
  T[1] = __builtin_addcll(T[1], (uint64_t)(c >> 64), 0, &cy);
  c = (u128)a.l[1] * a.l[1];
  T[2] = __builtin_addcll(T[2], (uint64_t)c, cy, &cy);
  T[3] = __builtin_addcll(T[3], (uint64_t)(c >> 64), cy, &cy);
  c = (u128)a.l[2] * a.l[2];
  T[4] = __builtin_addcll(T[4], (uint64_t)c, cy, &cy);
  T[5] = __builtin_addcll(T[5], (uint64_t)(c >> 64), cy, &cy);
  c = (u128)a.l[3] * a.l[3];
  T[6] = __builtin_addcll(T[6], (uint64_t)c, cy, &cy);
  T[7] = __builtin_addcll(T[7], (uint64_t)(c >> 64), cy, &cy);
  mont_reduce_wide(T, o);   // 4 r^2 / R + r < 1.76 r
  return F{{o[0], o[1], o[2], o[3]}};
}
inline F pow5w(const F &x) { return mulw(sqrw(sqrw(x)), x); }
inline F canon(const F &a) {   // below 2 r -> below r
  return reduce_once(a.l[0], a.l[1], a.l[2], a.l[3]);
}

static const int T = 3, RATE = 2, R_F = 8, R_P = 57, ROUNDS = R_F + R_P;

struct Constants {
  F rc[ROUNDS][T];   // plain round constants (Grain order)
  F mds[T][T];
  // Partial rounds in sparse form.  With M = S_r D_r, D_r = diag(1, M^_r) and S_r = [[m00, v^], [w, I]], the dense factor
  // D_r commutes with the partial S-box and is pushed into the round before; what is left of it after the first partial
  // round is folded into the last full round of the first half (pre = D_0 M).  z_{r+1} = S_r sbox0(z_r + D_r c_r): the same
  // permutation as the plain form (tests/test_poseidon.py), 7 products per partial round instead of 12.
  F pre[T][T];
  F pc[R_P][T];      // D_r c_r
  F s_row[R_P][T];   // (m00, v^_1, v^_2)
  F s_col[R_P][T - 1];  // w
};

// Grain LFSR (x^80 + x^62 + x^51 + x^38 + x^23 + x^13 + 1), self-shrinking: of each pair of bits the second is output when
// the first is 1.  Seed: field tag 1 (2 bits), S-box tag 0 (4), field size 254 (12), t (12), R_F (10), R_P (10), thirty 1s;
// the first 160 bits are discarded.  Field elements are read most significant bit first.
class Grain {
 public:
  Grain() {
    int n = 0;
    auto put = [&](unsigned v, int w) {
      for (int i = w - 1; i >= 0; --i) s[n++] = (v >> i) & 1;
    };
    put(1, 2);
    put(0, 4);
    put(254, 12);
    put(T, 12);
    put(R_F, 10);
    put(R_P, 10);
    put((1u << 30) - 1, 30);
    for (int i = 0; i < 160; ++i) step();
  }
  U256 next_bits254() {
    U256 v = fe::zero();
    for (int i = 253; i >= 0; --i)
      if (next_bit()) v.l[i >> 6] |= (uint64_t)1 << (i & 63);
    return v;
  }
  U256 next_field_element() {  // rejection sampling
    for (;;) {
      const U256 v = next_bits254();
      if (v < fe::MOD) return v;
    }
  }
  U256 next_field_element_without_rejection() {  // reduced instead: 2^254 < 2 r
    U256 v = next_bits254();
    if (!(v < fe::MOD)) fe::sub_raw(v, v, fe::MOD);
    return v;
  }

 private:
  uint8_t s[80];
  int head = 0;  // s[(head + i) % 80] is bit i of the register
  int step() {
    auto at = [&](int i) { return s[(head + i) % 80]; };
    const int b = at(62) ^ at(51) ^ at(38) ^ at(23) ^ at(13) ^ at(0);
    s[head] = (uint8_t)b;  // drop bit 0, append the new bit
    head = (head + 1) % 80;
    return b;
  }
  int next_bit() {
    for (;;) {
      const int a = step(), b = step();
      if (a) return b;
    }
  }
};

inline F sub(const F &a, const F &b) {
  unsigned long long br, c;
  const uint64_t t0 = __builtin_subcll(a.l[0], b.l[0], 0, &br);
  const uint64_t t1 = __builtin_subcll(a.l[1], b.l[1], br, &br);
  const uint64_t t2 = __builtin_subcll(a.l[2], b.l[2], br, &br);
  const uint64_t t3 = __builtin_subcll(a.l[3], b.l[3], br, &br);
  const uint64_t m = (uint64_t)0 - (uint64_t)br;  // add r back after a borrow
  F r;
  r.l[0] = __builtin_addcll(t0, P[0] & m, 0, &c);
  r.l[1] = __builtin_addcll(t1, P[1] & m, c, &c);
  r.l[2] = __builtin_addcll(t2, P[2] & m, c, &c);
  r.l[3] = __builtin_addcll(t3, P[3] & m, c, &c);
  return r;
}

inline const Constants &constants() {
  static const Constants C = [] {
    Constants c;
    Grain g;
    for (int r = 0; r < ROUNDS; ++r)
      for (int i = 0; i < T; ++i) c.rc[r][i] = from_canon(g.next_field_element());
    F xs[T], ys[T];
    for (int i = 0; i < T; ++i) xs[i] = from_canon(g.next_field_element_without_rejection());
    for (int i = 0; i < T; ++i) ys[i] = from_canon(g.next_field_element_without_rejection());
    for (int i = 0; i < T; ++i)
      for (int j = 0; j < T; ++j) c.mds[i][j] = inv(add(xs[i], ys[j]));  // Cauchy matrix 1 / (x_i + y_j)
    // sparse factors, from the last partial round backwards: M_r = S_r D_r, M_{r-1} = D_r M
    const F zero = {{0, 0, 0, 0}};
    F cur[T][T];
    memcpy(cur, c.mds, sizeof(cur));
    for (int r = R_P - 1; r >= 0; --r) {
      // inverse of the lower-right 2 x 2 block
      const F a = cur[1][1], b = cur[1][2], cc = cur[2][1], d = cur[2][2];
      const F det_inv = inv(sub(mul(a, d), mul(b, cc)));
      const F i00 = mul(d, det_inv), i01 = mul(sub(zero, b), det_inv), i10 = mul(sub(zero, cc), det_inv), i11 = mul(a, det_inv);
      c.s_row[r][0] = cur[0][0];
      c.s_row[r][1] = add(mul(cur[0][1], i00), mul(cur[0][2], i10));  // v^T M^^-1
      c.s_row[r][2] = add(mul(cur[0][1], i01), mul(cur[0][2], i11));
      c.s_col[r][0] = cur[1][0];
      c.s_col[r][1] = cur[2][0];
      const F *k = c.rc[R_F / 2 + r];
      c.pc[r][0] = k[0];
      c.pc[r][1] = add(mul(a, k[1]), mul(b, k[2]));
      c.pc[r][2] = add(mul(cc, k[1]), mul(d, k[2]));
      // next (earlier) round's matrix: D_r M
      F nxt[T][T];
      for (int j = 0; j < T; ++j) {
        nxt[0][j] = c.mds[0][j];
        nxt[1][j] = add(mul(a, c.mds[1][j]), mul(b, c.mds[2][j]));
        nxt[2][j] = add(mul(cc, c.mds[1][j]), mul(d, c.mds[2][j]));
      }
      memcpy(cur, nxt, sizeof(cur));
    }
    memcpy(c.pre, cur, sizeof(cur));
    return c;
  }();
  return C;
}

inline void full_round(F s[T], const F rc[T], const F m[T][T]) {
  const F v[T] = {pow5w(addw(s[0], rc[0])), pow5w(addw(s[1], rc[1])), pow5w(addw(s[2], rc[2]))};
  s[0] = dot3w(m[0], v);
  s[1] = dot3w(m[1], v);
  s[2] = dot3w(m[2], v);
}

// poseidon_ifma.cpp: the permutation with the multiplications by constants on AVX-512 IFMA lanes beside the S-box chain;
// false (nothing done) on a CPU without IFMA or with ZKFHE_POSEIDON_SCALAR set
bool permute_ifma(F s[T]);

inline void partial_rounds_scalar(F s[T]) {
  const Constants &c = constants();
  for (int r = 0; r < R_P; ++r) {
    const F x = pow5w(addw(s[0], c.pc[r][0]));
    const F y = addw(s[1], c.pc[r][1]), z = addw(s[2], c.pc[r][2]);
    const F v[T] = {x, y, z};
    s[0] = dot3w(c.s_row[r], v);
    s[1] = addw(mulw(c.s_col[r][0], x), y);
    s[2] = addw(mulw(c.s_col[r][1], x), z);
  }
}

// state in, state out: below 2 r (weak); the constants are canonical
inline void permute_scalar(F s[T]) {
  const Constants &c = constants();
  const int half = R_F / 2;
  for (int r = 0; r < half; ++r) full_round(s, c.rc[r], r == half - 1 ? c.pre : c.mds);
  partial_rounds_scalar(s);
  for (int r = half + R_P; r < ROUNDS; ++r) full_round(s, c.rc[r], c.mds);
}
inline void permute(F s[T]) {
  if (!permute_ifma(s)) permute_scalar(s);
}

// poseidon_x8.cpp: eight sponges in lockstep, one per AVX-512 lane.  A job is "absorb n_pairs full chunks into this state"; the
// hash service runs the jobs of all proofs in flight side by side (a job joins a free lane between two permutations).
struct AbsorbJob {
  F st[T];             // in: sponge state (2^256 form, below 2 r); out: the state n_pairs permutations later
  const U256 *data;    // 2 * n_pairs canonical values, alive until wait() returns
  size_t n_pairs = 0;
  void wait();
  void finish();       // the service's side

 private:
  std::mutex mu;
  std::condition_variable cv;
  bool done = false;
};
bool x8_available();                                   // AVX-512 IFMA present, not disabled (ZKFHE_POSEIDON_X8=0 / ZKFHE_POSEIDON_SCALAR)
void x8_submit(AbsorbJob *j);                          // asynchronous: the hash service's worker threads
bool x8_absorb_now(AbsorbJob *const *jobs, size_t n);  // on the calling thread (tests, parity hook); false without IFMA
// How the transcripts of concurrent provers hash (zkfhe_host_hash_mode): 0 = "latency": every sponge on its own thread, the
// single-sponge path (4.6 us per permutation on an EPYC 9575F); 1 = "shared": long runs go to the eight-lane service (9.4 us per
// step of eight lanes = 1.2 us per permutation when the lanes are full, but every sponge now waits 9.4 us per permutation).
// Measured on one MI355X with 16 host CPUs (wave of 20 proofs): latency 195 proofs/s at 28 ms of host CPU per proof, shared
// 175 proofs/s at 16 ms -- the mode for hosts with few CPUs per GPU.  Initial value: ZKFHE_HASH_MODE=latency|shared.
inline std::atomic<int> &hash_mode() {
  static std::atomic<int> m{[] {
    const char *e = getenv("ZKFHE_HASH_MODE");
    return e && e[0] == 's' ? 1 : 0;
  }()};
  return m;
}
// Provers in flight in this process: a sponge that has the lanes to itself is better off on the single-sponge path.
inline std::atomic<int> &bulk_clients() {
  static std::atomic<int> n{0};
  return n;
}
struct BulkClientScope {
  BulkClientScope() { bulk_clients().fetch_add(1, std::memory_order_relaxed); }
  ~BulkClientScope() { bulk_clients().fetch_sub(1, std::memory_order_relaxed); }
};

// snark-verifier util/hash/poseidon.rs `Poseidon<F, L, 3, 2>`.  Full chunks are permuted as they arrive (the result is the
// same as buffering until the squeeze, and lets the public inputs be absorbed while the GPU is busy).
class Sponge {
 public:
  Sponge() {
    U256 v = fe::zero();
    v.l[1] = 1;  // 2^64
    st[0] = from_canon(v);
    st[1] = st[2] = F{{0, 0, 0, 0}};
  }
  void update(const U256 &x) {
    buf[n_buf++] = from_canon(x);
    if (n_buf == RATE) {
      st[1] = addw(st[1], buf[0]);
      st[2] = addw(st[2], buf[1]);
      permute(st);
      n_buf = 0;
      ++n_perm;
    }
  }
  U256 squeeze() {
    // what is left is a chunk of 0 or 1 elements: absorbed with a 1 in the first free word
    if (n_buf == 1) {
      st[1] = addw(st[1], buf[0]);
      st[2] = addw(st[2], ONE);
    } else {
      st[1] = addw(st[1], ONE);
    }
    permute(st);
    n_buf = 0;
    ++n_perm;
    return to_canon(st[1]);
  }
  // A long run of scalars through the hash service: begin_bulk() returns at once, end_bulk() waits.  `data` must stay alive and
  // the sponge untouched in between.  Same state afterwards as n update() calls.
  void begin_bulk(const U256 *data, size_t n, AbsorbJob &job) {
    begin_bulk_prepare(data, n, job);
    x8_submit(&job);
  }
  void begin_bulk_prepare(const U256 *data, size_t n, AbsorbJob &job) {   // the job filled in, not yet submitted
    size_t i = 0;
    if (n_buf == 1 && n) update(data[i++]);
    job.n_pairs = (n - i) / 2;
    job.data = data + i;
    for (int t = 0; t < T; ++t) job.st[t] = st[t];
    bulk_tail = (n - i) & 1 ? data + n - 1 : nullptr;
  }
  void end_bulk(AbsorbJob &job) {
    job.wait();
    for (int t = 0; t < T; ++t) st[t] = job.st[t];
    n_perm += job.n_pairs;
    if (bulk_tail) update(*bulk_tail);
    bulk_tail = nullptr;
  }
  size_t n_perm = 0;

 private:
  F st[T], buf[RATE];
  int n_buf = 0;
  const U256 *bulk_tail = nullptr;
};

}  // namespace pos
}  // namespace zkhost
