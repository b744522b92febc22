// BN254 optimal-ate pairing on the host, for the `verify` command (reference README.md:48-52; the reference's
// verifier is halo2_proofs `verify_proof` + halo2curves' pairing -- third-party, CPU).  Textbook construction,
// chosen for being easy to check rather than fast (one verification = one product of two Miller loops and one
// final exponentiation):  Fq12 = Fq[w]/(w^12 - 18 w^6 + 82);  G2 on the sextic twist y^2 = x^3 + 3/(9+i) with
// Fq2 = Fq[i]/(i^2+1), mapped into Fq12;  Miller loop over 6t+2 with the two Frobenius corrections;  final
// exponentiation by (q^12-1)/r as a plain square-and-multiply.
#pragma once
#include <array>
#include <vector>

#include "fe.hpp"

namespace zkhost {
namespace pairing {

using zk::Fq;
typedef zk::FqP QP;

inline Fq fq_from_u64(uint64_t v) {
  Fq t = Fq::zero();
  t.l[0] = (uint32_t)v;
  t.l[1] = (uint32_t)(v >> 32);
  return zk::fp_to_mont<QP>(t);
}
inline Fq fq_from_canon(const U256 &c) {
  Fq t;
  memcpy(t.l, c.l, 32);
  return zk::fp_to_mont<QP>(t);
}
inline Fq fq_neg(const Fq &a) { return zk::fp_neg<QP>(a); }
inline Fq fq_inv(const Fq &a) { return zk::fp_inv<QP>(a); }

// ---------------------------------------------------------------------------------- Fq[w] / (w^D - ...)
template <int D>
struct FQP {
  std::array<Fq, D> c;
  static FQP zero() {
    FQP r;
    for (auto &x : r.c) x = Fq::zero();
    return r;
  }
  static FQP one() {
    FQP r = zero();
    r.c[0] = Fq::one();
    return r;
  }
  bool is_zero() const {
    for (const auto &x : c)
      if (!x.is_zero()) return false;
    return true;
  }
  bool operator==(const FQP &o) const {
    for (int i = 0; i < D; ++i)
      if (c[i] != o.c[i]) return false;
    return true;
  }
  FQP operator+(const FQP &o) const {
    FQP r;
    for (int i = 0; i < D; ++i) r.c[i] = c[i] + o.c[i];
    return r;
  }
  FQP operator-(const FQP &o) const {
    FQP r;
    for (int i = 0; i < D; ++i) r.c[i] = c[i] - o.c[i];
    return r;
  }
  FQP operator-() const {
    FQP r;
    for (int i = 0; i < D; ++i) r.c[i] = fq_neg(c[i]);
    return r;
  }
  FQP scale(const Fq &k) const {
    FQP r;
    for (int i = 0; i < D; ++i) r.c[i] = c[i] * k;
    return r;
  }
};

// modulus coefficients m[i] (w^D = -sum m[i] w^i):  Fq2: w^2 = -1 ;  Fq12: w^12 = 18 w^6 - 82
template <int D>
inline const std::array<Fq, D> &modulus_coeffs();
template <>
inline const std::array<Fq, 2> &modulus_coeffs<2>() {
  static const std::array<Fq, 2> m = {fq_from_u64(1), Fq::zero()};
  return m;
}
template <>
inline const std::array<Fq, 12> &modulus_coeffs<12>() {
  static const std::array<Fq, 12> m = [] {
    std::array<Fq, 12> a;
    for (auto &x : a) x = Fq::zero();
    a[0] = fq_from_u64(82);
    a[6] = fq_neg(fq_from_u64(18));
    return a;
  }();
  return m;
}

template <int D>
inline FQP<D> operator*(const FQP<D> &a, const FQP<D> &b) {
  Fq t[2 * D - 1];
  for (auto &x : t) x = Fq::zero();
  for (int i = 0; i < D; ++i) {
    if (a.c[i].is_zero()) continue;
    for (int j = 0; j < D; ++j) t[i + j] = t[i + j] + a.c[i] * b.c[j];
  }
  const auto &m = modulus_coeffs<D>();
  for (int e = 2 * D - 2; e >= D; --e) {
    const Fq top = t[e];
    if (top.is_zero()) continue;
    for (int i = 0; i < D; ++i)
      if (!m[i].is_zero()) t[e - D + i] = t[e - D + i] - top * m[i];
  }
  FQP<D> r;
  for (int i = 0; i < D; ++i) r.c[i] = t[i];
  return r;
}

// inverse by the extended Euclidean algorithm on polynomials over Fq
template <int D>
inline FQP<D> inv(const FQP<D> &a) {
  auto deg = [](const std::vector<Fq> &p) {
    int k = (int)p.size() - 1;
    while (k > 0 && p[k].is_zero()) --k;
    return k;
  };
  std::vector<Fq> lm(D + 1, Fq::zero()), hm(D + 1, Fq::zero()), low(D + 1, Fq::zero()), high(D + 1, Fq::zero());
  lm[0] = Fq::one();
  for (int i = 0; i < D; ++i) {
    low[i] = a.c[i];
    high[i] = modulus_coeffs<D>()[i];
  }
  high[D] = Fq::one();
  while (deg(low) > 0) {
    // r = high / low (rounded polynomial division)
    const int dl = deg(low);
    std::vector<Fq> temp = high, r(D + 1, Fq::zero());
    const Fq linv = fq_inv(low[dl]);
    for (int i = deg(high) - dl; i >= 0; --i) {
      r[i] = r[i] + temp[dl + i] * linv;
      for (int c = 0; c <= dl; ++c) temp[c + i] = temp[c + i] - r[i] * low[c];
    }
    std::vector<Fq> nm = hm, nw = high;
    for (int i = 0; i <= D; ++i)
      for (int j = 0; j + i <= D; ++j) {
        nm[i + j] = nm[i + j] - lm[i] * r[j];
        nw[i + j] = nw[i + j] - low[i] * r[j];
      }
    hm = lm;
    high = low;
    lm = nm;
    low = nw;
  }
  const Fq i0 = fq_inv(low[0]);
  FQP<D> out;
  for (int i = 0; i < D; ++i) out.c[i] = lm[i] * i0;
  return out;
}

typedef FQP<2> Fq2;
typedef FQP<12> Fq12;

template <int D>
inline FQP<D> pow_bits(const FQP<D> &base, const std::vector<uint32_t> &e_le) {
  FQP<D> out = FQP<D>::one();
  bool started = false;
  for (int i = (int)e_le.size() * 32 - 1; i >= 0; --i) {
    if (started) out = out * out;
    if ((e_le[i >> 5] >> (i & 31)) & 1) {
      out = started ? out * base : base;
      started = true;
    }
  }
  return out;
}

// ---------------------------------------------------------------------------------- curve points over FQP
template <class F>
struct Pt {
  F x, y;
  bool inf = false;
};
template <class F>
inline F times(const F &a, uint64_t k) {
  return a.scale(fq_from_u64(k));
}
template <class F>
inline Pt<F> ec_double(const Pt<F> &p) {
  if (p.inf || p.y.is_zero()) return Pt<F>{p.x, p.y, true};
  const F lam = times(p.x * p.x, 3) * inv(times(p.y, 2));
  Pt<F> r;
  r.x = lam * lam - times(p.x, 2);
  r.y = lam * (p.x - r.x) - p.y;
  return r;
}
template <class F>
inline Pt<F> ec_add(const Pt<F> &a, const Pt<F> &b) {
  if (a.inf) return b;
  if (b.inf) return a;
  if (a.x == b.x) {
    if (a.y == b.y) return ec_double(a);
    return Pt<F>{a.x, a.y, true};
  }
  const F lam = (b.y - a.y) * inv(b.x - a.x);
  Pt<F> r;
  r.x = lam * lam - a.x - b.x;
  r.y = lam * (a.x - r.x) - a.y;
  return r;
}
template <class F>
inline Pt<F> ec_mul(Pt<F> p, const U256 &k) {
  Pt<F> acc;
  acc.x = F::zero();
  acc.y = F::zero();
  acc.inf = true;
  for (int i = 0; i < 256; ++i) {
    if ((k.l[i >> 6] >> (i & 63)) & 1) acc = ec_add(acc, p);
    p = ec_double(p);
  }
  return acc;
}

inline Fq fq_from_dec(const char *s) { return fq_from_canon(fe::from_bigint(BigInt::parse_dec(s))); }

inline Pt<Fq2> g2_generator() {
  Pt<Fq2> g;
  g.x.c = {fq_from_dec("10857046999023057135944570762232829481370756359578518086990519993285655852781"),
           fq_from_dec("11559732032986387107991004021392285783925812861821192530917403151452391805634")};
  g.y.c = {fq_from_dec("8495653923123431417604973247489272438418190587263600148770280649306958101930"),
           fq_from_dec("4082367875863433681332203403145435568316851327593401208105741076214120093531")};
  return g;
}

// y^2 = x^3 + 3/(9+i) (the sextic twist)
inline bool g2_on_curve(const Pt<Fq2> &p) {
  if (p.inf) return false;
  Fq2 nine_i, three = Fq2::zero();
  nine_i.c = {fq_from_u64(9), fq_from_u64(1)};
  three.c[0] = fq_from_u64(3);
  const Fq2 b2 = three * inv(nine_i);
  return p.y * p.y == p.x * p.x * p.x + b2;
}

inline Fq12 w_pow(int e) {
  Fq12 w = Fq12::zero();
  w.c[e] = Fq::one();
  return w;
}
// G2 point over Fq2 on the twist -> point on y^2 = x^3 + 3 over Fq12
inline Pt<Fq12> twist(const Pt<Fq2> &p) {
  Pt<Fq12> r;
  r.inf = p.inf;
  r.x = Fq12::zero();
  r.y = Fq12::zero();
  if (p.inf) return r;
  const Fq nine = fq_from_u64(9);
  Fq12 nx = Fq12::zero(), ny = Fq12::zero();
  nx.c[0] = p.x.c[0] - p.x.c[1] * nine;
  nx.c[6] = p.x.c[1];
  ny.c[0] = p.y.c[0] - p.y.c[1] * nine;
  ny.c[6] = p.y.c[1];
  r.x = nx * w_pow(2);
  r.y = ny * w_pow(3);
  return r;
}
inline Pt<Fq12> cast_g1(const Fq &x, const Fq &y) {
  Pt<Fq12> r;
  r.x = Fq12::zero();
  r.y = Fq12::zero();
  r.x.c[0] = x;
  r.y.c[0] = y;
  return r;
}
inline Fq12 linefunc(const Pt<Fq12> &p1, const Pt<Fq12> &p2, const Pt<Fq12> &t) {
  if (!(p1.x == p2.x)) {
    const Fq12 m = (p2.y - p1.y) * inv(p2.x - p1.x);
    return m * (t.x - p1.x) - (t.y - p1.y);
  } else if (p1.y == p2.y) {
    const Fq12 m = times(p1.x * p1.x, 3) * inv(times(p1.y, 2));
    return m * (t.x - p1.x) - (t.y - p1.y);
  }
  return t.x - p1.x;
}

inline std::vector<uint32_t> q_limbs() {
  std::vector<uint32_t> v(8);
  for (int i = 0; i < 8; ++i) v[i] = QP::MOD[i];
  return v;
}

inline Fq12 miller_loop(const Pt<Fq12> &Q, const Pt<Fq12> &P) {
  if (Q.inf || P.inf) return Fq12::one();
  const unsigned __int128 ate = ((unsigned __int128)1 << 64) + (unsigned __int128)11347224129447541672ULL;  // 6t+2 = 29793968203157093288
  Pt<Fq12> R = Q;
  Fq12 f = Fq12::one();
  for (int i = 63; i >= 0; --i) {
    f = f * f * linefunc(R, R, P);
    R = ec_double(R);
    if ((ate >> i) & 1) {
      f = f * linefunc(R, Q, P);
      R = ec_add(R, Q);
    }
  }
  const std::vector<uint32_t> q = q_limbs();
  Pt<Fq12> Q1{pow_bits(Q.x, q), pow_bits(Q.y, q), false};
  Pt<Fq12> nQ2{pow_bits(Q1.x, q), -pow_bits(Q1.y, q), false};
  f = f * linefunc(R, Q1, P);
  R = ec_add(R, Q1);
  f = f * linefunc(R, nQ2, P);
  return f;
}

// (q^12 - 1) / r as little-endian 32-bit limbs, computed once with BigInt arithmetic
inline const std::vector<uint32_t> &final_exponent() {
  static const std::vector<uint32_t> e = [] {
    BigInt q = fe::to_bigint(U256{{0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}});
    BigInt q12(1);
    for (int i = 0; i < 12; ++i) q12 = q12 * q;
    BigInt num = q12 - BigInt(1);
    // divide by r (254-bit): schoolbook long division, bit by bit (one-off, ~3000 iterations)
    const BigInt r = fe::to_bigint(fe::MOD);
    BigInt quo, rem;
    const int nbits = (int)num.bits();
    std::vector<uint32_t> ql((nbits + 31) / 32, 0);
    for (int i = nbits - 1; i >= 0; --i) {
      rem = rem + rem;
      if ((num.mag[i >> 5] >> (i & 31)) & 1) rem = rem + BigInt(1);
      if (!(rem < r)) {
        rem = rem - r;
        ql[i >> 5] |= 1u << (i & 31);
      }
    }
    if (!rem.is_zero()) throw std::logic_error("r does not divide q^12 - 1");
    return ql;
  }();
  return e;
}

struct G1 {
  U256 x, y;  // canonical affine; identity = (0,0)
  bool is_identity() const { return x.is_zero() && y.is_zero(); }
};

// prod_i e(P_i, Q_i) == 1 with one shared final exponentiation
inline bool pairing_product_is_one(const std::vector<std::pair<G1, Pt<Fq2>>> &pairs) {
  Fq12 f = Fq12::one();
  for (const auto &pq : pairs) {
    if (pq.first.is_identity() || pq.second.inf) continue;
    f = f * miller_loop(twist(pq.second), cast_g1(fq_from_canon(pq.first.x), fq_from_canon(pq.first.y)));
  }
  return pow_bits(f, final_exponent()) == Fq12::one();
}

}  // namespace pairing
}  // namespace zkhost
