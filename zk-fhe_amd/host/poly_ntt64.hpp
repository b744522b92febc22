// Exact integer product of two short polynomials with 32-bit coefficients on the host: an NTT convolution over the 64-bit
// prime p = 2^64 - 2^32 + 1.  Used by Poly::mul (src/poly.rs:75-103 restated in poly.hpp) for the two products pk_i * u of the
// phase-0 witness at N <= 2048, Q < 2^32 (BASELINE configs[0..2]): 0.25 ms on one core, against ten dependent kernel launches
// and a host round trip per product on the GPU -- with twenty proofs starting together those launches queued on the
// runtime's submission path for 5 ms.  Longer or wider polynomials (N = 4096 / 32768, Q = 2^60 - 93) go to the GPU
// (zkfhe_witness_poly_mul_u64).
//
// a is split into 16-bit halves: every coefficient of (a_half * b) is below 2^16 * 2^32 * 2048 = 2^59 < p, so the residues
// ARE the integers, and c = a_lo * b + 2^16 (a_hi * b) is exact in 128 bits.
#pragma once
#include <cstdint>
#include <vector>

namespace zkhost {
namespace gl {

typedef unsigned __int128 u128;
static const uint64_t P = 0xffffffff00000001ULL;
static const uint64_t EPS = 0xffffffffULL;   // 2^64 mod p

inline uint64_t add(uint64_t a, uint64_t b) {   // a, b < p
  uint64_t r = a + b;
  if (r < a) return r + EPS;   // wrapped: the true sum is r + 2^64, minus p is r + 2^32 - 1 (below p)
  return r >= P ? r - P : r;
}
inline uint64_t sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a - b + P; }
inline uint64_t reduce128(u128 x) {   // 2^64 = 2^32 - 1, 2^96 = -1 (mod p)
  const uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
  const uint64_t hh = hi >> 32, hl = hi & EPS;
  uint64_t r = lo - hh;
  if (lo < hh) r += P;            // lo - hh + p, in [0, p)
  const uint64_t t = hl * EPS;    // below 2^64
  uint64_t s = r + t;
  if (s < t) s += EPS;            // wrapped once; cannot wrap again (s + 2^32 - 1 <= 2^64 - 2^32 - 1)
  return s >= P ? s - P : s;
}
inline uint64_t mul(uint64_t a, uint64_t b) { return reduce128((u128)a * b); }
inline uint64_t pow(uint64_t b, uint64_t e) {
  uint64_t r = 1;
  for (; e; e >>= 1, b = mul(b, b))
    if (e & 1) r = mul(r, b);
  return r;
}

struct Plan {   // size-m transform: twiddles w^k, k < m/2, for the forward and the inverse root
  size_t m = 0;
  std::vector<uint64_t> fwd, inv;
  uint64_t m_inv = 0;
};
inline const Plan &plan(size_t m) {
  static thread_local Plan cache[13];
  int lg = 0;
  while (((size_t)1 << lg) < m) ++lg;
  Plan &p = cache[lg];
  if (p.m != m) {
    p.m = m;
    const uint64_t w = pow(7, (P - 1) / m), wi = pow(w, P - 2);   // 7 generates the multiplicative group
    p.fwd.resize(m / 2);
    p.inv.resize(m / 2);
    uint64_t a = 1, b = 1;
    for (size_t k = 0; k < m / 2; ++k) {
      p.fwd[k] = a;
      p.inv[k] = b;
      a = mul(a, w);
      b = mul(b, wi);
    }
    p.m_inv = pow((uint64_t)m, P - 2);
  }
  return p;
}
// in place, natural order in and out (bit reversal, then decimation-in-time butterflies)
inline void ntt(std::vector<uint64_t> &x, bool inverse) {
  const size_t m = x.size();
  const Plan &p = plan(m);
  for (size_t i = 1, j = 0; i < m; ++i) {
    size_t bit = m >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) std::swap(x[i], x[j]);
  }
  const std::vector<uint64_t> &tw = inverse ? p.inv : p.fwd;
  for (size_t len = 2; len <= m; len <<= 1) {
    const size_t half = len >> 1, step = m / len;
    for (size_t i = 0; i < m; i += len)
      for (size_t k = 0; k < half; ++k) {
        const uint64_t u = x[i + k], v = mul(x[i + k + half], tw[k * step]);
        x[i + k] = add(u, v);
        x[i + k + half] = sub(u, v);
      }
  }
  if (inverse)
    for (size_t i = 0; i < m; ++i) x[i] = mul(x[i], p.m_inv);
}

inline bool fits(const std::vector<uint64_t> &a, const std::vector<uint64_t> &b) {
  const size_t n = a.size();
  if (n != b.size() || n < 2 || n > 2048 || (n & (n - 1))) return false;
  for (size_t i = 0; i < n; ++i)
    if ((a[i] | b[i]) >> 32) return false;
  return true;
}
// c[k] = sum_{i+j=k} a[i] b[j], 2n - 1 coefficients as (low, high) 64-bit words; requires fits(a, b)
inline void poly_mul_u32(const std::vector<uint64_t> &a, const std::vector<uint64_t> &b, std::vector<uint64_t> &lo, std::vector<uint64_t> &hi) {
  const size_t n = a.size(), m = 2 * n;
  std::vector<uint64_t> a0(m, 0), a1(m, 0), fb(m, 0);
  for (size_t i = 0; i < n; ++i) {
    a0[i] = a[i] & 0xffff;
    a1[i] = a[i] >> 16;
    fb[i] = b[i];
  }
  ntt(a0, false);
  ntt(a1, false);
  ntt(fb, false);
  for (size_t i = 0; i < m; ++i) {
    a0[i] = mul(a0[i], fb[i]);
    a1[i] = mul(a1[i], fb[i]);
  }
  ntt(a0, true);
  ntt(a1, true);
  lo.resize(2 * n - 1);
  hi.resize(2 * n - 1);
  for (size_t k = 0; k < 2 * n - 1; ++k) {
    const u128 c = (u128)a0[k] + ((u128)a1[k] << 16);
    lo[k] = (uint64_t)c;
    hi[k] = (uint64_t)(c >> 64);
  }
}

}  // namespace gl
}  // namespace zkhost
