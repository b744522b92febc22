// Latency of dependent Montgomery products / squarings: the CIOS of poseidon.hpp against a product-scanning product followed by
// a reduction whose quotient digits are the only sequential part.
#include <chrono>
#include <cstdio>
#include "poseidon.hpp"
using namespace zkhost; using namespace zkhost::pos;
typedef unsigned __int128 u128;

// column sums with a three-word accumulator
#define COL_ADD(x, y) do { u128 _p = (u128)(x) * (y); unsigned long long _c; c0 = __builtin_addcll(c0, (uint64_t)_p, 0, &_c); c1 = __builtin_addcll(c1, (uint64_t)(_p >> 64), _c, &_c); c2 += _c; } while (0)
#define COL_OUT(dst) do { dst = c0; c0 = c1; c1 = c2; c2 = 0; } while (0)
static inline void prod_scan(const F &a, const F &b, uint64_t T[8]) {
  uint64_t c0 = 0, c1 = 0, c2 = 0;
  COL_ADD(a.l[0], b.l[0]); COL_OUT(T[0]);
  COL_ADD(a.l[0], b.l[1]); COL_ADD(a.l[1], b.l[0]); COL_OUT(T[1]);
  COL_ADD(a.l[0], b.l[2]); COL_ADD(a.l[1], b.l[1]); COL_ADD(a.l[2], b.l[0]); COL_OUT(T[2]);
  COL_ADD(a.l[0], b.l[3]); COL_ADD(a.l[1], b.l[2]); COL_ADD(a.l[2], b.l[1]); COL_ADD(a.l[3], b.l[0]); COL_OUT(T[3]);
  COL_ADD(a.l[1], b.l[3]); COL_ADD(a.l[2], b.l[2]); COL_ADD(a.l[3], b.l[1]); COL_OUT(T[4]);
  COL_ADD(a.l[2], b.l[3]); COL_ADD(a.l[3], b.l[2]); COL_OUT(T[5]);
  COL_ADD(a.l[3], b.l[3]); COL_OUT(T[6]);
  T[7] = c0;
}
static inline F mul_ps(const F &a, const F &b) {
  uint64_t T[8], o[4];
  prod_scan(a, b, T);
  mont_reduce_wide(T, o);
  return F{{o[0], o[1], o[2], o[3]}};
}
// reduction with the four quotient digits computed first (each needs only the running low word), then one pass of products
static inline F mul_ps2(const F &a, const F &b) {
  uint64_t T[8];
  prod_scan(a, b, T);
  // quotient digits: m_i = (T_i + carry-ins) * INV; the low-word recurrence only
  uint64_t m[4];
  u128 acc;
  // step 0
  m[0] = T[0] * INV;
  acc = (u128)m[0] * P[0] + T[0];
  uint64_t k1 = (uint64_t)(acc >> 64);                       // carry into word 1 from m0*P0
  u128 w1 = (u128)m[0] * P[1] + T[1] + k1;                    // word 1 after step 0
  m[1] = (uint64_t)w1 * INV;
  u128 w1b = (u128)m[1] * P[0] + (uint64_t)w1;                // low word of step 1 (zero), carry out
  u128 w2 = (u128)m[0] * P[2] + T[2] + (uint64_t)(w1 >> 64);
  u128 w2b = (u128)m[1] * P[1] + (uint64_t)w2 + (uint64_t)(w1b >> 64);
  m[2] = (uint64_t)w2b * INV;
  u128 w2c = (u128)m[2] * P[0] + (uint64_t)w2b;
  u128 w3 = (u128)m[0] * P[3] + T[3] + (uint64_t)(w2 >> 64);
  u128 w3b = (u128)m[1] * P[2] + (uint64_t)w3 + (uint64_t)(w2b >> 64);
  u128 w3c = (u128)m[2] * P[1] + (uint64_t)w3b + (uint64_t)(w2c >> 64);
  m[3] = (uint64_t)w3c * INV;
  u128 w3d = (u128)m[3] * P[0] + (uint64_t)w3c;
  // words 4..7: T[4..7] + the remaining products and the carries
  u128 w4 = (u128)T[4] + (uint64_t)(w3 >> 64) + (uint64_t)(w3b >> 64) + (uint64_t)(w3c >> 64) + (uint64_t)(w3d >> 64);
  w4 += (u128)m[1] * P[3];
  u128 x4 = (u128)m[2] * P[2] + (uint64_t)w4;
  u128 y4 = (u128)m[3] * P[1] + (uint64_t)x4;
  uint64_t o0 = (uint64_t)y4;
  u128 w5 = (u128)T[5] + (uint64_t)(w4 >> 64) + (uint64_t)(x4 >> 64) + (uint64_t)(y4 >> 64);
  w5 += (u128)m[2] * P[3];
  u128 x5 = (u128)m[3] * P[2] + (uint64_t)w5;
  uint64_t o1 = (uint64_t)x5;
  u128 w6 = (u128)T[6] + (uint64_t)(w5 >> 64) + (uint64_t)(x5 >> 64);
  w6 += (u128)m[3] * P[3];
  uint64_t o2 = (uint64_t)w6;
  uint64_t o3 = T[7] + (uint64_t)(w6 >> 64);
  return F{{o0, o1, o2, o3}};
}
int main() {
  F a = from_canon(U256{{7, 1, 2, 3}}), b = from_canon(U256{{11, 5, 9, 1}});
  for (int i = 0; i < 1000; ++i) {
    const F r0 = mulw(a, b), r1 = mul_ps(a, b), r2 = mul_ps2(a, b);
    const U256 u0 = to_canon(r0), u1 = to_canon(r1), u2 = to_canon(r2);
    if (memcmp(u0.l, u1.l, 32) || memcmp(u0.l, u2.l, 32)) { printf("MISMATCH %d\n", i); return 1; }
    a = r0; b = addw(b, a);
  }
  const int n = 20000000;
  F s = a;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) s = mulw(s, b);
  double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
  printf("CIOS (mulw)            %.2f ns  (%llx)\n", ns / n, (unsigned long long)s.l[0]);
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) s = mul_ps(s, b);
  ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
  printf("product scan + reduce   %.2f ns  (%llx)\n", ns / n, (unsigned long long)s.l[0]);
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) s = mul_ps2(s, b);
  ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
  printf("product scan + digits   %.2f ns  (%llx)\n", ns / n, (unsigned long long)s.l[0]);
}
