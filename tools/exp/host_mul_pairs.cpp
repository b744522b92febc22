#include <cstdint>
#include <cstdio>
#include <cstring>
#include <chrono>
typedef unsigned __int128 u128;
typedef unsigned long long ull;
static const uint64_t P[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const uint64_t INV = 0xc2e1f593efffffffULL;
struct F { uint64_t l[4]; };
#define AI static inline __attribute__((always_inline))
AI F reduce_once(uint64_t t0, uint64_t t1, uint64_t t2, uint64_t t3) {
  unsigned long long br;
  const uint64_t r0 = __builtin_subcll(t0, P[0], 0, &br);
  const uint64_t r1 = __builtin_subcll(t1, P[1], br, &br);
  const uint64_t r2 = __builtin_subcll(t2, P[2], br, &br);
  const uint64_t r3 = __builtin_subcll(t3, P[3], br, &br);
  const uint64_t keep = (uint64_t)0 - (uint64_t)br;
  return F{{(t0 & keep) | (r0 & ~keep), (t1 & keep) | (r1 & ~keep), (t2 & keep) | (r2 & ~keep), (t3 & keep) | (r3 & ~keep)}};
}
struct Acc { uint64_t t0, t1, t2, t3; };
AI void row(Acc &t, const F &a, uint64_t bi) {
  u128 A = (u128)a.l[0] * bi + t.t0;
  const uint64_t m = (uint64_t)A * INV;
  u128 C = (u128)m * P[0] + (uint64_t)A;
  A = (u128)a.l[1] * bi + t.t1 + (uint64_t)(A >> 64);
  C = (u128)m * P[1] + (uint64_t)A + (uint64_t)(C >> 64);
  t.t0 = (uint64_t)C;
  A = (u128)a.l[2] * bi + t.t2 + (uint64_t)(A >> 64);
  C = (u128)m * P[2] + (uint64_t)A + (uint64_t)(C >> 64);
  t.t1 = (uint64_t)C;
  A = (u128)a.l[3] * bi + t.t3 + (uint64_t)(A >> 64);
  C = (u128)m * P[3] + (uint64_t)A + (uint64_t)(C >> 64);
  t.t2 = (uint64_t)C;
  t.t3 = (uint64_t)(C >> 64) + (uint64_t)(A >> 64);
}
AI F mul(const F &a, const F &b) {
  Acc t = {0,0,0,0};
  row(t,a,b.l[0]); row(t,a,b.l[1]); row(t,a,b.l[2]); row(t,a,b.l[3]);
  return reduce_once(t.t0,t.t1,t.t2,t.t3);
}
AI void mul2(const F &a1, const F &b1, const F &a2, const F &b2, F &r1, F &r2) {
  Acc t = {0,0,0,0}, u = {0,0,0,0};
  row(t,a1,b1.l[0]); row(u,a2,b2.l[0]);
  row(t,a1,b1.l[1]); row(u,a2,b2.l[1]);
  row(t,a1,b1.l[2]); row(u,a2,b2.l[2]);
  row(t,a1,b1.l[3]); row(u,a2,b2.l[3]);
  r1 = reduce_once(t.t0,t.t1,t.t2,t.t3);
  r2 = reduce_once(u.t0,u.t1,u.t2,u.t3);
}
AI void mul3(const F &a1, const F &b1, const F &a2, const F &b2, const F &a3, const F &b3, F &r1, F &r2, F &r3) {
  Acc t = {0,0,0,0}, u = {0,0,0,0}, v = {0,0,0,0};
  for (int i = 0; i < 4; ++i) { row(t,a1,b1.l[i]); row(u,a2,b2.l[i]); row(v,a3,b3.l[i]); }
  r1 = reduce_once(t.t0,t.t1,t.t2,t.t3);
  r2 = reduce_once(u.t0,u.t1,u.t2,u.t3);
  r3 = reduce_once(v.t0,v.t1,v.t2,v.t3);
}
int main() {
  const int N = 10000000;
  F a = {{1,2,3,4}}, b = {{5,6,7,0x0fffffffffffffffULL}}, c = {{7,7,7,7}}, d = {{9,1,2,3}};
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; ++i) { a = mul(a, b); }
  auto t1 = std::chrono::steady_clock::now();
  printf("mul dependent: %.2f ns (%llx)\n", std::chrono::duration<double, std::nano>(t1 - t0).count() / N, (ull)a.l[0]);
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; ++i) { mul2(a, b, c, b, a, c); }
  t1 = std::chrono::steady_clock::now();
  printf("mul2 dependent: %.2f ns per pair (%llx)\n", std::chrono::duration<double, std::nano>(t1 - t0).count() / N, (ull)(a.l[0]^c.l[0]));
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; ++i) { mul3(a, b, c, b, d, b, a, c, d); }
  t1 = std::chrono::steady_clock::now();
  printf("mul3 dependent: %.2f ns per triple (%llx)\n", std::chrono::duration<double, std::nano>(t1 - t0).count() / N, (ull)(a.l[0]^c.l[0]^d.l[0]));
}
