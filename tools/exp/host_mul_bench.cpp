#include <cstdint>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <random>
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
AI F mul_c(const F &a, const F &b) {
  Acc t = {0,0,0,0};
  row(t,a,b.l[0]); row(t,a,b.l[1]); row(t,a,b.l[2]); row(t,a,b.l[3]);
  return reduce_once(t.t0,t.t1,t.t2,t.t3);
}
// t(4 limbs) + x * y(4 limbs) -> (t0..t3, A), two carry chains
#define ROW(x0, x1, x2, x3, mult)                                                                                  \
  asm("xorl %%eax, %%eax\n\t"                                                                                     \
      "mulx %[a0], %[lo], %[hi]\n\t"                                                                              \
      "adox %[lo], %[t0]\n\t"                                                                                     \
      "adcx %[hi], %[t1]\n\t"                                                                                     \
      "mulx %[a1], %[lo], %[hi]\n\t"                                                                              \
      "adox %[lo], %[t1]\n\t"                                                                                     \
      "adcx %[hi], %[t2]\n\t"                                                                                     \
      "mulx %[a2], %[lo], %[hi]\n\t"                                                                              \
      "adox %[lo], %[t2]\n\t"                                                                                     \
      "adcx %[hi], %[t3]\n\t"                                                                                     \
      "mulx %[a3], %[lo], %[hi]\n\t"                                                                              \
      "adox %[lo], %[t3]\n\t"                                                                                     \
      "adcx %[hi], %[A]\n\t"                                                                                      \
      "movl $0, %%eax\n\t"                                                                                        \
      "adox %%rax, %[A]\n\t"                                                                                      \
      : [t0] "+r"(t0), [t1] "+r"(t1), [t2] "+r"(t2), [t3] "+r"(t3), [A] "+r"(A), [lo] "=&r"(lo), [hi] "=&r"(hi) \
      : [a0] "rm"(x0), [a1] "rm"(x1), [a2] "rm"(x2), [a3] "rm"(x3), "d"(mult)                                    \
      : "rax", "cc")
__attribute__((target("bmi2,adx"))) AI F mul_adx(const F &a, const F &b) {
  uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, A, lo, hi;
  const uint64_t a0 = a.l[0], a1 = a.l[1], a2 = a.l[2], a3 = a.l[3];
  const uint64_t q0 = P[0], q1 = P[1], q2 = P[2], q3 = P[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    A = 0;
    ROW(a0, a1, a2, a3, b.l[i]);
    const uint64_t m = t0 * INV;
    uint64_t B = 0;
    // (t0..t3, A) + m * q, low limb vanishes
    asm("xorl %%eax, %%eax\n\t"
        "mulx %[q0], %[lo], %[hi]\n\t"
        "adox %[lo], %[t0]\n\t"
        "adcx %[hi], %[t1]\n\t"
        "mulx %[q1], %[lo], %[hi]\n\t"
        "adox %[lo], %[t1]\n\t"
        "adcx %[hi], %[t2]\n\t"
        "mulx %[q2], %[lo], %[hi]\n\t"
        "adox %[lo], %[t2]\n\t"
        "adcx %[hi], %[t3]\n\t"
        "mulx %[q3], %[lo], %[hi]\n\t"
        "adox %[lo], %[t3]\n\t"
        "adcx %[hi], %[A]\n\t"
        "movl $0, %%eax\n\t"
        "adox %%rax, %[A]\n\t"
        : [t0] "+r"(t0), [t1] "+r"(t1), [t2] "+r"(t2), [t3] "+r"(t3), [A] "+r"(A), [lo] "=&r"(lo), [hi] "=&r"(hi)
        : [q0] "rm"(q0), [q1] "rm"(q1), [q2] "rm"(q2), [q3] "rm"(q3), "d"(m)
        : "rax", "cc");
    (void)B;
    t0 = t1; t1 = t2; t2 = t3; t3 = A;
  }
  return reduce_once(t0, t1, t2, t3);
}
template <class FN> void bench(const char *name, FN fn) {
  const int N = 20000000;
  F a = {{1,2,3,4}}, b = {{5,6,7,0x0fffffffffffffffULL}};
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; ++i) a = fn(a, b);
  auto t1 = std::chrono::steady_clock::now();
  printf("%s dependent: %.2f ns (%llx)", name, std::chrono::duration<double, std::nano>(t1 - t0).count() / N, (ull)a.l[0]);
  F x[4] = {{{1,2,3,4}},{{2,3,4,5}},{{3,4,5,6}},{{9,9,9,9}}};
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < N/4; ++i) { for (int k=0;k<4;++k) x[k] = fn(x[k], b); }
  t1 = std::chrono::steady_clock::now();
  printf("   4 independent: %.2f ns (%llx)\n", std::chrono::duration<double, std::nano>(t1 - t0).count() / N, (ull)(x[0].l[0]^x[1].l[0]^x[2].l[0]^x[3].l[0]));
}
int main() {
  std::mt19937_64 g(1);
  for (int it = 0; it < 200000; ++it) {
    F a, b;
    for (int i = 0; i < 4; ++i) { a.l[i] = g(); b.l[i] = g(); }
    a.l[3] &= 0x0fffffffffffffffULL; b.l[3] &= 0x0fffffffffffffffULL;
    if (it == 0) { for (int i=0;i<4;++i) a.l[i] = P[i]; a.l[0] -= 1; b = a; }
    F x = mul_c(a, b), y = [&]() __attribute__((target("bmi2,adx"))) { return mul_adx(a, b); }();
    if (memcmp(&x, &y, 32)) { printf("MISMATCH at %d\n", it); return 1; }
  }
  printf("adx == c on 200000 random pairs\n");
  bench("C  ", [](const F&a,const F&b){return mul_c(a,b);});
  bench("ADX", [](const F&a,const F&b) __attribute__((target("bmi2,adx"))) {return mul_adx(a,b);});
}
