// Prototype: carry-free Montgomery product in radix 2^29 (9 limbs, R = 2^261) against the 8x32 FIPS product.
#include "../../zk-fhe_amd/csrc/bn254.hip.hpp"
#include <cstdio>
#include <vector>
using namespace zk;

struct F29 { u32 l[9]; };
constexpr u32 M29 = (1u << 29) - 1;
struct Q29 {  // Fq in radix 2^29
  static __host__ __device__ constexpr u32 limb(int i) {
    // filled at runtime on the host and passed by value; placeholder
    return 0;
  }
};
struct Mod29 { u32 p[9]; u32 inv; };

__device__ __forceinline__ F29 f29_mul(const F29 &a, const F29 &b, const Mod29 &md) {
  u32 m[9];
  F29 r;
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int j = 0; j < k; ++j) {
      acc += (u64)a.l[j] * b.l[k - j];
      acc += (u64)m[j] * md.p[k - j];
    }
    acc += (u64)a.l[k] * b.l[0];
    m[k] = ((u32)acc * md.inv) & M29;
    acc += (u64)m[k] * md.p[0];
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 18; ++k) {
#pragma unroll
    for (int j = k - 8; j < 9; ++j) {
      acc += (u64)a.l[j] * b.l[k - j];
      acc += (u64)m[j] * md.p[k - j];
    }
    r.l[k - 9] = (u32)acc & M29;
    acc >>= 29;
  }
  return r;
}

__global__ void k29(const F29 *in, F29 *out, Mod29 md, int iters, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  F29 x = in[i], y = in[(i + 1) % n];
  for (int it = 0; it < iters; ++it) x = f29_mul(x, y, md);
  out[i] = x;
}
__global__ void k32(const Fq *in, Fq *out, int iters, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fq x = in[i], y = in[(i + 1) % n];
  for (int it = 0; it < iters; ++it) x = x * y;
  out[i] = x;
}

int main() {
  const size_t n = 1 << 20;
  const int iters = 256;
  Mod29 md;
  // p in radix 2^29 and -p^-1 mod 2^29
  unsigned __int128 dummy = 0; (void)dummy;
  u32 pw[8];
  for (int i = 0; i < 8; ++i) pw[i] = FqP::MOD[i];
  for (int i = 0; i < 9; ++i) {
    const int bit = 29 * i, w = bit >> 5, sh = bit & 31;
    u64 v = pw[w] >> sh;
    if (w + 1 < 8) v |= (u64)pw[w + 1] << (32 - sh);
    md.p[i] = (u32)v & M29;
  }
  md.inv = (0u - (0u - FqP::INV)) ;  // FqP::INV = -p^-1 mod 2^32
  md.inv = FqP::INV & M29;           // -p^-1 mod 2^29
  std::vector<F29> h(n);
  std::vector<Fq> h32(n);
  unsigned s = 12345;
  for (size_t i = 0; i < n; ++i) {
    for (int j = 0; j < 9; ++j) { s = s * 1664525u + 1013904223u; h[i].l[j] = (s >> 3) & M29; }
    h[i].l[8] &= (1u << 21) - 1;  // < 2^253
    for (int j = 0; j < 8; ++j) { s = s * 1664525u + 1013904223u; h32[i].l[j] = s; }
    h32[i].l[7] &= 0x0fffffff;
  }
  F29 *d_in, *d_out;
  Fq *e_in, *e_out;
  hipMalloc(&d_in, n * sizeof(F29)); hipMalloc(&d_out, n * sizeof(F29));
  hipMalloc(&e_in, n * sizeof(Fq)); hipMalloc(&e_out, n * sizeof(Fq));
  hipMemcpy(d_in, h.data(), n * sizeof(F29), hipMemcpyHostToDevice);
  hipMemcpy(e_in, h32.data(), n * sizeof(Fq), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    k29<<<n / 256, 256>>>(d_in, d_out, md, iters, n);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("radix 2^29 (9 limbs): %.3f ms  %.1f G modmul/s\n", ms, (double)n * iters / ms / 1e6);
    hipEventRecord(e0);
    k32<<<n / 256, 256>>>(e_in, e_out, iters, n);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("radix 2^32 FIPS (8 limbs): %.3f ms  %.1f G modmul/s\n", ms, (double)n * iters / ms / 1e6);
  }
  // correctness sample: one product, printed as integers for an offline check
  k29<<<1, 64>>>(d_in, d_out, md, 1, 64);
  std::vector<F29> o(64);
  hipMemcpy(o.data(), d_out, 64 * sizeof(F29), hipMemcpyDeviceToHost);
  for (int t = 0; t < 2; ++t) {
    printf("a=["); for (int j = 0; j < 9; ++j) printf("%u,", h[t].l[j]); printf("] b=["); for (int j = 0; j < 9; ++j) printf("%u,", h[t + 1].l[j]);
    printf("] r=["); for (int j = 0; j < 9; ++j) printf("%u,", o[t].l[j]); printf("]\n");
  }
  return 0;
}
