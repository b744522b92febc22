// Would a 2^13 tile of radix-4 passes with four coefficients per thread (<= 128 VGPRs: four waves per SIMD) issue more field
// products per second than the radix-8 passes of k_ntt13 (eight coefficients per thread, 246 VGPRs: two waves per SIMD)?
// Both kernels run the real arithmetic of csrc/lz29.hip.hpp -- butterflies, twiddle products with table loads, the weak
// reduction -- in a loop, without the LDS exchanges (which are 2-3 % of k_ntt13): products per second is what is compared.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../zk-fhe_amd/csrc/lz29.hip.hpp"
using namespace zk;

#define ZK_F __builtin_amdgcn_sched_barrier(0);
template <int LO, int HI, int V>
__device__ __forceinline__ LzT mulw(const Lz<LO, HI, V> &x, const Lw &w) {
  __builtin_amdgcn_sched_barrier(0);
  const LzT r = lz_mul(x, w);
  __builtin_amdgcn_sched_barrier(0);
  return r;
}
struct Consts {
  Lw w4, w8, w83;
};
#define ZK_DFT8_CORE(x, K)                                                                                           \
  const auto a0 = lz_add(x[0], x[4]);                                                                                \
  const auto a1 = lz_sub(x[0], x[4]);                                                                                \
  const auto b0 = lz_add(x[2], x[6]);                                                                                \
  ZK_F const LzT b1 = mulw(lz_sub(x[2], x[6]), K.w4);                                                                \
  ZK_F const auto E0 = lz_norm(lz_add(a0, b0));                                                                      \
  ZK_F const auto E2 = lz_norm(lz_sub(a0, b0));                                                                      \
  ZK_F const auto E1 = lz_norm(lz_add(a1, b1));                                                                      \
  ZK_F const auto E3 = lz_norm(lz_sub(a1, b1));                                                                      \
  ZK_F const auto c0 = lz_add(x[1], x[5]);                                                                           \
  const auto c1 = lz_sub(x[1], x[5]);                                                                                \
  const auto d0 = lz_add(x[3], x[7]);                                                                                \
  ZK_F const LzT d1 = mulw(lz_sub(x[3], x[7]), K.w4);                                                                \
  ZK_F const auto O0 = lz_norm(lz_add(c0, d0));                                                                      \
  ZK_F const LzT O2 = mulw(lz_sub(c0, d0), K.w4);                                                                    \
  ZK_F const LzT O1 = mulw(lz_add(c1, d1), K.w8);                                                                    \
  ZK_F const LzT O3 = mulw(lz_sub(c1, d1), K.w83);                                                                   \
  ZK_F const auto y0 = lz_add(E0, O0);                                                                               \
  const auto y4 = lz_sub(E0, O0);                                                                                    \
  ZK_F const auto y1 = lz_add(E1, O1);                                                                               \
  const auto y5 = lz_sub(E1, O1);                                                                                    \
  ZK_F const auto y2 = lz_add(E2, O2);                                                                               \
  const auto y6 = lz_sub(E2, O2);                                                                                    \
  ZK_F const auto y3 = lz_add(E3, O3);                                                                               \
  const auto y7 = lz_sub(E3, O3);                                                                                    \
  ZK_F

__device__ __forceinline__ void pass8_tw(LzT (&x)[8], const Consts &K, const LwMem *__restrict__ tw, int stride) {
  const Lw t1 = lw_load(tw[0]), t2 = lw_load(tw[stride]), t3 = lw_load(tw[2 * stride]), t4 = lw_load(tw[3 * stride]), t5 = lw_load(tw[4 * stride]),
           t6 = lw_load(tw[5 * stride]), t7 = lw_load(tw[6 * stride]);
  ZK_F
  ZK_DFT8_CORE(x, K)
  x[0] = lz_weak(y0); ZK_F
  x[1] = mulw(y1, t1);
  x[2] = mulw(y2, t2);
  x[3] = mulw(y3, t3);
  x[4] = mulw(y4, t4);
  x[5] = mulw(y5, t5);
  x[6] = mulw(y6, t6);
  x[7] = mulw(y7, t7);
}
// radix 4: no carry propagation at all (limb ranges stay inside (2, 2), values below 8 r)
__device__ __forceinline__ void pass4_tw(LzT (&x)[4], const Lw &w4, const LwMem *__restrict__ tw, int stride) {
  const Lw t1 = lw_load(tw[0]), t2 = lw_load(tw[stride]), t3 = lw_load(tw[2 * stride]);
  ZK_F
  const auto a0 = lz_add(x[0], x[2]);
  const auto a1 = lz_sub(x[0], x[2]);
  const auto b0 = lz_add(x[1], x[3]);
  ZK_F const LzT b1 = mulw(lz_sub(x[1], x[3]), w4);
  ZK_F const auto y0 = lz_add(a0, b0);
  const auto y2 = lz_sub(a0, b0);
  const auto y1 = lz_add(a1, b1);
  const auto y3 = lz_sub(a1, b1);
  ZK_F
  x[0] = lz_weak(y0); ZK_F
  x[1] = mulw(y1, t1);
  x[2] = mulw(y2, t2);
  x[3] = mulw(y3, t3);
}

__global__ void __launch_bounds__(512, 2) k_r8(const Fr *in, const LwMem *tw, Fr *out, int iters) {
  const int tid = threadIdx.x, gid = blockIdx.x * 512 + tid;
  Consts K;
  K.w8 = lw_load(tw[1]), K.w4 = lw_load(tw[2]), K.w83 = lw_load(tw[3]);
  LzT x[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) x[m] = lz_weak(lz_load(in[(gid * 8 + m) & 65535]));
  for (int it = 0; it < iters; ++it) pass8_tw(x, K, tw + ((tid + it * 7) & 511), 512);
  Fr acc = Fr::zero();
#pragma unroll
  for (int m = 0; m < 8; ++m) acc = acc + lz_store(x[m]);
  out[gid] = acc;
}
__global__ void __launch_bounds__(1024, 4) k_r4(const Fr *in, const LwMem *tw, Fr *out, int iters) {
  const int tid = threadIdx.x, gid = blockIdx.x * 1024 + tid;
  const Lw w4 = lw_load(tw[2]);
  LzT x[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) x[m] = lz_weak(lz_load(in[(gid * 4 + m) & 65535]));
  for (int it = 0; it < iters; ++it) pass4_tw(x, w4, tw + ((tid + it * 7) & 1023), 1024);
  Fr acc = Fr::zero();
#pragma unroll
  for (int m = 0; m < 4; ++m) acc = acc + lz_store(x[m]);
  out[gid] = acc;
}

int main() {
  Fr *in, *out;
  LwMem *tw;
  hipMalloc(&in, 65536 * 32), hipMalloc(&out, (size_t)1 << 26), hipMalloc(&tw, 8192 * sizeof(LwMem));
  hipMemset(in, 1, 65536 * 32), hipMemset(tw, 1, 8192 * sizeof(LwMem));
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int iters = 200;
  for (int rep = 0; rep < 2; ++rep) {
    // the same number of coefficients in flight: 512 workgroups x 4096 points (one half column each), as k_ntt13's grid for 256 columns
    float ms8, ms4;
    k_r8<<<512, 512>>>(in, tw, out, 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_r8<<<512, 512>>>(in, tw, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms8, e0, e1);
    k_r4<<<512, 1024>>>(in, tw, out, 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_r4<<<512, 1024>>>(in, tw, out, iters * 3 / 2);   // a radix-8 pass is three stages, a radix-4 pass two: equal stages
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms4, e0, e1);
    const double p8 = 512.0 * 512 * iters * 12, p4 = 512.0 * 1024 * (iters * 3 / 2) * 4;
    printf("radix 8, 8 per thread (2 waves/SIMD): %.3f ms, %.1f G products/s | radix 4, 4 per thread (4 waves/SIMD): %.3f ms, %.1f G products/s | time ratio %.3f\n", ms8,
           p8 / ms8 / 1e6, ms4, p4 / ms4 / 1e6, ms4 / ms8);
  }
  printf("(k_ntt13: 13 stages of 256 x 8192 points in 0.150 ms = 6 products per point -> 84 G products/s)\n");
  return 0;
}
