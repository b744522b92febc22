// Issue-rate probe: v_mad_u64_u32 alone, with the carry pair (s_nop + v_addc_co), v_mul_lo_u32, v_add_co chains.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned u32;
template <int MODE>
__global__ void k(u64 *out, u32 a, u32 b, int iters) {
  u64 acc[8];
  u32 top[8];
  for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i, top[i] = 0;
  u32 x = a + threadIdx.x, y = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y) : "vcc");
        if (MODE == 1) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc[i]), "+v"(top[i]) : "v"(x), "v"(y) : "vcc");
        if (MODE == 2) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(top[i]) : "v"(x), "v"(top[i]));
        if (MODE == 3) asm volatile("v_add_co_u32 %0, vcc, %1, %0\n\tv_addc_co_u32 %0, vcc, %1, %0, vcc" : "+v"(top[i]) : "v"(x) : "vcc");
        if (MODE == 5) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y) : "vcc");
        if (MODE == 6) asm volatile("v_ashrrev_i64 %0, 29, %0" : "+v"(acc[i]));
        if (MODE == 7) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) & 7]));
        if (MODE == 8) asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(top[i]));
        if (MODE == 9) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[0]) : "v"(x), "v"(y) : "vcc");   // one dependent chain
        if (MODE == 4) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_add_u32 %1, %1, %2" : "+v"(acc[i]), "+v"(top[i]) : "v"(x), "v"(y) : "vcc");
      }
  }
  u64 s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i] + top[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name, int per_slot) {
  u64 *d;
  hipMalloc(&d, 1024 * 256 * 8 * 8);
  const int blocks = 256 * 8, iters = 2000;
  k<MODE><<<blocks, 256>>>(d, 3, 5, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(d, 3, 5, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double ops = (double)blocks * 256 * iters * 32;
  // cycles per wave-instruction-group per SIMD: 256 CUs * 4 SIMD, 2.4 GHz assumed
  const double waves = (double)blocks * 4, per_simd = waves / 1024.0;
  const double cyc = ms * 1e-3 * 2.4e9 / (per_simd * iters * 32);
  printf("%-28s %.3f ms  %.1f Gop/s  ~%.1f cycles per op per wave (2.4 GHz)\n", name, ms, ops / ms / 1e6, cyc);
  hipFree(d);
}
int main() {
  run<0>("v_mad_u64_u32", 1);
  run<1>("mad + s_nop 1 + addc", 3);
  run<2>("v_mul_lo_u32", 1);
  run<3>("add_co + addc_co", 2);
  run<4>("mad + v_add_u32", 2);
  run<5>("v_mad_i64_i32", 1);
  run<6>("v_ashrrev_i64", 1);
  run<7>("v_lshl_add_u64", 1);
  run<8>("v_and_b32", 1);
  run<9>("v_mad_i64_i32 dependent", 1);
  return 0;
}
