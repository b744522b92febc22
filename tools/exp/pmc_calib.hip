// Calibration of the memory-side PMC counters (FETCH_SIZE, WRITE_SIZE) on THIS library's access patterns, against known byte counts.
// The guide's x2 correction for FETCH_SIZE is stated for wide coalesced streaming reads only; the MSM kernels gather one 64-byte
// table point per lane out of tens of GB, the NTT / quotient kernels stream 32-byte field elements.  Every kernel below moves an
// exactly known number of bytes through a buffer far larger than the 256 MiB Infinity Cache (and touches every line once, so no
// cache level can serve it twice):
//   k_gather64   one 64-byte point (four dwordx4) per lane at a pseudo-random index of a 16 GiB table      -- k_msm_table's gather
//   k_gather64_window  the same, indices confined to a 64 MiB window that moves with the workgroup          -- k_msm_accumulate's 10 MB table
//   k_stream16   16 B per lane, coalesced                                                                     -- the guide's reference pattern
//   k_stream32   32 B per lane (one Fr), coalesced: what every column kernel of the prover reads
//   k_store16 / k_store32   the same widths written
// Run under rocprofv3 --pmc FETCH_SIZE (one pass) and --pmc WRITE_SIZE (another); tools/exp/pmc_calib.sh prints counter / known.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e__ = (x);                                                         \
    if (e__ != hipSuccess) {                                                      \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__));                    \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

struct alignas(16) P64 {
  uint4 a, b, c, d;
};

__device__ __forceinline__ unsigned long long mix(unsigned long long x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

// every lane gathers `per` points; a permutation-like index (odd multiplier mod 2^bits) touches each point at most once
__global__ void __launch_bounds__(256) k_gather64(const P64 *__restrict__ T, unsigned long long n_points_mask, unsigned per, uint4 *__restrict__ sink) {
  const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (unsigned i = 0; i < per; ++i) {
    const unsigned long long idx = ((gid * per + i) * 0x9E3779B97F4A7C15ULL >> 7) & n_points_mask;   // odd multiplier, then a shift: scattered, near-unique
    const P64 p = T[idx];
    acc.x ^= p.a.x ^ p.b.y ^ p.c.z ^ p.d.w;
    acc.y += p.a.y + p.d.x;
  }
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[gid & 1023] = acc;   // never true for the fill pattern: keeps the loads
}

__global__ void __launch_bounds__(256) k_gather64_window(const P64 *__restrict__ T, unsigned window_mask, unsigned per, uint4 *__restrict__ sink) {
  const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (unsigned i = 0; i < per; ++i) {
    const unsigned idx = (unsigned)mix(gid * per + i) & window_mask;
    const P64 p = T[idx];
    acc.x ^= p.a.x ^ p.b.y ^ p.c.z ^ p.d.w;
    acc.y += p.a.y + p.d.x;
  }
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[gid & 1023] = acc;
}

template <int VEC>   // VEC uint4 per lane and step, lanes adjacent
__global__ void __launch_bounds__(256) k_stream(const uint4 *__restrict__ src, size_t n_vec /* uint4 count */, uint4 *__restrict__ sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i * VEC < n_vec; i += stride) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const uint4 t = src[i * VEC + v];
      acc.x ^= t.x ^ t.w;
      acc.y += t.y + t.z;
    }
  }
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[threadIdx.x] = acc;
}
template <int VEC>
__global__ void __launch_bounds__(256) k_store(uint4 *__restrict__ dst, size_t n_vec) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i * VEC < n_vec; i += stride) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) dst[i * VEC + v] = make_uint4((unsigned)i, v, 3, 4);
  }
}
__global__ void __launch_bounds__(256) k_fill(uint4 *__restrict__ dst, size_t n_vec) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) dst[i] = make_uint4((unsigned)i * 2654435761u, 1, 2, 3);
}

int main() {
  const size_t table_bytes = (size_t)16 << 30;           // 16 GiB: 2^28 points of 64 B
  const size_t stream_bytes = (size_t)4 << 30;           // 4 GiB streamed
  void *T, *sink;
  CHECK(hipMalloc(&T, table_bytes));
  CHECK(hipMalloc(&sink, 1 << 20));
  k_fill<<<4096, 256>>>((uint4 *)T, table_bytes / 16);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto timed = [&](const char *name, double known_read, double known_write, auto &&launch) {
    CHECK(hipEventRecord(e0));
    launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("KNOWN %-18s read_bytes=%.0f write_bytes=%.0f ms=%.3f GBps=%.0f\n", name, known_read, known_write, ms, (known_read + known_write) / ms / 1e6);
  };
  const unsigned per = 16;
  const unsigned long long n_gather = (unsigned long long)1 << 25;   // 2^25 gathers of 64 B = 2 GiB of points
  for (int rep = 0; rep < 2; ++rep) {
    timed("k_gather64", 64.0 * n_gather, 0, [&] { k_gather64<<<(unsigned)(n_gather / per / 256), 256>>>((const P64 *)T, (table_bytes / 64) - 1, per, (uint4 *)sink); });
    timed("k_gather64_window", 64.0 * n_gather, 0, [&] { k_gather64_window<<<(unsigned)(n_gather / per / 256), 256>>>((const P64 *)T, (unsigned)(((size_t)8 << 20) / 64) - 1, per, (uint4 *)sink); });
    timed("k_stream16", (double)stream_bytes, 0, [&] { k_stream<1><<<8192, 256>>>((const uint4 *)T, stream_bytes / 16, (uint4 *)sink); });
    timed("k_stream32", (double)stream_bytes, 0, [&] { k_stream<2><<<8192, 256>>>((const uint4 *)T + (stream_bytes / 16), stream_bytes / 16, (uint4 *)sink); });
    timed("k_store16", 0, (double)stream_bytes, [&] { k_store<1><<<8192, 256>>>((uint4 *)T + 2 * (stream_bytes / 16), stream_bytes / 16); });
    timed("k_store32", 0, (double)stream_bytes, [&] { k_store<2><<<8192, 256>>>((uint4 *)T + 3 * (stream_bytes / 16), stream_bytes / 16); });
  }
  CHECK(hipFree(T));
  CHECK(hipFree(sink));
  return 0;
}
