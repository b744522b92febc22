// Tile NTT kernel (2^3..2^13 points, one workgroup per column) -- see ntt.hip for the design notes.
// Instantiated one tile size per translation unit (ntt_tile_inst.hip, -DZK_TILE_LOGN=k) so the
// sizes build in parallel.
#pragma once
#include "ctx.hpp"

namespace zk {

__device__ __forceinline__ uint4 lo4(const Fr &a) { return make_uint4(a.l[0], a.l[1], a.l[2], a.l[3]); }
__device__ __forceinline__ uint4 hi4(const Fr &a) { return make_uint4(a.l[4], a.l[5], a.l[6], a.l[7]); }
__device__ __forceinline__ void set_lo(Fr &a, uint4 v) { a.l[0] = v.x; a.l[1] = v.y; a.l[2] = v.z; a.l[3] = v.w; }
__device__ __forceinline__ void set_hi(Fr &a, uint4 v) { a.l[4] = v.x; a.l[5] = v.y; a.l[6] = v.z; a.l[7] = v.w; }

// padded LDS slot of logical position pos (one 16-byte slot per position, +1 slot every 8)
__device__ __forceinline__ int pidx(int pos) { return pos + (pos >> 3); }

__device__ __forceinline__ void bfly(Fr &a, Fr &b) {
  Fr t = a - b;
  a = a + b;
  b = t;
}

// y[s] = sum_t v[t] w_R^(s t), natural order, in place.  w4 = w_R^(R/4), w8 = w_8, w83 = w_8^3.
__device__ __forceinline__ void dft2(Fr &v0, Fr &v1) { bfly(v0, v1); }

__device__ __forceinline__ void dft4(Fr &v0, Fr &v1, Fr &v2, Fr &v3, const Fr &w4) {
  bfly(v0, v2);  // v0 = a0, v2 = a1
  bfly(v1, v3);  // v1 = b0, v3 = (v1 - v3)
  v3 = v3 * w4;  // b1
  bfly(v0, v1);  // v0 = y0, v1 = y2
  bfly(v2, v3);  // v2 = y1, v3 = y3
  Fr t = v1;
  v1 = v2;
  v2 = t;
}

__device__ __forceinline__ void dft8(Fr (&r)[8], const Fr &w4, const Fr &w8, const Fr &w83) {
  Fr &v0 = r[0], &v1 = r[1], &v2 = r[2], &v3 = r[3], &v4 = r[4], &v5 = r[5], &v6 = r[6], &v7 = r[7];
  // even part E = DFT4(v0, v2, v4, v6), odd part O = DFT4(v1, v3, v5, v7)
  dft4(v0, v2, v4, v6, w4);  // v0=E0 v2=E1 v4=E2 v6=E3
  dft4(v1, v3, v5, v7, w4);  // v1=O0 v3=O1 v5=O2 v7=O3
  v3 = v3 * w8;
  v5 = v5 * w4;
  v7 = v7 * w83;
  // y[s] = E[s] + O'[s], y[s+4] = E[s] - O'[s]
  bfly(v0, v1);  // v0 = y0, v1 = y4
  bfly(v2, v3);  // v2 = y1, v3 = y5
  bfly(v4, v5);  // v4 = y2, v5 = y6
  bfly(v6, v7);  // v6 = y3, v7 = y7
  Fr y1 = v2, y2 = v4, y3 = v6, y4 = v1, y5 = v3, y6 = v5;
  v1 = y1; v2 = y2; v3 = y3; v4 = y4; v5 = y5; v6 = y6;
}

struct TileArgs {
  const Fr *in;        // column c, tile b at in + c*col_stride_in + b*in_tile_stride (contiguous run of N)
  size_t in_tile_stride;  // N for independent tiles, 0 when every tile of a column reads the same coefficients
  Fr *out;             // element q of (c,b) goes to out + c*col_stride_out + out_off(b) + q*out_stride
  size_t col_stride_in, col_stride_out;
  const Fr *tw;        // omega_N^j (or omega_N^-j), j < N
  const Fr *pre;       // optional per-position multiplier applied at load: x[q] *= pre[pre_off(c,b) + q]
  size_t pre_tile_stride;  // pre offset per tile (0 = same table for all tiles)
  const Fr *post;      // optional single multiplier applied at store (n^-1)
  int log_tiles;       // tiles per column = 2^log_tiles; out_off(b) = bitrev(b), out_stride = 2^log_tiles
  int in_len;          // coefficients actually present per tile (rest read as zero)
  int out_natural_tiles;  // 1: out_off(b) = b*N, stride 1 (independent tiles)
};

template <int LOGN, int R_LOG, int P_LOG>
__device__ __forceinline__ void stockham_pass(Fr (&reg)[8], const Fr *__restrict__ tw, uint4 *lds, int tid) {
  constexpr int N = 1 << LOGN, T = N / 8, R = 1 << R_LOG, U = 8 / R, P = 1 << P_LOG;
  constexpr bool LAST = (P_LOG + R_LOG == LOGN);
  const Fr w4 = tw[N / 4];
  // outer twiddles w_{PR}^(k t) = omega^(k t N/(P R))
  if (P_LOG > 0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = tid + u * T;
      const int k = i & (P - 1);
#pragma unroll
      for (int t = 1; t < R; ++t) {
        const int e = k * t * (N / (P * R));
        reg[u + t * U] = reg[u + t * U] * tw[e];
      }
    }
  }
  if (R == 8) {
    const Fr w8 = tw[N / 8], w83 = tw[3 * (N / 8)];
    dft8(reg, w4, w8, w83);
  } else if (R == 4) {
    dft4(reg[0], reg[2], reg[4], reg[6], w4);
    dft4(reg[1], reg[3], reg[5], reg[7], w4);
  } else {
#pragma unroll
    for (int u = 0; u < 4; ++u) dft2(reg[u], reg[u + 4]);
  }
  if (!LAST) {
    uint4 tlo[8];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = tid + u * T;
      const int k = i & (P - 1);
      const int j = ((i - k) << R_LOG) + k;
#pragma unroll
      for (int s = 0; s < R; ++s) lds[pidx(j + s * P)] = lo4(reg[u + s * U]);
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 8; ++m) tlo[m] = lds[pidx(tid + T * m)];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = tid + u * T;
      const int k = i & (P - 1);
      const int j = ((i - k) << R_LOG) + k;
#pragma unroll
      for (int s = 0; s < R; ++s) lds[pidx(j + s * P)] = hi4(reg[u + s * U]);
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      set_hi(reg[m], lds[pidx(tid + T * m)]);
      set_lo(reg[m], tlo[m]);
    }
  }
}

template <int LOGN, int P_LOG>
__device__ __forceinline__ void run_passes(Fr (&reg)[8], const Fr *__restrict__ tw, uint4 *lds, int tid) {
  if constexpr (P_LOG < LOGN) {
    constexpr int REM = LOGN - P_LOG;
    constexpr int R_LOG = REM >= 3 ? 3 : REM;
    stockham_pass<LOGN, R_LOG, P_LOG>(reg, tw, lds, tid);
    run_passes<LOGN, P_LOG + R_LOG>(reg, tw, lds, tid);
  }
}

__device__ __forceinline__ unsigned brev(unsigned x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

template <int LOGN>
__global__ void __launch_bounds__((1 << LOGN) / 8) k_ntt_tile(TileArgs a) {
  constexpr int N = 1 << LOGN, T = N / 8;
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int tid = threadIdx.x;
  const unsigned b = blockIdx.x;  // tile within column
  const size_t c = blockIdx.y;
  const Fr *__restrict__ src = a.in + c * a.col_stride_in + (size_t)b * a.in_tile_stride;
  Fr reg[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int q = tid + T * m;
    reg[m] = q < a.in_len ? src[q] : Fr::zero();
  }
  if (a.pre) {
    const Fr *__restrict__ pre = a.pre + (size_t)b * a.pre_tile_stride;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int q = tid + T * m;
      if (q < a.in_len) reg[m] = reg[m] * pre[q];
    }
  }
  run_passes<LOGN, 0>(reg, a.tw, lds, tid);
  if (a.post) {
    const Fr s = *a.post;
#pragma unroll
    for (int m = 0; m < 8; ++m) reg[m] = reg[m] * s;
  }
  Fr *__restrict__ dst = a.out + c * a.col_stride_out;
  if (a.out_natural_tiles) {
    dst += (size_t)b * N;
#pragma unroll
    for (int m = 0; m < 8; ++m) dst[tid + T * m] = reg[m];
  } else {
    const size_t off = brev(b, a.log_tiles);
    const size_t stride = (size_t)1 << a.log_tiles;
#pragma unroll
    for (int m = 0; m < 8; ++m) dst[off + (size_t)(tid + T * m) * stride] = reg[m];
  }
}

template <int LOGN>
inline int launch_tile(zkfhe_ctx *ctx, const TileArgs &a, unsigned tiles, unsigned cols) {
  constexpr int N = 1 << LOGN;
  constexpr size_t lds_bytes = (size_t)(N + N / 8) * 16;
  static bool attr_set = false;
  if (!attr_set) {
    ZK_HIP(ctx, hipFuncSetAttribute((const void *)k_ntt_tile<LOGN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_set = true;
  }
  dim3 grid(tiles, cols);
  zk_prof_begin(ctx);
  k_ntt_tile<LOGN><<<grid, N / 8, lds_bytes, ctx->stream>>>(a);
  ZK_LAUNCH_CHECK(ctx);
  zk_prof_end(ctx, 1, 64.0 * (double)N * (double)tiles * (double)cols);
  if (ctx->prof_on) ctx->prof_ops[1] += 0.5 * (double)N * LOGN * (double)tiles * (double)cols;  // butterflies
  return ZKFHE_OK;
}


}  // namespace zk
