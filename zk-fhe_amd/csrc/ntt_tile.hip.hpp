// Tile NTT kernel (2^3..2^13 points, one workgroup per column) -- see ntt.hip for the design notes.
// Instantiated one tile size per translation unit (ntt_tile_inst.hip, -DZK_TILE_LOGN=k) so the
// sizes build in parallel.
#pragma once
#include "ctx.hpp"
#include "fr29.hip.hpp"

namespace zk {

__device__ __forceinline__ uint4 lo4(const Fr &a) { return make_uint4(a.l[0], a.l[1], a.l[2], a.l[3]); }
__device__ __forceinline__ uint4 hi4(const Fr &a) { return make_uint4(a.l[4], a.l[5], a.l[6], a.l[7]); }
__device__ __forceinline__ void set_lo(Fr &a, uint4 v) { a.l[0] = v.x; a.l[1] = v.y; a.l[2] = v.z; a.l[3] = v.w; }
__device__ __forceinline__ void set_hi(Fr &a, uint4 v) { a.l[4] = v.x; a.l[5] = v.y; a.l[6] = v.z; a.l[7] = v.w; }

// padded LDS slot of logical position pos (one 16-byte slot per position, +1 slot every 8)
__device__ __forceinline__ int pidx(int pos) { return pos + (pos >> 3); }

// Arithmetic: radix 2^29, nine limbs per coefficient in registers (fr29.hip.hpp).  A coefficient is the standard value
// x 2^256 (any representative below 2 r between passes), a twiddle is stored as w 2^261, so the nine-limb Montgomery
// product of the two is the standard x w again.  Butterflies do not reduce: the comments give the bound of each value as a
// multiple of r; a product needs operands below 11 r and returns less than 2 r, a pass ends in one weak reduction (< 16 r
// in, < 2 r out).  The 8 x 32-bit multiply this replaces took 1.2 us per product in a lone wave, this one 0.37 us.
// One product at a time: the scheduler would otherwise interleave every independent product of a stage (good for latency,
// which a SIMD that already issues back to back does not need) and run out of registers.
__device__ __forceinline__ F29 mul_tw(const F29 &x, const Fr &w29) {
  __builtin_amdgcn_sched_barrier(0);
  const F29 r = fr29_mul(x, fr29_unpack(w29));
  __builtin_amdgcn_sched_barrier(0);
  return r;
}
__device__ __forceinline__ F29 weak(const F29 &x) {
  __builtin_amdgcn_sched_barrier(0);
  const F29 r = fr29_weak_reduce(x);
  __builtin_amdgcn_sched_barrier(0);
  return r;
}

__device__ __forceinline__ void bfly(F29 &a, F29 &b, const u32 (&kb)[9]) {  // kb: a multiple of r not below b
  const F29 t = f29_sub(a, b, kb);
  a = f29_add(a, b);
  b = t;
}

// y[s] = sum_t v[t] w_R^(s t), natural order, in place.  w4 = w_R^(R/4), w8 = w_8, w83 = w_8^3.
__device__ __forceinline__ void dft2(F29 &v0, F29 &v1) {   // < 2r in, < 4r out
  const u32 R2[9] = ZK_R29_2P;
  bfly(v0, v1, R2);
}

__device__ __forceinline__ void dft4(F29 &v0, F29 &v1, F29 &v2, F29 &v3, const Fr *__restrict__ w4) {   // < 2r in, < 8r out
  const u32 R2[9] = ZK_R29_2P, R4[9] = ZK_R29_4P;
  bfly(v0, v2, R2);  // v0 = a0, v2 = a1            < 4r
  bfly(v1, v3, R2);  // v1 = b0, v3 = (v1 - v3)     < 4r
  v3 = mul_tw(v3, *w4);   // b1                      < 2r
  bfly(v0, v1, R4);  // v0 = y0, v1 = y2            < 8r
  bfly(v2, v3, R2);  // v2 = y1, v3 = y3            < 6r
  const F29 t = v1;
  v1 = v2;
  v2 = t;
}

__device__ __forceinline__ void dft8(F29 (&r)[8], const Fr *__restrict__ w4, const Fr *__restrict__ w8, const Fr *__restrict__ w83) {   // < 2r in, < 16r out
  const u32 R2[9] = ZK_R29_2P, R8[9] = ZK_R29_8P;
  F29 &v0 = r[0], &v1 = r[1], &v2 = r[2], &v3 = r[3], &v4 = r[4], &v5 = r[5], &v6 = r[6], &v7 = r[7];
  // even part E = DFT4(v0, v2, v4, v6), odd part O = DFT4(v1, v3, v5, v7)
  dft4(v0, v2, v4, v6, w4);  // v0=E0 v2=E1 v4=E2 v6=E3   < 8r
  dft4(v1, v3, v5, v7, w4);  // v1=O0 v3=O1 v5=O2 v7=O3   < 8r
  v3 = mul_tw(v3, *w8);      // < 2r
  v5 = mul_tw(v5, *w4);
  v7 = mul_tw(v7, *w83);
  // y[s] = E[s] + O'[s], y[s+4] = E[s] - O'[s]
  bfly(v0, v1, R8);  // v0 = y0, v1 = y4   < 16r
  bfly(v2, v3, R2);  // v2 = y1, v3 = y5   < 10r
  bfly(v4, v5, R2);  // v4 = y2, v5 = y6
  bfly(v6, v7, R2);  // v6 = y3, v7 = y7
  const F29 y1 = v2, y2 = v4, y3 = v6, y4 = v1, y5 = v3, y6 = v5;
  v1 = y1; v2 = y2; v3 = y3; v4 = y4; v5 = y5; v6 = y6;
}

// radix 16: even / odd radix-8 transforms, the odd outputs times w_16^k, one more butterfly level.  < 2r in, < 4r out.
__device__ __forceinline__ void dft16(F29 (&r)[16], const Fr *__restrict__ tw, int step /* N / 16 */) {
  const u32 R2[9] = ZK_R29_2P;
  F29 e[8], o[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    e[k] = r[2 * k];
    o[k] = r[2 * k + 1];
  }
  dft8(e, tw + 4 * step, tw + 2 * step, tw + 6 * step);   // < 16r
  dft8(o, tw + 4 * step, tw + 2 * step, tw + 6 * step);
#pragma unroll
  for (int k = 0; k < 8; ++k) e[k] = weak(e[k]);   // < 2r
  o[0] = weak(o[0]);
#pragma unroll
  for (int k = 1; k < 8; ++k) o[k] = mul_tw(o[k], tw[k * step]);   // < 2r
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    bfly(e[k], o[k], R2);   // e = y[k], o = y[k + 8]   < 4r
    r[k] = e[k];
    r[k + 8] = o[k];
  }
}

struct TileArgs {
  const Fr *in;        // column c, tile b at in + c*col_stride_in + b*in_tile_stride (contiguous run of N)
  size_t in_tile_stride;  // N for independent tiles, 0 when every tile of a column reads the same coefficients
  Fr *out;             // element q of (c,b) goes to out + c*col_stride_out + out_off(b) + q*out_stride
  size_t col_stride_in, col_stride_out;
  // every multiplier table below holds its values in the 2^261 form (zk_fr_to_29: times 32), see the note on arithmetic
  const Fr *tw;        // omega_N^j (or omega_N^-j), j < N
  const Fr *pre;       // optional per-position multiplier applied at load: x[q] *= pre[pre_off(c,b) + q]
  size_t pre_tile_stride;  // pre offset per tile (0 = same table for all tiles)
  const Fr *post;      // optional single multiplier applied at store (n^-1)
  int log_tiles;       // tiles per column = 2^log_tiles; out_off(b) = bitrev(b), out_stride = 2^log_tiles
  int in_len;          // coefficients actually present per tile (rest read as zero)
  int out_natural_tiles;  // 1: out_off(b) = b*N, stride 1 (independent tiles)
  const void *pre13;      // the 2^13 tile (ntt13.hip) reads its pre-multipliers from the tables of zk_pre13 instead of `pre`
};

// one pass over F29 registers (values below 2 r in and out).  EPT coefficients per thread (8, or 16 for the 2^13 tile: 512
// threads may use 256 VGPRs each, which holds 16 nine-limb values and the temporaries of a product; 1024 threads with 128
// registers spilled 400 bytes per lane); radix R <= EPT, EPT / R independent transforms per thread.
template <int LOGN, int EPT, int R_LOG, int P_LOG>
__device__ __forceinline__ void stockham_pass(F29 (&reg)[EPT], const Fr *__restrict__ tw, uint4 *lds, int tid) {
  constexpr int N = 1 << LOGN, T = N / EPT, R = 1 << R_LOG, U = EPT / R, P = 1 << P_LOG;
  constexpr bool LAST = (P_LOG + R_LOG == LOGN);
  // outer twiddles w_{PR}^(k t) = omega^(k t N/(P R))
  if (P_LOG > 0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = tid + u * T;
      const int k = i & (P - 1);
#pragma unroll
      for (int t = 1; t < R; ++t) {
        const int e = k * t * (N / (P * R));
        reg[u + t * U] = mul_tw(reg[u + t * U], tw[e]);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if constexpr (R == 16) {
      F29 g[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) g[t] = reg[u + t * U];
      dft16(g, tw, N / 16);
#pragma unroll
      for (int t = 0; t < 16; ++t) reg[u + t * U] = g[t];
    } else if constexpr (R == 8) {
      F29 g[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) g[t] = reg[u + t * U];
      dft8(g, tw + N / 4, tw + N / 8, tw + 3 * (N / 8));
#pragma unroll
      for (int t = 0; t < 8; ++t) reg[u + t * U] = g[t];
    } else if constexpr (R == 4) {
      dft4(reg[u], reg[u + U], reg[u + 2 * U], reg[u + 3 * U], tw + N / 4);
    } else {
      dft2(reg[u], reg[u + U]);
    }
  }
#pragma unroll
  for (int m = 0; m < EPT; ++m) reg[m] = weak(reg[m]);   // < 16r -> < 2r
  if (!LAST) {
    // transpose through LDS, three limbs at a time (a 2^13 column of nine-limb values is 288 KiB, the LDS 160 KiB): a round
    // writes limbs 3r .. 3r+2 of the thread's coefficients and reads the transposed ones back into the same registers
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      __syncthreads();
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = tid + u * T;
        const int k = i & (P - 1);
        const int j = ((i - k) << R_LOG) + k;
#pragma unroll
        for (int s = 0; s < R; ++s) {
          const F29 &v = reg[u + s * U];
          lds[pidx(j + s * P)] = make_uint4(v.l[3 * r], v.l[3 * r + 1], v.l[3 * r + 2], 0u);
        }
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < EPT; ++m) {
        const uint4 t = lds[pidx(tid + T * m)];
        reg[m].l[3 * r] = t.x;
        reg[m].l[3 * r + 1] = t.y;
        reg[m].l[3 * r + 2] = t.z;
      }
    }
  }
}

template <int LOGN, int EPT, int P_LOG>
__device__ __forceinline__ void run_passes(F29 (&reg)[EPT], const Fr *__restrict__ tw, uint4 *lds, int tid) {
  if constexpr (P_LOG < LOGN) {
    constexpr int REM = LOGN - P_LOG;
    constexpr int MAXR = EPT == 16 ? 4 : 3;
    constexpr int R_LOG = REM >= MAXR ? MAXR : REM;
    stockham_pass<LOGN, EPT, R_LOG, P_LOG>(reg, tw, lds, tid);
    run_passes<LOGN, EPT, P_LOG + R_LOG>(reg, tw, lds, tid);
  }
}

// Coefficients per thread.  16 for the 2^13 tile (512 threads, radix-16 passes, three transposes instead of four) was built
// and measured: 0.193 ms per 256 columns against 0.181 ms with 8 (1024 threads), so 8 everywhere.
template <int LOGN>
constexpr int tile_ept() { return 8; }

__device__ __forceinline__ unsigned brev(unsigned x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

template <int LOGN>
__global__ void __launch_bounds__((1 << LOGN) / tile_ept<LOGN>()) k_ntt_tile(TileArgs a) {
  constexpr int N = 1 << LOGN, EPT = tile_ept<LOGN>(), T = N / EPT;
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int tid = threadIdx.x;
  const unsigned b = blockIdx.x;  // tile within column
  const size_t c = blockIdx.y;
  const Fr *__restrict__ src = a.in + c * a.col_stride_in + (size_t)b * a.in_tile_stride;
  F29 reg[EPT];
#pragma unroll
  for (int m = 0; m < EPT; ++m) {
    const int q = tid + T * m;
    reg[m] = q < a.in_len ? fr29_unpack(src[q]) : f29_zero();
  }
  if (a.pre) {
    const Fr *__restrict__ pre = a.pre + (size_t)b * a.pre_tile_stride;
#pragma unroll
    for (int m = 0; m < EPT; ++m) {
      const int q = tid + T * m;
      if (q < a.in_len) reg[m] = mul_tw(reg[m], pre[q]);
    }
  }
  run_passes<LOGN, EPT, 0>(reg, a.tw, lds, tid);
  if (a.post) {
#pragma unroll
    for (int m = 0; m < EPT; ++m) reg[m] = mul_tw(reg[m], *a.post);
  }
  Fr *__restrict__ dst = a.out + c * a.col_stride_out;
  if (a.out_natural_tiles) {
    dst += (size_t)b * N;
#pragma unroll
    for (int m = 0; m < EPT; ++m) dst[tid + T * m] = fr29_pack(fr29_canonical(reg[m]));   // columns in memory are canonical
  } else {
    const size_t off = brev(b, a.log_tiles);
    const size_t stride = (size_t)1 << a.log_tiles;
#pragma unroll
    for (int m = 0; m < EPT; ++m) dst[off + (size_t)(tid + T * m) * stride] = fr29_pack(fr29_canonical(reg[m]));
  }
}

template <int LOGN>
inline int launch_tile(zkfhe_ctx *ctx, const TileArgs &a, unsigned tiles, unsigned cols) {
  constexpr int N = 1 << LOGN;
  constexpr size_t lds_bytes = (size_t)(N + N / 8) * 16;
  ZK_CK(zk_func_max_lds(ctx, (const void *)k_ntt_tile<LOGN>, (int)lds_bytes));
  dim3 grid(tiles, cols);
  zk_prof_begin(ctx);
  k_ntt_tile<LOGN><<<grid, N / tile_ept<LOGN>(), lds_bytes, ctx->stream>>>(a);
  ZK_LAUNCH_CHECK(ctx);
  zk_prof_end(ctx, 1, 64.0 * (double)N * (double)tiles * (double)cols);
  if (ctx->prof_on) ctx->prof_ops[1] += 0.5 * (double)N * LOGN * (double)tiles * (double)cols;  // butterflies
  return ZKFHE_OK;
}


}  // namespace zk
