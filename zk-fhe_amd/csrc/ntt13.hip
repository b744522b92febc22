// The 2^13-point NTT tile (every column transform of a k = 13 proof; the tile of the larger domains) -- replaces the
// Stockham k_ntt_tile<13>, which needed 147 KB of LDS (one 1024-thread workgroup per CU), fifteen workgroup-wide barriers and
// spilled 400 bytes per lane.  Same seam as the rest of ntt.hip: halo2_proofs best_fft / EvaluationDomain (third-party,
// reached from reference examples/bfv.rs:311).
//
// Decomposition (decimation in frequency, n = 4096 m0 + 512 m1 + 64 m2 + 8 m3 + m4, k = s0 + 2 s1 + 16 s2 + 128 s3 + 1024 s4):
//   * radix 2 over m0 FUSED INTO THE LOAD: the column is cut into its even-k and odd-k halves, one workgroup of 512 threads
//     (eight waves, two per SIMD) each -- sub-tile s0 forms x[t] + x[t + 4096] resp. (x[t] - x[t + 4096]) w^t while loading.
//     With a coset pre-multiplier (coeff_to_extended) both halves come out of ONE fused two-product multiply per output,
//     x[t] A[t] + x[t + 4096] B[t], against tables that already hold the products of the coset powers and w^t: what used to be
//     a multiplication per input and one per odd output.  The two workgroups of a column sit on the same XCD (block ids 8
//     apart), so the second read of the column and the interleaved halves of the output lines meet in one L2.
//   * radix 8 over m1 in registers (thread t1 = 0..511 holds m1 = 0..7), twiddles w^(2 t1 s1), then the only workgroup-wide
//     exchange: wave s1 receives the 512 points of its independent 2^9 sub-transform.
//   * radix 8 over m2, m3, m4 inside a wave: the exchanges between them (register index <-> lane bits 5..3, then lane bits
//     2..0) go through the wave's own 9 KB of LDS with no barrier -- the waves of a workgroup drift apart and the SIMDs always
//     have a wave that issues multiply-adds.
//   * one last workgroup-wide exchange of the packed results, so that a wave stores 64 consecutive outputs of its half.
// LDS: 72 KB per workgroup (three limbs of a value per 16-byte slot, three rounds per exchange), two workgroups per CU.
// Arithmetic: lz29.hip.hpp -- signed lazy limbs, carries propagated five times per radix-8 butterfly instead of after each of
// its 24 additions, twiddles stored unpacked.  Bounds are in the types: if it compiles it cannot overflow.
#include <array>
#include <map>

#include "ctx.hpp"
#include "lz29.hip.hpp"
#include "dft8.hip.hpp"
#include "ntt_tile.hip.hpp"

using namespace zk;

namespace {

// twiddle pack of one direction (entries of LwMem, w = the 2^13-th root of that direction)
constexpr int T0_OFF = 0;                   // [t], t < 4096:             w^t
constexpr int T1_OFF = 4096;                // [s - 1][t1], t1 < 512:     w^(2 t1 s)
constexpr int T2_OFF = T1_OFF + 7 * 512;    // [s - 1][t2], t2 < 64:      w^(16 t2 s)
constexpr int T3_OFF = T2_OFF + 7 * 64;     // [s - 1][t3], t3 < 8:       w^(128 t3 s)
constexpr int C_OFF = T3_OFF + 7 * 8;       // w^1024 (w8), w^2048 (w4), w^3072 (w8^3)
constexpr int PACK_LEN = C_OFF + 3;

__global__ void __launch_bounds__(256) k_tw13_pack(const Fr *__restrict__ tw29 /* w^j 2^261, j < 8192 */, LwMem *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= PACK_LEN) return;
  int e;
  if (i < T1_OFF) e = i;
  else if (i < T2_OFF) e = 2 * ((i - T1_OFF) & 511) * (((i - T1_OFF) >> 9) + 1);
  else if (i < T3_OFF) e = 16 * ((i - T2_OFF) & 63) * (((i - T2_OFF) >> 6) + 1);
  else if (i < C_OFF) e = 128 * ((i - T3_OFF) & 7) * (((i - T3_OFF) >> 3) + 1);
  else e = 1024 * (i - C_OFF + 1);
  out[i] = lw_from_packed(tw29[e & 8191]);
}

// coset pre-multiplier tables of row k1 (shift h = g w_ext^k1), sub-tile 0 and 1, 8192 entries each:
//   sub 0: [t] = h^t                                   (t < 8192)      x[t] h^t + x[t+4096] h^(t+4096)
//   sub 1: [t] = h^t w^t, [t + 4096] = -h^(t+4096) w^t (t < 4096)      (x[t] h^t - x[t+4096] h^(t+4096)) w^t
__global__ void __launch_bounds__(256) k_pre13_pack(Fr h, Fr start /* 32 c: the powers come out as c h^t in the 2^261 form */, const Fr *__restrict__ fwd /* w^j, standard form */,
                                                    LwMem *__restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= 8192) return;
  Fr r = start;
  Fr b = h;
  for (int e = j; e; e >>= 1) {
    if (e & 1) r = r * b;
    b = fp_sqr<FrP>(b);
  }
  out[j] = lw_from_packed(r);
  Fr o = r * fwd[j & 4095];
  if (j >= 4096) o = fp_neg<FrP>(o);
  out[8192 + j] = lw_from_packed(o);
}

// radix-8 pass followed by the twiddles of the next level: x[s] <- y_s * tw[(s - 1) * stride]  (y_0: reduced only)
__device__ __forceinline__ void pass8_tw(LzT (&x)[8], const Consts &K, const LwMem *__restrict__ tw, int stride) {
  // the seven twiddles are requested before the butterfly (63 registers): their latency -- L2, mostly -- hides behind its nine
  // products; loaded where they are used each product would wait a microsecond for its operand with two waves per SIMD
  const Lw t1 = tw_at(tw), t2 = tw_at(tw + stride), t3 = tw_at(tw + 2 * stride), t4 = tw_at(tw + 3 * stride), t5 = tw_at(tw + 4 * stride),
           t6 = tw_at(tw + 5 * stride), t7 = tw_at(tw + 6 * stride);
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

// last radix-8 pass: canonical packed outputs, optionally times one constant (n^-1)
template <bool POST>
__device__ __forceinline__ void pass8_out(const LzT (&x)[8], const Consts &K, const Lw &p, Fr (&y)[8]) {
  ZK_DFT8_CORE(x, K)
  if (POST) {
    y[0] = lz_store(mulw_u(y0, p)); ZK_F
    y[1] = lz_store(mulw_u(y1, p)); ZK_F
    y[2] = lz_store(mulw_u(y2, p)); ZK_F
    y[3] = lz_store(mulw_u(y3, p)); ZK_F
    y[4] = lz_store(mulw_u(y4, p)); ZK_F
    y[5] = lz_store(mulw_u(y5, p)); ZK_F
    y[6] = lz_store(mulw_u(y6, p)); ZK_F
    y[7] = lz_store(mulw_u(y7, p)); ZK_F
  } else {
    y[0] = lz_store(y0); ZK_F
    y[1] = lz_store(y1); ZK_F
    y[2] = lz_store(y2); ZK_F
    y[3] = lz_store(y3); ZK_F
    y[4] = lz_store(y4); ZK_F
    y[5] = lz_store(y5); ZK_F
    y[6] = lz_store(y6); ZK_F
    y[7] = lz_store(y7); ZK_F
  }
}

// limbs 3 r .. 3 r + 2 of a value <-> one 16-byte LDS slot
__device__ __forceinline__ uint4 limbs3(const LzT &v, int r) { return make_uint4((u32)v.l[3 * r], (u32)v.l[3 * r + 1], (u32)v.l[3 * r + 2], 0u); }
__device__ __forceinline__ void set_limbs3(LzT &v, int r, const uint4 t) {
  v.l[3 * r] = (int)t.x;
  v.l[3 * r + 1] = (int)t.y;
  v.l[3 * r + 2] = (int)t.z;
}
// the LDS queue serves a wave's instructions in order: inside a wave a write is visible to the reads issued after it; the
// fence only keeps the compiler from moving them
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct Tile13Args {
  TileArgs t;
  const LwMem *tw;    // twiddle pack of the direction
  const LwMem *pre;   // coset tables [tile row][sub][8192], or nullptr
  unsigned tiles, cols;
};

#ifndef ZK_NTT13_WAVES
#define ZK_NTT13_WAVES 2   // waves per SIMD the register budget is cut for (2: up to 256 VGPRs, one workgroup per CU)
#endif
constexpr int WAVE_SLOTS = 576;   // 8 x 72 (512 slots + 8 per 64 of padding)
constexpr int OUT_SLOTS = 520;    // 512 + 8: one row of the output exchange
constexpr size_t LDS13 = (size_t)8 * WAVE_SLOTS * 16;

__global__ void __launch_bounds__(512, ZK_NTT13_WAVES) k_ntt13(Tile13Args A) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // block id -> (column, tile row, half): the two halves of a column on one XCD (ids 8 apart)
  const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const unsigned sub = slot & 1u, unit = (slot >> 1) * 8u + xcd;
  if (unit >= A.tiles * A.cols) return;
  const unsigned b = unit % A.tiles;
  const size_t c = unit / A.tiles;
  const TileArgs &a = A.t;
  const Fr *__restrict__ src = a.in + c * a.col_stride_in + (size_t)b * a.in_tile_stride;
  const LwMem *__restrict__ tw = A.tw;
  Consts K;
  K.w8 = tw_at(tw + C_OFF);
  K.w4 = tw_at(tw + C_OFF + 1);
  K.w83 = tw_at(tw + C_OFF + 2);

  // ---- load + radix 2 over m0 (this workgroup keeps the outputs k = sub mod 2) --------------------------------------------
  // The inputs are requested four iterations ahead and the table entries of a product two: their index is made to depend
  // (empty asm) on the result of an earlier iteration, otherwise the compiler hoists the loads of all eight iterations to the
  // top -- 320 registers of raw data.
  LzT x[8];
  Fr raw_lo[8], raw_hi[8];
  auto fetch = [&](int m) {
    int q = tid + 512 * m;
    if (m >= 4) asm volatile("" : "+v"(q) : "v"(x[m - 4].l[8]));   // four iterations (64 registers) of raw inputs in flight
    raw_lo[m] = q < a.in_len ? src[q] : Fr::zero();
    raw_hi[m] = q + 4096 < a.in_len ? src[q + 4096] : Fr::zero();
  };
#pragma unroll
  for (int m = 0; m < 4; ++m) fetch(m);
  if (A.pre) {
    const LwMem *__restrict__ pre = A.pre + ((size_t)b * 2 + sub) * 8192;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      int q = tid + 512 * m;
      if (m >= 2) asm volatile("" : "+v"(q) : "v"(x[m - 2].l[8]));
      const Lw wa = tw_at(pre + q), wb = tw_at(pre + q + 4096);
      const auto lo = lz_load(raw_lo[m]), hi = lz_load(raw_hi[m]);
      ZK_F
      x[m] = lz_mul2(lo, wa, hi, wb);
      ZK_F
      if (m + 4 < 8) fetch(m + 4);
    }
  } else if (sub == 0) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      x[m] = lz_norm(lz_add(lz_load(raw_lo[m]), lz_load(raw_hi[m])));   // (0,1), value < 2 r
      ZK_F
      if (m + 4 < 8) fetch(m + 4);
    }
  } else {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      int q = tid + 512 * m;
      if (m >= 2) asm volatile("" : "+v"(q) : "v"(x[m - 2].l[8]));
      const Lw w = tw_at(tw + T0_OFF + q);
      x[m] = mulw(lz_sub(lz_load(raw_lo[m]), lz_load(raw_hi[m])), w);
      if (m + 4 < 8) fetch(m + 4);
    }
  }

  // ---- radix 8 over m1, twiddles w^(2 t1 s1); exchange: wave s1 gets its 2^9 sub-transform, lane = t2, register = m2 -----
  pass8_tw(x, K, tw + T1_OFF + tid, 512);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    if (r) __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) lds[s * WAVE_SLOTS + tid] = limbs3(x[s], r);
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 8; ++m) set_limbs3(x[m], r, lds[wv * WAVE_SLOTS + m * 64 + lane]);
  }

  // ---- inside the wave from here on -----------------------------------------------------------------------------------------
  uint4 *__restrict__ mine = lds + wv * WAVE_SLOTS;
  const int a3 = lane >> 3, c3 = lane & 7;
  // radix 8 over m2, twiddles w^(16 t2 s2); exchange register index <-> lane bits 5..3: (s2 | m3 t3) -> (m3 | s2 t3)
  pass8_tw(x, K, tw + T2_OFF + lane, 64);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    wave_sync();
#pragma unroll
    for (int s = 0; s < 8; ++s) mine[s * 72 + lane] = limbs3(x[s], r);
    wave_sync();
#pragma unroll
    for (int m = 0; m < 8; ++m) set_limbs3(x[m], r, mine[a3 * 72 + m * 8 + c3]);
  }
  // radix 8 over m3, twiddles w^(128 t3 s3); exchange register index <-> lane bits 2..0: (s3 | s2 t3) -> (m4 | s2 s3)
  pass8_tw(x, K, tw + T3_OFF + c3, 8);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    wave_sync();
#pragma unroll
    for (int s = 0; s < 8; ++s) mine[a3 * 72 + s * 9 + c3] = limbs3(x[s], r);
    wave_sync();
#pragma unroll
    for (int m = 0; m < 8; ++m) set_limbs3(x[m], r, mine[a3 * 72 + c3 * 9 + m]);
  }
  // radix 8 over m4: register s4, lane (s2, s3), wave s1 -> output kappa = s1 + 8 s2 + 64 s3 + 512 s4 of this half
  Fr y[8];
  if (a.post) pass8_out<true>(x, K, lw_unpack(*a.post), y);
  else pass8_out<false>(x, K, K.w4, y);

  // ---- exchange of the packed results: thread kappa mod 512 stores kappa, kappa + 512, ... ------------------------------------
  const int kp = wv + 8 * a3 + 64 * c3;            // where this thread's results belong
  const int wslot = kp + (kp >> 6), rslot = tid + (tid >> 6);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) lds[s * OUT_SLOTS + wslot] = h ? hi4(y[s]) : lo4(y[s]);
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const uint4 t = lds[m * OUT_SLOTS + rslot];
      if (h) set_hi(y[m], t);
      else set_lo(y[m], t);
    }
  }
  Fr *__restrict__ dst = a.out + c * a.col_stride_out;
  if (a.out_natural_tiles) {
    dst += (size_t)b * 8192;
#pragma unroll
    for (int m = 0; m < 8; ++m) dst[sub + 2 * (tid + 512 * m)] = y[m];
  } else {
    const size_t off = brev(b, a.log_tiles);
    const size_t stride = (size_t)1 << a.log_tiles;
#pragma unroll
    for (int m = 0; m < 8; ++m) dst[off + (size_t)(sub + 2 * (tid + 512 * m)) * stride] = y[m];
  }
}


}  // namespace

// (A quarter-column variant -- four workgroups of 256 threads per column, two per CU -- was built and measured in round 5: bit-exact,
// 0.1542 against 0.1514 ms per 256 columns, not faster in the prover either.  It lives in tools/exp/patches/ntt13_quarter.patch.)

// coset tables for coeff_to_extended at n = 2^13 (rows cosets g w_ext^k1 of the 2^(13+lef) domain), built once per context
// (scaled: every entry times 2^-13 -- for input that is an inverse transform WITHOUT its n^-1, see zk_extend_lagrange)
int zk_pre13(zkfhe_ctx *ctx, const Fr &g, int lef, int rows, bool scaled, const void **out) {
  std::array<uint64_t, 6> key;
  for (int i = 0; i < 4; ++i) key[i] = (uint64_t)g.l[2 * i] | ((uint64_t)g.l[2 * i + 1] << 32);
  key[4] = (uint64_t)lef | (scaled ? 256u : 0u);
  key[5] = (uint64_t)rows;
  auto it = ctx->pre13.find(key);
  if (it == ctx->pre13.end()) {
    const NttDomain *dom, *edom;
    int rc = zk_domain(ctx, 13, &dom);
    if (rc) return rc;
    rc = zk_domain(ctx, 13 + lef, &edom);
    if (rc) return rc;
    // The generator comes from the caller of zkfhe_coset_ntt_batch: a bounded cache (the prover uses one or two keys; a caller
    // that sweeps g would otherwise grow device memory by rows * 786 KB per value for the life of the context).  On overflow
    // everything goes: the stream is drained first, queued transforms may still be reading the tables.
    if (ctx->pre13.size() >= 8) {
      ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
      for (auto &kv : ctx->pre13) (void)hipFree(kv.second);
      ctx->pre13.clear();
    }
    struct Owned {   // released unless handed to the cache
      LwMem *p = nullptr;
      ~Owned() {
        if (p) (void)hipFree(p);
      }
    } own;
    const size_t per_row = (size_t)2 * 8192;   // entries of one coset row's tables
    ZK_HIP(ctx, hipMalloc((void **)&own.p, (size_t)rows * per_row * sizeof(LwMem)));
    Fr shift = g;
    const Fr start = scaled ? dom->n_inv29 : zk_fr_to_29(Fr::one());
    for (int k1 = 0; k1 < rows; ++k1) {
      k_pre13_pack<<<32, 256, 0, ctx->stream>>>(shift, start, dom->fwd, own.p + (size_t)k1 * per_row);
      ZK_LAUNCH_CHECK(ctx);
      shift = shift * edom->omega;
    }
    it = ctx->pre13.emplace(key, (void *)own.p).first;
    own.p = nullptr;
  }
  *out = it->second;
  return ZKFHE_OK;
}

int zk_launch_tile_13(zkfhe_ctx *ctx, const TileArgs &a, unsigned tiles, unsigned cols) {
  if (a.pre && !a.pre13) return zk_fail_msg(ctx, ZKFHE_EINVAL, "2^13 tile: coset tables missing (zk_pre13)");
  // the two workgroups of a column both read all of it and write interleaved halves: never in place, and no partial overlap
  // either (a column's output would land in a column another workgroup has yet to read)
  {
    const size_t in_span = (size_t)(cols ? cols - 1 : 0) * a.col_stride_in + (size_t)(tiles ? tiles - 1 : 0) * a.in_tile_stride + 8192;
    const size_t out_span = (size_t)(cols ? cols - 1 : 0) * a.col_stride_out + (size_t)tiles * 8192;
    const Fr *i0 = a.in, *o0 = a.out;
    if (!(i0 + in_span <= o0 || o0 + out_span <= i0)) return zk_fail_msg(ctx, ZKFHE_EINVAL, "2^13 tile: input and output buffers must not overlap");
  }
  const void *tw_key = (const void *)a.tw;
  auto it = ctx->tw13.find(tw_key);
  if (it == ctx->tw13.end()) {
    LwMem *p = nullptr;
    const int len = PACK_LEN;
    ZK_HIP(ctx, hipMalloc((void **)&p, (size_t)len * sizeof(LwMem)));
    k_tw13_pack<<<zk_blocks(len, 256), 256, 0, ctx->stream>>>(a.tw, p);
    if (hipGetLastError() != hipSuccess) {
      (void)hipFree(p);
      return zk_fail_msg(ctx, ZKFHE_EHIP, "2^13 tile: twiddle pack launch failed");
    }
    it = ctx->tw13.emplace(tw_key, (void *)p).first;
  }
  Tile13Args A;
  A.t = a;
  A.tw = (const LwMem *)it->second;
  A.pre = (const LwMem *)a.pre13;
  A.tiles = tiles;
  A.cols = cols;
  const unsigned units = tiles * cols;
  zk_prof_begin(ctx);
  ZK_CK(zk_func_max_lds(ctx, (const void *)k_ntt13, (int)LDS13));
  k_ntt13<<<((units + 7) / 8) * 16, 512, LDS13, ctx->stream>>>(A);
  ZK_LAUNCH_CHECK(ctx);
  zk_prof_end(ctx, 1, 64.0 * 8192.0 * (double)tiles * (double)cols);
  if (ctx->prof_on) ctx->prof_ops[1] += 0.5 * 8192.0 * 13.0 * (double)tiles * (double)cols;  // butterflies
  return ZKFHE_OK;
}
