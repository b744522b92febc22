// Batched NTT / iNTT / coset NTT over BN254 Fr for gfx950.
//
// Replaces halo2_proofs `arithmetic::best_fft` and `EvaluationDomain::{lagrange_to_coeff,
// coeff_to_extended, extended_to_coeff}` (third-party; reached from reference examples/bfv.rs:311;
// semantics restated in oracle/oracle.c orc_fft / orc_ntt / orc_coset_ntt).  The DFT is unique, so
// the algorithm is free: this is an autosort Stockham transform, natural order in and out, with no
// bit-reversal pass.
//
// Tile kernel (2^3 <= n <= 2^13): ONE workgroup per column, n/8 threads, 8 coefficients per thread
// held in VGPRs (64 VGPRs of data).  Each pass is a radix-8 (or final radix-4/2) butterfly done
// entirely in registers; between passes the column is transposed through LDS.  A 2^13 column is
// 256 KiB -- more than the 160 KiB LDS -- so the transpose moves the low and the high 16 bytes of
// every element in two rounds of ds_write_b128 / ds_read_b128 (144 KiB with the +1/8 padding that
// keeps the stride-8 writes of the first pass bank-conflict-free).  HBM traffic is the
// algorithmic minimum: every coefficient is read once and written once, both as full 2 KiB-per-wave
// coalesced runs; twiddles come from an omega^j table that all columns share (L2 resident).
//
// Larger n: log2(n/2^13) radix-2 DIF stages over the whole vector (coalesced), then the tile kernel on
// each contiguous 2^13 block with a strided scatter to natural order (round-1 implementation; the
// prover itself never needs it, see zkfhe_coset_ntt_batch which keeps the extended domain coset-major).
#include <cstring>

#include "ctx.hpp"
#include "ntt_tile.hip.hpp"
#include "fr29.hip.hpp"

using namespace zk;

int zk_launch_tile_3(zkfhe_ctx *ctx, const zk::TileArgs &a, unsigned tiles, unsigned cols);
int zk_launch_tile_4(zkfhe_ctx *ctx, const zk::TileArgs &a, unsigned tiles, unsigned cols);
int zk_launch_tile_5(zkfhe_ctx *ctx, const zk::TileArgs &a, unsigned tiles, unsigned cols);
int zk_launch_tile_6(zkfhe_ctx *ctx, const zk::TileArgs &a, unsigned tiles, unsigned cols);
int zk_launch_tile_7(zkfhe_ctx *ctx, const zk::TileArgs &a, unsigned tiles, unsigned cols);
int zk_launch_tile_8(zkfhe_ctx *ctx, const zk::TileArgs &a, unsigned tiles, unsigned cols);
int zk_launch_tile_9(zkfhe_ctx *ctx, const zk::TileArgs &a, unsigned tiles, unsigned cols);
int zk_launch_tile_10(zkfhe_ctx *ctx, const zk::TileArgs &a, unsigned tiles, unsigned cols);
int zk_launch_tile_11(zkfhe_ctx *ctx, const zk::TileArgs &a, unsigned tiles, unsigned cols);
int zk_launch_tile_12(zkfhe_ctx *ctx, const zk::TileArgs &a, unsigned tiles, unsigned cols);
int zk_launch_tile_13(zkfhe_ctx *ctx, const zk::TileArgs &a, unsigned tiles, unsigned cols);

namespace {

// one radix-2 DIF stage over vectors of length n (all columns): pairs (j, j+half) inside blocks of 2*half
//   a' = a + b ; b' = (a - b) * omega_n^(j * n/(2*half)),  j = index inside the half
// pre != nullptr (first pass of a coset extension): the n_cols transforms are `rows` coset rows of n_cols / rows input vectors;
// transform c reads input vector c / rows and multiplies element i by pre[(c % rows) * n + i] on the way in (the coset powers).
__global__ void __launch_bounds__(256) k_dif_stage(const Fr *src, Fr *data, size_t n_cols, int log_n, int log_half,
                                                   const Fr *__restrict__ tw_n /* omega_n^j, j < n/2, 2^261 form */,
                                                   const Fr *__restrict__ pre = nullptr, unsigned rows = 1) {
  const size_t half = (size_t)1 << log_half;
  const size_t per_col = (size_t)1 << (log_n - 1);
  const size_t total = n_cols * per_col;
  for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t c = g >> (log_n - 1);
    const size_t i = g & (per_col - 1);
    const size_t j = i & (half - 1);
    const size_t blk = i >> log_half;
    const size_t w = (blk << (log_half + 1)) + j;
    const size_t o = (c << log_n) + w;
    Fr *p = data + o;
    Fr x, y;
    if (pre) {
      const size_t cs = c / rows, k1 = c - cs * rows;
      const Fr *ps = src + (cs << log_n) + w, *pp = pre + (k1 << log_n) + w;
      x = fr29_mul_const(ps[0], pp[0]);
      y = fr29_mul_const(ps[half], pp[half]);
    } else {
      x = src[o];   // src == data: in place
      y = src[o + half];
    }
    Fr s = x + y, d = x - y;
    const size_t e = j << (log_n - 1 - log_half);
    p[0] = s;
    p[half] = e ? fr29_mul_const(d, tw_n[e]) : d;
  }
}

// S consecutive radix-2 DIF stages (halves 2^log_half, 2^(log_half-1), ...) in one pass over memory: a thread holds the
// 2^S elements base + m * 2^(log_half - S + 1) of one block of 2^(log_half + 1) and runs the S butterfly levels in
// registers.  Long rows (k = 16 .. 19) used to make one full read+write of every column per stage.
template <int S>
__global__ void __launch_bounds__(256) k_dif_fused(const Fr *src, Fr *data, size_t n_cols, int log_n, int log_half, const Fr *__restrict__ tw_n,
                                                   const Fr *__restrict__ pre = nullptr, unsigned rows = 1 /* as in k_dif_stage */) {
  constexpr int R = 1 << S;
  const int log_q = log_half - S + 1;            // distance between the elements of a thread
  const size_t q = (size_t)1 << log_q;
  const size_t per_col = (size_t)1 << (log_n - S);
  const size_t total = n_cols * per_col;
  for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t c = g >> (log_n - S);
    const size_t i = g & (per_col - 1);
    const size_t j0 = i & (q - 1);
    const size_t blk = i >> log_q;
    const size_t w = (blk << (log_half + 1)) + j0;
    const size_t o = (c << log_n) + w;
    Fr *p = data + o;
    Fr x[R];
    if (pre) {
      const size_t cs = c / rows, k1 = c - cs * rows;
      const Fr *ps = src + (cs << log_n) + w, *pp = pre + (k1 << log_n) + w;
#pragma unroll
      for (int m = 0; m < R; ++m) x[m] = fr29_mul_const(ps[(size_t)m << log_q], pp[(size_t)m << log_q]);
    } else {
      const Fr *ps = src + o;   // src == data: in place; otherwise the first pass of an out-of-place transform
#pragma unroll
      for (int m = 0; m < R; ++m) x[m] = ps[(size_t)m << log_q];
    }
#pragma unroll
    for (int t = 0; t < S; ++t) {
      constexpr int dummy = 0;
      (void)dummy;
      const int h = R >> (t + 1);                // partner distance in units of q
      const int sh = log_n - 1 - (log_half - t); // twiddle exponent shift of this level
#pragma unroll
      for (int m = 0; m < R; ++m) {
        if (m & h) continue;
        const size_t j = j0 + ((size_t)(m & (h - 1)) << log_q);
        const Fr a = x[m], b = x[m + h];
        x[m] = a + b;
        const Fr d = a - b;
        const size_t e = j << sh;
        x[m + h] = e ? fr29_mul_const(d, tw_n[e]) : d;
      }
    }
#pragma unroll
    for (int m = 0; m < R; ++m) p[(size_t)m << log_q] = x[m];
  }
}

// 3 + SB radix-2 DIF stages in ONE pass over memory (rows of 2^17 .. 2^19: the four to six stages above the 2^13 tile were two
// passes of k_dif_fused, each a full read and write of every row -- 18.7 ms of a k = 19 proof at 2.5 TB/s).  A workgroup of 128
// threads owns a tile of M = 2^(3+SB) positions m (stride q = 2^(log_half - 2 - SB) elements) x JJ = 1024 / M consecutive
// offsets: a thread runs three levels on m = g + (M/8) r in registers, the tile goes through LDS (two 16-byte halves per element,
// each half array contiguous across the lanes: no bank conflicts), and the thread runs the last SB levels on the eight
// consecutive m = 8u + r.  Loads and stores are runs of JJ * 32 bytes.
template <int SB>
__global__ void __launch_bounds__(128) k_dif_lds(const Fr *src, Fr *data, size_t n_cols, int log_n, int log_half, const Fr *__restrict__ tw_n,
                                                 const Fr *__restrict__ pre = nullptr, unsigned rows = 1 /* as in k_dif_stage */) {
  constexpr int S = 3 + SB, M = 1 << S, JJ = 1024 / M, G = M / 8;
  __shared__ uint4 sh_lo[1024], sh_hi[1024];
  const int log_q = log_half - S + 1;
  const int sh0 = log_n - 1 - log_half;
  const size_t tiles_per_col = (size_t)1 << (log_n - 10);
  const size_t total = n_cols * tiles_per_col;
  const unsigned jj = threadIdx.x % JJ, gu = threadIdx.x / JJ;   // gu: g in the first half, u in the second
  for (size_t tile = blockIdx.x; tile < total; tile += gridDim.x) {
    const size_t c = tile >> (log_n - 10);
    const size_t i = (tile & (tiles_per_col - 1)) * JJ + jj;     // position among the n / M of the column
    const size_t j0 = i & (((size_t)1 << log_q) - 1);
    const size_t w = ((i >> log_q) << (log_half + 1)) + j0;
    const size_t o = (c << log_n) + w;
    Fr x[8];
    if (pre) {
      const size_t cs = c / rows, k1 = c - cs * rows;
      const Fr *ps = src + (cs << log_n) + w, *pp = pre + (k1 << log_n) + w;
#pragma unroll
      for (int r = 0; r < 8; ++r) x[r] = fr29_mul_const(ps[(size_t)(gu + G * r) << log_q], pp[(size_t)(gu + G * r) << log_q]);
    } else {
      const Fr *ps = src + o;
#pragma unroll
      for (int r = 0; r < 8; ++r) x[r] = ps[(size_t)(gu + G * r) << log_q];
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int hr = 4 >> t;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if (r & hr) continue;
        const size_t j = j0 + ((size_t)(gu + G * (r & (hr - 1))) << log_q);
        const Fr a = x[r], b = x[r + hr];
        x[r] = a + b;
        const Fr d = a - b;
        const size_t e = j << (sh0 + t);
        x[r + hr] = e ? fr29_mul_const(d, tw_n[e]) : d;
      }
    }
    __syncthreads();   // the previous tile has been read
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const unsigned idx = (gu + G * r) * JJ + jj;
      sh_lo[idx] = make_uint4(x[r].l[0], x[r].l[1], x[r].l[2], x[r].l[3]);
      sh_hi[idx] = make_uint4(x[r].l[4], x[r].l[5], x[r].l[6], x[r].l[7]);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const unsigned idx = (8 * gu + r) * JJ + jj;
      const uint4 lo = sh_lo[idx], hi = sh_hi[idx];
      x[r].l[0] = lo.x, x[r].l[1] = lo.y, x[r].l[2] = lo.z, x[r].l[3] = lo.w;
      x[r].l[4] = hi.x, x[r].l[5] = hi.y, x[r].l[6] = hi.z, x[r].l[7] = hi.w;
    }
#pragma unroll
    for (int t = 3; t < S; ++t) {
      const int hm = M >> (t + 1);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if (r & hm) continue;
        const size_t j = j0 + ((size_t)(r & (hm - 1)) << log_q);
        const Fr a = x[r], b = x[r + hm];
        x[r] = a + b;
        const Fr d = a - b;
        const size_t e = j << (sh0 + t);
        x[r + hm] = e ? fr29_mul_const(d, tw_n[e]) : d;
      }
    }
    Fr *p = data + o;
#pragma unroll
    for (int r = 0; r < 8; ++r) p[(size_t)(8 * gu + r) << log_q] = x[r];
  }
}

__global__ void __launch_bounds__(256) k_pow_table(Fr base, Fr *__restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    Fr r = Fr::one();
    Fr b = base;
    for (size_t e = i; e; e >>= 1) {
      if (e & 1) r = r * b;
      b = fp_sqr<FrP>(b);
    }
    out[i] = r;
  }
}

// out[i] = start * base^i
__global__ void __launch_bounds__(256) k_pow_table_scaled(Fr start, Fr base, Fr *__restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    Fr r = start;
    Fr b = base;
    for (size_t e = i; e; e >>= 1) {
      if (e & 1) r = r * b;
      b = fp_sqr<FrP>(b);
    }
    out[i] = r;
  }
}

// extended_to_coeff combine: for each (column, i2 < n): the 2^lef values A[k1][i2] (k1-major rows of
// length n) hold, after the row iNTTs (scaled by n^-1), (1/n) sum_k2 F[k1][k2] w^(-i2 k2).  Then
// coefficient i1*n + i2 = g^-(i1*n+i2) * 2^-lef * sum_k1 A[k1][i2] * w_ext^(-k1 (i1*n + i2))
//                       = scale[i1*n+i2] * sum_k1 (A[k1][i2] * w_ext^(-k1 i2)) * w_E^(-k1 i1),  E = 2^lef.
// LEF in {1,2,3}.
template <int LEF>
__global__ void __launch_bounds__(256) k_ext_combine(const Fr *__restrict__ rows, Fr *__restrict__ out, size_t n_cols, int log_n,
                                                     const Fr *__restrict__ tw_ext_inv /* w_ext^-j, j < n*E */,
                                                     const Fr *__restrict__ scale /* 2^-lef g^-j, j < n*E */) {   // both tables in the 2^261 form
  constexpr int E = 1 << LEF;
  const size_t n = (size_t)1 << log_n;
  const size_t total = n_cols * n;
  for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t c = g >> log_n, i2 = g & (n - 1);
    const Fr *src = rows + c * n * E;
    Fr v[E];
#pragma unroll
    for (int k1 = 0; k1 < E; ++k1) {
      Fr x = src[(size_t)k1 * n + i2];
      v[k1] = k1 ? fr29_mul_const(x, tw_ext_inv[(size_t)k1 * i2]) : x;
    }
    // DFT_E with root w_E^-1 = w_ext^-(n)
    Fr o[E];
#pragma unroll
    for (int i1 = 0; i1 < E; ++i1) {
      Fr acc = v[0];
#pragma unroll
      for (int k1 = 1; k1 < E; ++k1) {
        const int e = (k1 * i1) & (E - 1);
        acc = acc + (e ? fr29_mul_const(v[k1], tw_ext_inv[(size_t)e * n]) : v[k1]);
      }
      o[i1] = acc;
    }
    Fr *dst = out + c * n * E;
#pragma unroll
    for (int i1 = 0; i1 < E; ++i1) dst[(size_t)i1 * n + i2] = fr29_mul_const(o[i1], scale[(size_t)i1 * n + i2]);
  }
}

int launch_tile_dyn(zkfhe_ctx *ctx, int log_tile, const TileArgs &a, unsigned tiles, unsigned cols) {
  switch (log_tile) {
    case 3: return zk_launch_tile_3(ctx, a, tiles, cols);
    case 4: return zk_launch_tile_4(ctx, a, tiles, cols);
    case 5: return zk_launch_tile_5(ctx, a, tiles, cols);
    case 6: return zk_launch_tile_6(ctx, a, tiles, cols);
    case 7: return zk_launch_tile_7(ctx, a, tiles, cols);
    case 8: return zk_launch_tile_8(ctx, a, tiles, cols);
    case 9: return zk_launch_tile_9(ctx, a, tiles, cols);
    case 10: return zk_launch_tile_10(ctx, a, tiles, cols);
    case 11: return zk_launch_tile_11(ctx, a, tiles, cols);
    case 12: return zk_launch_tile_12(ctx, a, tiles, cols);
    case 13: return zk_launch_tile_13(ctx, a, tiles, cols);
  }
  return zk_fail_msg(ctx, ZKFHE_EINVAL, "tile size out of range");
}

}  // namespace

int zk_domain(zkfhe_ctx *ctx, int log_n, const NttDomain **out) {
  auto it = ctx->domains.find(log_n);
  if (it == ctx->domains.end()) {
    NttDomain d;
    struct Guard {   // the tables of a half-built domain are released on every early return
      NttDomain *d;
      ~Guard() {
        if (!d) return;
        (void)hipFree(d->fwd), (void)hipFree(d->inv), (void)hipFree(d->fwd29), (void)hipFree(d->inv29), (void)hipFree(d->n_inv29_dev);
      }
    } guard{&d};
    d.log_n = log_n;
    const size_t n = (size_t)1 << log_n;
    d.omega = zk_fr_root_of_unity(log_n);
    d.omega_inv = fp_inv<FrP>(d.omega);
    d.n_inv = fp_inv<FrP>(zk_fr_from_u64((uint64_t)n));
    ZK_HIP(ctx, hipMalloc((void **)&d.fwd, n * sizeof(Fr)));
    ZK_HIP(ctx, hipMalloc((void **)&d.inv, n * sizeof(Fr)));
    unsigned grid = zk_blocks(n, 256);
    if (grid > 4096) grid = 4096;
    k_pow_table<<<grid, 256, 0, ctx->stream>>>(d.omega, d.fwd, n);
    ZK_LAUNCH_CHECK(ctx);
    k_pow_table<<<grid, 256, 0, ctx->stream>>>(d.omega_inv, d.inv, n);
    ZK_LAUNCH_CHECK(ctx);
    const Fr c32 = zk_fr_to_29(Fr::one());   // 32 in the standard Montgomery form
    d.n_inv29 = d.n_inv * c32;
    ZK_HIP(ctx, hipMalloc((void **)&d.fwd29, n * sizeof(Fr)));
    ZK_HIP(ctx, hipMalloc((void **)&d.inv29, n * sizeof(Fr)));
    k_pow_table_scaled<<<grid, 256, 0, ctx->stream>>>(c32, d.omega, d.fwd29, n);
    ZK_LAUNCH_CHECK(ctx);
    k_pow_table_scaled<<<grid, 256, 0, ctx->stream>>>(c32, d.omega_inv, d.inv29, n);
    ZK_LAUNCH_CHECK(ctx);
    ZK_HIP(ctx, hipMalloc((void **)&d.n_inv29_dev, 64));
    ZK_HIP(ctx, hipMemcpyAsync(d.n_inv29_dev, &d.n_inv29, sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the source is a local of this function
    it = ctx->domains.emplace(log_n, d).first;
    guard.d = nullptr;
  }
  *out = &it->second;
  return ZKFHE_OK;
}

#define MAX_TILE_LOG 13

int zk_ntt_impl(zkfhe_ctx *ctx, const zkfhe_fr *src_dev, zkfhe_fr *cols_dev, size_t n_cols, int log_n, int inverse, const Fr *pre = nullptr, unsigned rows = 1,
                const Fr *shifts_host = nullptr);

// Rows of 2^16 and 2^19 (three / six stages above the 2^13 tile: BASELINE configs[3] and [4]) run those stages in four-step form
// (ntt_dif8.hip): constants for the size-8 / size-64 part, one streaming table product per element.  Every other length takes the
// radix-2 passes with gathered twiddles (k_dif_fused / k_dif_lds).
static bool dif8_rows(int log_n) { return log_n - MAX_TILE_LOG == 3 || log_n - MAX_TILE_LOG == 6; }

extern "C" {

int zkfhe_ntt_batch(zkfhe_ctx *ctx, zkfhe_fr *cols_dev, size_t n_cols, int log_n, int inverse) { return zk_ntt_impl(ctx, nullptr, cols_dev, n_cols, log_n, inverse); }

int zkfhe_ntt_batch_to(zkfhe_ctx *ctx, const zkfhe_fr *in_dev, zkfhe_fr *out_dev, size_t n_cols, int log_n, int inverse) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, in_dev != nullptr && out_dev != nullptr && log_n >= 0 && log_n < 40);
  {
    // "not overlapping" is enforced, not only documented: at n = 2^13 two workgroups per column read all of it while the other
    // writes, and a partial overlap would corrupt the result silently
    const Fr *i0 = (const Fr *)in_dev, *o0 = (const Fr *)out_dev;
    const size_t span = n_cols << log_n;
    if (!(i0 + span <= o0 || o0 + span <= i0)) return zk_fail_msg(ctx, ZKFHE_EINVAL, "zkfhe_ntt_batch_to: input and output must not overlap (zkfhe_ntt_batch transforms in place)");
  }
  return zk_ntt_impl(ctx, in_dev, out_dev, n_cols, log_n, inverse);
}

// src == nullptr: in place on cols_dev.  pre (long rows, forward, out of place only): the n_cols transforms are `rows` coset rows
// of the n_cols / rows vectors at src, each multiplied by its row of pre (2^261 form) while the first pass loads it.
extern "C++" int zk_ntt_impl(zkfhe_ctx *ctx, const zkfhe_fr *src_dev, zkfhe_fr *cols_dev, size_t n_cols, int log_n, int inverse, const Fr *pre, unsigned rows,
                             const Fr *shifts_host) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, (pre == nullptr && shifts_host == nullptr) || (src_dev != nullptr && !inverse && log_n > 13 && rows >= 1 && n_cols % rows == 0));
  ZK_ARG(ctx, !(pre && shifts_host));
  ZK_ARG(ctx, log_n >= 1 && log_n <= 26);
  if (!n_cols) return ZKFHE_OK;
  ZK_ARG(ctx, cols_dev != nullptr);
  ZK_ARG(ctx, n_cols < 65536);
  Fr *data = (Fr *)cols_dev;
  const Fr *src = (const Fr *)src_dev;
  const size_t n = (size_t)1 << log_n;
  const NttDomain *dom;
  int rc = zk_domain(ctx, log_n, &dom);
  if (rc) return rc;
  // n^-1 lives in device memory next to nothing else: keep a tiny scratch copy per call
  const Fr *ninv_dev = inverse ? dom->n_inv29_dev : nullptr;
  if (log_n < 3 && src) {
    rc = zk_copy_d2d(ctx, data, src, n_cols * n * sizeof(Fr));
    if (rc) return rc;
  }
  if (log_n < 3) {
    // tiny transforms: DIF stages then a bit-reversed gather is overkill; do log_n DIF stages and fix order on 2/4 points
    // n = 2: one stage is the whole transform (bitrev of 1 bit is identity).  n = 4: outputs 1 and 2 swapped.
    const Fr *tw = inverse ? dom->inv29 : dom->fwd29;
    for (int s = log_n - 1; s >= 0; --s) {
      k_dif_stage<<<zk_blocks(n_cols * (n / 2), 256), 256, 0, ctx->stream>>>(data, data, n_cols, log_n, s, tw);
      ZK_LAUNCH_CHECK(ctx);
    }
    if (log_n == 2 || inverse) {
      // finish on the host side of the stream with a trivial kernel-free path: reuse tile kernel is impossible (< 8);
      // use the scale kernel for n^-1 and a swap through scratch.
      void *p;
      rc = zk_scratch(ctx, 0, n_cols * n * sizeof(Fr), &p);
      if (rc) return rc;
      Fr *tmp = (Fr *)p;
      rc = zk_copy_d2d(ctx, tmp, data, n_cols * n * sizeof(Fr));
      if (rc) return rc;
      if (log_n == 2) {
        // swap elements 1 and 2 of every column: strided 2D copies
        ZK_HIP(ctx, hipMemcpy2DAsync(data + 1, 4 * sizeof(Fr), tmp + 2, 4 * sizeof(Fr), sizeof(Fr), n_cols, hipMemcpyDeviceToDevice, ctx->stream));
        ZK_HIP(ctx, hipMemcpy2DAsync(data + 2, 4 * sizeof(Fr), tmp + 1, 4 * sizeof(Fr), sizeof(Fr), n_cols, hipMemcpyDeviceToDevice, ctx->stream));
      }
      if (inverse) {
        rc = zkfhe_fr_scale(ctx, (const zkfhe_fr *)data, (const zkfhe_fr *)&dom->n_inv, (zkfhe_fr *)data, n_cols * n);
        if (rc) return rc;
      }
    }
    return ZKFHE_OK;
  }
  if (log_n <= MAX_TILE_LOG) {
    TileArgs a{};
    a.in = src ? src : data;
    a.out = data;
    a.in_tile_stride = n;
    a.col_stride_in = a.col_stride_out = n;
    a.tw = inverse ? dom->inv29 : dom->fwd29;
    a.pre = nullptr;
    a.post = ninv_dev;
    a.log_tiles = 0;
    a.in_len = (int)n;
    a.out_natural_tiles = 1;
    if (log_n == 13 && a.in == a.out) {
      // the 2^13 tile cuts a column into two workgroups that both read all of it: in place it goes through scratch
      void *p;
      rc = zk_scratch(ctx, 0, n_cols * n * sizeof(Fr), &p);
      if (rc) return rc;
      a.out = (Fr *)p;
      rc = launch_tile_dyn(ctx, log_n, a, 1, (unsigned)n_cols);
      if (rc) return rc;
      return zk_copy_d2d(ctx, data, p, n_cols * n * sizeof(Fr));
    }
    return launch_tile_dyn(ctx, log_n, a, 1, (unsigned)n_cols);
  }
  // large: DIF stages down to 2^13 blocks, then tile NTT per block with bit-reversed strided scatter.
  // Out of place (src given): the first pass reads src and writes the scratch arena, the later passes run in place there and the
  // tile kernel scatters into the destination -- no copy anywhere.  In place: passes on the data, tiles into scratch, one copy back.
  void *p;
  rc = zk_scratch(ctx, 0, n_cols * n * sizeof(Fr), &p);
  if (rc) return rc;
  Fr *work = src ? (Fr *)p : data;
  const Fr *pass_in = src ? src : data;
  const int log_tiles = log_n - MAX_TILE_LOG;
  const Fr *tw = inverse ? dom->inv29 : dom->fwd29;
  int s_top = log_n - 1;
  if (dif8_rows(log_n) && !pre) {
    // all stages above the tile in one four-step pass (the coset shifts, if any, are folded into its tables)
    rc = zk_dif8_pass(ctx, pass_in, work, n_cols, log_n, log_n - MAX_TILE_LOG, inverse, shifts_host, shifts_host ? rows : 1u);
    if (rc) return rc;
    pass_in = work;
    s_top = MAX_TILE_LOG - 1;
  } else {
    ZK_ARG(ctx, shifts_host == nullptr);   // the radix-2 passes take the coset powers as a device table (pre)
  }
  for (int s = s_top; s >= MAX_TILE_LOG;) {
    const int left = s - MAX_TILE_LOG + 1;
    static const bool lds_pass = !(getenv("ZKFHE_NTT_LDS_PASS") && getenv("ZKFHE_NTT_LDS_PASS")[0] == '0');
    if (left >= 4 && lds_pass) {   // four to six stages in one pass through LDS (seven: four, then three in registers)
      const int S = left == 7 ? 4 : (left > 6 ? 6 : left);
      const size_t tiles = n_cols * (n >> 10);
      unsigned grid = (unsigned)(tiles < (size_t)ctx->num_cu * 64 ? tiles : (size_t)ctx->num_cu * 64);
      if (S == 6) k_dif_lds<3><<<grid, 128, 0, ctx->stream>>>(pass_in, work, n_cols, log_n, s, tw, pre, rows);
      else if (S == 5) k_dif_lds<2><<<grid, 128, 0, ctx->stream>>>(pass_in, work, n_cols, log_n, s, tw, pre, rows);
      else k_dif_lds<1><<<grid, 128, 0, ctx->stream>>>(pass_in, work, n_cols, log_n, s, tw, pre, rows);
      ZK_LAUNCH_CHECK(ctx);
      pre = nullptr;
      pass_in = work;
      s -= S;
      continue;
    }
    const int S = left >= 3 ? 3 : left;       // fuse up to three stages per pass over memory
    size_t wk = n_cols * (n >> S);
    unsigned grid = zk_blocks(wk, 256);
    unsigned cap = (unsigned)ctx->num_cu * 16;
    if (grid > cap) grid = cap;
    if (S == 3) k_dif_fused<3><<<grid, 256, 0, ctx->stream>>>(pass_in, work, n_cols, log_n, s, tw, pre, rows);
    else if (S == 2) k_dif_fused<2><<<grid, 256, 0, ctx->stream>>>(pass_in, work, n_cols, log_n, s, tw, pre, rows);
    else k_dif_stage<<<grid, 256, 0, ctx->stream>>>(pass_in, work, n_cols, log_n, s, tw, pre, rows);
    ZK_LAUNCH_CHECK(ctx);
    pre = nullptr;   // only the first pass reads the unscaled input
    pass_in = work;
    s -= S;
  }
  const NttDomain *tdom;
  rc = zk_domain(ctx, MAX_TILE_LOG, &tdom);
  if (rc) return rc;
  TileArgs a{};
  a.in = work;
  a.out = src ? data : (Fr *)p;
  a.in_tile_stride = (size_t)1 << MAX_TILE_LOG;
  a.col_stride_in = a.col_stride_out = n;
  a.tw = inverse ? tdom->inv29 : tdom->fwd29;
  a.post = ninv_dev;
  a.log_tiles = log_tiles;
  a.in_len = 1 << MAX_TILE_LOG;
  a.out_natural_tiles = 0;
  rc = launch_tile_dyn(ctx, MAX_TILE_LOG, a, 1u << log_tiles, (unsigned)n_cols);
  if (rc) return rc;
  if (!src) {
    rc = zk_copy_d2d(ctx, data, p, n_cols * n * sizeof(Fr));
    if (rc) return rc;
  }
  return ZKFHE_OK;
}

// Forward coset extension of the first `rows` cosets only (rows <= 2^lef), written DENSELY: column c, coset k1 at
// out + (c * rows + k1) * n.  The prover's quotient has degree < 3n, so three of the four cosets determine it.
extern "C++" int zk_coset_ntt_rows(zkfhe_ctx *ctx, const Fr *in_dev, Fr *out_dev, size_t n_cols, int log_n, int lef, const Fr &g, int rows) {
  const int E = 1 << lef;
  if (rows >= E) return zkfhe_coset_ntt_batch(ctx, (const zkfhe_fr *)in_dev, (zkfhe_fr *)out_dev, n_cols, log_n, lef, (const zkfhe_fr *)&g, 0);
  if (!n_cols) return ZKFHE_OK;
  const size_t n = (size_t)1 << log_n, nr = n * (size_t)rows;
  const NttDomain *dom, *edom;
  int rc = zk_domain(ctx, log_n, &dom);
  if (rc) return rc;
  rc = zk_domain(ctx, log_n + lef, &edom);
  if (rc) return rc;
  void *p;
  rc = zk_scratch(ctx, 1, nr * sizeof(Fr), &p);
  if (rc) return rc;
  Fr *pre = (Fr *)p;
  const void *pre13 = nullptr;
  if (log_n == 13) {   // the 2^13 tile keeps its own tables (coset powers times the first-stage twiddles), built once
    rc = zk_pre13(ctx, g, lef, rows, false, &pre13);
    if (rc) return rc;
  } else if (dif8_rows(log_n) && rows <= 4) {
    // the four-step pass folds the coset powers into its own (cached) tables: no table of powers per call
    Fr shifts[4];
    Fr shift = g;
    for (int k1 = 0; k1 < rows; ++k1) {
      shifts[k1] = shift;
      shift = shift * edom->omega;
    }
    return zk_ntt_impl(ctx, (const zkfhe_fr *)in_dev, (zkfhe_fr *)out_dev, n_cols * (size_t)rows, log_n, 0, nullptr, (unsigned)rows, shifts);
  } else {
    Fr shift = g;
    const Fr c32 = zk_fr_to_29(Fr::one());
    for (int k1 = 0; k1 < rows; ++k1) {   // (g w_ext^k1)^i as constant operands of the nine-limb multiply (2^261 form)
      k_pow_table_scaled<<<zk_blocks(n, 256), 256, 0, ctx->stream>>>(c32, shift, pre + (size_t)k1 * n, n);
      ZK_LAUNCH_CHECK(ctx);
      shift = shift * edom->omega;
    }
  }
  if (log_n > MAX_TILE_LOG) {
    // one out-of-place transform of all (column, coset) rows into the destination; its first pass multiplies by the coset powers
    // as it loads (it used to be a separate pre-scaling pass through a scratch arena: one more read and write of every row)
    return zk_ntt_impl(ctx, (const zkfhe_fr *)in_dev, (zkfhe_fr *)out_dev, n_cols * (size_t)rows, log_n, 0, pre, (unsigned)rows);
  }
  TileArgs a{};
  a.in = in_dev;
  a.out = out_dev;
  a.in_tile_stride = 0;
  a.col_stride_in = n;
  a.col_stride_out = nr;
  a.tw = dom->fwd29;
  a.pre = pre;
  a.pre13 = pre13;
  a.pre_tile_stride = n;
  a.post = nullptr;
  a.log_tiles = lef;
  a.in_len = (int)n;
  a.out_natural_tiles = 1;
  return launch_tile_dyn(ctx, log_n, a, (unsigned)rows, (unsigned)n_cols);
}

// lagrange_to_coeff followed by coeff_to_extended on `rows` cosets, for the prover's extend step.  At n = 2^13 the inverse
// transform runs out of place WITHOUT its final n^-1 (one product per coefficient less) and the coset tables carry the factor.
extern "C++" int zk_extend_lagrange(zkfhe_ctx *ctx, const Fr *lagr_dev, Fr *tmp_dev, Fr *out_dev, size_t n_cols, int log_n, int lef, const Fr &g, int rows) {
  if (!n_cols) return ZKFHE_OK;
  const size_t n = (size_t)1 << log_n;
  if (log_n != 13 || rows > (1 << lef)) {
    int rc = zk_ntt_impl(ctx, (const zkfhe_fr *)lagr_dev, (zkfhe_fr *)tmp_dev, n_cols, log_n, 1);   // out of place: no copy
    if (rc) return rc;
    return zk_coset_ntt_rows(ctx, tmp_dev, out_dev, n_cols, log_n, lef, g, rows);
  }
  const NttDomain *dom;
  int rc = zk_domain(ctx, log_n, &dom);
  if (rc) return rc;
  const void *pre13 = nullptr;
  rc = zk_pre13(ctx, g, lef, rows, true, &pre13);
  if (rc) return rc;
  TileArgs a{};
  a.in = lagr_dev;
  a.out = tmp_dev;
  a.in_tile_stride = n;
  a.col_stride_in = a.col_stride_out = n;
  a.tw = dom->inv29;
  a.log_tiles = 0;
  a.in_len = (int)n;
  a.out_natural_tiles = 1;
  rc = launch_tile_dyn(ctx, log_n, a, 1, (unsigned)n_cols);
  if (rc) return rc;
  TileArgs b{};
  b.in = tmp_dev;
  b.out = out_dev;
  b.in_tile_stride = 0;
  b.col_stride_in = n;
  b.col_stride_out = n * (size_t)rows;
  b.tw = dom->fwd29;
  b.pre = tmp_dev;   // not read: the tile takes its multipliers from pre13
  b.pre13 = pre13;
  b.pre_tile_stride = n;
  b.log_tiles = lef;
  b.in_len = (int)n;
  b.out_natural_tiles = 1;
  return launch_tile_dyn(ctx, log_n, b, (unsigned)rows, (unsigned)n_cols);
}

int zkfhe_coset_ntt_batch(zkfhe_ctx *ctx, const zkfhe_fr *in_dev, zkfhe_fr *out_dev, size_t n_cols, int log_n,
                          int log_ext_factor, const zkfhe_fr *g_host, int inverse) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, log_n >= 3 && log_n + log_ext_factor <= 26);
  ZK_ARG(ctx, log_ext_factor >= 1 && log_ext_factor <= 3);
  ZK_ARG(ctx, g_host != nullptr);
  if (!n_cols) return ZKFHE_OK;
  ZK_ARG(ctx, in_dev != nullptr && out_dev != nullptr && n_cols < 65536);
  const int lef = log_ext_factor, E = 1 << lef;
  const size_t n = (size_t)1 << log_n, ne = n * E;
  Fr g;
  memcpy(&g, g_host, 32);
  const NttDomain *dom, *edom;
  int rc = zk_domain(ctx, log_n, &dom);
  if (rc) return rc;
  rc = zk_domain(ctx, log_n + lef, &edom);
  if (rc) return rc;
  void *p;
  if (!inverse) {
    // row k1: NTT_n of x[i] * (g * w_ext^k1)^i.  pre table: [k1][i]
    rc = zk_scratch(ctx, 1, ne * sizeof(Fr), &p);
    if (rc) return rc;
    Fr *pre = (Fr *)p;
    const void *pre13 = nullptr;
    if (log_n == 13) {
      rc = zk_pre13(ctx, g, lef, E, false, &pre13);
      if (rc) return rc;
    } else if (dif8_rows(log_n) && E <= 4) {
      Fr shifts[4];
      Fr shift = g;
      for (int k1 = 0; k1 < E; ++k1) {
        shifts[k1] = shift;
        shift = shift * edom->omega;
      }
      return zk_ntt_impl(ctx, in_dev, out_dev, n_cols * E, log_n, 0, nullptr, (unsigned)E, shifts);
    } else {
      Fr shift = g;
      const Fr c32 = zk_fr_to_29(Fr::one());
      for (int k1 = 0; k1 < E; ++k1) {
        k_pow_table_scaled<<<zk_blocks(n, 256), 256, 0, ctx->stream>>>(c32, shift, pre + (size_t)k1 * n, n);
        ZK_LAUNCH_CHECK(ctx);
        shift = shift * edom->omega;
      }
    }
    if (log_n > MAX_TILE_LOG) {
      // rows longer than one tile: pre-scale into the output rows, then a batched size-n NTT over all (column, k1) rows
      return zk_ntt_impl(ctx, in_dev, out_dev, n_cols * E, log_n, 0, pre, (unsigned)E);
    }
    TileArgs a{};
    a.in = (const Fr *)in_dev;
    a.out = (Fr *)out_dev;
    a.in_tile_stride = 0;  // every coset row of a column starts from the same n coefficients
    a.col_stride_in = n;
    a.col_stride_out = ne;
    a.tw = dom->fwd29;
    a.pre = pre;
    a.pre13 = pre13;
    a.pre_tile_stride = n;
    a.post = nullptr;
    a.log_tiles = lef;
    a.in_len = (int)n;
    a.out_natural_tiles = 1;
    return launch_tile_dyn(ctx, log_n, a, (unsigned)E, (unsigned)n_cols);
  }
  // inverse: rows iNTT (size n, scaled by n^-1) into scratch, then combine across k1
  if (log_n > MAX_TILE_LOG) {
    // rows transformed out of place straight into scratch slot 2 (the transform's own work arena is slot 0), combined from there
    rc = zk_scratch(ctx, 2, n_cols * ne * sizeof(Fr), &p);
    if (rc) return rc;
    Fr *rows2 = (Fr *)p;
    rc = zk_ntt_impl(ctx, in_dev, (zkfhe_fr *)rows2, n_cols * E, log_n, 1);
    if (rc) return rc;
    rc = zk_scratch(ctx, 1, ne * sizeof(Fr), &p);
    if (rc) return rc;
    Fr *scale2 = (Fr *)p;
    const Fr einv2 = fp_inv<FrP>(zk_fr_from_u64((uint64_t)E));
    const Fr ginv2 = fp_inv<FrP>(g);
    k_pow_table_scaled<<<zk_blocks(ne, 256), 256, 0, ctx->stream>>>(zk_fr_to_29(einv2), ginv2, scale2, ne);
    ZK_LAUNCH_CHECK(ctx);
    unsigned grid2 = zk_blocks(n_cols * n, 256);
    if (lef == 1) k_ext_combine<1><<<grid2, 256, 0, ctx->stream>>>(rows2, (Fr *)out_dev, n_cols, log_n, edom->inv29, scale2);
    else if (lef == 2) k_ext_combine<2><<<grid2, 256, 0, ctx->stream>>>(rows2, (Fr *)out_dev, n_cols, log_n, edom->inv29, scale2);
    else k_ext_combine<3><<<grid2, 256, 0, ctx->stream>>>(rows2, (Fr *)out_dev, n_cols, log_n, edom->inv29, scale2);
    ZK_LAUNCH_CHECK(ctx);
    return ZKFHE_OK;
  }
  rc = zk_scratch(ctx, 0, n_cols * ne * sizeof(Fr), &p);
  if (rc) return rc;
  Fr *rows = (Fr *)p;
  const void *q = dom->n_inv29_dev;
  TileArgs a{};
  a.in = (const Fr *)in_dev;
  a.out = rows;
  a.in_tile_stride = n;
  a.col_stride_in = ne;
  a.col_stride_out = ne;
  a.tw = dom->inv29;
  a.post = (const Fr *)q;
  a.log_tiles = lef;
  a.in_len = (int)n;
  a.out_natural_tiles = 1;
  rc = launch_tile_dyn(ctx, log_n, a, (unsigned)E, (unsigned)n_cols);
  if (rc) return rc;
  // scale[j] = 2^-lef * g^-j
  rc = zk_scratch(ctx, 1, ne * sizeof(Fr), &p);
  if (rc) return rc;
  Fr *scale = (Fr *)p;
  Fr einv = fp_inv<FrP>(zk_fr_from_u64((uint64_t)E));
  Fr ginv = fp_inv<FrP>(g);
  k_pow_table_scaled<<<zk_blocks(ne, 256), 256, 0, ctx->stream>>>(zk_fr_to_29(einv), ginv, scale, ne);
  ZK_LAUNCH_CHECK(ctx);
  unsigned grid = zk_blocks(n_cols * n, 256);
  if (lef == 1) k_ext_combine<1><<<grid, 256, 0, ctx->stream>>>(rows, (Fr *)out_dev, n_cols, log_n, edom->inv29, scale);
  else if (lef == 2) k_ext_combine<2><<<grid, 256, 0, ctx->stream>>>(rows, (Fr *)out_dev, n_cols, log_n, edom->inv29, scale);
  else k_ext_combine<3><<<grid, 256, 0, ctx->stream>>>(rows, (Fr *)out_dev, n_cols, log_n, edom->inv29, scale);
  ZK_LAUNCH_CHECK(ctx);
  return ZKFHE_OK;
}

}  // extern "C"
