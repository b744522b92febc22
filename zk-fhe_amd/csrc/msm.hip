// Batched multi-scalar multiplication over BN254 G1 for gfx950 (the KZG commit of the prover).
//
// Replaces halo2_proofs `arithmetic::best_multiexp(&[Fr], &[G1Affine]) -> G1` as called by
// `ParamsKZG::{commit, commit_lagrange}` (third-party; reached from reference examples/bfv.rs:311).
// The result is a group element, so any exact algorithm gives identical bytes once normalised to
// affine; parity is checked against oracle/oracle.c orc_msm.
//
// Two algorithms.  (1) The bucket pipeline -- Pippenger with ONE bucket set per MSM and per-window precomputed bases (wide calls
// at n >= 2^15, every call at n >= 2^18):
//   * The SRS is fixed, HBM is 288 GB: zkfhe_basis_create stores T[w][i] = 2^(c*w) * P_i for every
//     signed window w (n*W*64 bytes; 10 MiB at n = 2^13, c = 13).  Digit d of scalar i in window w then
//     contributes sign(d) * T[w][i] to bucket |d| -- all windows share the same 2^(c-1) buckets, so
//     there is one bucket reduction per MSM and no window-combining doublings at all.
//   * Scalars are taken out of Montgomery form and folded to sign-magnitude (s > r/2 -> r - s with the
//     point negated), so the "negative small" witness values (r - x) cost as little as small ones,
//     and zero digits are skipped.
//   * entries (bucket, table index, sign) are counting-sorted per MSM (LDS-privatised histogram per 2048-scalar chunk,
//     exclusive scan, scatter).  Every bucket is cut into equal slices of <= E entries (E ~ bucket load / 3); the task
//     list is counting-sorted by slice length, so the 64 lanes of a wave sum slices of the same length with XYZZ mixed
//     additions and the dependent chain is bounded however skewed a column is (witness columns hold thousands of 0/1
//     cells).  A bucket's partials are merged by a thread (<= 8), by eight lanes (<= 128) or by a wave.
//   * bucket reduction sum_b b*B_b for K <= 32768 buckets: idx = 64 a + b; row and column marginal sums, then per MSM a
//     lane-parallel double-and-add over the marginals and a shuffle tree; the result is normalised to affine in the
//     same kernel (k_msm_marginals, k_msm_weighted; k_msm_small for fewer than 64 buckets).
// All MSMs of a batch (columns sharing the basis) run through each stage in ONE launch.
//
// Field arithmetic of every kernel between the tables and the final normalisation: radix 2^29, nine limbs, Montgomery
// constant 2^261 (fq29.hip.hpp: 162 carry-free multiply-adds per product instead of 128 multiply-add + carry pairs, lazy
// reduction in the point formulas).  The tables, partials, buckets and marginals hold packed 256-bit words in that
// form; k_msm_weighted / k_msm_table_fold convert the one result per MSM back before normalising it.
//
// (2) When the digit-multiple table of a basis fits its HBM budget (zkfhe_basis_create: 43 GB per SRS half at n = 2^13, 13-bit
// digits) the whole pipeline above is replaced by k_msm_table: T[i][w][j] = j * 2^(13 w) * P_i is resident, so an MSM is a
// plain sum of <= 20 n table points per column -- no sort, no buckets, no weights, one launch (+ a fold) instead of
// thirteen.  Same additions as the bucket pipeline's accumulation, none of the rest: 0.9x the VALU instructions on full-width
// columns, 0.75x on witness columns, and 0.2 ms instead of 0.54 for a lone column (profiles/r2b_msm_table.md).  The bucket
// pipeline stays for bases whose table would not fit (n >= 2^18) or would force so many more windows that it loses.
#include <cstdio>
#include <vector>
#include <cstring>

#include "ctx.hpp"
#include "fq29.hip.hpp"

using namespace zk;

namespace {


// (r-1)/2 as canonical limbs: scalars above it are negated
__device__ __forceinline__ bool fr_gt_half(const Fr &s) {
  const u32 H[8] = {0xf8000000u, 0xa1f0fac9u, 0x3cdcb848u, 0x9419f424u, 0x40c0ac2eu, 0xdc2822dbu, 0x7098d014u, 0x18322739u};
  // compare s > H from the top limb
#pragma unroll
  for (int i = 7; i >= 0; --i) {
    if (s.l[i] > H[i]) return true;
    if (s.l[i] < H[i]) return false;
  }
  return false;
}

struct Digits {
  int c, windows;
};

// signed-window digits of the sign-magnitude scalar.  Calls f(w, bucket (1..2^(c-1)), negative)
template <class F>
__device__ __forceinline__ void for_each_digit(const Fr &mont, int c, int windows, F f) {
  Fr s = fp_from_mont<FrP>(mont);
  bool neg = fr_gt_half(s);
  if (neg) s = fp_neg<FrP>(s);  // r - s  (s != 0 here)
  u32 carry = 0;
  const u32 mask = (1u << c) - 1, halfv = 1u << (c - 1);
  // the scalar is shifted down by c bits per window (eight v_alignbit): indexing its limbs by the window's bit position put the
  // value into scratch memory (48 bytes per lane in every sort kernel)
  for (int w = 0; w < windows; ++w) {
    u32 v = (s.l[0] & mask) + carry;
#pragma unroll
    for (int i = 0; i < 7; ++i) s.l[i] = __builtin_amdgcn_alignbit(s.l[i + 1], s.l[i], (u32)c);
    s.l[7] >>= c;
    bool dneg = false;
    if (v > halfv) {
      v = (1u << c) - v;
      dneg = true;
      carry = 1;
    } else {
      carry = 0;
    }
    if (v) f(w, v, neg != dneg);
  }
}

// Histogram / scatter with LDS-privatised counters.  A workgroup owns a chunk of CHUNK scalars of ONE column and
// counts their digits in LDS (K+1 counters); only the non-zero bins touch global memory, with one atomic per
// (workgroup, bucket) instead of one per entry -- device-scope atomics are memory transactions on this chip
// (profiles/r1_pmc_traffic.md: 144 MB of writes per launch before this change).
constexpr size_t MSM_MAX_COLS = 4096;   // columns per call (ticket counters of the table path)
constexpr unsigned SORT_CHUNK = 2048;   // scalars per workgroup
constexpr unsigned SORT_THREADS = 512;

__global__ void __launch_bounds__(SORT_THREADS) k_msm_hist(const Fr *__restrict__ scalars, size_t col_stride, size_t n, unsigned chunks_per_col, int c, int windows,
                                                          unsigned *__restrict__ hist /* [n_cols][K+1] */, unsigned K1) {
  extern __shared__ unsigned lh[];
  const size_t col = blockIdx.x / chunks_per_col;
  const unsigned chunk = blockIdx.x % chunks_per_col;
  for (unsigned b = threadIdx.x; b < K1; b += SORT_THREADS) lh[b] = 0;
  __syncthreads();
  const size_t i0 = (size_t)chunk * SORT_CHUNK, i1 = min(n, i0 + SORT_CHUNK);
  for (size_t i = i0 + threadIdx.x; i < i1; i += SORT_THREADS)
    for_each_digit(scalars[col * col_stride + i], c, windows, [&](int, u32 b, bool) { atomicAdd(&lh[b], 1u); });
  __syncthreads();
  unsigned *h = hist + col * K1;
  for (unsigned b = threadIdx.x; b < K1; b += SORT_THREADS) {
    const unsigned v = lh[b];
    if (v) atomicAdd(&h[b], v);
  }
}

// per column exclusive scan of hist[1..K] -> off[0..K]: bucket value v (1..K) owns sorted entries
// [off[v-1], off[v]); cursor[v] starts at off[v-1].  One block per column.
__global__ void __launch_bounds__(256) k_msm_scan(const unsigned *__restrict__ hist, unsigned *__restrict__ off,
                                                  unsigned *__restrict__ cursor, unsigned K1) {
  __shared__ unsigned part[256];
  const unsigned *h = hist + (size_t)blockIdx.x * K1;
  unsigned *o = off + (size_t)blockIdx.x * K1;
  unsigned *cu = cursor + (size_t)blockIdx.x * K1;
  const unsigned K = K1 - 1;
  const unsigned per = (K + 255) / 256;
  const unsigned lo = threadIdx.x * per, hi = min(lo + per, K);
  unsigned s = 0;
  for (unsigned b = lo; b < hi; ++b) s += h[b + 1];
  part[threadIdx.x] = s;
  __syncthreads();
  // exclusive scan of part (simple serial by one wave lane 0 is fine: 256 values)
  if (threadIdx.x == 0) {
    unsigned acc = 0;
    for (int i = 0; i < 256; ++i) {
      unsigned t = part[i];
      part[i] = acc;
      acc += t;
    }
  }
  __syncthreads();
  unsigned acc = part[threadIdx.x];
  for (unsigned b = lo; b < hi; ++b) {
    o[b] = acc;       // start of bucket b+1
    cu[b + 1] = acc;  // scatter cursor of bucket b+1
    acc += h[b + 1];
  }
  if (hi == K) o[K] = acc;  // total (every thread past the end writes the same value)
}

// scatter: count in LDS again, reserve one contiguous range per (workgroup, bucket) with a single global atomic,
// then rank the entries inside the range with LDS atomics: entries of a bucket coming from one workgroup land next
// to each other (fewer partial-line stores), and global atomics drop from one per entry to one per non-empty bin.
__global__ void __launch_bounds__(SORT_THREADS) k_msm_scatter(const Fr *__restrict__ scalars, size_t col_stride, size_t n, unsigned chunks_per_col, int c, int windows,
                                                             unsigned *__restrict__ cursor, unsigned K1, unsigned *__restrict__ entries,
                                                             size_t col_entries) {
  extern __shared__ unsigned lh[];  // [K1] counts, then running positions
  const size_t col = blockIdx.x / chunks_per_col;
  const unsigned chunk = blockIdx.x % chunks_per_col;
  for (unsigned b = threadIdx.x; b < K1; b += SORT_THREADS) lh[b] = 0;
  __syncthreads();
  const size_t i0 = (size_t)chunk * SORT_CHUNK, i1 = min(n, i0 + SORT_CHUNK);
  for (size_t i = i0 + threadIdx.x; i < i1; i += SORT_THREADS)
    for_each_digit(scalars[col * col_stride + i], c, windows, [&](int, u32 b, bool) { atomicAdd(&lh[b], 1u); });
  __syncthreads();
  unsigned *cu = cursor + col * K1;
  for (unsigned b = threadIdx.x; b < K1; b += SORT_THREADS) {
    const unsigned v = lh[b];
    lh[b] = v ? atomicAdd(&cu[b], v) : 0u;  // base position of this workgroup's run in bucket b
  }
  __syncthreads();
  unsigned *e = entries + col * col_entries;
  for (size_t i = i0 + threadIdx.x; i < i1; i += SORT_THREADS)
    for_each_digit(scalars[col * col_stride + i], c, windows, [&](int w, u32 b, bool neg) {
      const unsigned pos = atomicAdd(&lh[b], 1u);
      e[pos] = ((unsigned)w * (unsigned)n + (unsigned)i) | (neg ? 0x80000000u : 0u);
    });
}

// ---- two-level counting sort (K >= 2048 buckets: the long bases, n >= 2^15) -------------------------------------------
// With K = 8192 ... 32768 buckets a workgroup's 2048 scalars put about one entry into each LDS counter: the privatised
// histogram above saves nothing (one device-scope atomic per entry and bin, 128 KB of counters = one workgroup per CU) and every
// sorted entry is a lone 4-byte store into a 33 MB array (k = 19: 20 ms of a 170 ms proof in k_msm_hist + k_msm_scatter).
// Here the bucket id is split into a coarse part (NB = K >> L bins) and L fine bits that ride in the spare bits of the entry
// word (entry indices need log2(W n) bits):
//   k_msm_chist     coarse histogram per column (NB counters per workgroup, NB atomics per 2048 scalars);
//   k_msm_cscan     exclusive scan -> coarse segment bounds;
//   k_msm_cscatter  512 scalars per workgroup: entries ranked by coarse bin in LDS and copied out as contiguous runs
//                   (one reserved range per workgroup and bin) into the staging array, fine bits attached;
//   k_msm_fine      one workgroup per (column, coarse bin): histogram of the segment over its F = 2^L buckets, scan
//                   (-> off[], the bucket bounds every later kernel reads), and the placement inside the segment -- a
//                   window of a few hundred KB that the L2 holds until its lines are complete.
constexpr unsigned CH_SCALARS = 2048, CS_SCALARS = 512, CS_THREADS = 256, FINE_THREADS = 512, FINE_MAX = 256, FINE_COPIES = 8, FINE_CTRS = FINE_MAX * FINE_COPIES, FINE_TILE = 14336;
static_assert(FINE_CTRS == 4 * FINE_THREADS && FINE_TILE % (4 * FINE_THREADS) == 0, "k_msm_fine: four counters per thread, whole 16-byte loads per tile");
constexpr int FINE_BITS_MAX = 8;   // F = 256 buckets per coarse bin: enough workgroups in k_msm_fine on a basis of 2^15 points

__global__ void __launch_bounds__(CS_THREADS) k_msm_chist(const Fr *__restrict__ scalars, size_t col_stride, size_t n, unsigned chunks_per_col, int c, int windows,
                                                         int L, unsigned NB, unsigned *__restrict__ chist /* [n_cols][NB] */) {
  extern __shared__ unsigned lh[];
  const size_t col = blockIdx.x / chunks_per_col;
  const unsigned chunk = blockIdx.x % chunks_per_col;
  for (unsigned b = threadIdx.x; b < NB; b += CS_THREADS) lh[b] = 0;
  __syncthreads();
  const size_t i0 = (size_t)chunk * CH_SCALARS, i1 = min(n, i0 + CH_SCALARS);
  for (size_t i = i0 + threadIdx.x; i < i1; i += CS_THREADS)
    for_each_digit(scalars[col * col_stride + i], c, windows, [&](int, u32 b, bool) { atomicAdd(&lh[(b - 1) >> L], 1u); });
  __syncthreads();
  for (unsigned g = threadIdx.x; g < NB; g += CS_THREADS) {
    const unsigned v = lh[g];
    if (v) atomicAdd(&chist[col * NB + g], v);
  }
}

// one thread per column: coff[col][0..NB] = exclusive scan of the coarse counts, ccursor = the segment starts
__global__ void __launch_bounds__(64) k_msm_cscan(const unsigned *__restrict__ chist, unsigned NB, size_t n_cols, unsigned *__restrict__ coff /* [n_cols][NB+1] */,
                                                  unsigned *__restrict__ ccursor /* [n_cols][NB] */) {
  const size_t col = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (col >= n_cols) return;
  unsigned acc = 0;
  for (unsigned g = 0; g < NB; ++g) {
    coff[col * (NB + 1) + g] = acc;
    ccursor[col * NB + g] = acc;
    acc += chist[col * NB + g];
  }
  coff[col * (NB + 1) + NB] = acc;
}

__global__ void __launch_bounds__(CS_THREADS) k_msm_cscatter(const Fr *__restrict__ scalars, size_t col_stride, size_t n, unsigned chunks_per_col, int c, int windows,
                                                            int L, unsigned NB, unsigned *__restrict__ ccursor, unsigned *__restrict__ stage, size_t stage_stride /* col_entries rounded up to four */) {
  extern __shared__ unsigned sh[];
  unsigned *cnt = sh, *loff = sh + NB, *gb = sh + 2 * NB, *buf = sh + 3 * NB;   // buf: CS_SCALARS * windows words
  const size_t col = blockIdx.x / chunks_per_col;
  const unsigned chunk = blockIdx.x % chunks_per_col;
  for (unsigned b = threadIdx.x; b < NB; b += CS_THREADS) cnt[b] = 0;
  __syncthreads();
  const size_t i0 = (size_t)chunk * CS_SCALARS, i1 = min(n, i0 + CS_SCALARS);
  for (size_t i = i0 + threadIdx.x; i < i1; i += CS_THREADS)
    for_each_digit(scalars[col * col_stride + i], c, windows, [&](int, u32 b, bool) { atomicAdd(&cnt[(b - 1) >> L], 1u); });
  __syncthreads();
  if (threadIdx.x < 64) {   // exclusive scan of cnt[0..NB) by the first wave: `per` consecutive bins per lane
    const unsigned per = (NB + 63) / 64, lo = threadIdx.x * per, hi = min(lo + per, NB);
    unsigned s = 0;
    for (unsigned g = lo; g < hi; ++g) s += cnt[g];
    unsigned inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned t = __shfl_up(inc, d);
      if ((int)threadIdx.x >= d) inc += t;
    }
    unsigned acc = inc - s;
    for (unsigned g = lo; g < hi; ++g) {
      loff[g] = acc;
      acc += cnt[g];
    }
  }
  __syncthreads();
  for (unsigned g = threadIdx.x; g < NB; g += CS_THREADS) {
    const unsigned v = cnt[g];
    gb[g] = v ? atomicAdd(&ccursor[col * NB + g], v) : 0u;   // this workgroup's run inside the coarse segment
    cnt[g] = loff[g];                                        // from here on: the running position inside buf
  }
  __syncthreads();
  const unsigned fmask = (1u << L) - 1;
  const int fsh = 31 - L;
  for (size_t i = i0 + threadIdx.x; i < i1; i += CS_THREADS)
    for_each_digit(scalars[col * col_stride + i], c, windows, [&](int w, u32 b, bool neg) {
      const unsigned key = b - 1;
      const unsigned pos = atomicAdd(&cnt[key >> L], 1u);
      buf[pos] = ((unsigned)w * (unsigned)n + (unsigned)i) | ((key & fmask) << fsh) | (neg ? 0x80000000u : 0u);
    });
  __syncthreads();
  unsigned *dst_col = stage + col * stage_stride;
  const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (unsigned g = wv; g < NB; g += CS_THREADS / 64) {
    const unsigned base = loff[g], len = cnt[g] - base;
    unsigned *dst = dst_col + gb[g];
    for (unsigned k = lane; k < len; k += 64) dst[k] = buf[base + k];
  }
}

// exclusive scan over FINE_CTRS = 4 * FINE_THREADS counters in place (four consecutive counters per thread); returns nothing,
// every thread calls it
__device__ __forceinline__ void fine_scan_inplace(unsigned *ctr, unsigned *wsum /* [FINE_THREADS / 64] */) {
  const unsigned t = threadIdx.x, lane = t & 63;
  const uint4 c = *(const uint4 *)(ctr + 4 * t);
  const unsigned s = c.x + c.y + c.z + c.w;
  unsigned inc = s;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned u = __shfl_up(inc, d);
    if ((int)lane >= d) inc += u;
  }
  if (lane == 63) wsum[t >> 6] = inc;
  __syncthreads();
  unsigned pre = inc - s;
  for (unsigned w = 0; w < (t >> 6); ++w) pre += wsum[w];
  *(uint4 *)(ctr + 4 * t) = make_uint4(pre, pre + c.x, pre + c.x + c.y, pre + c.x + c.y + c.z);
  __syncthreads();
}

// One workgroup per (column, coarse bin).  Scattering the entries of the segment straight to their buckets is a lone 4-byte store
// per entry, and the chip retires ~105 G of those per second whatever the window they fall into (measured: 3.0 ms per 32 columns of
// 2^19, 0.46 ms with the stores left out).  So the segment is sorted tile by tile in LDS -- FINE_TILE entries: one returning atomic
// per entry gives its arrival rank, a scan the run starts, then placement in the LDS buffer -- and leaves as one contiguous run per
// bucket and tile (~56 entries on a full-width column).
// Witness columns and narrow top windows pile thousands of entries onto a few buckets, and LDS atomics on one address serialise
// (~10 cycles per lane, measured): every bucket has FINE_COPIES counters, lane l uses copy l % FINE_COPIES -- sub-buckets that sort
// next to each other, so the order inside a bucket changes and nothing else.
__global__ void __launch_bounds__(FINE_THREADS, 4) k_msm_fine(const unsigned *__restrict__ coff, unsigned NB, int L, const unsigned *__restrict__ stage, size_t stage_stride, size_t col_entries,
                                                             unsigned K1, unsigned *__restrict__ off, unsigned *__restrict__ entries) {
  extern __shared__ unsigned buf[];   // FINE_TILE words
  __shared__ __attribute__((aligned(16))) unsigned ctr[FINE_CTRS + 4];
  __shared__ unsigned gcur[FINE_MAX], wsum[FINE_THREADS / 64];
  const size_t col = blockIdx.x / NB;
  const unsigned g = blockIdx.x % NB;
  const unsigned F = 1u << L;
  const unsigned lo = coff[col * (NB + 1) + g], hi = coff[col * (NB + 1) + g + 1];
  const unsigned *src = stage + col * stage_stride;   // 16-byte aligned: the staging stride is a multiple of four entries
  const int fsh = 31 - L;
  const unsigned fmask = F - 1;
  const unsigned t = threadIdx.x, lane = t & 63, copy = lane % FINE_COPIES;
  for (unsigned i = t; i < FINE_CTRS + 4; i += FINE_THREADS) ctr[i] = 0;
  __syncthreads();
  const unsigned a0 = lo & ~3u;
  for (unsigned p0 = a0; p0 < hi; p0 += 4 * FINE_THREADS) {   // histogram of the whole segment
    const unsigned p = p0 + 4 * t;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (p < hi) v = *(const uint4 *)(src + p);
    const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (p + j >= lo && p + j < hi) atomicAdd(&ctr[((w4[j] >> fsh) & fmask) * FINE_COPIES + copy], 1u);
  }
  __syncthreads();
  fine_scan_inplace(ctr, wsum);
  if (t < F) {
    const unsigned pre = lo + ctr[t * FINE_COPIES];
    off[col * K1 + (size_t)g * F + t] = pre;
    gcur[t] = pre;                       // where the next run of bucket t goes
  }
  if (g == NB - 1 && t == 0) off[col * K1 + (K1 - 1)] = hi;   // the column's total
  unsigned *e = entries + col * col_entries;
  const unsigned keep = 0x80000000u | ((1u << fsh) - 1);
  constexpr unsigned R = FINE_TILE / (4 * FINE_THREADS);
  for (unsigned t0 = a0; t0 < hi; t0 += FINE_TILE) {
    uint4 v[R];
#pragma unroll
    for (unsigned r = 0; r < R; ++r) {
      const unsigned p = t0 + 4 * (r * FINE_THREADS + t);
      v[r] = make_uint4(0, 0, 0, 0);
      if (p < hi) v[r] = *(const uint4 *)(src + p);
    }
    __syncthreads();   // the previous tile's runs are out, ctr is free
    for (unsigned i = t; i < FINE_CTRS + 4; i += FINE_THREADS) ctr[i] = 0;
    __syncthreads();
    unsigned arr[R][4];   // arrival rank of the entry inside its sub-bucket's run of this tile
#pragma unroll
    for (unsigned r = 0; r < R; ++r) {
      const unsigned p = t0 + 4 * (r * FINE_THREADS + t);
      const unsigned w4[4] = {v[r].x, v[r].y, v[r].z, v[r].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        arr[r][j] = 0;
        if (p + j >= lo && p + j < hi) arr[r][j] = atomicAdd(&ctr[((w4[j] >> fsh) & fmask) * FINE_COPIES + copy], 1u);
      }
    }
    __syncthreads();
    fine_scan_inplace(ctr, wsum);   // ctr[FINE_CTRS] stays 0: the last bucket's end comes from the tile's length below
#pragma unroll
    for (unsigned r = 0; r < R; ++r) {
      const unsigned p = t0 + 4 * (r * FINE_THREADS + t);
      const unsigned w4[4] = {v[r].x, v[r].y, v[r].z, v[r].w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (p + j >= lo && p + j < hi) buf[ctr[((w4[j] >> fsh) & fmask) * FINE_COPIES + copy] + arr[r][j]] = w4[j] & keep;
    }
    __syncthreads();
    const unsigned tile_lo = t0 < lo ? lo : t0, tile_hi = t0 + FINE_TILE < hi ? t0 + FINE_TILE : hi;
    for (unsigned f = t >> 6; f < F; f += FINE_THREADS / 64) {   // one wave per bucket run
      const unsigned b0 = ctr[f * FINE_COPIES], b1 = f + 1 < F ? ctr[(f + 1) * FINE_COPIES] : tile_hi - tile_lo;
      const unsigned *sp = buf + b0;
      unsigned *d = e + gcur[f];
      for (unsigned k = lane; k < b1 - b0; k += 64) d[k] = sp[k];
    }
    __syncthreads();
    if (t < F) gcur[t] += (t + 1 < F ? ctr[(t + 1) * FINE_COPIES] : tile_hi - tile_lo) - ctr[t * FINE_COPIES];
  }
}

// ---- bounded-length accumulation tasks -------------------------------------------------------------
// A bucket with cnt entries is cut into ceil(cnt / TASK_E) tasks; one thread sums one task (<= TASK_E mixed
// additions), so the longest dependent chain in the kernel no longer depends on how skewed a column is
// (witness columns hold thousands of 0/1 cells).  k_msm_task_count: tasks per column; k_msm_task_fill:
// the task list (bucket id, slice); k_msm_accumulate: one thread per task -> partial sum;
// k_msm_merge: one thread per bucket adds its partials (few), buckets with many partials go to a wave each.
// The task length is chosen per basis geometry (task_len below): ~1/5 of the bucket load of a full-width column.
constexpr int TASK_E_MAX = 64;
constexpr int MERGE_LIGHT = 8;  // partials merged by a single thread; more -> one wave per bucket
static unsigned task_len(size_t col_entries, unsigned K) {
  static int forced = -1;
  if (forced < 0) {
    const char *s = getenv("ZKFHE_TASK_E");
    forced = s ? atoi(s) : 0;
  }
  if (forced > 0) return (unsigned)forced;
  // ~3 tasks per bucket of a full-width column (profiles/r1_task_len.md): with the length-sorted task list every wave
  // runs equal chains whatever E is, so E only trades the number of partials the merge has to add against the number
  // of tasks available to fill the chip; 16 at k = 13 (40 entries per bucket) measured best end to end.
  const size_t load = col_entries / K;
  // narrow windows on a long basis (the prover's c = 10 tables for columns of small values): the nominal load says 64,
  // but those columns fill a few buckets with thousands of entries and leave the rest nearly empty -- 16 keeps the chains
  // of the full buckets short (measured: 117 proofs/s against 115 at 32 and 114 at 64)
  if (K <= 1024 && load > 128) return 16;
  unsigned e = 8;
  while (e < (unsigned)TASK_E_MAX && (size_t)e * 3 < load) e <<= 1;
  return e;
}

// Slices of a bucket with cnt entries: nt = ceil(cnt / E) slices, the first r = cnt % nt of length a + 1, the others of
// length a = cnt / nt.  The task list is ordered by slice length, longest first (counting sort over the E length
// classes): the 64 lanes of a wave then run chains of the SAME length -- with the bucket-ordered list a wave mixed
// lengths between E/2 and E and idled behind its longest lane (measured: 54 % of the modmul peak at E = 32 against
// 85 % at E = 8, where slices are nearly equal but every bucket costs five partials in the merge).
constexpr unsigned TASK_BINS = 65;  // slice lengths 0..64
struct Slices {
  unsigned nt, a, r;
};
__device__ __forceinline__ Slices slices_of(unsigned cnt, unsigned TASK_E) {
  Slices s;
  s.nt = (cnt + TASK_E - 1) / TASK_E;
  s.a = s.nt ? cnt / s.nt : 0;
  s.r = s.nt ? cnt - s.a * s.nt : 0;
  return s;
}
// position of partial j of a bucket in the task / partial arrays
__device__ __forceinline__ unsigned partial_pos(const Slices &s, unsigned posA, unsigned posB, unsigned j) { return j < s.r ? posA + j : posB + (j - s.r); }

// per column: number of slices of every length
// zero the bucket histogram and the heavy-bucket counter of a call (one launch instead of two runtime fills)
__global__ void __launch_bounds__(256) k_msm_clear(unsigned *__restrict__ hist, size_t words, unsigned *__restrict__ heavy_count) {
  const size_t i0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i0 < 4) heavy_count[i0] = 0;
  for (size_t i = i0; i < words; i += (size_t)gridDim.x * blockDim.x) hist[i] = 0;
}

__global__ void __launch_bounds__(256) k_msm_task_count(const unsigned *__restrict__ off, unsigned K, unsigned TASK_E, unsigned *__restrict__ col_hist /* [n_cols][TASK_BINS] */) {
  __shared__ unsigned h[TASK_BINS];
  for (unsigned i = threadIdx.x; i < TASK_BINS; i += 256) h[i] = 0;
  __syncthreads();
  const unsigned *o = off + (size_t)blockIdx.x * (K + 1);
  for (unsigned b = threadIdx.x; b < K; b += 256) {
    const Slices s = slices_of(o[b + 1] - o[b], TASK_E);
    if (!s.nt) continue;
    if (s.r) atomicAdd(&h[s.a + 1], s.r);
    atomicAdd(&h[s.a], s.nt - s.r);
  }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < TASK_BINS; i += 256) col_hist[(size_t)blockIdx.x * TASK_BINS + i] = h[i];
}
// first task of every (length, column): lengths descending, columns ascending inside a length.  One block of 128 threads.
__global__ void __launch_bounds__(128) k_msm_task_colscan(const unsigned *__restrict__ col_hist, unsigned n_cols, unsigned *__restrict__ col_base /* [n_cols][TASK_BINS] */,
                                                          unsigned *__restrict__ n_tasks) {
  __shared__ unsigned bin_total[TASK_BINS], bin_base[TASK_BINS];
  const unsigned L = threadIdx.x;
  if (L < TASK_BINS) {
    unsigned s = 0;
    for (unsigned c = 0; c < n_cols; ++c) s += col_hist[(size_t)c * TASK_BINS + L];
    bin_total[L] = s;
  }
  __syncthreads();
  if (L == 0) {
    unsigned acc = 0;
    for (int l = (int)TASK_BINS - 1; l >= 0; --l) {
      bin_base[l] = acc;
      acc += bin_total[l];
    }
    *n_tasks = acc;
  }
  __syncthreads();
  if (L < TASK_BINS) {
    unsigned acc = bin_base[L];
    for (unsigned c = 0; c < n_cols; ++c) {
      col_base[(size_t)c * TASK_BINS + L] = acc;
      acc += col_hist[(size_t)c * TASK_BINS + L];
    }
  }
}
// per column: hand out the positions (LDS cursors per length class) and write the task list
constexpr unsigned MERGE_MEDIUM = 128;  // up to this many partials: eight lanes per bucket; beyond: one wave per bucket
__global__ void __launch_bounds__(256) k_msm_task_fill(const unsigned *__restrict__ off, unsigned K, unsigned TASK_E, const unsigned *__restrict__ col_base,
                                                       unsigned *__restrict__ bucket_posA, unsigned *__restrict__ bucket_posB /* [n_cols][K] each */,
                                                       uint2 *__restrict__ tasks, unsigned *__restrict__ heavy_count /* [0] medium, [1] large */,
                                                       unsigned *__restrict__ medium_list, unsigned *__restrict__ large_list, unsigned heavy_cap) {
  __shared__ unsigned cur[TASK_BINS];
  const size_t col = blockIdx.x;
  for (unsigned i = threadIdx.x; i < TASK_BINS; i += 256) cur[i] = col_base[col * TASK_BINS + i];
  __syncthreads();
  const unsigned *o = off + col * (K + 1);
  for (unsigned b = threadIdx.x; b < K; b += 256) {
    const Slices s = slices_of(o[b + 1] - o[b], TASK_E);
    if (!s.nt) continue;
    const unsigned g = (unsigned)(col * K + b);
    const unsigned posA = s.r ? atomicAdd(&cur[s.a + 1], s.r) : 0u;
    const unsigned posB = atomicAdd(&cur[s.a], s.nt - s.r);
    bucket_posA[g] = posA;
    bucket_posB[g] = posB;
    if (s.nt > (unsigned)MERGE_LIGHT) {   // the merge kernel's block ranges for buckets with many partials
      const bool large = s.nt > MERGE_MEDIUM;
      const unsigned slot = atomicAdd(&heavy_count[large ? 1 : 0], 1u);
      if (slot < heavy_cap) (large ? large_list : medium_list)[slot] = g;
    }
    for (unsigned j = 0; j < s.r; ++j) tasks[posA + j] = make_uint2(g, j);
    for (unsigned j = s.r; j < s.nt; ++j) tasks[posB + (j - s.r)] = make_uint2(g, j);
  }
}

// 16-byte load of four consecutive entry words at a 4-byte aligned address (the entry array is padded by 32 bytes)
__device__ __forceinline__ uint4 ld_entries(const unsigned *p) {
  uint4 v;
  __builtin_memcpy(&v, p, 16);
  return v;
}

__global__ void __launch_bounds__(256) k_msm_accumulate(const uint2 *__restrict__ tasks, const unsigned *__restrict__ n_tasks_ptr,
                                                        const unsigned *__restrict__ off, const unsigned *__restrict__ entries,
                                                        size_t col_entries, const G1Affine *__restrict__ table, unsigned K, unsigned TASK_E,
                                                        G1X *__restrict__ partials) {
  const unsigned n_tasks = *n_tasks_ptr;
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n_tasks; t += (size_t)gridDim.x * blockDim.x) {
    const uint2 tk = tasks[t];
    const size_t col = tk.x / K;
    const unsigned b = tk.x - (unsigned)col * K;
    const unsigned *o = off + col * (K + 1);
    const Slices s = slices_of(o[b + 1] - o[b], TASK_E);
    const unsigned lo = o[b] + (tk.y < s.r ? tk.y * (s.a + 1) : tk.y * s.a + s.r);
    const unsigned hi = lo + s.a + (tk.y < s.r ? 1u : 0u);
    const unsigned *e = entries + col * col_entries;
    // software pipeline: the table gather of entry k+1 is in flight while entry k is added (PMC showed 40 % of the wave
    // cycles of the plain loop waiting on this dependent load chain).  The entry indices themselves come in 16-byte
    // loads into an 8-entry shift window: with one 4-byte load per entry and lanes 4*E bytes apart, every lane pulled
    // its own cache line through L1 once per entry.
    G1X29 acc = G1X29::identity();
    const unsigned *ep = e + lo;
    uint4 w0 = ld_entries(ep), w1 = ld_entries(ep + 4);
    unsigned valid = 8;
    G1Affine p = table[w0.x & 0x7fffffffu];
    const unsigned cnt_e = hi - lo;
    for (unsigned i = 0; i < cnt_e; ++i) {
      const unsigned en = w0.x, en1 = w0.y;
      G1Affine pn = p;
      if (i + 1 < cnt_e) pn = table[en1 & 0x7fffffffu];
      g1x29_add_affine(acc, g1a29_load(p), (en >> 31) != 0);
      p = pn;
      w0.x = w0.y, w0.y = w0.z, w0.z = w0.w, w0.w = w1.x;
      w1.x = w1.y, w1.y = w1.z, w1.z = w1.w;
      if (--valid == 4) {
        w1 = ld_entries(ep + i + 5);   // entries i+5 .. i+8 of the slice (window now starts at i+1)
        valid = 8;
      }
    }
    partials[t] = g1x29_store(acc);
  }
}

__device__ __forceinline__ G1X29 g1x_shfl_down(const G1X29 &p, int delta) {
  G1X29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    r.x.l[i] = __shfl_down(p.x.l[i], delta);
    r.y.l[i] = __shfl_down(p.y.l[i], delta);
    r.zz.l[i] = __shfl_down(p.zz.l[i], delta);
    r.zzz.l[i] = __shfl_down(p.zzz.l[i], delta);
  }
  return r;
}

__device__ __forceinline__ G1X29 g1x_shfl_xor(const G1X29 &p, int mask) {
  G1X29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    r.x.l[i] = __shfl_xor(p.x.l[i], mask);
    r.y.l[i] = __shfl_xor(p.y.l[i], mask);
    r.zz.l[i] = __shfl_xor(p.zz.l[i], mask);
    r.zzz.l[i] = __shfl_xor(p.zzz.l[i], mask);
  }
  return r;
}

// bucket sum = sum of its partials, ONE launch with three block ranges that run side by side (they used to be two
// kernels and two loops, one after the other: 650 us of dependent chains per call):
//   large_blocks: skewed witness columns (thousands of 0/1 cells in one bucket): one wave per bucket, 6-step tree;
//   medium_blocks: buckets with <= MERGE_MEDIUM partials -- every column has ~2^(254 mod c) of them, the narrow top
//     window piles its digits onto them -- eight lanes per bucket, lanes stride over the partials, 3-step butterfly;
//   the rest: one thread per bucket with <= MERGE_LIGHT partials (every bucket of a full-width column).
// The medium / large lists are written by k_msm_task_fill.
__global__ void __launch_bounds__(256) k_msm_merge(const unsigned *__restrict__ off, const unsigned *__restrict__ bucket_posA, const unsigned *__restrict__ bucket_posB,
                                                   const G1X *__restrict__ partials, unsigned K, unsigned TASK_E, size_t n_cols, G1X *__restrict__ buckets,
                                                   const unsigned *__restrict__ heavy_count, const unsigned *__restrict__ medium_list,
                                                   const unsigned *__restrict__ large_list, unsigned heavy_cap, unsigned large_blocks, unsigned medium_blocks) {
  // block order = dispatch order: the long chains (large, then medium) go first, the light range fills the chip behind them
  const unsigned lane = threadIdx.x & 63;
  if (blockIdx.x < large_blocks) {
    unsigned cnt = heavy_count[1];
    if (cnt > heavy_cap) cnt = heavy_cap;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned n_waves = (large_blocks * blockDim.x) >> 6;
    for (unsigned h = wave; h < cnt; h += n_waves) {
      const size_t g = large_list[h];
      const size_t col = g / K;
      const unsigned b = (unsigned)(g - col * K);
      const unsigned *o = off + col * (K + 1);
      const Slices s = slices_of(o[b + 1] - o[b], TASK_E);
      const unsigned nt = s.nt;
      const unsigned pa = bucket_posA[g], pb = bucket_posB[g];
      G1X29 acc = G1X29::identity();
      for (unsigned j = lane; j < nt; j += 64) g1x29_add(acc, g1x29_load(partials[partial_pos(s, pa, pb, j)]));
      for (int d = 32; d > 0; d >>= 1) {
        const G1X29 other = g1x_shfl_down(acc, d);
        if ((int)lane < d) g1x29_add(acc, other);
      }
      if (lane == 0) buckets[g] = g1x29_store(acc);
    }
    return;
  }
  if (blockIdx.x < large_blocks + medium_blocks) {
    unsigned cnt = heavy_count[0];
    if (cnt > heavy_cap) cnt = heavy_cap;
    const unsigned wave = ((blockIdx.x - large_blocks) * blockDim.x + threadIdx.x) >> 6;
    const unsigned n_waves = (medium_blocks * blockDim.x) >> 6;
    for (unsigned base = wave * 8; base < cnt; base += n_waves * 8) {
      const unsigned h = base + (lane >> 3), sub = lane & 7;
      size_t g = 0;
      unsigned nt = 0, pa = 0, pb = 0;
      Slices s = {0, 0, 0};
      if (h < cnt) {
        g = medium_list[h];
        const size_t col = g / K;
        const unsigned b = (unsigned)(g - col * K);
        const unsigned *o = off + col * (K + 1);
        s = slices_of(o[b + 1] - o[b], TASK_E);
        nt = s.nt;
        pa = bucket_posA[g];
        pb = bucket_posB[g];
      }
      G1X29 acc = G1X29::identity();
      for (unsigned j = sub; j < nt; j += 8) g1x29_add(acc, g1x29_load(partials[partial_pos(s, pa, pb, j)]));
      for (int m = 1; m < 8; m <<= 1) {
        const G1X29 other = g1x_shfl_xor(acc, m);
        g1x29_add(acc, other);
      }
      if (nt && sub == 0) buckets[g] = g1x29_store(acc);
    }
    return;
  }
  const unsigned first = large_blocks + medium_blocks, light_blocks = gridDim.x - first;
  const size_t total = (size_t)K * n_cols;
  for (size_t g = (blockIdx.x - first) * (size_t)blockDim.x + threadIdx.x; g < total; g += (size_t)light_blocks * blockDim.x) {
    const size_t col = g / K;
    const unsigned b = (unsigned)(g - col * K);
    const unsigned *o = off + col * (K + 1);
    const Slices s = slices_of(o[b + 1] - o[b], TASK_E);
    const unsigned nt = s.nt;
    if (nt > (unsigned)MERGE_LIGHT) continue;
    G1X first = G1X::identity();
    if (nt) {
      const unsigned pa = bucket_posA[g], pb = bucket_posB[g];
      first = partials[partial_pos(s, pa, pb, 0)];
      if (nt > 1) {
        G1X29 acc = g1x29_load(first);
        G1X nxt = partials[partial_pos(s, pa, pb, 1)];
        for (unsigned j = 1; j < nt; ++j) {  // the load of partial j+1 is in flight while partial j is added
          const G1X cur = nxt;
          if (j + 1 < nt) nxt = partials[partial_pos(s, pa, pb, j + 1)];
          g1x29_add(acc, g1x29_load(cur));
        }
        first = g1x29_store(acc);
      }
    }
    buckets[g] = first;
  }
}

__device__ __forceinline__ G1X g1x_mul_pow2(G1X p, int k) {
  for (int i = 0; i < k; ++i) p = g1x_dbl(p);
  return p;
}
__device__ __forceinline__ G1X29 g1x29_mul_pow2(G1X29 p, int k) {
  for (int i = 0; i < k; ++i) p = g1x29_dbl(p);
  return p;
}

// ---- fast bucket reduction for 64 <= K <= 32768 -------------------------------------------------------
// bucket index idx = 64 a + b carries weight idx + 1, so
//     sum (idx+1) B = 64 * sum_a a R_a  +  sum_b (b+1) C_b ,   R_a = sum_b B[a][b],  C_b = sum_a B[a][b].
// k_msm_marginals: L lanes per row / column sum -- every lane adds up to 64/L buckets serially, then a butterfly over
// the L lanes.  L = 8 for calls with many columns (1.4 lane-additions per bucket and marginal instead of the 6 of a
// whole-wave butterfly), L = 64 for calls of a few columns (shortest dependent chain).  lane = sub * 8 + g: the 8 outputs of a wave sit in the low lane bits so that column sums read
// consecutive buckets across lanes.
// k_msm_weighted: two waves per MSM; lane a forms a * R_a by double-and-add (<= 7 bits) and a 6-step butterfly sums
// the lanes; then one lane normalises.
__global__ void __launch_bounds__(256) k_msm_marginals(const G1X *__restrict__ buckets, unsigned K, size_t n_cols, unsigned L /* lanes per output: 8 or 64 */,
                                                       unsigned w0, unsigned w_cnt /* outputs w0 .. w0 + w_cnt - 1 of every column */,
                                                       G1X *__restrict__ marg /* [n_cols][A + 64] */) {
  const unsigned A = K >> 6;
  const unsigned per_col = A + 64;
  const unsigned G = 64 / L;  // outputs per wave
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
  const unsigned lane = threadIdx.x & 63, g = lane & (G - 1), sub = lane / G;
  const size_t o = wave * G + g;  // output index over all columns, within the [w0, w0 + w_cnt) slice
  const bool live = o < n_cols * w_cnt;
  const size_t col = live ? o / w_cnt : 0;
  const unsigned w = w0 + (unsigned)(o - col * w_cnt);
  const G1X *B = buckets + col * K;
  size_t base, stride;
  unsigned cnt;
  if (w < A) {  // row w: buckets w*64 + sub + L i
    base = (size_t)w * 64 + sub;
    stride = L;
    cnt = 64 / L;
  } else {      // column w - A: buckets (sub + L i) * 64 + (w - A)
    base = (size_t)sub * 64 + (w - A);
    stride = (size_t)L * 64;
    cnt = A > sub ? (A - sub + L - 1) / L : 0;
  }
  if (!live) cnt = 0;
  G1X29 v = G1X29::identity();
  for (unsigned i = 0; i < cnt; ++i) g1x29_add(v, g1x29_load(B[base + i * stride]));
  for (int m = (int)G; m < 64; m <<= 1) {
    const G1X29 other = g1x_shfl_xor(v, m);
    g1x29_add(v, other);
  }
  if (live && sub == 0) marg[col * per_col + w] = g1x29_store(v);
}

// one block per MSM: waves 0 .. ceil(A/64)-1 weight the row sums (64 a * R_a), the last wave the column sums ((b+1) * C_b).
// Every lane runs a double-and-add over exactly the bits its wave needs (log2 A for the rows, 7 for the columns); the row
// lanes then double six more times (the factor 64) while the column wave is still busy, so that after the butterflies one
// lane only adds the wave sums and normalises: 30 dependent point operations at K = 4096, 26 at K = 512 (it was 37).
// XYZZ: the sum leaves the kernel as it is (128 B, standard Montgomery form) and the caller normalises a whole round's points with
// one inversion on the host (zkfhe_g1_xyzz_to_affine) -- the 40 us Bernstein-Yang inversion of one lane is the tail of every call.
template <bool XYZZ>
__global__ void __launch_bounds__(576) k_msm_weighted(const G1X *__restrict__ marg, unsigned K, void *__restrict__ out) {
  __shared__ G1X sh[9];
  const unsigned A = K >> 6;
  const unsigned row_waves = (A + 63) / 64;
  const size_t col = blockIdx.x;
  const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const G1X *M = marg + col * (A + 64);
  const bool rows = wv < row_waves;
  const unsigned a = wv * 64 + lane;
  const G1X29 P = g1x29_load(rows ? (a < A ? M[a] : G1X::identity()) : M[A + lane]);
  const unsigned k = rows ? a : lane + 1;  // weights a (rows, times 64 below) and b + 1 (columns)
  int nbits = 7;
  if (rows) {
    nbits = 0;
    while ((1u << nbits) < A) ++nbits;
  }
  G1X29 W = G1X29::identity();
  for (int bit = nbits - 1; bit >= 0; --bit) {
    W = g1x29_dbl(W);
    if ((k >> bit) & 1) g1x29_add(W, P);
  }
  if (rows) W = g1x29_mul_pow2(W, 6);
  for (int m = 1; m < 64; m <<= 1) {
    const G1X29 other = g1x_shfl_xor(W, m);
    g1x29_add(W, other);
  }
  if (lane == 0) sh[wv] = g1x29_store(W);
  __syncthreads();
  if (threadIdx.x == 0) {
    G1X29 t = g1x29_load(sh[0]);
    for (unsigned w = 1; w <= row_waves; ++w) g1x29_add(t, g1x29_load(sh[w]));
    // back to the standard Montgomery form, then one inversion (or none: XYZZ)
    if (XYZZ) ((G1X *)out)[col] = g1x29_to_std(t);
    else ((G1Affine *)out)[col] = g1x_to_affine(g1x29_to_std(t));
  }
}

// K < 64 buckets (window_bits < 7: tiny bases and tests): one wave per MSM, lane b weights bucket b by b + 1
template <bool XYZZ>
__global__ void __launch_bounds__(64) k_msm_small(const G1X *__restrict__ buckets, unsigned K, void *__restrict__ out) {
  const size_t col = blockIdx.x;
  const unsigned lane = threadIdx.x;
  const G1X29 P = g1x29_load(lane < K ? buckets[col * K + lane] : G1X::identity());
  const unsigned k = lane + 1;
  G1X29 W = G1X29::identity();
  for (int bit = 6; bit >= 0; --bit) {
    W = g1x29_dbl(W);
    if ((k >> bit) & 1) g1x29_add(W, P);
  }
  for (int m = 1; m < 64; m <<= 1) {
    const G1X29 other = g1x_shfl_xor(W, m);
    g1x29_add(W, other);
  }
  if (lane == 0) {
    if (XYZZ) ((G1X *)out)[col] = g1x29_to_std(W);
    else ((G1Affine *)out)[col] = g1x_to_affine(g1x29_to_std(W));
  }
}

// ---- digit-multiple table path ---------------------------------------------------------------------------------------
// With T[(i*W + w)*J + j-1] = j * 2^(c*w) * P_i  (j = 1..J = 2^(c-1), W = ceil(255/c) signed windows) resident in HBM an MSM is
// a plain sum of at most n*W table points: no sort, no buckets, no weights, no doublings.  At n = 2^13, c = 12 the table is
// 23.6 GB of the 288 -- the SRS is fixed for the life of the prover, HBM is what this chip has most of.
struct BiasArg {
  u32 l[9];   // sum_w 2^(c*w + c-1): with it, window w of (s + bias) minus 2^(c-1) is a signed digit in [-J, J-1] and the digits
};            // sum to s without a carry chain; the zero windows of a short scalar stay zero digits

__device__ __forceinline__ u32 bits_at(const u32 (&l)[9], int bit, int c) {
  const int limb = bit >> 5, sh = bit & 31;
  u32 lo = 0, hi = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {   // selects, not indexed loads: the limbs stay in registers
    if (k == limb) lo = l[k];
    if (k == limb + 1) hi = l[k];
  }
  return (u32)((((u64)hi << 32) | lo) >> sh) & ((1u << c) - 1);
}

// |s| (sign-magnitude, < 2^253) + bias; returns the sign, and in `bits` the bit length of |s|: windows above
// bits / c + 1 hold zero digits (a negative digit carries one into the next window, not further)
__device__ __forceinline__ bool biased_scalar(const Fr &mont, const BiasArg &B, u32 (&out)[9], int &bits) {
  Fr s = fp_from_mont<FrP>(mont);
  const bool neg = fr_gt_half(s);
  if (neg) s = fp_neg<FrP>(s);
  bits = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (s.l[k]) bits = 32 * k + 32 - __clz(s.l[k]);
  u32 carry = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const u64 v = (u64)s.l[k] + B.l[k] + carry;
    out[k] = (u32)v;
    carry = (u32)(v >> 32);
  }
  out[8] = B.l[8] + carry;
  return neg;
}

// One thread per (point, window) row: j * B for j = 1..J, each normalised on its own (a 40 us inversion per entry and lane;
// 369 M entries at n = 2^13, c = 12 -- a quarter of a second per basis, once).
__global__ void __launch_bounds__(64) k_basis_multiples(const G1Affine *__restrict__ bases, size_t n, int c, int W, G1Affine *__restrict__ T) {
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t >= n * (size_t)W) return;
  const size_t i = t / (size_t)W;
  const int w = (int)(t % (size_t)W);
  const size_t J = (size_t)1 << (c - 1);
  G1Affine *row = T + t * J;
  G1Affine b = bases[i];
  if (!b.is_identity() && w) b = g1x_to_affine(g1x_mul_pow2(g1x_from_affine(b), c * w));
  if (b.is_identity()) {
    for (size_t j = 0; j < J; ++j) row[j] = b;
    return;
  }
  row[0] = g1_affine_to_29(b);   // stored in the 2^261 Montgomery form of the MSM kernels
  if (J == 1) return;
  G1X acc = g1x_from_affine_dbl(b);
  row[1] = g1_affine_to_29(g1x_to_affine(acc));
  for (size_t j = 2; j < J; ++j) {
    g1x_add_affine(acc, b, false);
    row[j] = g1_affine_to_29(g1x_to_affine(acc));
  }
}

// Sum of 256 XYZZ points held one per thread -> thread 0.  Across the waves first, through LDS (waves 2, 3 hand theirs to
// waves 0, 1, then wave 1 to wave 0: three wave-wide additions), then a 6-step butterfly in wave 0 alone: 9 wave-wide
// additions where a 256-lane butterfly issues 26.
__device__ __forceinline__ G1X29 block_sum_256(G1X29 v, G1X *sh /* [128] */) {
  const unsigned wv = threadIdx.x >> 6;
  __syncthreads();
  if (wv >= 2) sh[threadIdx.x - 128] = g1x29_store(v);
  __syncthreads();
  if (wv < 2) g1x29_add(v, g1x29_load(sh[threadIdx.x]));
  __syncthreads();
  if (wv == 1) sh[threadIdx.x - 64] = g1x29_store(v);
  __syncthreads();
  if (wv == 0) {
    g1x29_add(v, g1x29_load(sh[threadIdx.x]));
    for (int m = 1; m < 64; m <<= 1) {
      const G1X29 other = g1x_shfl_xor(v, m);
      g1x29_add(v, other);
    }
  }
  return v;
}

// Work item = chunk of P <= 256 consecutive points of one column.  Every column has a chunk counter; a grid of persistent
// workgroups (three per CU) is spread over the columns, each draws chunks of its column until the counter runs out and then
// moves on to the next column that has chunks left -- a workgroup on a full-width column draws few chunks, one on a column of
// 0/1 cells many, and the accumulators live as long as the workgroup stays with a column (witness columns mix 0/1 cells,
// 8-bit range cells and full-width values: 0.5 to 22 additions per scalar).  Per chunk: the non-zero digits of the P
// scalars are compacted into an LDS list of table offsets (digit mask per thread, prefix sum over the workgroup), then
// the 256 threads take the list entries round-robin: every lane has the same number of mixed additions (+-1).
// Leaving a column, the workgroup appends to that column's partial list: with TREE one butterfly and one partial (calls of a
// few columns: thousands of short visits per column, the fold must stay short); without, every thread stores its accumulator
// as it is -- 256 partials per visit and no butterfly (eight dependent point additions with most lanes idle cost as much as
// eleven useful ones), and no point-addition code besides the loop's in the kernel: 167 VGPRs (three waves per SIMD; rocprof
// shows the arch / accumulation halves of the unified file), 12 bytes of scratch -- two values of the digit extraction that
// live across the addition loop.  k_msm_table_fold sums the lists.
template <bool TREE>
__global__ void __launch_bounds__(256, TREE ? 1 : 3) k_msm_table(const Fr *__restrict__ scalars, size_t col_stride, size_t n, const G1Affine *__restrict__ T, int c, int W, BiasArg B,
                                                   unsigned P, unsigned cpc /* chunks per column */, unsigned n_cols, unsigned max_part,
                                                   G1X *__restrict__ partials /* [n_cols][max_part][TREE ? 1 : 256] */,
                                                   unsigned *__restrict__ n_part /* [n_cols] visits recorded, zero on entry */,
                                                   unsigned *__restrict__ col_next /* [n_cols] next chunk, zero on entry */,
                                                   unsigned *__restrict__ lists /* [gridDim.x][P * W] */, unsigned long long *__restrict__ adds) {
  // the entry list of the chunk in flight: in global memory (written and read by this workgroup only: it stays in this CU's
  // L1 / this XCD's L2), not in LDS -- with no LDS to its name the kernel shares a CU with the NTT tile kernel (147 KB of the
  // 160), whose waves wait on LDS while these issue multiply-adds
  unsigned *__restrict__ lst = lists + (size_t)blockIdx.x * P * (unsigned)W;
  __shared__ G1X sh[TREE ? 128 : 1];
  __shared__ unsigned wave_cnt[4], item_sh;
  const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int half = 1 << (c - 1);
  G1X29 acc = G1X29::identity();
  unsigned col = (unsigned)(((unsigned long long)blockIdx.x * n_cols) / gridDim.x), done = 0, total = 0;
  for (;;) {
    __syncthreads();   // the previous chunk's list is consumed
    if (threadIdx.x == 0) item_sh = atomicAdd(&col_next[col], 1u);
    __syncthreads();
    const unsigned chunk = item_sh;
    if (chunk >= cpc) {
      // this column has no chunks left: hand over what was summed, then look for the next column that has some
      if (done) {
        if (TREE) acc = block_sum_256(acc, sh);
        __syncthreads();
        if (threadIdx.x == 0) item_sh = atomicAdd(&n_part[col], 1u);
        __syncthreads();
        const size_t slot = (size_t)col * max_part + item_sh;
        if (TREE) {
          if (threadIdx.x == 0) partials[slot] = g1x29_store(acc);
        } else {
          partials[slot * 256 + threadIdx.x] = g1x29_store(acc);
        }
        acc = G1X29::identity();
        done = 0;
      }
      // next: the column with the most chunks left (not the neighbour: the workgroups would pile up on it for a chunk each,
      // and every visit costs the fold 256 additions)
      __syncthreads();
      if (threadIdx.x == 0) item_sh = 0;
      __syncthreads();
      // A column is joined only while it has at least MIN_JOIN chunks left (or has not been started): the last chunks of a
      // column stay with the workgroups that are on it.  Without the floor every idle workgroup of the call's tail jumped onto
      // the last busy columns for ONE chunk each -- up to 32 visits on a column of 32 chunks, and the fold's duration is that
      // of its longest column (measured: visits per column 6 on average, 30 at the maximum; 320 - 360 us per wide fold).
      // ... and among the columns that qualify every workgroup has its own order of preference (a hash of column and
      // workgroup): choosing "the column with the most chunks left" sent all workgroups that went idle at the same moment to the
      // same column.
      const unsigned MIN_JOIN = cpc < 4u ? cpc : 4u;
      unsigned best = 0;   // (preference << 12) | column, n_cols <= 4096
      for (unsigned j = threadIdx.x; j < n_cols; j += 256) {
        const unsigned nx = __hip_atomic_load(&col_next[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (nx < cpc && (nx == 0 || cpc - nx >= MIN_JOIN)) {
          const unsigned h = ((j * 2654435761u) ^ (blockIdx.x * 40503u + 12345u) * 2246822519u) >> 13;   // 19 bits, never zero with the +1 below
          best = max(best, (((h & 0x7ffffu) + 1u) << 12) | j);
        }
      }
      if (best) atomicMax(&item_sh, best);
      __syncthreads();
      const unsigned found = item_sh;
      if (!found) break;
      col = found & 4095u;
      continue;
    }
    ++done;
    {
      const size_t i = (size_t)chunk * P + threadIdx.x;
      const bool valid = threadIdx.x < P && i < n;
      u32 sb[9];
      bool neg = false;
      unsigned long long mask = 0;
      if (valid) {
        int bits;
        neg = biased_scalar(scalars[(size_t)col * col_stride + i], B, sb, bits);
        const int wtop = min(W, bits / c + 2);
        for (int w = 0; w < wtop; ++w)
          if ((int)bits_at(sb, w * c, c) != half) mask |= 1ull << w;
      }
      const unsigned mine = (unsigned)__popcll(mask);
      unsigned incl = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned t = __shfl_up(incl, d);
        if ((int)lane >= d) incl += t;
      }
      if (lane == 63) wave_cnt[wv] = incl;
      __syncthreads();
      unsigned off = incl - mine;
      for (unsigned v = 0; v < wv; ++v) off += wave_cnt[v];
      // table offsets relative to the chunk's first point (31 bits + sign: P * W * 2^(c-1) <= 256 * 32 * 2^14; the whole table of a
      // 15-bit Lagrange half has more than 2^31 entries)
      const unsigned row0 = threadIdx.x * (unsigned)W;
      while (mask) {
        const int w = __builtin_ctzll(mask);
        mask &= mask - 1;
        const int d = (int)bits_at(sb, w * c, c) - half;
        lst[off++] = (((row0 + (unsigned)w) << (c - 1)) + (unsigned)(d < 0 ? -d : d) - 1u) | ((neg != (d < 0)) ? 0x80000000u : 0u);
      }
    }
    __syncthreads();
    const unsigned M = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    total += M;
    // software pipeline: the entry two steps ahead and the table point one step ahead are in flight during an addition
    const G1Affine *__restrict__ Tc = T + (((size_t)chunk * P * (size_t)W) << (c - 1));   // this chunk's rows of the table
    unsigned e = threadIdx.x, en = 0, en2 = 0;
    G1Affine p;
    if (e < M) {
      en = lst[e];
      p = Tc[en & 0x7fffffffu];
    }
    if (e + 256 < M) en2 = lst[e + 256];
    while (e < M) {
      const unsigned e2 = e + 256;
      unsigned en3 = 0;
      G1Affine p2;
      if (e2 < M) p2 = Tc[en2 & 0x7fffffffu];
      if (e2 + 256 < M) en3 = lst[e2 + 256];
      g1x29_add_affine(acc, g1a29_load(p), (en >> 31) != 0);
      e = e2;
      en = en2;
      en2 = en3;
      p = p2;
    }
  }
  if (adds && threadIdx.x == 0 && total) atomicAdd(adds, (unsigned long long)total);
}

// out[col] = sum of the column's partial list (n_part[col] * L entries), normalised; the counters go back to zero.
// 512 threads: with L = 256 (one accumulator per thread and visit) the two halves of the workgroup take the even and the odd
// visits -- a column of the wide calls collects 25 to 30 visits (every workgroup that drew a chunk of it), and the visits are
// the sequential part of this kernel: visits / 2 + 9 dependent point additions instead of visits + 8.  The next partial is in
// flight during an addition.
template <bool XYZZ>
__global__ void __launch_bounds__(512) k_msm_table_fold(const G1X *__restrict__ partials, unsigned max_part, unsigned L, unsigned *__restrict__ n_part,
                                                        unsigned *__restrict__ col_next, void *__restrict__ out) {
  __shared__ G1X sh[256];
  const unsigned col = blockIdx.x, t = threadIdx.x & 255u, q = threadIdx.x >> 8;
  const unsigned np = n_part[col] * L;
  const G1X *mine = partials + (size_t)col * max_part * L;
  G1X29 f = G1X29::identity();
  {
    // L = 256: entry v * 256 + t, v = q, q + 2, ...; L = 1: entries threadIdx.x, threadIdx.x + 512, ...
    const unsigned first = L == 256 ? q * 256u + t : threadIdx.x, step = 512u;
    unsigned b = first;
    G1X cur;
    if (b < np) cur = mine[b];
    while (b < np) {
      const unsigned nb = b + step;
      G1X nxt;
      if (nb < np) nxt = mine[nb];
      g1x29_add(f, g1x29_load(cur));
      cur = nxt;
      b = nb;
    }
  }
  // 512 -> 256 -> 128 -> 64 through LDS, then a butterfly in the first wave
  if (q == 1) sh[t] = g1x29_store(f);
  __syncthreads();
  if (q == 0) g1x29_add(f, g1x29_load(sh[t]));
  __syncthreads();
  if (q == 0 && t >= 128) sh[t - 128] = g1x29_store(f);
  __syncthreads();
  if (q == 0 && t < 128) g1x29_add(f, g1x29_load(sh[t]));
  __syncthreads();
  if (q == 0 && t >= 64 && t < 128) sh[t - 64] = g1x29_store(f);
  __syncthreads();
  if (threadIdx.x < 64) {
    g1x29_add(f, g1x29_load(sh[t]));
    for (int m = 1; m < 64; m <<= 1) {
      const G1X29 other = g1x_shfl_xor(f, m);
      g1x29_add(f, other);
    }
    if (threadIdx.x == 0) {
      if (XYZZ) ((G1X *)out)[col] = g1x29_to_std(f);   // normalised by the caller, a round's points at a time
      else ((G1Affine *)out)[col] = g1x_to_affine(g1x29_to_std(f));
      n_part[col] = 0;
      col_next[col] = 0;
    }
  }
}

// table[w][i] = 2^(c*w) * P_i
__global__ void __launch_bounds__(256) k_basis_table(const G1Affine *__restrict__ bases, size_t n, int c, int windows,
                                                     G1Affine *__restrict__ table) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  G1Affine p = bases[i];
  table[i] = g1_affine_to_29(p);   // the tables hold the 2^261 Montgomery form of the MSM kernels (fq29.hip.hpp)
  G1X cur = g1x_from_affine(p);
  for (int w = 1; w < windows; ++w) {
    cur = g1x_mul_pow2(cur, c);
    G1Affine a = g1x_to_affine(cur);
    table[(size_t)w * n + i] = g1_affine_to_29(a);
    cur = g1x_from_affine(a);
  }
}

__global__ void __launch_bounds__(256) k_g1_add(const G1Affine *__restrict__ a, const G1Affine *__restrict__ b,
                                                G1Affine *__restrict__ out, size_t n) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  G1X acc = g1x_from_affine(a[i]);
  g1x_add_affine(acc, b[i], false);
  out[i] = g1x_to_affine(acc);
}

__global__ void __launch_bounds__(256) k_g1_mul(const G1Affine *__restrict__ p, const Fr *__restrict__ k,
                                                G1Affine *__restrict__ out, size_t n) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Fr s = fp_from_mont<FrP>(k[i]);
  const G1Affine base = p[i];
  G1X acc = G1X::identity();
  for (int bit = 255; bit >= 0; --bit) {
    acc = g1x_dbl(acc);
    if ((s.l[bit >> 5] >> (bit & 31)) & 1) g1x_add_affine(acc, base, false);
  }
  out[i] = g1x_to_affine(acc);
}

// A handful of non-zero scalars against a basis with a digit-multiple table: out[slot] = sum over the cells of that slot of
// scalar * P_row.  One wave per slot, lanes over the windows, a butterfly, one normalisation.  (The prover's early
// phase-1 commitment: the 16 challenge-dependent gate cells of the constrain_mul gates, as corrections to <= 4 columns.)
template <bool XYZZ>
__global__ void __launch_bounds__(64) k_msm_sparse(const zkfhe_sparse_term *__restrict__ terms, unsigned n_terms, const G1Affine *__restrict__ T, int c, int W,
                                                   BiasArg B, void *__restrict__ out) {
  const unsigned slot = blockIdx.x, lane = threadIdx.x;
  const int half = 1 << (c - 1);
  G1X29 acc = G1X29::identity();
  for (unsigned t = 0; t < n_terms; ++t) {
    if (terms[t].slot != slot) continue;
    Fr s;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s.l[2 * i] = (u32)terms[t].scalar.l[i];
      s.l[2 * i + 1] = (u32)(terms[t].scalar.l[i] >> 32);
    }
    u32 sb[9];
    int bits;
    const bool neg = biased_scalar(s, B, sb, bits);
    const int d = (int)lane < W ? (int)bits_at(sb, (int)lane * c, c) - half : 0;   // window = lane (W <= 64)
    if (d) {
      const G1Affine p = T[(((size_t)terms[t].row * W + lane) << (c - 1)) + (size_t)(d < 0 ? -d : d) - 1];
      g1x29_add_affine(acc, g1a29_load(p), neg != (d < 0));
    }
  }
  for (int m = 1; m < 64; m <<= 1) {
    const G1X29 other = g1x_shfl_xor(acc, m);
    g1x29_add(acc, other);
  }
  if (lane == 0) {
    if (XYZZ) ((G1X *)out)[slot] = g1x29_to_std(acc);
    else ((G1Affine *)out)[slot] = g1x_to_affine(g1x29_to_std(acc));
  }
}

// Window width of the digit-multiple table: the widest (<= 15 bits) whose table fits the per-basis budget.
//   * ZKFHE_TABLE_GB unset (the library default): 48 GB, and never more than a quarter of the memory that is free on the device when
//     the basis is made -- n = 2^13: 13-bit digits for the Lagrange half of an SRS (43 GB), 11 for the monomial half, which gets
//     0.3 of the budget (srs.hip; 13 GB): 56 GB per SRS.  A library that is one tenant of the device among others (a second key, another process, a
//     k = 16 key next to a k = 13 one) must not take two thirds of it by default.
//   * ZKFHE_TABLE_GB=<GB>: an explicit budget -- the SERVICE profile of a prover that owns the GPU is 160 (bench.py sets it): 15-bit
//     digits at n = 2^13, 17 windows x 16 384 multiples x 64 B per base point, 146 GB + 43 GB (13 bits) = 189 GB per SRS; the wide
//     commitment calls then need 21.4 M additions per proof against 24.9 M with 13 bits (profiles/r5_table_budget.md: rate against
//     resident GB and build time).  A table may hold more than 2^31 entries: k_msm_table addresses it relative to the chunk it sums.
// Longer bases get narrower digits from the same budget (2^16: 11 bits, 2^19: 8 bits for the Lagrange half alone); their wide calls
// take the bucket pipeline and the table serves the calls of a few columns.  ZKFHE_TABLE_BITS forces a width (0 = no table).
// Whatever the budget says, zk_basis_create_scaled narrows the table until it fits beside a reserve for proving keys and workspaces
// (table_reserve_bytes) and reports what it built (zkfhe_basis_table_bits / _bytes; one line on stderr with ZKFHE_VERBOSE).
int table_bits(size_t n, double budget_scale) {
  // read per basis (creation is rare): tests switch widths inside one process
  const char *e = getenv("ZKFHE_TABLE_BITS");
  const int forced = e ? atoi(e) : -1;
  const char *g = getenv("ZKFHE_TABLE_GB");
  double gb = g ? atof(g) : 48.0;
  if (!g) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (double)free_b / 4.0 < gb * 1073741824.0) gb = (double)free_b / 4.0 / 1073741824.0;
  }
  const double budget = gb * 1073741824.0 * budget_scale;
  auto fits = [&](int c) {
    const double entries = (double)n * (double)((255 + c - 1) / c) * (double)(1u << (c - 1));
    return c >= 8 && c <= 15 && entries * sizeof(G1Affine) <= budget;
  };
  if (forced >= 0) return forced > 0 && fits(forced) ? forced : 0;
  for (int c = 15; c >= 8; --c)
    if (fits(c)) return c;
  return 0;
}

// What stays free on the device after a table is allocated: room for proving keys (k = 13: 0.34 GB, k = 19: 11 GB) and prover
// workspaces (0.5 GB per stream at k = 13, 16 - 20 streams; several GB each at k = 19) -- an eighth of the device, at least 24 GB
// (ZKFHE_TABLE_RESERVE_GB overrides).
size_t table_reserve_bytes(size_t total_b) {
  if (const char *r = getenv("ZKFHE_TABLE_RESERVE_GB")) return (size_t)(atof(r) * 1073741824.0);
  const size_t eighth = total_b / 8, floor_b = (size_t)24 << 30;
  return eighth > floor_b ? eighth : floor_b;
}

BiasArg table_bias(int c, int W) {
  BiasArg B;
  memset(&B, 0, sizeof(B));
  for (int w = 0; w < W; ++w) {
    const int bit = c * w + c - 1;
    B.l[bit >> 5] |= 1u << (bit & 31);
  }
  return B;
}

int msm_table(zkfhe_ctx *ctx, const zkfhe_basis *basis, const Fr *scalars, size_t col_stride, size_t n_cols, void *out, bool xyzz) {
  const size_t n = basis->n;
  const int c = basis->mc, W = basis->mw;
  // n_part | col_next | adds
  const size_t words = 2 * MSM_MAX_COLS + 4;
  if (!ctx->tickets) {
    ZK_HIP(ctx, hipMalloc((void **)&ctx->tickets, words * sizeof(unsigned)));
    ZK_HIP(ctx, hipMemsetAsync(ctx->tickets, 0, words * sizeof(unsigned), ctx->stream));
  }
  unsigned *n_part = ctx->tickets, *col_next = ctx->tickets + MSM_MAX_COLS;
  unsigned long long *adds = (unsigned long long *)(ctx->tickets + 2 * MSM_MAX_COLS);
  const size_t grid_max = (size_t)ctx->num_cu * 3;
  // chunk: 256 points when the call fills the chip that way, else 128 (a lone column of 2^13: 64 workgroups, 11 additions
  // per thread and a 9-addition butterfly each -- a smaller chunk shortens the chain and multiplies the butterflies)
  size_t P = 256;
  while (P > 128 && n_cols * ((n + P - 1) / P) < grid_max) P >>= 1;
  const size_t cpc = (n + P - 1) / P, n_items = n_cols * cpc;
  const size_t grid = n_items < grid_max ? n_items : grid_max;
  const size_t max_part = cpc < grid ? cpc : grid;   // visits to one column
  const bool tree = n_cols <= 16;
  const unsigned L = tree ? 1 : 256;
  void *p0, *p1;
  int rc = zk_scratch(ctx, 0, n_cols * max_part * L * sizeof(G1X), &p0);
  if (rc) return rc;
  rc = zk_scratch(ctx, 1, grid * P * (size_t)W * 4, &p1);
  if (rc) return rc;
  const bool big = n_cols * n > ((size_t)1 << 16);
  const int slot = big ? 0 : 2;
  if (ctx->prof_on) ZK_HIP(ctx, hipMemsetAsync(adds, 0, 8, ctx->stream));
  zk_prof_begin(ctx);
  if (tree)
    k_msm_table<true><<<(unsigned)grid, 256, 0, ctx->stream>>>(scalars, col_stride, n, basis->mult, c, W, table_bias(c, W), (unsigned)P, (unsigned)cpc, (unsigned)n_cols,
                                                             (unsigned)max_part, (G1X *)p0, n_part, col_next, (unsigned *)p1, ctx->prof_on ? adds : nullptr);
  else
    k_msm_table<false><<<(unsigned)grid, 256, 0, ctx->stream>>>(scalars, col_stride, n, basis->mult, c, W, table_bias(c, W), (unsigned)P, (unsigned)cpc, (unsigned)n_cols,
                                                              (unsigned)max_part, (G1X *)p0, n_part, col_next, (unsigned *)p1, ctx->prof_on ? adds : nullptr);
  ZK_LAUNCH_CHECK(ctx);
  zk_prof_end(ctx, slot, 96.0 * (double)n * (double)n_cols);
  {
    static const bool dbg = getenv("ZKFHE_DEBUG_NPART") != nullptr;   // visits per column (the fold's work), for tools/exp
    if (dbg) {
      std::vector<unsigned> np(n_cols);
      ZK_HIP(ctx, hipMemcpy(np.data(), n_part, n_cols * sizeof(unsigned), hipMemcpyDeviceToHost));
      unsigned long long sum = 0;
      unsigned mx = 0;
      for (unsigned v : np) {
        sum += v;
        mx = v > mx ? v : mx;
      }
      fprintf(stderr, "[msm_table] cols %zu  chunks/col %zu  grid %zu  visits/col avg %.1f max %u\n", n_cols, cpc, grid, (double)sum / n_cols, mx);
    }
  }
  if (xyzz) k_msm_table_fold<true><<<(unsigned)n_cols, 512, 0, ctx->stream>>>((const G1X *)p0, (unsigned)max_part, L, n_part, col_next, out);
  else k_msm_table_fold<false><<<(unsigned)n_cols, 512, 0, ctx->stream>>>((const G1X *)p0, (unsigned)max_part, L, n_part, col_next, out);
  ZK_LAUNCH_CHECK(ctx);
  if (ctx->prof_on) {
    unsigned long long h = 0;
    ZK_HIP(ctx, hipMemcpy(&h, adds, 8, hipMemcpyDeviceToHost));
    ctx->prof_ops[slot] += (double)h;
  }
  return ZKFHE_OK;
}

// Smallest bucket count that takes the two-level sort.  K = 8192 (n = 2^15 .. 2^17, 14-bit windows) stays with the one-pass sort:
// 32 KB of counters, four workgroups per CU, eight entries per counter and workgroup -- measured equal (k = 16: 3.3 ms of sorting per
// proof either way); at K = 32768 the two-level sort halves it (k = 19: 19 -> 13 ms).  ZKFHE_SORT=2:<bits> lowers it to 2048 (tests).
static unsigned two_level_min_k() {
  const char *e = getenv("ZKFHE_SORT");
  return e && e[0] == '2' ? 2048u : 16384u;
}

int default_window_bits(size_t n) {
  if (n <= 64) return 5;
  if (n <= 1024) return 8;
  if (n <= 4096) return 11;
  if (n <= 16384) return 13;
  if (n <= 131072) return 14;
  return 16;
}

}  // namespace

extern "C" {

int zkfhe_basis_create(zkfhe_ctx *ctx, const zkfhe_g1_affine *bases_host, size_t n, int window_bits, zkfhe_basis **out) {
  return zk_basis_create_scaled(ctx, bases_host, n, window_bits, 1.0, out);
}

}  // extern "C"

// table_budget_scale: share of the per-basis table budget this basis may use (srs.hip: the monomial half of an SRS gets half)
int zk_basis_create_scaled(zkfhe_ctx *ctx, const zkfhe_g1_affine *bases_host, size_t n, int window_bits, double table_budget_scale, zkfhe_basis **out) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, out != nullptr && bases_host != nullptr && n > 0);
  ZK_ARG(ctx, window_bits == 0 || (window_bits >= 2 && window_bits <= 16));
  const int c = window_bits ? window_bits : default_window_bits(n);
  const int windows = (254 + c - 1) / c;
  ZK_ARG(ctx, (size_t)windows * n < ((size_t)1 << 31));
  zkfhe_basis *b = new zkfhe_basis();
  b->n = n;
  b->c = c;
  b->windows = windows;
  hipError_t e = hipMalloc((void **)&b->table, (size_t)windows * n * sizeof(G1Affine));
  if (e != hipSuccess) {
    delete b;
    return zk_fail(ctx, e == hipErrorOutOfMemory ? ZKFHE_ENOMEM : ZKFHE_EHIP, "hipMalloc(basis table)", e, __FILE__, __LINE__);
  }
  void *tmp;
  int rc = zk_scratch(ctx, 0, n * sizeof(G1Affine), &tmp);
  if (rc) return rc;
  ZK_HIP(ctx, hipMemcpyAsync(tmp, bases_host, n * sizeof(G1Affine), hipMemcpyHostToDevice, ctx->stream));
  k_basis_table<<<zk_blocks(n, 256), 256, 0, ctx->stream>>>((const G1Affine *)tmp, n, c, windows, b->table);
  ZK_LAUNCH_CHECK(ctx);
  int mc = window_bits == 0 && n >= 256 ? table_bits(n, table_budget_scale) : 0;
  // a digit-multiple table (k_msm_table): every call against this basis becomes a plain sum of table points.  It is an
  // accelerator, not a requirement: on a device that does not have the room (other tenants, many SRS alive) the width drops
  // until it fits, and without any table the calls take the bucket pipeline
  const int mc_budget = mc;
  bool narrowed = false;
  for (; mc >= 8; --mc) {
    const int mw = (255 + mc - 1) / mc;
    const size_t bytes = (n * (size_t)mw << (mc - 1)) * sizeof(G1Affine);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + table_reserve_bytes(total_b)) {
      narrowed = true;   // the device does not have the room right now (other keys, other tenants): the next narrower width
      continue;
    }
    e = hipMalloc((void **)&b->mult, bytes);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      b->mult = nullptr;
      narrowed = true;
      continue;
    }
    b->mc = mc;
    b->mw = mw;
    b->mult_bytes = bytes;
    k_basis_multiples<<<zk_blocks(n * (size_t)mw, 64), 64, 0, ctx->stream>>>((const G1Affine *)tmp, n, mc, mw, b->mult);
    ZK_LAUNCH_CHECK(ctx);
    break;
  }
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  b->narrowed = narrowed && b->mc < mc_budget;
  if (getenv("ZKFHE_VERBOSE"))
    fprintf(stderr, "[zkfhe] basis n = %zu: digit-multiple table %d bits, %.1f GB resident%s\n", n, b->mc, (double)b->mult_bytes / 1073741824.0,
            b->narrowed ? " (narrower than the budget allows: the device did not have the room)" : (b->mc ? "" : " (none: bucket pipeline)"));
  *out = b;
  return ZKFHE_OK;
}

extern "C" {

int zkfhe_basis_destroy(zkfhe_ctx *ctx, zkfhe_basis *basis) {
  ZK_ENTER(ctx);
  if (!basis) return ZKFHE_OK;
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  hipFree(basis->table);
  if (basis->mult) hipFree(basis->mult);
  delete basis;
  return ZKFHE_OK;
}

size_t zkfhe_basis_len(const zkfhe_basis *basis) { return basis ? basis->n : 0; }

int zkfhe_msm_batch(zkfhe_ctx *ctx, const zkfhe_basis *basis, const zkfhe_fr *scalars_dev, size_t n_cols, zkfhe_g1_affine *out_dev) {
  return zk_msm_batch_strided(ctx, basis, scalars_dev, basis ? basis->n : 0, n_cols, out_dev);
}

// The same sums left in the accumulator form (XYZZ, 128 B each, x = X / ZZ, y = Y / ZZZ, identity ZZ = 0): the last kernel of the
// call skips its field inversion -- one lane, 40 us, at the end of every call -- and the caller normalises the points of a whole
// Fiat-Shamir round with ONE inversion (zkfhe_g1_xyzz_to_affine, host).  halo2 does the same one level up: `commit_lagrange`
// returns projective points and create_proof batch-normalises a round's commitments.
int zkfhe_msm_batch_xyzz(zkfhe_ctx *ctx, const zkfhe_basis *basis, const zkfhe_fr *scalars_dev, size_t n_cols, zkfhe_g1_xyzz *out_dev) {
  return zk_msm_batch_strided_form(ctx, basis, scalars_dev, basis ? basis->n : 0, n_cols, out_dev, true);
}

}  // extern "C"

// Column c of the call holds its basis->n scalars at scalars_dev + c * col_stride: col_stride > n selects a row range of longer
// columns (the point-range shard of one rank, comm.hip) without copying it out.
int zk_msm_batch_strided(zkfhe_ctx *ctx, const zkfhe_basis *basis, const zkfhe_fr *scalars_dev, size_t col_stride, size_t n_cols, zkfhe_g1_affine *out_dev) {
  return zk_msm_batch_strided_form(ctx, basis, scalars_dev, col_stride, n_cols, out_dev, false);
}

// out_dev: n_cols affine points (64 B each) or, with xyzz, n_cols accumulator-form sums (zkfhe_g1_xyzz, 128 B each)
int zk_msm_batch_strided_form(zkfhe_ctx *ctx, const zkfhe_basis *basis, const zkfhe_fr *scalars_dev, size_t col_stride, size_t n_cols, void *out_dev, bool xyzz) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, basis != nullptr);
  if (!n_cols) return ZKFHE_OK;
  ZK_ARG(ctx, scalars_dev != nullptr && out_dev != nullptr && col_stride >= basis->n);
  const size_t n = basis->n;
  ZK_ARG(ctx, n_cols <= MSM_MAX_COLS);
  // the table path when its windows are at most two more than the bucket pipeline's (whose sort and bucket reduction cost
  // about that much), and always for calls of a few columns, where the pipeline's thirteen dependent launches are the cost
  static const char *wide_env = getenv("ZKFHE_TABLE_WIDE");   // 1 / 0: force the choice for calls of many columns
  const bool wide = wide_env ? wide_env[0] == '1' : basis->mw <= basis->windows + 2;
  if (basis->mult && (wide || n_cols <= 8))
    return msm_table(ctx, basis, (const Fr *)scalars_dev, col_stride, n_cols, out_dev, xyzz);
  const int c = basis->c, W = basis->windows;
  const unsigned K = 1u << (c - 1), K1 = K + 1;
  const size_t col_entries = n * (size_t)W;
  const unsigned TASK_E = task_len(col_entries, K);
  const size_t max_tasks = (n_cols * col_entries) / TASK_E + (size_t)K * n_cols;      // upper bound on accumulation tasks
  const size_t heavy_cap = (n_cols * col_entries) / ((size_t)TASK_E * MERGE_LIGHT) + 1;  // buckets with > MERGE_LIGHT partials
  // scratch 1: hist | off | cursor | bucket_posA | bucket_posB | col_hist | col_base | n_tasks | heavy_count | heavy_list | tasks
  // scratch 2: entries     scratch 0: buckets | partials
  const size_t words = 3 * n_cols * K1 + 2 * (size_t)K * n_cols + 2 * n_cols * TASK_BINS + 16 + 2 * heavy_cap + 2 * max_tasks;
  void *p1, *p2, *p0;
  int rc = zk_scratch(ctx, 1, words * sizeof(unsigned), &p1);
  if (rc) return rc;
  rc = zk_scratch(ctx, 2, n_cols * col_entries * sizeof(unsigned) + 64, &p2);
  if (rc) return rc;
  // two-level sort (see k_msm_chist): L fine bits of the bucket id travel in the entry word above the log2(W n) index bits
  int idx_bits = 1;
  while (((size_t)1 << idx_bits) < col_entries) ++idx_bits;
  int L = 31 - idx_bits;
  if (L > FINE_BITS_MAX) L = FINE_BITS_MAX;
  const char *sort_env = getenv("ZKFHE_SORT");   // "1": the one-pass sort for every basis; "2:<bits>": fewer fine bits (tests: the geometry of n = 2^19 on a short basis)
  if (sort_env && sort_env[0] == '2' && sort_env[1] == ':' && atoi(sort_env + 2) >= 6 && atoi(sort_env + 2) < L) L = atoi(sort_env + 2);
  const bool two_level = K >= two_level_min_k() && L >= 6 && !(sort_env && sort_env[0] == '1');
  const unsigned NB = two_level ? K >> L : 0;
  // the staging array of the two-level sort lives where the buckets and partials will be (dead before they are written)
  size_t s0 = (n_cols * (size_t)K + max_tasks) * sizeof(G1X);
  const size_t stage_stride = (col_entries + 3) & ~(size_t)3;   // k_msm_fine reads the staged entries in 16-byte loads
  if (two_level && s0 < n_cols * stage_stride * sizeof(unsigned)) s0 = n_cols * stage_stride * sizeof(unsigned);
  rc = zk_scratch(ctx, 0, s0, &p0);
  if (rc) return rc;
  unsigned *hist = (unsigned *)p1;
  unsigned *off = hist + n_cols * K1;
  unsigned *cursor = off + n_cols * K1;
  unsigned *bucket_posA = cursor + n_cols * K1;
  unsigned *bucket_posB = bucket_posA + (size_t)K * n_cols;
  unsigned *col_hist = bucket_posB + (size_t)K * n_cols;
  unsigned *col_base = col_hist + n_cols * TASK_BINS;
  unsigned *n_tasks_dev = col_base + n_cols * TASK_BINS;
  unsigned *heavy_count = n_tasks_dev + 4;
  unsigned *medium_list = heavy_count + 4;
  unsigned *large_list = medium_list + heavy_cap;
  uint2 *tasks = (uint2 *)(((uintptr_t)(large_list + heavy_cap) + 7) & ~(uintptr_t)7);
  unsigned *entries = (unsigned *)p2;
  G1X *buckets = (G1X *)p0;
  G1X *partials = buckets + n_cols * (size_t)K;
  if (two_level) {
    unsigned *chist = hist, *coff = cursor, *ccursor = cursor + n_cols * (NB + 1);   // inside the regions of the one-pass sort
    unsigned *stage = (unsigned *)p0;
    k_msm_clear<<<zk_blocks(n_cols * NB, 256), 256, 0, ctx->stream>>>(chist, n_cols * NB, heavy_count);
    ZK_LAUNCH_CHECK(ctx);
    const unsigned ch_chunks = (unsigned)((n + CH_SCALARS - 1) / CH_SCALARS), cs_chunks = (unsigned)((n + CS_SCALARS - 1) / CS_SCALARS);
    k_msm_chist<<<(unsigned)(n_cols * ch_chunks), CS_THREADS, NB * sizeof(unsigned), ctx->stream>>>((const Fr *)scalars_dev, col_stride, n, ch_chunks, c, W, L, NB, chist);
    ZK_LAUNCH_CHECK(ctx);
    k_msm_cscan<<<zk_blocks(n_cols, 64), 64, 0, ctx->stream>>>(chist, NB, n_cols, coff, ccursor);
    ZK_LAUNCH_CHECK(ctx);
    const size_t cs_lds = ((size_t)3 * NB + (size_t)CS_SCALARS * W) * sizeof(unsigned);
    if (cs_lds > 48 * 1024) ZK_CK(zk_func_max_lds(ctx, (const void *)k_msm_cscatter, 64 * 1024));
    k_msm_cscatter<<<(unsigned)(n_cols * cs_chunks), CS_THREADS, cs_lds, ctx->stream>>>((const Fr *)scalars_dev, col_stride, n, cs_chunks, c, W, L, NB, ccursor, stage,
                                                                                         stage_stride);
    ZK_LAUNCH_CHECK(ctx);
    ZK_CK(zk_func_max_lds(ctx, (const void *)k_msm_fine, (int)(FINE_TILE * sizeof(unsigned))));
    k_msm_fine<<<(unsigned)(n_cols * NB), FINE_THREADS, FINE_TILE * sizeof(unsigned), ctx->stream>>>(coff, NB, L, stage, stage_stride, col_entries, K1, off, entries);
    ZK_LAUNCH_CHECK(ctx);
  } else {
    {
      unsigned gc = zk_blocks(n_cols * K1, 256);
      if (gc > (unsigned)ctx->num_cu * 8) gc = (unsigned)ctx->num_cu * 8;
      k_msm_clear<<<gc, 256, 0, ctx->stream>>>(hist, n_cols * K1, heavy_count);
      ZK_LAUNCH_CHECK(ctx);
    }
    const unsigned chunks_per_col = (unsigned)((n + SORT_CHUNK - 1) / SORT_CHUNK);
    const unsigned grid = (unsigned)(n_cols * chunks_per_col);
    const size_t sort_lds = (size_t)K1 * sizeof(unsigned);
    if (sort_lds > 48 * 1024) {
      ZK_CK(zk_func_max_lds(ctx, (const void *)k_msm_hist, 160 * 1024));
      ZK_CK(zk_func_max_lds(ctx, (const void *)k_msm_scatter, 160 * 1024));
    }
    k_msm_hist<<<grid, SORT_THREADS, sort_lds, ctx->stream>>>((const Fr *)scalars_dev, col_stride, n, chunks_per_col, c, W, hist, K1);
    ZK_LAUNCH_CHECK(ctx);
    k_msm_scan<<<(unsigned)n_cols, 256, 0, ctx->stream>>>(hist, off, cursor, K1);
    ZK_LAUNCH_CHECK(ctx);
    k_msm_scatter<<<grid, SORT_THREADS, sort_lds, ctx->stream>>>((const Fr *)scalars_dev, col_stride, n, chunks_per_col, c, W, cursor, K1, entries, col_entries);
    ZK_LAUNCH_CHECK(ctx);
  }
  k_msm_task_count<<<(unsigned)n_cols, 256, 0, ctx->stream>>>(off, K, TASK_E, col_hist);
  ZK_LAUNCH_CHECK(ctx);
  k_msm_task_colscan<<<1, 128, 0, ctx->stream>>>(col_hist, (unsigned)n_cols, col_base, n_tasks_dev);
  ZK_LAUNCH_CHECK(ctx);
  k_msm_task_fill<<<(unsigned)n_cols, 256, 0, ctx->stream>>>(off, K, TASK_E, col_base, bucket_posA, bucket_posB, tasks, heavy_count, medium_list, large_list,
                                                             (unsigned)heavy_cap);
  ZK_LAUNCH_CHECK(ctx);
  unsigned gridt = zk_blocks(max_tasks, 256);
  const unsigned capt = (unsigned)ctx->num_cu * 32;
  static int acc_blocks = -1;
  if (acc_blocks < 0) {
    const char *s = getenv("ZKFHE_ACC_BLOCKS");
    acc_blocks = s ? atoi(s) : 0;
  }
  const unsigned capa = acc_blocks > 0 ? (unsigned)ctx->num_cu * acc_blocks : capt;
  if (gridt > capa) gridt = capa;
  zk_prof_begin(ctx);
  k_msm_accumulate<<<gridt, 256, 0, ctx->stream>>>(tasks, n_tasks_dev, off, entries, col_entries, basis->table, K, TASK_E, partials);
  ZK_LAUNCH_CHECK(ctx);
  zk_prof_end(ctx, 0, 96.0 * (double)n * (double)n_cols);
  if (ctx->prof_on) {  // mixed additions of this launch = sorted entries = off[col][K] summed over the columns
    std::vector<unsigned> tot(n_cols);
    ZK_HIP(ctx, hipMemcpy2D(tot.data(), sizeof(unsigned), off + K, (size_t)K1 * sizeof(unsigned), sizeof(unsigned), n_cols, hipMemcpyDeviceToHost));
    for (unsigned t : tot) ctx->prof_ops[0] += (double)t;
  }
  {
    // light buckets: one thread each; medium: a wave per eight buckets; large: a wave per bucket -- three block ranges of one launch
    const size_t nb = (size_t)K * n_cols;
    unsigned gridb = zk_blocks(nb, 256);
    if (gridb > capt) gridb = capt;
    unsigned gridm = (unsigned)((heavy_cap / 8 + 3) / 4);
    if (gridm > 512) gridm = 512;
    unsigned gridl = (unsigned)((heavy_cap + 3) / 4);
    if (gridl > 512) gridl = 512;
    k_msm_merge<<<gridl + gridm + gridb, 256, 0, ctx->stream>>>(off, bucket_posA, bucket_posB, partials, K, TASK_E, n_cols, buckets, heavy_count, medium_list,
                                                                large_list, (unsigned)heavy_cap, gridl, gridm);
    ZK_LAUNCH_CHECK(ctx);
  }
  if (K >= 64 && K <= 32768) {
    const unsigned A = K >> 6, per_col = A + 64;
    G1X *marg = partials;  // the accumulation partials are dead once the buckets are merged
    // row sums (64 buckets each) and column sums (A buckets each).  Few columns: whole-wave butterflies (shortest chain);
    // many: 8 lanes per sum (1.4 lane-additions per bucket) -- except column sums over more than 64 rows, whose serial
    // part would be A / 8 additions deep
    const unsigned Lr = n_cols <= 16 ? 64 : 8;
    const unsigned Lc = (n_cols <= 16 || A > 64) ? 64 : 8;
    if (Lr == Lc) {   // rows and columns side by side in one launch
      const size_t waves = (n_cols * (A + 64) + 64 / Lr - 1) / (64 / Lr);
      k_msm_marginals<<<zk_blocks(waves * 64, 256), 256, 0, ctx->stream>>>(buckets, K, n_cols, Lr, 0, A + 64, marg);
      ZK_LAUNCH_CHECK(ctx);
    } else {
      size_t waves = (n_cols * A + 64 / Lr - 1) / (64 / Lr);
      k_msm_marginals<<<zk_blocks(waves * 64, 256), 256, 0, ctx->stream>>>(buckets, K, n_cols, Lr, 0, A, marg);
      ZK_LAUNCH_CHECK(ctx);
      waves = (n_cols * 64 + 64 / Lc - 1) / (64 / Lc);
      k_msm_marginals<<<zk_blocks(waves * 64, 256), 256, 0, ctx->stream>>>(buckets, K, n_cols, Lc, A, 64, marg);
      ZK_LAUNCH_CHECK(ctx);
    }
    if (xyzz) k_msm_weighted<true><<<(unsigned)n_cols, 64 * ((A + 63) / 64 + 1), 0, ctx->stream>>>(marg, K, out_dev);
    else k_msm_weighted<false><<<(unsigned)n_cols, 64 * ((A + 63) / 64 + 1), 0, ctx->stream>>>(marg, K, out_dev);
    ZK_LAUNCH_CHECK(ctx);
    (void)per_col;
    return ZKFHE_OK;
  }
  if (K < 64) {
    if (xyzz) k_msm_small<true><<<(unsigned)n_cols, 64, 0, ctx->stream>>>(buckets, K, out_dev);
    else k_msm_small<false><<<(unsigned)n_cols, 64, 0, ctx->stream>>>(buckets, K, out_dev);
    ZK_LAUNCH_CHECK(ctx);
    return ZKFHE_OK;
  }
  return zk_fail_msg(ctx, ZKFHE_EINVAL, "MSM basis with more than 32768 buckets (window_bits > 16) is not supported");
}

extern "C" {

static int msm_sparse_form(zkfhe_ctx *ctx, const zkfhe_basis *basis, const zkfhe_sparse_term *terms_dev, size_t n_terms, size_t n_slots, void *out_dev, bool xyzz) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, basis != nullptr && basis->mult != nullptr && terms_dev != nullptr && out_dev != nullptr && n_slots > 0 && n_slots <= 65535 && n_terms <= 4096);
  if (xyzz)
    k_msm_sparse<true><<<(unsigned)n_slots, 64, 0, ctx->stream>>>(terms_dev, (unsigned)n_terms, basis->mult, basis->mc, basis->mw, table_bias(basis->mc, basis->mw), out_dev);
  else
    k_msm_sparse<false><<<(unsigned)n_slots, 64, 0, ctx->stream>>>(terms_dev, (unsigned)n_terms, basis->mult, basis->mc, basis->mw, table_bias(basis->mc, basis->mw), out_dev);
  ZK_LAUNCH_CHECK(ctx);
  return ZKFHE_OK;
}

int zkfhe_msm_sparse(zkfhe_ctx *ctx, const zkfhe_basis *basis, const zkfhe_sparse_term *terms_dev, size_t n_terms, size_t n_slots, zkfhe_g1_affine *out_dev) {
  return msm_sparse_form(ctx, basis, terms_dev, n_terms, n_slots, out_dev, false);
}
int zkfhe_msm_sparse_xyzz(zkfhe_ctx *ctx, const zkfhe_basis *basis, const zkfhe_sparse_term *terms_dev, size_t n_terms, size_t n_slots, zkfhe_g1_xyzz *out_dev) {
  return msm_sparse_form(ctx, basis, terms_dev, n_terms, n_slots, out_dev, true);
}

// Accumulator-form sums -> affine points, on the host: Montgomery's trick over the ZZZ of the whole array (3 products per point and
// ONE Bernstein-Yang inversion), then x = X (ZZ / ZZZ)^2, y = Y / ZZZ; ZZ = 0 -> the identity (0, 0).  In and out are raw Montgomery
// coordinates (what the kernels store and zkfhe_msm_batch returns).  in == out element-wise aliasing is not supported.
int zkfhe_g1_xyzz_to_affine(const zkfhe_g1_xyzz *in, size_t n, zkfhe_g1_affine *out) {
  if ((!in || !out) && n) return ZKFHE_EINVAL;
  g1x_normalize_batch((const G1X *)in, n, (G1Affine *)out);
  return ZKFHE_OK;
}

int zkfhe_basis_has_multiples(const zkfhe_basis *basis) { return basis && basis->mult ? 1 : 0; }

int zkfhe_basis_table_bits(const zkfhe_basis *basis, int *wide_calls) {
  const char *wide_env = getenv("ZKFHE_TABLE_WIDE");
  if (wide_calls) *wide_calls = basis && basis->mult && (wide_env ? wide_env[0] == '1' : basis->mw <= basis->windows + 2) ? 1 : 0;
  return basis && basis->mult ? basis->mc : 0;
}

// bytes of the basis' digit-multiple table resident in HBM; *narrowed (optional) = 1 when the table is narrower than the budget
// would have allowed because the device did not have the room when the basis was made
size_t zkfhe_basis_table_bytes(const zkfhe_basis *basis, int *narrowed) {
  if (narrowed) *narrowed = basis && basis->narrowed ? 1 : 0;
  return basis ? basis->mult_bytes : 0;
}

int zkfhe_g1_add(zkfhe_ctx *ctx, const zkfhe_g1_affine *a, const zkfhe_g1_affine *b, zkfhe_g1_affine *out, size_t n) {
  ZK_ENTER(ctx);
  if (!n) return ZKFHE_OK;
  k_g1_add<<<zk_blocks(n, 256), 256, 0, ctx->stream>>>((const G1Affine *)a, (const G1Affine *)b, (G1Affine *)out, n);
  ZK_LAUNCH_CHECK(ctx);
  return ZKFHE_OK;
}

int zkfhe_g1_mul(zkfhe_ctx *ctx, const zkfhe_g1_affine *p, const zkfhe_fr *k, zkfhe_g1_affine *out, size_t n) {
  ZK_ENTER(ctx);
  if (!n) return ZKFHE_OK;
  k_g1_mul<<<zk_blocks(n, 256), 256, 0, ctx->stream>>>((const G1Affine *)p, (const Fr *)k, (G1Affine *)out, n);
  ZK_LAUNCH_CHECK(ctx);
  return ZKFHE_OK;
}

}  // extern "C"
