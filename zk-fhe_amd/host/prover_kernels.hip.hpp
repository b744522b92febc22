// Device kernels of the prover steps P1, P4-P10: everything between the column NTT/MSM
// primitives of csrc/ and the host-side transcript.  All polynomials stay resident in HBM; the
// extended (quotient) domain is kept COSET-MAJOR ([k1][k2], see zkfhe_coset_ntt_batch), so a rotation by
// omega is an index shift inside a row and X^n - 1 is constant per row.
#pragma once
#include "../csrc/ctx.hpp"
#include "../csrc/fr29.hip.hpp"

namespace zkp {

using zk::Fr;

// ------------------------------------------------------------------------------------------- small utilities
// out[i] = start * base^i
static __global__ void __launch_bounds__(256) k_powers(Fr start, Fr base, Fr *__restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    Fr r = start, b = base;
    for (size_t e = i; e; e >>= 1) {
      if (e & 1) r = r * b;
      b = zk::fp_sqr<zk::FrP>(b);
    }
    out[i] = r;
  }
}

// SRS: den[i] = n (s - w^i)   (inverted by the caller);  then li[i] = w^i (s^n - 1) * inv[i]
// ---- the blinding stream on the device ------------------------------------------------------------------------------
// draw(i) = Blake2b-512(person "zkfhe-rng", seed[32] || i as u64 LE) reduced mod r (transcript.hpp `Rng`, oracle/halo2_ref.py
// `Rng`): counter based, so every blinding row / random coefficient of a proof is a pure function of (seed, index) and is
// written where it is needed -- no host hashing (it was 50 k hashes = 12 ms of host CPU per k = 13 proof) and no upload.
__device__ __forceinline__ unsigned long long rotr64(unsigned long long x, int n) { return (x >> n) | (x << (64 - n)); }
__device__ __forceinline__ Fr rng_draw(const unsigned long long seed[4], unsigned long long ctr) {
  typedef unsigned long long u64;
  const u64 IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                     0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
  const unsigned char S[12][16] = {
      {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
      {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
      {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
      {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
      {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
      {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
  u64 h[8], m[16], v[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] = IV[i];
  h[0] ^= 0x01010040ULL;                 // digest length 64, no key, fanout = depth = 1
  h[6] ^= 0x6e722d6568666b7aULL;         // personalisation "zkfhe-rn"
  h[7] ^= 0x0000000000000067ULL;         //                 "g" + zero padding
#pragma unroll
  for (int i = 0; i < 16; ++i) m[i] = 0;
  m[0] = seed[0], m[1] = seed[1], m[2] = seed[2], m[3] = seed[3], m[4] = ctr;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i] = h[i];
    v[i + 8] = IV[i];
  }
  v[12] ^= 40;        // 40 message bytes
  v[14] = ~v[14];     // last block
#define ZK_B2G(a, b, c, d, x, y)          \
  v[a] = v[a] + v[b] + (x);               \
  v[d] = rotr64(v[d] ^ v[a], 32);         \
  v[c] = v[c] + v[d];                     \
  v[b] = rotr64(v[b] ^ v[c], 24);         \
  v[a] = v[a] + v[b] + (y);               \
  v[d] = rotr64(v[d] ^ v[a], 16);         \
  v[c] = v[c] + v[d];                     \
  v[b] = rotr64(v[b] ^ v[c], 63);
#pragma unroll
  for (int r = 0; r < 12; ++r) {
    ZK_B2G(0, 4, 8, 12, m[S[r][0]], m[S[r][1]])
    ZK_B2G(1, 5, 9, 13, m[S[r][2]], m[S[r][3]])
    ZK_B2G(2, 6, 10, 14, m[S[r][4]], m[S[r][5]])
    ZK_B2G(3, 7, 11, 15, m[S[r][6]], m[S[r][7]])
    ZK_B2G(0, 5, 10, 15, m[S[r][8]], m[S[r][9]])
    ZK_B2G(1, 6, 11, 12, m[S[r][10]], m[S[r][11]])
    ZK_B2G(2, 7, 8, 13, m[S[r][12]], m[S[r][13]])
    ZK_B2G(3, 4, 9, 14, m[S[r][14]], m[S[r][15]])
  }
#undef ZK_B2G
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
  // from_bytes_wide: lo + hi * 2^256 mod r, as a Montgomery value
  Fr lo, hi;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    lo.l[2 * i] = (zk::u32)h[i];
    lo.l[2 * i + 1] = (zk::u32)(h[i] >> 32);
    hi.l[2 * i] = (zk::u32)h[4 + i];
    hi.l[2 * i + 1] = (zk::u32)(h[4 + i] >> 32);
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) {   // 2^256 < 6 r
    zk::fp_reduce_once<zk::FrP>(lo.l);
    zk::fp_reduce_once<zk::FrP>(hi.l);
  }
  // lo, hi canonical integers: (lo + hi 2^256) 2^256 = lo * R2 / R ... as Montgomery values: mont(lo) + mont(hi) * mont(2^256)
  const Fr r2 = Fr::r2();
  const Fr lom = zk::fp_mul<zk::FrP>(lo, r2);                        // lo in Montgomery form
  const Fr him = zk::fp_mul<zk::FrP>(zk::fp_mul<zk::FrP>(hi, r2), r2);   // (hi * 2^256) in Montgomery form
  return zk::fp_add<zk::FrP>(lom, him);
}
struct RngSeed {
  unsigned long long w[4];
};
// dst[c * col_stride + j] = draw(ctr0 + c * ctr_col_stride + j)   (Montgomery), j < per_col, c < n_cols
static __global__ void __launch_bounds__(256) k_rng_fill(RngSeed seed, unsigned long long ctr0, unsigned long long ctr_col_stride, Fr *__restrict__ dst,
                                                  size_t per_col, size_t col_stride, size_t n_cols) {
  const size_t total = per_col * n_cols;
  for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t c = g / per_col, j = g - c * per_col;
    dst[c * col_stride + j] = rng_draw(seed.w, ctr0 + c * ctr_col_stride + j);
  }
}

static __global__ void __launch_bounds__(256) k_srs_den(const Fr *__restrict__ wpow, Fr s, Fr nn, Fr *__restrict__ out, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = (s - wpow[i]) * nn;
}
static __global__ void __launch_bounds__(256) k_srs_li(const Fr *__restrict__ wpow, Fr snm1, Fr *__restrict__ inv_inout, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) inv_inout[i] = wpow[i] * snm1 * inv_inout[i];
}
static __global__ void __launch_bounds__(256) k_fill_point(zk::G1Affine p, zk::G1Affine *__restrict__ out, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = p;
}

// sigma_l[cell] = delta^(col of target) * omega^(row of target)
static __global__ void __launch_bounds__(256) k_sigma_values(const uint32_t *__restrict__ target, const Fr *__restrict__ dpow,
                                                      const Fr *__restrict__ wpow, Fr *__restrict__ out, size_t cells, int log_n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= cells) return;
  const uint32_t t = target[i];
  out[i] = dpow[t >> log_n] * wpow[t & ((1u << log_n) - 1)];
}

// ------------------------------------------------------------------------------------------- grand products
struct PermArgs {
  const Fr *adv;      // [n_advice][n] Lagrange
  const Fr *constcol; // [n]
  const Fr *inst;     // [n]
  const Fr *sigma;    // [n_perm][n]
  const Fr *wpow;     // omega^i
  const Fr *beta_delta;  // [n_perm] beta * delta^c
  Fr beta, gamma;
  unsigned n_advice, n_perm, chunk, n_chunks;
  size_t n;
};
__device__ __forceinline__ const Fr *perm_col(const PermArgs &a, unsigned c) {
  return c < a.n_advice ? a.adv + (size_t)c * a.n : (c == a.n_advice ? a.constcol : a.inst);
}
// num[j][i] = prod_c (v_c + beta delta^c w^i + gamma), den[j][i] = prod_c (v_c + beta sigma_c + gamma)
static __global__ void __launch_bounds__(256) k_perm_num_den(PermArgs a, Fr *__restrict__ num, Fr *__restrict__ den) {
  const size_t total = (size_t)a.n_chunks * a.n;
  for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const unsigned j = (unsigned)(g / a.n);
    const size_t i = g - (size_t)j * a.n;
    Fr nu = Fr::one(), de = Fr::one();
    const Fr w = a.wpow[i];
    for (unsigned c = j * a.chunk; c < (j + 1) * a.chunk && c < a.n_perm; ++c) {
      const Fr v = perm_col(a, c)[i];
      nu = nu * (v + a.beta_delta[c] * w + a.gamma);
      de = de * (v + a.beta * a.sigma[(size_t)c * a.n + i] + a.gamma);
    }
    num[g] = nu;
    den[g] = de;
  }
}
// lookup: num = (a + beta)(s + gamma), den = (a' + beta)(s' + gamma)
static __global__ void __launch_bounds__(256) k_lookup_num_den(const Fr *__restrict__ a_cols, const Fr *__restrict__ table, const Fr *__restrict__ la,
                                                        const Fr *__restrict__ ls, Fr beta, Fr gamma, unsigned n_lookup, size_t n,
                                                        Fr *__restrict__ num, Fr *__restrict__ den) {
  const size_t total = (size_t)n_lookup * n;
  for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t i = g % n;
    num[g] = (a_cols[g] + beta) * (table[i] + gamma);
    den[g] = (la[g] + beta) * (ls[g] + gamma);
  }
}

// Exclusive running product per column: z[0] = 1, z[i+1] = z[i] * r[i] for i < u; z has u+1 defined entries.
// One workgroup of 1024 threads per column; thread t owns rows [t*per, (t+1)*per).  `total[col]` = z[u].
static __global__ void __launch_bounds__(1024) k_prefix_product(const Fr *__restrict__ ratio, Fr *__restrict__ z, Fr *__restrict__ total, size_t n,
                                                         unsigned u) {
  __shared__ Fr sh[1024];
  const size_t col = blockIdx.x;
  const Fr *r = ratio + col * n;
  Fr *o = z + col * n;
  const unsigned per = (unsigned)((n + 1023) / 1024);
  const unsigned lo = threadIdx.x * per;
  Fr local = Fr::one();
  for (unsigned k = 0; k < per; ++k) {
    const unsigned i = lo + k;
    if (i < u) local = local * r[i];
  }
  sh[threadIdx.x] = local;
  __syncthreads();
  // inclusive Hillis-Steele scan of the 1024 partial products
  for (unsigned d = 1; d < 1024; d <<= 1) {
    Fr v = sh[threadIdx.x];
    Fr other = threadIdx.x >= d ? sh[threadIdx.x - d] : Fr::one();
    __syncthreads();
    if (threadIdx.x >= d) sh[threadIdx.x] = other * v;
    __syncthreads();
  }
  Fr acc = threadIdx.x ? sh[threadIdx.x - 1] : Fr::one();  // product of everything before this thread's rows
  for (unsigned k = 0; k < per; ++k) {
    const unsigned i = lo + k;
    if (i <= u) o[i] = acc;
    if (i < u) acc = acc * r[i];
  }
  if (threadIdx.x == 1023) total[col] = sh[1023];
}
// Long columns (k >= 17): the column is cut into `segs` segments of `seg_len` rows, one workgroup each, so that a call fills the chip
// (at k = 19 one workgroup per column meant 100 workgroups walking 512 rows per thread twice: 3.2 ms per call).  Three launches:
// k_prefix_seg_totals (product of every segment), k_prefix_seg_scan (per column: exclusive scan of the segment products, `total[col]`),
// k_prefix_seg_apply (the running product inside a segment, started from its carry).  Same values as k_prefix_product: field
// multiplication is associative and the results are canonical.
static __global__ void __launch_bounds__(1024) k_prefix_seg_totals(const Fr *__restrict__ ratio, Fr *__restrict__ seg_total, size_t n, unsigned u, unsigned seg_len) {
  __shared__ Fr sh[1024];
  const size_t col = blockIdx.y;
  const unsigned seg = blockIdx.x, base = seg * seg_len;
  const Fr *r = ratio + col * n;
  const unsigned per = (seg_len + 1023) / 1024, lo = base + threadIdx.x * per;
  Fr local = Fr::one();
  for (unsigned k = 0; k < per; ++k) {
    const unsigned i = lo + k;
    if (i < u && i < base + seg_len) local = local * r[i];
  }
  sh[threadIdx.x] = local;
  __syncthreads();
  for (unsigned d = 512; d > 0; d >>= 1) {   // product reduction (order is irrelevant: commutative)
    if (threadIdx.x < d) sh[threadIdx.x] = sh[threadIdx.x] * sh[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) seg_total[col * gridDim.x + seg] = sh[0];
}
static __global__ void __launch_bounds__(64) k_prefix_seg_scan(Fr *__restrict__ seg_total, unsigned segs, unsigned n_cols, Fr *__restrict__ total) {
  const unsigned col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= n_cols) return;
  Fr acc = Fr::one();
  for (unsigned s = 0; s < segs; ++s) {
    const Fr t = seg_total[(size_t)col * segs + s];
    seg_total[(size_t)col * segs + s] = acc;   // carry into segment s
    acc = acc * t;
  }
  total[col] = acc;
}
static __global__ void __launch_bounds__(1024) k_prefix_seg_apply(const Fr *__restrict__ ratio, const Fr *__restrict__ seg_carry, Fr *__restrict__ z, size_t n, unsigned u,
                                                            unsigned seg_len) {
  __shared__ Fr sh[1024];
  const size_t col = blockIdx.y;
  const unsigned seg = blockIdx.x, base = seg * seg_len, end = base + seg_len;
  const Fr *r = ratio + col * n;
  Fr *o = z + col * n;
  const unsigned per = (seg_len + 1023) / 1024, lo = base + threadIdx.x * per;
  Fr local = Fr::one();
  for (unsigned k = 0; k < per; ++k) {
    const unsigned i = lo + k;
    if (i < u && i < end) local = local * r[i];
  }
  sh[threadIdx.x] = local;
  __syncthreads();
  for (unsigned d = 1; d < 1024; d <<= 1) {
    Fr v = sh[threadIdx.x];
    Fr other = threadIdx.x >= d ? sh[threadIdx.x - d] : Fr::one();
    __syncthreads();
    if (threadIdx.x >= d) sh[threadIdx.x] = other * v;
    __syncthreads();
  }
  Fr acc = seg_carry[col * gridDim.x + seg];
  if (threadIdx.x) acc = acc * sh[threadIdx.x - 1];
  for (unsigned k = 0; k < per; ++k) {
    const unsigned i = lo + k;
    if (i >= end) break;
    if (i <= u) o[i] = acc;
    if (i < u) acc = acc * r[i];
  }
}
// Chunk carries of the permutation argument on the device (was: download the totals, multiply on the host, upload):
// carry[j] = prod_{i < j} total[i] in place, *closes = (the product of all of them == 1).  With check_ones every total itself
// has to be one (the lookup products) and the array is left alone.  One workgroup of 1024 threads; count <= 4096.
static __global__ void __launch_bounds__(1024) k_chunk_carry(Fr *__restrict__ total, unsigned count, int check_ones, int *__restrict__ closes) {
  __shared__ Fr sh[1024];
  __shared__ int bad;
  const unsigned t = threadIdx.x;
  if (t == 0) bad = 0;
  __syncthreads();
  const unsigned per = (count + 1023) / 1024, lo = t * per;
  if (check_ones) {
    const Fr one = Fr::one();
    for (unsigned k = 0; k < per; ++k)
      if (lo + k < count && !(total[lo + k] == one)) bad = 1;
    __syncthreads();
    if (t == 0) *closes = bad ? 0 : 1;
    return;
  }
  Fr vals[4];
  Fr local = Fr::one();
#pragma unroll
  for (unsigned k = 0; k < 4; ++k) {   // per <= 4; unrolled so that vals stays in registers
    vals[k] = k < per && lo + k < count ? total[lo + k] : Fr::one();
    local = local * vals[k];
  }
  sh[t] = local;
  __syncthreads();
  const unsigned active = (count + per - 1) / per;   // threads that hold totals: the scan stops once it spans them (66 chunks: 7 steps, not 10)
  for (unsigned d = 1; d < active; d <<= 1) {
    const Fr v = sh[t];
    const Fr other = t >= d ? sh[t - d] : Fr::one();
    __syncthreads();
    if (t >= d) sh[t] = other * v;
    __syncthreads();
  }
  Fr acc = t ? sh[t - 1] : Fr::one();
#pragma unroll
  for (unsigned k = 0; k < 4; ++k) {
    if (k < per && lo + k < count) total[lo + k] = acc;
    acc = acc * vals[k];
  }
  if (active == 0 ? t == 0 : t == active - 1) *closes = (active == 0 || sh[t] == Fr::one()) ? 1 : 0;   // the threads past it hold ones: sh[active - 1] is the product of all
}
// z[col][0..u] *= carry[col]
static __global__ void __launch_bounds__(256) k_scale_rows(Fr *__restrict__ z, const Fr *__restrict__ carry, size_t n, unsigned rows, unsigned n_cols) {
  const size_t total = (size_t)n_cols * rows;
  for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const unsigned col = (unsigned)(g / rows);
    const unsigned i = (unsigned)(g - (size_t)col * rows);
    Fr *p = z + (size_t)col * n + i;
    *p = *p * carry[col];
  }
}

// ------------------------------------------------------------------------------------------- quotient
enum { QG_GATE = 0, QG_RLC = 1, QG_PERM_HEAD = 2, QG_PERM_C = 3, QG_PERM_D = 4, QG_LOOKUP = 5, QG_PERM_FIRST = 6, QG_PERM_LAST = 7 };
struct QGroup {
  int type, first, count, pad;
};
struct QArgs {
  const Fr *adv, *fix, *sig, *pz, *lz, *la, *ls, *inst, *lext, *xs;  // extended (coset-major) evaluations
  const Fr *beta_delta;                                               // [n_perm]
  const QGroup *groups;
  Fr *partials;  // [n_groups][4n]
  Fr y, beta, gamma, gamma_rlc;
  unsigned log_n, u, n_gate, n_rlc, adv_rlc0, fix_qrlc0, fix_const, fix_table, adv_lookup0, n_lookup, n_advice, n_perm, chunk, n_chunks;
  unsigned rows;  // cosets evaluated (3 or 4) = rows per column of every extended array (column stride rows * n)
  // the points of this launch: [pt0, pt0 + pt_count) of the rows * n (everything, or one coset row at a time when the rows'
  // shares are gathered across ranks while the next row is evaluated)
  size_t pt0, pt_count;
};

static __global__ void __launch_bounds__(256) k_quotient_partials(QArgs a) {
  const size_t n = (size_t)1 << a.log_n, ne = n * a.rows;
  const size_t p = a.pt0 + blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (p >= a.pt0 + a.pt_count || p >= ne) return;
  const QGroup g = a.groups[blockIdx.y];
  const size_t row0 = p & ~(n - 1);           // k1 * n
  const size_t k2 = p & (n - 1);
  auto at = [&](const Fr *base, unsigned col, unsigned rot) -> Fr { return base[(size_t)col * ne + row0 + ((k2 + rot) & (n - 1))]; };
  const Fr one = Fr::one();
  Fr acc = Fr::zero();
  // acc * y + u * v with one Montgomery reduction (bn254.hip.hpp fp_mul2): the Horner step of every expression
  auto horner = [&](const Fr &u, const Fr &v) { acc = zk::fp_mul2<zk::FrP>(acc, a.y, u, v); };
  switch (g.type) {
    case QG_GATE:
      for (int j = g.first; j < g.first + g.count; ++j) {
        const Fr q = at(a.fix, j, 0);
        if (!q.is_zero()) horner(q, at(a.adv, j, 0) + at(a.adv, j, 1) * at(a.adv, j, 2) - at(a.adv, j, 3));
        else acc = acc * a.y;
      }
      break;
    case QG_RLC:
      for (int j = g.first; j < g.first + g.count; ++j) {
        const unsigned col = a.adv_rlc0 + j;
        const Fr q = at(a.fix, a.fix_qrlc0 + j, 0);
        horner(q, at(a.adv, col, 0) * a.gamma_rlc + at(a.adv, col, 1) - at(a.adv, col, 2));
      }
      break;
    case QG_PERM_HEAD: {
      const Fr l0 = a.lext[p], ll = a.lext[ne + p];
      const Fr z0 = at(a.pz, 0, 0), zm = at(a.pz, a.n_chunks - 1, 0);
      acc = l0 * (one - z0);
      horner(ll, zm * zm - zm);
      break;
    }
    case QG_PERM_FIRST:   // the two expressions of QG_PERM_HEAD as groups of their own (they belong to different ranks when the
      acc = a.lext[p] * (one - at(a.pz, 0, 0));   // quotient is sharded by column)
      break;
    case QG_PERM_LAST: {
      const Fr zm = at(a.pz, a.n_chunks - 1, 0);
      acc = a.lext[ne + p] * (zm * zm - zm);
      break;
    }
    case QG_PERM_C: {
      const Fr l0 = a.lext[p];
      for (int j = g.first; j < g.first + g.count; ++j) horner(l0, at(a.pz, j, 0) - at(a.pz, j - 1, a.u));
      break;
    }
    case QG_PERM_D: {
      const Fr lact = a.lext[2 * ne + p];
      const Fr x = a.xs[p];
      for (int j = g.first; j < g.first + g.count; ++j) {
        Fr left = at(a.pz, j, 1), right = at(a.pz, j, 0);
        for (unsigned c = j * a.chunk; c < (j + 1) * a.chunk && c < a.n_perm; ++c) {
          const Fr v = c < a.n_advice ? at(a.adv, c, 0) : (c == a.n_advice ? at(a.fix, a.fix_const, 0) : a.inst[p]);
          left = left * (v + a.beta * at(a.sig, c, 0) + a.gamma);
          right = right * (v + a.beta_delta[c] * x + a.gamma);
        }
        horner(lact, left - right);
      }
      break;
    }
    case QG_LOOKUP: {
      const Fr l0 = a.lext[p], ll = a.lext[ne + p], lact = a.lext[2 * ne + p];
      const Fr s = at(a.fix, a.fix_table, 0);
      for (int i = g.first; i < g.first + g.count; ++i) {
        const Fr z0 = at(a.lz, i, 0), z1 = at(a.lz, i, 1);
        const Fr av = at(a.adv, a.adv_lookup0 + i, 0);
        const Fr ap = at(a.la, i, 0), apm = at(a.la, i, (unsigned)(n - 1)), sp = at(a.ls, i, 0);
        horner(l0, one - z0);
        horner(ll, z0 * z0 - z0);
        horner(lact, zk::fp_mul2<zk::FrP>(z1, (ap + a.beta) * (sp + a.gamma), zk::fp_neg<zk::FrP>(z0), (av + a.beta) * (s + a.gamma)));
        horner(l0, ap - sp);
        horner(lact, (ap - sp) * (ap - apm));
      }
      break;
    }
  }
  a.partials[(size_t)blockIdx.y * ne + p] = acc;
}

// h_ext[p] = (sum_g ypow[g] * partials[g][p]) * zinv[k1]
static __global__ void __launch_bounds__(256) k_quotient_combine(const Fr *__restrict__ partials, const Fr *__restrict__ ypow, unsigned n_groups,
                                                          const Fr *__restrict__ zinv, unsigned log_n, unsigned rows, size_t pt0, size_t pt_count, Fr *__restrict__ h_ext) {
  const size_t n = (size_t)1 << log_n, ne = n * rows;
  const size_t p = pt0 + blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (p >= pt0 + pt_count || p >= ne) return;
  // ypow and zinv come in the 2^261 form (constants of the call): nine-limb products, two groups per reduction, a lazy sum
  zk::F29 acc;
#pragma unroll
  for (int l = 0; l < 9; ++l) acc.l[l] = 0;
  unsigned g = 0;
  for (; g + 1 < n_groups; g += 2)
    acc = zk::fr29_weak_reduce(zk::f29_add(acc, zk::fr29_mul2(zk::fr29_unpack(partials[(size_t)g * ne + p]), zk::fr29_unpack(ypow[g]),
                                                             zk::fr29_unpack(partials[(size_t)(g + 1) * ne + p]), zk::fr29_unpack(ypow[g + 1]))));
  if (g < n_groups) acc = zk::fr29_weak_reduce(zk::f29_add(acc, zk::fr29_mul(zk::fr29_unpack(partials[(size_t)g * ne + p]), zk::fr29_unpack(ypow[g]))));
  h_ext[p] = zk::fr29_pack(zk::fr29_canonical(zk::fr29_mul(acc, zk::fr29_unpack(zinv[p >> log_n]))));
}

// Quotient from three cosets.  rows3[k1][i] = i-th coefficient (already scaled by 1/n) of h restricted to the coset
// g_k1 <w>, i.e. g_k1^i * sum_m h_m[i] c_k1^m with c_k1 = g_k1^n; pw[k1][i] = g_k1^-i; vinv = inverse of the 3x3
// Vandermonde matrix (c_k1^m).  h_c[m * n + i] = sum_k1 vinv[m][k1] * rows3[k1][i] * pw[k1][i]; the fourth piece is zero.
struct Mat3 {
  Fr v[9];
};
static __global__ void __launch_bounds__(256) k_ext3_combine(const Fr *__restrict__ rows3, const Fr *__restrict__ pw, Mat3 vinv, size_t n, Fr *__restrict__ h_c) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr t[3];
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1) t[k1] = rows3[(size_t)k1 * n + i] * pw[(size_t)k1 * n + i];
#pragma unroll
  for (int m = 0; m < 3; ++m) h_c[(size_t)m * n + i] = vinv.v[3 * m] * t[0] + vinv.v[3 * m + 1] * t[1] + vinv.v[3 * m + 2] * t[2];
  h_c[3 * n + i] = Fr::zero();
}

// ------------------------------------------------------------------------------------------- evaluations
// barycentric weights: d[r][i] = z_r - w^i (inverted by the caller), then b[r][i] = c * w^i * inv
static __global__ void __launch_bounds__(256) k_bary_den(const Fr *__restrict__ wpow, const Fr *__restrict__ pts, unsigned n_pts, size_t n,
                                                  Fr *__restrict__ out) {
  const size_t total = (size_t)n_pts * n;
  for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x)
    out[g] = pts[g / n] - wpow[g % n];
}
static __global__ void __launch_bounds__(256) k_bary_weights(const Fr *__restrict__ wpow, Fr c, size_t total, size_t n, Fr *__restrict__ inout) {
  for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x)
    inout[g] = inout[g] * wpow[g % n] * c;
}
// one workgroup per job: out[job][r] = sum_i col[i] * bw[rot_index[r]][i], up to 4 rotations per job
struct EvalJob {
  const Fr *col;
  int n_rot;
  int rot[4];  // indices into the weight table
};
// gridDim.y row slices (long columns): slice y sums rows [y n / Y, (y + 1) n / Y) into out[(y * jobs + job) * 4 + r]
static __global__ void __launch_bounds__(256) k_eval_jobs(const EvalJob *__restrict__ jobs, const Fr *__restrict__ bw /* 2^261 form: k_bary_weights with c * 32 */,
                                                          size_t n, Fr *__restrict__ out) {
  __shared__ Fr sh[256];
  const EvalJob job = jobs[blockIdx.x];
  // Nine-limb products (csrc/fr29.hip.hpp): the weights are the constant operand, stored in the 2^261 form, and two rows share one
  // Montgomery reduction (fr29_mul2); the accumulators stay below 2 r in limb form.  r < n_rot is tested inside fully unrolled
  // loops: with a loop bound read from the job the four accumulators were indexed dynamically and lived in scratch memory (144
  // bytes per lane; 5.4 ms of a k = 19 proof, 3.8 ms without the scratch, with the 8 x 32-bit product).
  zk::F29 acc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int l = 0; l < 9; ++l) acc[r].l[l] = 0;
  const size_t lo = (n / gridDim.y) * blockIdx.y, hi = blockIdx.y + 1 == gridDim.y ? n : lo + n / gridDim.y;
  size_t i = lo + threadIdx.x;
  for (; i + 256 < hi; i += 512) {
    const zk::F29 v0 = zk::fr29_unpack(job.col[i]), v1 = zk::fr29_unpack(job.col[i + 256]);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r < job.n_rot) {
        const Fr *w = bw + (size_t)job.rot[r] * n + i;
        acc[r] = zk::fr29_weak_reduce(zk::f29_add(acc[r], zk::fr29_mul2(v0, zk::fr29_unpack(w[0]), v1, zk::fr29_unpack(w[256]))));
      }
  }
  if (i < hi) {
    const zk::F29 v0 = zk::fr29_unpack(job.col[i]);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r < job.n_rot) acc[r] = zk::fr29_weak_reduce(zk::f29_add(acc[r], zk::fr29_mul(v0, zk::fr29_unpack(bw[(size_t)job.rot[r] * n + i]))));
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (r >= job.n_rot) break;
    sh[threadIdx.x] = zk::fr29_pack(zk::fr29_canonical(acc[r]));
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) sh[threadIdx.x] = sh[threadIdx.x] + sh[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + r] = sh[0];
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------- SHPLONK
// out[i] = sum_m s[m] * ptr[m][i].  The scalars are constants of the call, uploaded in the 2^261 form (zk_fr_to_29): nine-limb
// products, two columns per Montgomery reduction (fr29_mul2), the sum kept below 2 r in limb form (csrc/fr29.hip.hpp) -- these
// sums were 6 % of the summed kernel time of the k = 13 wave with the 8 x 32-bit product and an addition mod r per term.
static __device__ __forceinline__ Fr lincomb29(const Fr *const *__restrict__ ptrs, const Fr *__restrict__ s29, unsigned k0, unsigned k1, size_t i) {
  zk::F29 acc;
#pragma unroll
  for (int l = 0; l < 9; ++l) acc.l[l] = 0;
  unsigned k = k0;
  for (; k + 1 < k1; k += 2)
    acc = zk::fr29_weak_reduce(zk::f29_add(acc, zk::fr29_mul2(zk::fr29_unpack(ptrs[k][i]), zk::fr29_unpack(s29[k]), zk::fr29_unpack(ptrs[k + 1][i]), zk::fr29_unpack(s29[k + 1]))));
  if (k < k1) acc = zk::fr29_weak_reduce(zk::f29_add(acc, zk::fr29_mul(zk::fr29_unpack(ptrs[k][i]), zk::fr29_unpack(s29[k]))));
  return zk::fr29_pack(zk::fr29_canonical(acc));
}
static __global__ void __launch_bounds__(256) k_lincomb_ptrs(const Fr *const *__restrict__ ptrs, const Fr *__restrict__ s29, unsigned m, size_t n,
                                                      Fr *__restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = lincomb29(ptrs, s29, 0, m, i);
}
// the same sum split over blockIdx.y chunks of `per` pointers: partial[chunk][i]; k_sum_rows adds the chunks up.  One thread
// looping over ~600 columns is a 0.7 ms dependent chain; 13 chunks of 48 run side by side.
static __global__ void __launch_bounds__(256) k_lincomb_ptrs_chunked(const Fr *const *__restrict__ ptrs, const Fr *__restrict__ s29, unsigned m, unsigned per, size_t n,
                                                              Fr *__restrict__ partial) {
  const unsigned k0 = blockIdx.y * per, k1 = min(k0 + per, m);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    partial[(size_t)blockIdx.y * n + i] = lincomb29(ptrs, s29, k0, k1, i);
}
static __global__ void __launch_bounds__(256) k_sum_rows(const Fr *__restrict__ partial, unsigned rows, size_t n, Fr *__restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    Fr acc = partial[i];
    for (unsigned r = 1; r < rows; ++r) acc = acc + partial[(size_t)r * n + i];
    out[i] = acc;
  }
}
struct ShSet {
  Fr rc[4];    // r_j coefficients (ascending), unused ones zero
  Fr pts[4];   // points of S_j
  Fr vj;       // v^j
  Fr coef;     // v^j * Z_{T\S_j}(u)
  Fr r_u;      // r_j(u)
  int n_pts, pad[3];
};
// zs[j][i] = prod_{p in S_j} (w^i - p)
static __global__ void __launch_bounds__(256) k_sh_zs(const ShSet *__restrict__ sets, unsigned n_sets, const Fr *__restrict__ wpow, size_t n,
                                               Fr *__restrict__ zs) {
  const size_t total = (size_t)n_sets * n;
  for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const ShSet &s = sets[g / n];
    const Fr w = wpow[g % n];
    Fr acc = Fr::one();
    for (int t = 0; t < s.n_pts; ++t) acc = acc * (w - s.pts[t]);
    zs[g] = acc;
  }
}
// hq[i] = sum_j v^j (F_j[i] - r_j(w^i)) * zs_inv[j][i]
static __global__ void __launch_bounds__(256) k_sh_h(const ShSet *__restrict__ sets, unsigned n_sets, const Fr *__restrict__ F, const Fr *__restrict__ zs_inv,
                                              const Fr *__restrict__ wpow, size_t n, Fr *__restrict__ hq) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const Fr w = wpow[i];
    Fr acc = Fr::zero();
    for (unsigned j = 0; j < n_sets; ++j) {
      const ShSet &s = sets[j];
      Fr r = s.rc[3];
      r = r * w + s.rc[2];
      r = r * w + s.rc[1];
      r = r * w + s.rc[0];
      acc = acc + s.vj * (F[(size_t)j * n + i] - r) * zs_inv[(size_t)j * n + i];
    }
    hq[i] = acc;
  }
}
// den[i] = w^i - u (inverted by the caller);  W[i] = (sum_j coef_j (F_j[i] - r_j(u)) - ztu * hq[i]) * inv[i]
static __global__ void __launch_bounds__(256) k_sh_den(const Fr *__restrict__ wpow, Fr u, size_t n, Fr *__restrict__ out) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = wpow[i] - u;
}
static __global__ void __launch_bounds__(256) k_sh_w(const ShSet *__restrict__ sets, unsigned n_sets, const Fr *__restrict__ F, const Fr *__restrict__ hq,
                                              Fr ztu, const Fr *__restrict__ inv, size_t n, Fr *__restrict__ W) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    Fr acc = Fr::zero();
    for (unsigned j = 0; j < n_sets; ++j) acc = acc + sets[j].coef * (F[(size_t)j * n + i] - sets[j].r_u);
    W[i] = (acc - ztu * hq[i]) * inv[i];
  }
}

}  // namespace zkp
