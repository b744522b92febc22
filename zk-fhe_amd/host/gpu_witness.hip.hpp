// Witness generation on the GPU (the per-coefficient gadget loops of reference
// src/poly_chip.rs that produce 1.23 M of the circuit's 1.29 M advice cells).
//
// The cell layout of the circuit does not depend on the input: every gadget of PolyChip emits a fixed number of
// cells per coefficient (halo2-base layouts; restated on the host in halo2_base.hpp).  So the
// phase-1 gate stream is produced by one kernel launch per gadget call, ONE THREAD PER COEFFICIENT, each thread
// writing its cells at  stream + base + i * cells_per_coefficient  -- no host loop, no 50 MB upload.  The emitters
// below are the device twins of GateChip / RangeChip (same cell order); they are templates over a writer so the host
// can run them in "count" mode to get cells_per_coefficient from the very same code.
// Cell values are handled as canonical integers mod r held in an Fr struct and converted to Montgomery form when
// stored.  The deferred 1/x cells of is_zero are written as denominators and inverted afterwards in one batch.
#pragma once
#include "../csrc/ctx.hpp"
#include "../csrc/fr29.hip.hpp"

namespace zkw {

using zk::Fr;
using zk::FrP;
typedef unsigned long long u64;

struct DevWriter {
  Fr *o;
  __device__ __forceinline__ void put(const Fr &canon) { *o++ = zk::fr29_to_mont(canon); }
};
struct CountWriter {
  size_t n = 0;
  ZK_HD void put(const Fr &) { ++n; }
};

ZK_HD Fr c_u64(u64 v) {
  Fr t = Fr::zero();
  t.l[0] = (zk::u32)v;
  t.l[1] = (zk::u32)(v >> 32);
  return t;
}
// limbs are selected, not indexed: a limb index computed at run time puts the value into scratch memory on the device
ZK_HD Fr c_pow2(unsigned bits) {
  Fr t;
  for (unsigned i = 0; i < 8; ++i) t.l[i] = i == (bits >> 5) ? 1u << (bits & 31) : 0u;
  return t;
}
ZK_HD unsigned c_bits(const Fr &a) {
  for (int i = 7; i >= 0; --i)
    if (a.l[i]) return 32 * i + (32 - (unsigned)__builtin_clz(a.l[i]));
  return 0;
}
ZK_HD zk::u32 c_byte(const Fr &a, unsigned shift) {
  zk::u32 w = 0;
  for (unsigned i = 0; i < 8; ++i)
    if (i == (shift >> 5)) w = a.l[i];
  return (w >> (shift & 31)) & 0xffu;
}
// low `bits` bits of a (bits multiple of 8, < 256)
ZK_HD Fr c_low(const Fr &a, unsigned bits) {
  Fr t = Fr::zero();
  for (unsigned i = 0; i < 8; ++i) {
    if (32 * (i + 1) <= bits) t.l[i] = a.l[i];
    else if (32 * i < bits) t.l[i] = a.l[i] & ((1u << (bits - 32 * i)) - 1u);
  }
  return t;
}
ZK_HD Fr c_add(const Fr &a, const Fr &b) { return zk::fp_add<FrP>(a, b); }
ZK_HD Fr c_sub(const Fr &a, const Fr &b) { return zk::fp_sub<FrP>(a, b); }
ZK_HD Fr c_mul(const Fr &a, const Fr &b) { return zk::fp_mul<FrP>(zk::fp_to_mont<FrP>(a), b); }  // (aR) * b * R^-1

// ---- GateChip ----------------------------------------------------------------------------------------------
template <class W>
ZK_HD Fr e_add(W &w, const Fr &a, const Fr &b) {
  const Fr s = c_add(a, b);
  w.put(a), w.put(b), w.put(c_u64(1)), w.put(s);
  return s;
}
template <class W>
ZK_HD Fr e_sub(W &w, const Fr &a, const Fr &b) {
  const Fr d = c_sub(a, b);
  w.put(d), w.put(b), w.put(c_u64(1)), w.put(a);
  return d;
}
template <class W>
ZK_HD Fr e_mul(W &w, const Fr &a, const Fr &b) {
  const Fr p = c_mul(a, b);
  w.put(c_u64(0)), w.put(a), w.put(b), w.put(p);
  return p;
}
template <class W>
ZK_HD Fr e_not(W &w, const Fr &a) { return e_sub(w, c_u64(1), a); }
template <class W>
ZK_HD Fr e_or(W &w, const Fr &a, const Fr &b) {
  const Fr nb = c_sub(c_u64(1), b);
  const Fr out = c_sub(c_add(a, b), c_mul(a, b));
  w.put(nb), w.put(c_u64(1)), w.put(b), w.put(c_u64(1)), w.put(b), w.put(a), w.put(nb), w.put(out);
  return out;
}
// cell 2 holds the DENOMINATOR (or 1): inverted later in one batch over the structural list of such cells
template <class W>
ZK_HD Fr e_is_zero(W &w, const Fr &a) {
  const bool z = a.is_zero();
  const Fr zv = c_u64(z ? 1 : 0);
  w.put(zv), w.put(a), w.put(z ? c_u64(1) : a), w.put(c_u64(1)), w.put(c_u64(0)), w.put(a), w.put(zv), w.put(c_u64(0));
  return zv;
}
// ---- RangeChip (lookup_bits = 8) -----------------------------------------------------------------------------
template <class W>
ZK_HD void e_range_check(W &w, const Fr &a, unsigned range_bits) {
  const unsigned k = range_bits / 8;
  if (k <= 1) return;
  w.put(c_u64(c_byte(a, 0)));
  for (unsigned i = 1; i < k; ++i) {
    w.put(c_u64(c_byte(a, 8 * i)));
    w.put(c_pow2(8 * i));
    w.put(c_low(a, 8 * (i + 1)));
  }
}
template <class W>
ZK_HD void e_check_less_than(W &w, const Fr &a, const Fr &b, unsigned num_bits) {
  const Fr p2 = c_pow2(num_bits);
  const Fr shift_a = c_add(p2, a);
  const Fr chk = c_sub(shift_a, b);
  w.put(chk), w.put(b), w.put(c_u64(1)), w.put(shift_a), w.put(zk::fp_neg<FrP>(p2)), w.put(c_u64(1)), w.put(a);
  e_range_check(w, chk, num_bits);
}
template <class W>
ZK_HD void e_check_less_than_safe(W &w, const Fr &a, const Fr &b) {
  const unsigned rb = (c_bits(b) + 7) / 8 * 8;
  e_range_check(w, a, rb);
  e_check_less_than(w, a, b, rb);
}
template <class W>
ZK_HD Fr e_is_less_than(W &w, const Fr &a, const Fr &b, unsigned num_bits) {
  const unsigned padded = (num_bits + 7) / 8 * 8;
  const Fr pp = c_pow2(padded);
  const Fr shift_a = c_add(pp, a);
  const Fr shifted = c_sub(shift_a, b);
  w.put(shifted), w.put(b), w.put(c_u64(1)), w.put(shift_a), w.put(zk::fp_neg<FrP>(pp)), w.put(c_u64(1)), w.put(a);
  e_range_check(w, shifted, padded + 8);
  return e_is_zero(w, c_u64(c_byte(shifted, padded)));  // the top limb is the last lookup cell
}
// floor division of a canonical value < 2^192 by q < 2^63
ZK_HD void c_divmod(const Fr &a, u64 q, Fr &div, u64 &rem) {
  const u64 limb[3] = {(u64)a.l[0] | ((u64)a.l[1] << 32), (u64)a.l[2] | ((u64)a.l[3] << 32), (u64)a.l[4] | ((u64)a.l[5] << 32)};
  u64 out[3], r = 0;
  for (int li = 2; li >= 0; --li) {
    u64 qd = 0;
    for (int b = 63; b >= 0; --b) {
      r = (r << 1) | ((limb[li] >> b) & 1);
      qd <<= 1;
      if (r >= q) {
        r -= q;
        qd |= 1;
      }
    }
    out[li] = qd;
  }
  div = Fr::zero();
  for (int i = 0; i < 3; ++i) {
    div.l[2 * i] = (zk::u32)out[i];
    div.l[2 * i + 1] = (zk::u32)(out[i] >> 32);
  }
  rem = r;
}
template <class W>
ZK_HD Fr e_div_mod(W &w, const Fr &a, u64 q, const Fr &div_bound) {
  Fr div;
  u64 rem;
  c_divmod(a, q, div, rem);
  const Fr r = c_u64(rem);
  w.put(r), w.put(c_u64(q)), w.put(div), w.put(a);
  e_check_less_than_safe(w, div, div_bound);
  e_check_less_than_safe(w, r, c_u64(q));
  return r;
}
// ---- PolyChip per-coefficient bodies (reference src/poly_chip.rs:270-366) ---------------------------------------
template <class W>
ZK_HD void e_coeff_in_range(W &w, const Fr &c, u64 z, u64 y) {  // constrain_coefficients_in_range :270-317
  const unsigned y_bits = c_bits(c_u64(y));
  e_check_less_than_safe(w, c, c_u64(y));
  const Fr in1 = e_is_less_than(w, c, c_u64(z + 1), y_bits);
  const Fr nin2 = e_is_less_than(w, c, c_u64(y - z), y_bits);
  const Fr in2 = e_not(w, nin2);
  e_or(w, in1, in2);
}
template <class W>
ZK_HD void e_coeff_chi_key(W &w, const Fr &c, u64 z) {  // constrain_from_distribution_chi_key :320-354
  const Fr f1 = e_sub(w, c, c_u64(0));
  const Fr f2 = e_sub(w, c, c_u64(1));
  const Fr f3 = e_sub(w, c, c_u64(z));
  const Fr f12 = e_mul(w, f1, f2);
  e_mul(w, f12, f3);
}
template <class W>
ZK_HD void e_coeff_equal(W &w, const Fr &a, const Fr &b) {  // constrain_equality :255-264
  const Fr d = e_sub(w, a, b);
  e_is_zero(w, d);
}

// cells per coefficient of every gadget, from the emitters themselves (host, count mode)
inline size_t cpc_in_range(u64 z, u64 y) {
  CountWriter w;
  e_coeff_in_range(w, c_u64(1), z, y);
  return w.n;
}
inline size_t cpc_chi_key() {
  CountWriter w;
  e_coeff_chi_key(w, c_u64(1), 5);
  return w.n;
}
inline size_t cpc_div_mod(u64 q, const Fr &bound) {
  CountWriter w;
  e_div_mod(w, c_u64(1), q, bound);
  return w.n;
}
inline size_t cpc_in_field(u64 q) {
  CountWriter w;
  e_check_less_than_safe(w, c_u64(1), c_u64(q));
  return w.n;
}
inline size_t cpc_equal() {
  CountWriter w;
  e_coeff_equal(w, c_u64(1), c_u64(2));
  return w.n;
}
// offsets (inside one coefficient's block) of the deferred-inverse cells
inline void inv_slots_in_range(u64 z, u64 y, std::vector<unsigned> &out) {
  // two is_less_than calls: each ends with an 8-cell is_zero whose cell 2 is the slot
  CountWriter w;
  e_check_less_than_safe(w, c_u64(1), c_u64(y));
  const unsigned y_bits = c_bits(c_u64(y));
  e_is_less_than(w, c_u64(1), c_u64(z + 1), y_bits);
  out.push_back((unsigned)w.n - 8 + 2);
  e_is_less_than(w, c_u64(1), c_u64(y - z), y_bits);
  out.push_back((unsigned)w.n - 8 + 2);
}

// ---- kernels: one thread per coefficient ---------------------------------------------------------------------
enum { G_IN_RANGE = 0, G_CHI_KEY, G_DIV_MOD, G_IN_FIELD, G_ADD, G_SCALAR_MUL, G_EQUAL };
struct GadgetArgs {
  int type;
  const Fr *a, *b;   // input coefficient arrays (canonical); b: second operand (add / equal), or a single scalar (scalar_mul)
  Fr *out;           // output coefficient array (canonical) or nullptr
  Fr *stream;        // Montgomery cells
  size_t base, cpc, count;
  u64 p0, p1;        // z, y  |  z  |  q
  Fr bound;          // div_mod bound
};
// blockIdx.y selects one of the gadget calls of a dependency level: calls whose inputs are ready run side by side in one
// launch (24 calls, 5 levels: see GpuPhase1::launch)
static __global__ void __launch_bounds__(256) k_gadget(const GadgetArgs *__restrict__ calls) {
  const GadgetArgs g = calls[blockIdx.y];
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= g.count) return;
  DevWriter w{g.stream + g.base + i * g.cpc};
  const Fr a = g.a[i];
  switch (g.type) {
    case G_IN_RANGE: e_coeff_in_range(w, a, g.p0, g.p1); break;
    case G_CHI_KEY: e_coeff_chi_key(w, a, g.p0); break;
    case G_DIV_MOD: {
      const Fr r = e_div_mod(w, a, g.p0, g.bound);
      if (g.out) g.out[i] = r;
      break;
    }
    case G_IN_FIELD: e_check_less_than_safe(w, a, c_u64(g.p0)); break;
    case G_ADD: {
      const Fr s = e_add(w, a, g.b[i]);
      if (g.out) g.out[i] = s;
      break;
    }
    case G_SCALAR_MUL: {
      const Fr s = e_mul(w, a, g.b[0]);
      if (g.out) g.out[i] = s;
      break;
    }
    case G_EQUAL: e_coeff_equal(w, a, g.b[i]); break;
  }
}

// stream -> columns: column c, rows 0..len_c-1  <-  stream[start_c ..]  (a break-point duplicate is simply the next cell)
static __global__ void __launch_bounds__(256) k_place(const Fr *__restrict__ stream, const unsigned *__restrict__ col_start,
                                               const unsigned *__restrict__ col_len, unsigned n_cols, size_t n, Fr *__restrict__ cols) {
  const size_t total = (size_t)n_cols * n;
  for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const unsigned c = (unsigned)(g / n);
    const unsigned r = (unsigned)(g - (size_t)c * n);
    cols[g] = r < col_len[c] ? stream[col_start[c] + r] : Fr::zero();
  }
}
// the four gate cells of every constrain_mul depend on the phase-1 challenge: they are written after the rest of the column was
// committed (prove.hip "early phase-1 commitment") -- into the advice columns and into the sparse correction columns
struct PatchCell {
  Fr *dst_adv, *dst_patch;
  unsigned value;   // index into the values array
};
static __global__ void __launch_bounds__(64) k_patch_cells(const PatchCell *__restrict__ cells, unsigned count, const Fr *__restrict__ values) {
  const unsigned i = threadIdx.x;
  if (i >= count) return;
  const Fr v = values[cells[i].value];
  *cells[i].dst_adv = v;
  if (cells[i].dst_patch) *cells[i].dst_patch = v;
}
// up to four runs of four reserved cells back to zero (one launch instead of four fills)
struct ZeroRuns {
  Fr *at[4];
  unsigned count;
};
static __global__ void __launch_bounds__(64) k_zero_runs(ZeroRuns z) {
  const unsigned i = threadIdx.x >> 2, j = threadIdx.x & 3;
  if (i < z.count) z.at[i][j] = Fr::zero();
}
// lookup advice columns: the k-th looked-up cell goes to column k / max_rows, row k % max_rows
static __global__ void __launch_bounds__(256) k_place_lookups(const Fr *__restrict__ stream, const unsigned *__restrict__ src_off, size_t n_lookups,
                                                       unsigned max_rows, size_t n, unsigned n_lookup_cols, Fr *__restrict__ cols) {
  const size_t total = (size_t)n_lookup_cols * n;
  for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t c = g / n, r = g - c * n;
    const size_t k = c * max_rows + r;
    cols[g] = (r < max_rows && k < n_lookups) ? stream[src_off[k]] : Fr::zero();
  }
}
// deferred inverses: gather the slots, (batch invert), scatter back
static __global__ void __launch_bounds__(256) k_gather(const Fr *__restrict__ src, const unsigned *__restrict__ idx, size_t count, Fr *__restrict__ dst) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < count) dst[i] = src[idx[i]];
}
static __global__ void __launch_bounds__(256) k_scatter(Fr *__restrict__ dst, const unsigned *__restrict__ idx, size_t count, const Fr *__restrict__ src) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < count) dst[idx[i]] = src[i];
}

// halo2 permute_expression_pair for an 8-bit table, one workgroup per lookup column.
// in: Lagrange column (Montgomery), rows < u are the inputs.  out_a / out_s rows < u (Montgomery); err set if a value > 255.
static __global__ void __launch_bounds__(1024) k_lookup_permute(const Fr *__restrict__ in, size_t n, unsigned u, Fr *__restrict__ out_a, Fr *__restrict__ out_s,
                                                        int *__restrict__ err) {
  __shared__ unsigned cnt[256], start[257], hole0[257], left0[257];
  __shared__ Fr mont[256];
  const Fr *col = in + (size_t)blockIdx.x * n;
  Fr *oa = out_a + (size_t)blockIdx.x * n, *os = out_s + (size_t)blockIdx.x * n;
  const unsigned t = threadIdx.x, T = blockDim.x;  // T >= 256: the first 256 threads own the table values
  if (t < 256) {
    cnt[t] = 0;
    mont[t] = zk::fp_to_mont<FrP>(c_u64(t));
  }
  __syncthreads();
  for (unsigned i = t; i < u; i += T) {
    const Fr v = zk::fp_from_mont<FrP>(col[i]);
    if (v.l[1] | v.l[2] | v.l[3] | v.l[4] | v.l[5] | v.l[6] | v.l[7] || v.l[0] > 255u) {
      *(volatile int *)err = 1;   // a plain store: the flag may live in pinned host memory (every writer stores the same value)
      continue;
    }
    atomicAdd(&cnt[v.l[0]], 1u);
  }
  __syncthreads();
  if (t == 0) {
    // start[v]: first row of value v in the sorted column; hole0[v]: index of its first hole among all holes;
    // left0[v]: index of its first leftover table value among all leftovers (table = {0..255} once, then zeros)
    unsigned s = 0, h = 0, l = 0;
    for (unsigned v = 0; v < 256; ++v) {
      start[v] = s;
      hole0[v] = h;
      left0[v] = l;
      s += cnt[v];
      h += cnt[v] ? cnt[v] - 1 : 0;
      const unsigned in_table = v == 0 ? u - 255u : 1u;
      l += in_table - (cnt[v] ? 1u : 0u);
    }
    start[256] = s;
    hole0[256] = h;
    left0[256] = l;
  }
  __syncthreads();
  // row i of the sorted column: value v = the run containing i; first row of a run keeps v in S', the others are holes
  // that take the leftover table values in ascending order
  for (unsigned i = t; i < u; i += T) {
    if (i >= start[256]) {  // only when an out-of-table value was skipped: leave the row, err is set
      continue;
    }
    unsigned lo = 0, hi = 256;  // largest v with start[v] <= i and cnt[v] > 0
    while (hi - lo > 1) {
      const unsigned mid = (lo + hi) / 2;
      if (start[mid] <= i) lo = mid; else hi = mid;
    }
    while (cnt[lo] == 0) --lo;  // start[] is flat over empty runs: step back to the run that owns row i
    const unsigned v = lo, k = i - start[v];
    oa[i] = mont[v];
    if (k == 0) {
      os[i] = mont[v];
    } else {
      const unsigned j = hole0[v] + k - 1;
      unsigned a = 0, b = 256;  // w with left0[w] <= j < left0[w+1]
      while (b - a > 1) {
        const unsigned mid = (a + b) / 2;
        if (left0[mid] <= j) a = mid; else b = mid;
      }
      // a is the LAST value with left0[a] <= j; values with no leftover share left0 with their successor, and the last
      // of such a tie is the one that owns j
      os[i] = mont[a];
    }
  }
}

// The same for long columns (n > 2^15: one workgroup per column walks 2^19 rows twice -- 2 ms of a k = 19 proof on 17 of the 256
// CUs): gridDim.y row slices count into a global histogram, a second launch rebuilds the run tables from it and fills its slice.
static __global__ void __launch_bounds__(1024) k_lookup_count(const Fr *__restrict__ in, size_t n, unsigned u, unsigned *__restrict__ gcnt /* [cols][256], zero */,
                                                      int *__restrict__ err) {
  __shared__ unsigned cnt[256];
  const Fr *col = in + (size_t)blockIdx.x * n;
  const unsigned t = threadIdx.x;
  if (t < 256) cnt[t] = 0;
  __syncthreads();
  const unsigned per = (u + gridDim.y - 1) / gridDim.y, lo = per * blockIdx.y, hi = min(u, lo + per);
  for (unsigned i = lo + t; i < hi; i += blockDim.x) {
    const Fr v = zk::fp_from_mont<FrP>(col[i]);
    if (v.l[1] | v.l[2] | v.l[3] | v.l[4] | v.l[5] | v.l[6] | v.l[7] || v.l[0] > 255u) {
      *(volatile int *)err = 1;
      continue;
    }
    atomicAdd(&cnt[v.l[0]], 1u);
  }
  __syncthreads();
  if (t < 256 && cnt[t]) atomicAdd(&gcnt[blockIdx.x * 256 + t], cnt[t]);
}
static __global__ void __launch_bounds__(1024) k_lookup_fill(const unsigned *__restrict__ gcnt, size_t n, unsigned u, Fr *__restrict__ out_a, Fr *__restrict__ out_s) {
  __shared__ unsigned cnt[256], start[257], hole0[257], left0[257];
  __shared__ Fr mont[256];
  Fr *oa = out_a + (size_t)blockIdx.x * n, *os = out_s + (size_t)blockIdx.x * n;
  const unsigned t = threadIdx.x;
  if (t < 256) {
    cnt[t] = gcnt[blockIdx.x * 256 + t];
    mont[t] = zk::fp_to_mont<FrP>(c_u64(t));
  }
  __syncthreads();
  if (t == 0) {   // as in k_lookup_permute
    unsigned s = 0, h = 0, l = 0;
    for (unsigned v = 0; v < 256; ++v) {
      start[v] = s;
      hole0[v] = h;
      left0[v] = l;
      s += cnt[v];
      h += cnt[v] ? cnt[v] - 1 : 0;
      const unsigned in_table = v == 0 ? u - 255u : 1u;
      l += in_table - (cnt[v] ? 1u : 0u);
    }
    start[256] = s;
    hole0[256] = h;
    left0[256] = l;
  }
  __syncthreads();
  const unsigned per = (u + gridDim.y - 1) / gridDim.y, lo_r = per * blockIdx.y, hi_r = min(u, lo_r + per);
  for (unsigned i = lo_r + t; i < hi_r; i += blockDim.x) {
    if (i >= start[256]) continue;
    unsigned lo = 0, hi = 256;
    while (hi - lo > 1) {
      const unsigned mid = (lo + hi) / 2;
      if (start[mid] <= i) lo = mid; else hi = mid;
    }
    while (cnt[lo] == 0) --lo;
    const unsigned v = lo, k = i - start[v];
    oa[i] = mont[v];
    if (k == 0) {
      os[i] = mont[v];
    } else {
      const unsigned j = hole0[v] + k - 1;
      unsigned a = 0, b = 256;
      while (b - a > 1) {
        const unsigned mid = (a + b) / 2;
        if (left0[mid] <= j) a = mid; else b = mid;
      }
      os[i] = mont[a];
    }
  }
}
// the launch: one workgroup per column up to n = 2^15, row slices of 2^14 beyond
static inline int lookup_permute(zkfhe_ctx *ctx, const Fr *in, size_t n, unsigned u, unsigned n_cols, Fr *out_a, Fr *out_s, int *err,
                                 unsigned *hist = nullptr /* n_cols * 256 words of the caller's, or scratch slot 3 */) {
  if (!n_cols) return ZKFHE_OK;
  if (n <= 32768) {
    k_lookup_permute<<<n_cols, 1024, 0, ctx->stream>>>(in, n, u, out_a, out_s, err);
    ZK_LAUNCH_CHECK(ctx);
    return ZKFHE_OK;
  }
  void *p = hist;
  if (!p) {
    const int rc = zk_scratch(ctx, 3, (size_t)n_cols * 256 * sizeof(unsigned), &p);
    if (rc) return rc;
  }
  ZK_HIP(ctx, hipMemsetAsync(p, 0, (size_t)n_cols * 256 * sizeof(unsigned), ctx->stream));
  const dim3 grid(n_cols, (unsigned)(n >> 14));
  k_lookup_count<<<grid, 1024, 0, ctx->stream>>>(in, n, u, (unsigned *)p, err);
  ZK_LAUNCH_CHECK(ctx);
  k_lookup_fill<<<grid, 1024, 0, ctx->stream>>>((const unsigned *)p, n, u, out_a, out_s);
  ZK_LAUNCH_CHECK(ctx);
  return ZKFHE_OK;
}

}  // namespace zkw
