// The stages above the 2^13 tile of a long-row NTT (n = 2^16: three, n = 2^19: six) in FOUR-STEP form -- replaces k_dif_fused<3> /
// k_dif_lds<3> (ntt.hip) on those rows.  Same seam: halo2_proofs best_fft / EvaluationDomain::coeff_to_extended (third-party, reached
// from reference examples/bfv.rs:311); BASELINE configs[3] / [4] ("large-domain NTT, LDS twiddle tiling").
//
// With n = M q (M = 8 or 64), i = m q + j0, k = k1 + M k2:
//     X[k1 + M k2] = sum_j0 w_q^(j0 k2) * ( w_n^(j0 k1) * sum_m x[m q + j0] w_M^(m k1) )
// i.e. (1) a size-M DFT down every column j0 of the M x q matrix -- its twiddles are powers of w_M: at most 63 constants, none of
// them a function of j0; (2) ONE product per element with w_n^(j0 k1); (3) the size-q transforms of the rows, which the 2^13 tile
// kernel runs on row-slot brev(k1).  The radix-2 passes this replaces gathered a 32-byte twiddle w_n^(j 2^t) per butterfly from the
// n-entry table (stride 2^t entries across the lanes: k_dif_lds moved 1.9 x its algorithmic bytes, profiles/r4_roofline.md) and the
// coset extension read a second n-entry table of coset powers.  Here:
//   * the size-8 DFTs are the butterfly of the 2^13 tile (dft8.hip.hpp: five products by w_8 powers, signed lazy limbs); M = 64 is two
//     of them around one exchange through LDS, with the 7 x 8 constants w_64^(g s) in between;
//   * step (2) and the j0-part of the coset factor h^i = (h^q)^m h^j0 are ONE table T[slot][j0] = h^j0 w_n^(j0 k1) laid out exactly
//     where the result is stored: a streaming, coalesced read; the m-part (h^q)^m is M constants applied at the load;
//   * tiles are numbered position-major (all transforms of one j0 range next to each other), so the workgroups in flight read the
//     same 32 KB slice of T and it is served by the L2s: one table for hundreds of columns.
// Products per element: 2 (two DFT8 levels) + 1 (T) [+ 1 (coset constant)], as before; no gathered loads are left.
#include <array>
#include <map>

#include "ctx.hpp"
#include "dft8.hip.hpp"

using namespace zk;

namespace {

__host__ __device__ inline unsigned brev3(unsigned v) { return ((v & 1u) << 2) | (v & 2u) | ((v >> 2) & 1u); }

// T[slot * q + j0] = start * h^j0 * w^(j0 * brev_S(slot) mod n), S = log2 M; all in the standard form except `start` (32: the 2^261 form)
__global__ void __launch_bounds__(256) k_dif8_table(Fr start, Fr h, Fr w, int log_n, int S, Fr *__restrict__ out) {
  const size_t n = (size_t)1 << log_n;
  const int log_q = log_n - S;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t slot = idx >> log_q, j0 = idx & (((size_t)1 << log_q) - 1);
    size_t k1 = 0;
    for (int b = 0; b < S; ++b) k1 |= ((slot >> b) & 1) << (S - 1 - b);
    Fr r = start, b1 = h;
    for (size_t e = j0; e; e >>= 1) {
      if (e & 1) r = r * b1;
      b1 = fp_sqr<FrP>(b1);
    }
    Fr b2 = w;
    for (size_t e = (j0 * k1) & (n - 1); e; e >>= 1) {
      if (e & 1) r = r * b2;
      b2 = fp_sqr<FrP>(b2);
    }
    out[idx] = r;
  }
}

struct Dif8Args {
  const Fr *src;
  Fr *dst;
  size_t n_cols;      // transforms: input vector c / rows, coset row c % rows
  int log_n;
  unsigned rows;
  const LwMem *consts;   // [0..2] w8, w4, w8^3; [3 + 7 g + (s - 1)] w64^(g s) (M = 64 only)
  const Fr *tab[4];      // per coset row: [M constants (h^q)^m | n entries T], 2^261 form; rows == 1 and no coset: the constants are not read
  int coset;
};

__device__ __forceinline__ LzT lz_t(const Lz<0, 1, 1> &a) {   // the same limbs under the wider bound of a product's result
  LzT r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = a.l[i];
  return r;
}

// M = 8: one thread per (transform, j0); elements j0 + r q in, row-slot brev3(s) out
__global__ void __launch_bounds__(256, 2) k_dif8_one(Dif8Args A) {
  const int log_q = A.log_n - 3;
  const size_t q = (size_t)1 << log_q, n = (size_t)1 << A.log_n;
  const size_t blocks_per_col = q >> 8, total = A.n_cols * blocks_per_col;
  Consts K;
  K.w8 = tw_at(A.consts);
  K.w4 = tw_at(A.consts + 1);
  K.w83 = tw_at(A.consts + 2);
  for (size_t wi = blockIdx.x; wi < total; wi += gridDim.x) {
    const size_t c = wi % A.n_cols, pb = wi / A.n_cols;   // position-major: neighbours in the grid share their slice of T
    const size_t j0 = (pb << 8) + threadIdx.x;
    const size_t cs = c / A.rows, k1r = c - cs * A.rows;
    const Fr *ps = A.src + cs * n + j0;
    const Fr *__restrict__ tb = A.tab[k1r];
    Fr raw[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) raw[r] = ps[(size_t)r << log_q];
    LzT x[8];
    if (A.coset) {
#pragma unroll
      for (int r = 0; r < 8; ++r) x[r] = mulw(lz_load(raw[r]), lw_unpack(tb[r]));
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) x[r] = lz_t(lz_load(raw[r]));
    }
    const Fr *__restrict__ T = tb + 8 + j0;
    Fr t[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) t[s] = T[(size_t)brev3(s) << log_q];   // requested before the butterfly: the latency hides behind its products
    ZK_F
    ZK_DFT8_CORE(x, K)
    Fr *pd = A.dst + c * n + j0;
    pd[(size_t)brev3(0) << log_q] = lz_store(mulw(y0, lw_unpack(t[0]))); ZK_F
    pd[(size_t)brev3(1) << log_q] = lz_store(mulw(y1, lw_unpack(t[1]))); ZK_F
    pd[(size_t)brev3(2) << log_q] = lz_store(mulw(y2, lw_unpack(t[2]))); ZK_F
    pd[(size_t)brev3(3) << log_q] = lz_store(mulw(y3, lw_unpack(t[3]))); ZK_F
    pd[(size_t)brev3(4) << log_q] = lz_store(mulw(y4, lw_unpack(t[4]))); ZK_F
    pd[(size_t)brev3(5) << log_q] = lz_store(mulw(y5, lw_unpack(t[5]))); ZK_F
    pd[(size_t)brev3(6) << log_q] = lz_store(mulw(y6, lw_unpack(t[6]))); ZK_F
    pd[(size_t)brev3(7) << log_q] = lz_store(mulw(y7, lw_unpack(t[7]))); ZK_F
  }
}

// M = 64: a workgroup of 128 threads owns the 64 rows x 16 consecutive j0 of one transform.  Thread (g = tid / 16, jj): DFT8 over
// r of the rows m = g + 8 r, times w64^(g s); exchange through LDS (nine limbs: two 16-byte arrays and a 4-byte one, contiguous
// across the lanes); thread (u, jj): DFT8 over g of the values s = u, times T, row-slot 8 brev3(u) + brev3(s').
__global__ void __launch_bounds__(128, 2) k_dif8_two(Dif8Args A) {
  __shared__ uint4 sh_a[1024], sh_b[1024];
  __shared__ int sh_c[1024];
  const int log_q = A.log_n - 6;
  const size_t q = (size_t)1 << log_q, n = (size_t)1 << A.log_n;
  const size_t pos_per_col = q >> 4, total = A.n_cols * pos_per_col;
  const unsigned jj = threadIdx.x & 15u, g = threadIdx.x >> 4;
  Consts K;
  K.w8 = tw_at(A.consts);
  K.w4 = tw_at(A.consts + 1);
  K.w83 = tw_at(A.consts + 2);
  for (size_t wi = blockIdx.x; wi < total; wi += gridDim.x) {
    const size_t c = wi % A.n_cols, pos = wi / A.n_cols;
    const size_t j0 = (pos << 4) + jj;
    const size_t cs = c / A.rows, k1r = c - cs * A.rows;
    const Fr *ps = A.src + cs * n + j0;
    const Fr *__restrict__ tb = A.tab[k1r];
    LzT x[8];
    {
      Fr raw[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) raw[r] = ps[(size_t)(g + 8u * r) << log_q];
      if (A.coset) {
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = mulw(lz_load(raw[r]), lw_unpack(tb[g + 8u * r]));
      } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = lz_t(lz_load(raw[r]));
      }
    }
    {
      const LwMem *__restrict__ t64 = A.consts + 3 + 7 * g;
      const Lw t1 = tw_at(t64), t2 = tw_at(t64 + 1), t3 = tw_at(t64 + 2), t4 = tw_at(t64 + 3), t5 = tw_at(t64 + 4), t6 = tw_at(t64 + 5), t7 = tw_at(t64 + 6);
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
    __syncthreads();   // the previous tile has been read
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const unsigned idx = (s * 8u + g) * 16u + jj;
      sh_a[idx] = make_uint4((u32)x[s].l[0], (u32)x[s].l[1], (u32)x[s].l[2], (u32)x[s].l[3]);
      sh_b[idx] = make_uint4((u32)x[s].l[4], (u32)x[s].l[5], (u32)x[s].l[6], (u32)x[s].l[7]);
      sh_c[idx] = x[s].l[8];
    }
    __syncthreads();
    const unsigned u = g;   // this thread's s from here on
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const unsigned idx = (u * 8u + r) * 16u + jj;
      const uint4 lo = sh_a[idx], hi = sh_b[idx];
      x[r].l[0] = (int)lo.x, x[r].l[1] = (int)lo.y, x[r].l[2] = (int)lo.z, x[r].l[3] = (int)lo.w;
      x[r].l[4] = (int)hi.x, x[r].l[5] = (int)hi.y, x[r].l[6] = (int)hi.z, x[r].l[7] = (int)hi.w;
      x[r].l[8] = sh_c[idx];
    }
    const size_t row0 = (size_t)(8u * brev3(u)) << log_q;
    const Fr *__restrict__ T = tb + 64 + row0 + j0;
    Fr t[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) t[s] = T[(size_t)brev3(s) << log_q];
    ZK_F
    ZK_DFT8_CORE(x, K)
    Fr *pd = A.dst + c * n + row0 + j0;
    pd[(size_t)brev3(0) << log_q] = lz_store(mulw(y0, lw_unpack(t[0]))); ZK_F
    pd[(size_t)brev3(1) << log_q] = lz_store(mulw(y1, lw_unpack(t[1]))); ZK_F
    pd[(size_t)brev3(2) << log_q] = lz_store(mulw(y2, lw_unpack(t[2]))); ZK_F
    pd[(size_t)brev3(3) << log_q] = lz_store(mulw(y3, lw_unpack(t[3]))); ZK_F
    pd[(size_t)brev3(4) << log_q] = lz_store(mulw(y4, lw_unpack(t[4]))); ZK_F
    pd[(size_t)brev3(5) << log_q] = lz_store(mulw(y5, lw_unpack(t[5]))); ZK_F
    pd[(size_t)brev3(6) << log_q] = lz_store(mulw(y6, lw_unpack(t[6]))); ZK_F
    pd[(size_t)brev3(7) << log_q] = lz_store(mulw(y7, lw_unpack(t[7]))); ZK_F
  }
}

Fr fr_pow_u64(Fr b, uint64_t e) {
  Fr r = Fr::one();
  while (e) {
    if (e & 1) r = r * b;
    b = b * b;
    e >>= 1;
  }
  return r;
}

}  // namespace

// The top S (3 or 6) DIF stages of `n_cols` transforms of 2^log_n points, src -> dst (out of place or in place: a workgroup reads
// all it writes before it writes).  shifts_host: nullptr, or the `rows` coset shifts h_k1 -- transform c then reads input vector
// c / rows and is the transform of x[i] h_(c % rows)^i.  Tables are built on first use and kept with the context.
int zk_dif8_pass(zkfhe_ctx *ctx, const Fr *src, Fr *dst, size_t n_cols, int log_n, int S, int inverse, const Fr *shifts_host, unsigned rows) {
  if (!(S == 3 || S == 6) || log_n - S < 8 || rows < 1 || rows > 4 || (shifts_host == nullptr && rows != 1)) return zk_fail_msg(ctx, ZKFHE_EINVAL, "zk_dif8_pass: unsupported shape");
  const int M = 1 << S;
  const size_t n = (size_t)1 << log_n, q = n >> S;
  const Fr w = inverse ? fp_inv<FrP>(zk_fr_root_of_unity(log_n)) : zk_fr_root_of_unity(log_n);
  const Fr c32 = zk_fr_to_29(Fr::one());
  Dif8Args A{};
  A.src = src;
  A.dst = dst;
  A.n_cols = n_cols;
  A.log_n = log_n;
  A.rows = rows;
  A.coset = shifts_host != nullptr;
  // constants of the butterflies: w8, w4, w8^3 and w64^(g s)
  {
    std::array<uint64_t, 8> key{};
    key[0] = (uint64_t)log_n | ((uint64_t)S << 8) | ((uint64_t)(inverse ? 1 : 0) << 16) | ((uint64_t)1 << 32);
    auto it = ctx->dif8.find(key);
    if (it == ctx->dif8.end()) {
      std::vector<LwMem> h(3 + 7 * 8);
      const Fr w8 = fr_pow_u64(w, n >> 3), w64 = fr_pow_u64(w, n >> 6);
      h[0] = lw_from_packed(zk_fr_to_29(w8));
      h[1] = lw_from_packed(zk_fr_to_29(w8 * w8));
      h[2] = lw_from_packed(zk_fr_to_29(w8 * w8 * w8));
      for (unsigned g = 0; g < 8; ++g)
        for (unsigned s = 1; s < 8; ++s) h[3 + 7 * g + (s - 1)] = lw_from_packed(zk_fr_to_29(fr_pow_u64(w64, (uint64_t)g * s)));
      void *d = nullptr;
      ZK_HIP(ctx, hipMalloc(&d, h.size() * sizeof(LwMem)));
      // blocking copy + a wait on the null stream it ran on: the kernels that read the table run on ctx->stream, which is
      // hipStreamNonBlocking and would not order itself behind the copy (once per table: they are cached with the context)
      if (hipMemcpy(d, h.data(), h.size() * sizeof(LwMem), hipMemcpyHostToDevice) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
        (void)hipFree(d);
        return zk_fail_msg(ctx, ZKFHE_EHIP, "zk_dif8_pass: constant upload failed");
      }
      it = ctx->dif8.emplace(key, d).first;
    }
    A.consts = (const LwMem *)it->second;
  }
  for (unsigned k1 = 0; k1 < rows; ++k1) {
    const Fr h = shifts_host ? shifts_host[k1] : Fr::one();
    std::array<uint64_t, 8> key{};
    key[0] = (uint64_t)log_n | ((uint64_t)S << 8) | ((uint64_t)(inverse ? 1 : 0) << 16);
    for (int i = 0; i < 4; ++i) key[1 + i] = (uint64_t)h.l[2 * i] | ((uint64_t)h.l[2 * i + 1] << 32);
    auto it = ctx->dif8.find(key);
    if (it == ctx->dif8.end()) {
      // a bounded cache: the prover uses one plain table per direction and one per coset row; a caller that sweeps the coset
      // generator must not grow device memory for the life of the context
      if (ctx->dif8.size() >= 24) {
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (auto &kv : ctx->dif8) (void)hipFree(kv.second);
        ctx->dif8.clear();
        return zk_dif8_pass(ctx, src, dst, n_cols, log_n, S, inverse, shifts_host, rows);   // rebuild what this call needs
      }
      Fr *d = nullptr;
      ZK_HIP(ctx, hipMalloc((void **)&d, ((size_t)M + n) * sizeof(Fr)));
      std::vector<Fr> cp(M);
      const Fr hq = fr_pow_u64(h, q);
      Fr cur = c32;
      for (int m = 0; m < M; ++m) {
        cp[m] = cur;
        cur = cur * hq;
      }
      if (hipMemcpy(d, cp.data(), M * sizeof(Fr), hipMemcpyHostToDevice) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
        (void)hipFree(d);
        return zk_fail_msg(ctx, ZKFHE_EHIP, "zk_dif8_pass: constant upload failed");
      }
      unsigned grid = zk_blocks(n, 256);
      if (grid > 4096) grid = 4096;
      k_dif8_table<<<grid, 256, 0, ctx->stream>>>(c32, h, w, log_n, S, d + M);
      if (hipGetLastError() != hipSuccess) {
        (void)hipFree(d);
        return zk_fail_msg(ctx, ZKFHE_EHIP, "zk_dif8_pass: table launch failed");
      }
      it = ctx->dif8.emplace(key, (void *)d).first;
    }
    A.tab[k1] = (const Fr *)it->second;
  }
  if (S == 3) {
    const size_t total = n_cols * (q >> 8);
    const unsigned grid = (unsigned)(total < (size_t)ctx->num_cu * 8 ? total : (size_t)ctx->num_cu * 8);
    k_dif8_one<<<grid, 256, 0, ctx->stream>>>(A);
  } else {
    const size_t total = n_cols * (q >> 4);
    const unsigned grid = (unsigned)(total < (size_t)ctx->num_cu * 16 ? total : (size_t)ctx->num_cu * 16);
    k_dif8_two<<<grid, 128, 0, ctx->stream>>>(A);
  }
  ZK_LAUNCH_CHECK(ctx);
  return ZKFHE_OK;
}
