// Round 6 probe (kill line "< 10 % faster on one expression group"): the two largest expression kinds of
// k_quotient_partials (host/prover_kernels.hip.hpp) -- the gate  q (a + b c - d)  and the permutation term
// z(wX) prod (v + beta sigma + gamma) - z(X) prod (v + beta delta^c X + gamma)  -- with the arithmetic the kernel has today
// (8 x 32-bit Montgomery product, fp_mul2 Horner steps, data in the 2^256 form) against the nine-limb lazy arithmetic of
// csrc/lz29.hip.hpp on data stored in the 2^261 form (what the coset extension would have to produce: no 2^-5 drift in
// data x data products, every challenge constant uploaded in that form).  Same loads, same grid (one thread per extended point,
// one block row per group of eight gates / four chunks), same Horner structure; only the arithmetic differs.  The second variant's
// VALUES are not checked here (random words are not canonical 2^261-form data: timing only) -- the kill line is about speed.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zk-fhe_amd/csrc tools/exp/quotient_gate_probe.hip -o /tmp/qprobe && /tmp/qprobe [log_n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../zk-fhe_amd/csrc/lz29.hip.hpp"
using namespace zk;

#define CHECK(e)                                                                 \
  do {                                                                           \
    hipError_t _e = (e);                                                         \
    if (_e != hipSuccess) {                                                      \
      fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e));                   \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

// general nine-limb products: first operands lazy signed limbs (|l| < 2^30 for one product, < 2^29 for two), second operands
// with tight limbs (a canonical constant, a fresh load, a product's result)
template <class A, class B>
__device__ __forceinline__ LzT mul_g(const A &a, const B &b) {
  constexpr u32 P[9] = ZK_R29_P;
  int m[9];
  LzT r;
  long long acc = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int j = 0; j < k; ++j) {
      acc += (long long)(int)a.l[j] * (long long)(int)b.l[k - j];
      acc += (long long)m[j] * (long long)(int)P[k - j];
    }
    acc += (long long)(int)a.l[k] * (long long)(int)b.l[0];
    m[k] = (int)(((u32)acc * r29::INV) & q29::MASK);
    acc += (long long)m[k] * (long long)(int)P[0];
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int j = k - 8; j < 9; ++j) {
      acc += (long long)(int)a.l[j] * (long long)(int)b.l[k - j];
      acc += (long long)m[j] * (long long)(int)P[k - j];
    }
    r.l[k - 9] = (int)((u32)acc & q29::MASK);
    acc >>= 29;
  }
  r.l[8] = (int)acc;
  return r;
}
template <class A, class W, class B, class V>
__device__ __forceinline__ LzT mul2_g(const A &a, const W &w, const B &b, const V &v) {
  constexpr u32 P[9] = ZK_R29_P;
  int m[9];
  LzT r;
  long long acc = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int j = 0; j < k; ++j) {
      acc += (long long)(int)a.l[j] * (long long)(int)w.l[k - j];
      acc += (long long)(int)b.l[j] * (long long)(int)v.l[k - j];
      acc += (long long)m[j] * (long long)(int)P[k - j];
    }
    acc += (long long)(int)a.l[k] * (long long)(int)w.l[0];
    acc += (long long)(int)b.l[k] * (long long)(int)v.l[0];
    m[k] = (int)(((u32)acc * r29::INV) & q29::MASK);
    acc += (long long)m[k] * (long long)(int)P[0];
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int j = k - 8; j < 9; ++j) {
      acc += (long long)(int)a.l[j] * (long long)(int)w.l[k - j];
      acc += (long long)(int)b.l[j] * (long long)(int)v.l[k - j];
      acc += (long long)m[j] * (long long)(int)P[k - j];
    }
    r.l[k - 9] = (int)((u32)acc & q29::MASK);
    acc >>= 29;
  }
  r.l[8] = (int)acc;
  return r;
}
#define ZK_F __builtin_amdgcn_sched_barrier(0);

struct Args {
  const Fr *adv, *fix, *sig, *pz, *xs, *lact;
  Fr *out;
  Fr y, beta, gamma;
  const Fr *beta_delta;
  unsigned log_n, rows, per_group;
};

// ---- today's arithmetic ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gate_std(Args a) {
  const size_t n = (size_t)1 << a.log_n, ne = n * a.rows;
  const size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (p >= ne) return;
  const size_t row0 = p & ~(n - 1), k2 = p & (n - 1);
  auto at = [&](const Fr *base, unsigned col, unsigned rot) -> Fr { return base[(size_t)col * ne + row0 + ((k2 + rot) & (n - 1))]; };
  Fr acc = Fr::zero();
  const int first = blockIdx.y * a.per_group;
  for (int j = first; j < first + (int)a.per_group; ++j) {
    const Fr q = at(a.fix, j, 0);
    if (!q.is_zero()) acc = fp_mul2<FrP>(acc, a.y, q, at(a.adv, j, 0) + at(a.adv, j, 1) * at(a.adv, j, 2) - at(a.adv, j, 3));
    else acc = acc * a.y;
  }
  a.out[(size_t)blockIdx.y * ne + p] = acc;
}
__global__ void __launch_bounds__(256) k_perm_std(Args a) {
  const size_t n = (size_t)1 << a.log_n, ne = n * a.rows;
  const size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (p >= ne) return;
  const size_t row0 = p & ~(n - 1), k2 = p & (n - 1);
  auto at = [&](const Fr *base, unsigned col, unsigned rot) -> Fr { return base[(size_t)col * ne + row0 + ((k2 + rot) & (n - 1))]; };
  Fr acc = Fr::zero();
  const Fr lact = a.lact[p], x = a.xs[p];
  const int first = blockIdx.y * a.per_group;
  for (int j = first; j < first + (int)a.per_group; ++j) {
    Fr left = at(a.pz, j, 1), right = at(a.pz, j, 0);
    for (unsigned c = 2 * j; c < 2 * j + 2; ++c) {
      const Fr v = at(a.adv, c, 0);
      left = left * (v + a.beta * at(a.sig, c, 0) + a.gamma);
      right = right * (v + a.beta_delta[c] * x + a.gamma);
    }
    acc = fp_mul2<FrP>(acc, a.y, lact, left - right);
  }
  a.out[(size_t)blockIdx.y * ne + p] = acc;
}

// ---- today's arithmetic, ONE load of the advice column per thread: the rotations X w, X w^2, X w^3 of the gate come from the lanes to the
// right (wave-wide rotate by one lane, v_mov_b32 dpp wave_rol:1), the last three lanes of the wave from three extra values the first three
// lanes load.  5 load instructions per gate -> 2 (+ one of three active lanes).
__device__ __forceinline__ Fr rol1(const Fr &v) {
  Fr r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.l[i] = (u32)__builtin_amdgcn_update_dpp(0, (int)v.l[i], 0x134 /* wave_rol:1 */, 0xf, 0xf, false);
  return r;
}
__device__ __forceinline__ Fr pick(bool c, const Fr &a, const Fr &b) {
  Fr r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.l[i] = c ? a.l[i] : b.l[i];
  return r;
}
__global__ void __launch_bounds__(256) k_gate_dpp(Args a) {
  const size_t n = (size_t)1 << a.log_n, ne = n * a.rows;
  const size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;   // ne is a multiple of 256: every lane is live
  const size_t row0 = p & ~(n - 1), k2 = p & (n - 1);
  const unsigned lane = threadIdx.x & 63u;
  Fr acc = Fr::zero();
  const int first = blockIdx.y * a.per_group;
  for (int j = first; j < first + (int)a.per_group; ++j) {
    const Fr *col = a.adv + (size_t)j * ne + row0;
    const Fr q = a.fix[(size_t)j * ne + p];
    const Fr a0 = col[k2];
    Fr ex = Fr::zero();
    if (lane < 3) ex = col[(k2 + 64) & (n - 1)];          // lane t < 3: the value 64 rows further (wraps inside the coset row)
    const Fr A1 = rol1(a0), E1 = rol1(ex);
    const Fr A2 = rol1(A1), E2 = rol1(E1);
    const Fr A3 = rol1(A2), E3 = rol1(E2);
    const Fr r1 = pick(lane >= 63, E1, A1), r2 = pick(lane >= 62, E2, A2), r3 = pick(lane >= 61, E3, A3);
    if (!q.is_zero()) acc = fp_mul2<FrP>(acc, a.y, q, a0 + r1 * r2 - r3);
    else acc = acc * a.y;
  }
  a.out[(size_t)blockIdx.y * ne + p] = acc;
}

// ---- nine lazy limbs, data and constants in the 2^261 form ---------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gate_lz(Args a) {
  const size_t n = (size_t)1 << a.log_n, ne = n * a.rows;
  const size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (p >= ne) return;
  const size_t row0 = p & ~(n - 1), k2 = p & (n - 1);
  auto at = [&](const Fr *base, unsigned col, unsigned rot) { return lz_load(base[(size_t)col * ne + row0 + ((k2 + rot) & (n - 1))]); };
  const Lw y = lw_unpack(a.y);
  LzT acc;
#pragma unroll
  for (int i = 0; i < 9; ++i) acc.l[i] = 0;
  const int first = blockIdx.y * a.per_group;
  for (int j = first; j < first + (int)a.per_group; ++j) {
    const auto q = at(a.fix, j, 0);
    const auto a0 = at(a.adv, j, 0), a1 = at(a.adv, j, 1), a2 = at(a.adv, j, 2), a3 = at(a.adv, j, 3);
    ZK_F
    const LzT bc = mul_g(a1, a2);                                   // (-r, 2 r)
    ZK_F
    const auto e = lz_norm(lz_add(lz_sub(a0, a3), bc));             // |e| < 4 r, tight limbs
    ZK_F
    acc = mul2_g(acc, y, e, q);                                     // acc y + e q: |.| < (2 + 4) r r
    ZK_F
  }
  a.out[(size_t)blockIdx.y * ne + p] = lz_store(acc);
}
__global__ void __launch_bounds__(256) k_perm_lz(Args a) {
  const size_t n = (size_t)1 << a.log_n, ne = n * a.rows;
  const size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (p >= ne) return;
  const size_t row0 = p & ~(n - 1), k2 = p & (n - 1);
  auto at = [&](const Fr *base, unsigned col, unsigned rot) { return lz_load(base[(size_t)col * ne + row0 + ((k2 + rot) & (n - 1))]); };
  const Lw y = lw_unpack(a.y), beta = lw_unpack(a.beta);
  const auto gamma = lz_load(a.gamma);
  const auto lact = lz_load(a.lact[p]), x = lz_load(a.xs[p]);
  LzT acc;
#pragma unroll
  for (int i = 0; i < 9; ++i) acc.l[i] = 0;
  const int first = blockIdx.y * a.per_group;
  for (int j = first; j < first + (int)a.per_group; ++j) {
    LzT left, right;
    {
      const auto l0 = at(a.pz, j, 1), r0 = at(a.pz, j, 0);
#pragma unroll
      for (int i = 0; i < 9; ++i) left.l[i] = l0.l[i], right.l[i] = r0.l[i];
    }
    for (unsigned c = 2 * j; c < 2 * j + 2; ++c) {
      const auto vg = lz_add(at(a.adv, c, 0), gamma);               // < 2 r, limbs (0, 2)
      const auto s = at(a.sig, c, 0);
      const Lw bd = lw_unpack(a.beta_delta[c]);
      ZK_F
      const auto fl = lz_norm(lz_add(vg, mul_g(s, beta)));          // v + gamma + beta sigma: |.| < 4 r
      ZK_F
      left = mul_g(left, fl);
      ZK_F
      const auto fr = lz_norm(lz_add(vg, mul_g(x, bd)));
      ZK_F
      right = mul_g(right, fr);
      ZK_F
    }
    const auto d = lz_norm(lz_sub(left, right));                    // |.| < 4 r
    ZK_F
    acc = mul2_g(acc, y, d, lact);
    ZK_F
  }
  a.out[(size_t)blockIdx.y * ne + p] = lz_store(acc);
}

template <class K>
static float time_it(K kern, dim3 grid, Args a, int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  kern<<<grid, 256>>>(a);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) kern<<<grid, 256>>>(a);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char **argv) {
  const unsigned log_n = argc > 1 ? (unsigned)atoi(argv[1]) : 13;
  const unsigned rows = 3, cols = log_n >= 17 ? 64 : 192, gate_groups = cols / 8, chunks = cols / 2, perm_groups = chunks / 4;
  const size_t n = (size_t)1 << log_n, ne = n * rows;
  Args a{};
  auto fill = [&](size_t count) {
    std::vector<uint32_t> h(count * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)rand() * 2654435761u;
    for (size_t i = 0; i < count; ++i) h[8 * i + 7] &= 0x0fffffffu;   // below r
    Fr *d;
    CHECK(hipMalloc((void **)&d, count * 32));
    CHECK(hipMemcpy(d, h.data(), count * 32, hipMemcpyHostToDevice));
    return d;
  };
  a.adv = fill(cols * ne);
  a.fix = fill(cols * ne);
  a.sig = fill(cols * ne);
  a.pz = fill(chunks * ne);
  a.xs = fill(ne);
  a.lact = fill(ne);
  a.beta_delta = fill(cols);
  CHECK(hipMalloc((void **)&a.out, (size_t)gate_groups * ne * 32));
  const Fr *k = fill(3);
  Fr hk[3];
  CHECK(hipMemcpy(hk, k, 96, hipMemcpyDeviceToHost));
  a.y = hk[0], a.beta = hk[1], a.gamma = hk[2];
  a.log_n = log_n, a.rows = rows;
  const unsigned bx = (unsigned)((ne + 255) / 256);
  a.per_group = 8;
  const float g_std = time_it(k_gate_std, dim3(bx, gate_groups), a, 20), g_lz = time_it(k_gate_lz, dim3(bx, gate_groups), a, 20);
  const float g_dpp = time_it(k_gate_dpp, dim3(bx, gate_groups), a, 20);
  {   // the rotated form must give the bytes of the plain one
    std::vector<uint32_t> h0((size_t)gate_groups * ne * 8), h1(h0.size());
    k_gate_std<<<dim3(bx, gate_groups), 256>>>(a);
    CHECK(hipMemcpy(h0.data(), a.out, h0.size() * 4, hipMemcpyDeviceToHost));
    k_gate_dpp<<<dim3(bx, gate_groups), 256>>>(a);
    CHECK(hipMemcpy(h1.data(), a.out, h1.size() * 4, hipMemcpyDeviceToHost));
    printf("gate, one load + lane rotations: %.3f ms (ratio to five loads %.3f), same output: %s\n", g_dpp, g_dpp / g_std, h0 == h1 ? "yes" : "NO");
  }
  a.per_group = 4;
  const float p_std = time_it(k_perm_std, dim3(bx, perm_groups), a, 20), p_lz = time_it(k_perm_lz, dim3(bx, perm_groups), a, 20);
  const double gates = (double)cols * ne, chunk_terms = (double)chunks * ne;
  printf("log_n %u, %u columns, %zu extended points\n", log_n, cols, ne);
  printf("gate   q (a + b c - d):  8x32 %.3f ms (%.2f G gates/s)   nine lazy limbs %.3f ms (%.2f G gates/s)   ratio %.3f\n", g_std, gates / g_std / 1e6, g_lz,
         gates / g_lz / 1e6, g_lz / g_std);
  printf("permutation term:        8x32 %.3f ms (%.2f G chunks/s)  nine lazy limbs %.3f ms (%.2f G chunks/s)  ratio %.3f\n", p_std, chunk_terms / p_std / 1e6, p_lz,
         chunk_terms / p_lz / 1e6, p_lz / p_std);
  return 0;
}
