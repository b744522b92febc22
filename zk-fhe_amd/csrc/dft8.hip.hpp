// The size-8 DFT butterfly of the NTT kernels on signed lazy limbs (lz29.hip.hpp), shared by the 2^13 tile (ntt13.hip) and the
// four-step passes of the longer rows (ntt_dif8.hip).  Same seam as the rest of the NTT code: halo2_proofs best_fft (third-party,
// reached from reference examples/bfv.rs:311).
#pragma once
#include "lz29.hip.hpp"

namespace zk {

struct Consts {
  Lw w4, w8, w83;
};

// one product at a time (see ntt_tile.hip.hpp mul_tw: the scheduler would interleave the independent products of a stage and
// run out of registers)
template <int LO, int HI, int V>
__device__ __forceinline__ LzT mulw(const Lz<LO, HI, V> &x, const Lw &w) {
  __builtin_amdgcn_sched_barrier(0);
  const LzT r = lz_mul(x, w);
  __builtin_amdgcn_sched_barrier(0);
  return r;
}
// the same against a constant that is uniform across the wave (the butterfly constants K, n^-1): its limbs stay in SGPRs
template <int LO, int HI, int V>
__device__ __forceinline__ LzT mulw_u(const Lz<LO, HI, V> &x, const Lw &w) {
  __builtin_amdgcn_sched_barrier(0);
  const LzT r = lz_mul<true>(x, w);
  __builtin_amdgcn_sched_barrier(0);
  return r;
}
__device__ __forceinline__ Lw tw_at(const LwMem *__restrict__ p) { return lw_load(*p); }

// The eight outputs of a size-8 DFT of x (natural order in and out), as typed values:
//   E = DFT4(x0, x2, x4, x6), O = DFT4(x1, x3, x5, x7), O_s *= w8^s, y_s = E_s + O_s, y_(s+4) = E_s - O_s
// Limb ranges in units of 2^29 (see lz29.hip.hpp) are in the comments; the five carry propagations are the lz_norm calls.
// ZK_F fences the instruction scheduler after every step: one wave's dependent multiply-add chain already issues back to back
// (0.34 us per product alone in a wave, profiles/r2b_microbench.md), so there is nothing to gain from interleaving steps and a
// lot to lose -- left alone the scheduler hoists the cheap additions of all eight outputs above the products and keeps three
// times the live values the source order needs.
#define ZK_F __builtin_amdgcn_sched_barrier(0);
#define ZK_DFT8_CORE(x, K)                                                                                           \
  const auto a0 = lz_add(x[0], x[4]);                        /* (0,2) */                                             \
  const auto a1 = lz_sub(x[0], x[4]);                        /* (1,1) */                                             \
  const auto b0 = lz_add(x[2], x[6]);                        /* (0,2) */                                             \
  ZK_F const LzT b1 = mulw_u(lz_sub(x[2], x[6]), K.w4);        /* (0,1) */                                             \
  ZK_F const auto E0 = lz_norm(lz_add(a0, b0));              /* (0,4) -> (0,1), |v| < 8 r */                         \
  ZK_F const auto E2 = lz_norm(lz_sub(a0, b0));              /* (2,2) -> (0,1) */                                    \
  ZK_F const auto E1 = lz_norm(lz_add(a1, b1));              /* (1,2) -> (0,1) */                                    \
  ZK_F const auto E3 = lz_norm(lz_sub(a1, b1));              /* (2,1) -> (0,1) */                                    \
  ZK_F const auto c0 = lz_add(x[1], x[5]);                                                                           \
  const auto c1 = lz_sub(x[1], x[5]);                                                                                \
  const auto d0 = lz_add(x[3], x[7]);                                                                                \
  ZK_F const LzT d1 = mulw_u(lz_sub(x[3], x[7]), K.w4);                                                                \
  ZK_F const auto O0 = lz_norm(lz_add(c0, d0));              /* (0,1), |v| < 8 r */                                  \
  ZK_F const LzT O2 = mulw_u(lz_sub(c0, d0), K.w4);            /* (2,2) in */                                          \
  ZK_F const LzT O1 = mulw_u(lz_add(c1, d1), K.w8);            /* (1,2) in */                                          \
  ZK_F const LzT O3 = mulw_u(lz_sub(c1, d1), K.w83);           /* (2,1) in */                                          \
  ZK_F const auto y0 = lz_add(E0, O0);                       /* (0,2), |v| < 16 r */                                 \
  const auto y4 = lz_sub(E0, O0);                            /* (1,1) */                                             \
  ZK_F const auto y1 = lz_add(E1, O1);                                                                               \
  const auto y5 = lz_sub(E1, O1);                                                                                    \
  ZK_F const auto y2 = lz_add(E2, O2);                                                                               \
  const auto y6 = lz_sub(E2, O2);                                                                                    \
  ZK_F const auto y3 = lz_add(E3, O3);                                                                               \
  const auto y7 = lz_sub(E3, O3);                                                                                    \
  ZK_F


}  // namespace zk
