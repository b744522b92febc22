// Lazy radix-2^29 arithmetic over BN254 Fr with COMPILE-TIME bounds, for the butterflies of the NTT kernel (ntt13.hip).
//
// fr29.hip.hpp normalises the limbs after every addition and subtraction (9 adds + 24 carry operations, and a multiple of r
// so that a difference stays positive) although the nine-limb product does not need normalised inputs.  Here a value is nine
// SIGNED 32-bit limbs, value = sum l[i] 2^(29 i), of either sign:
//   * addition and subtraction are nine v_add / v_sub, nothing else;
//   * the Montgomery product takes a signed first operand (v_mad_i64_i32) and a canonical constant w 2^261 (a twiddle, limbs in
//     [0, 2^29)): a column is nine a_j w_(k-j) below 2^30 2^29 in magnitude plus nine m_j r_(k-j) below 2^58, inside the signed
//     64-bit accumulator; the result has limbs 0..7 in [0, 2^29) and a signed top limb, value in (-r, 2 r);
//   * carries are propagated (24 operations) only where a bound below would otherwise be exceeded.
// Lz<LO, HI, V>: -LO 2^29 < l[i] < HI 2^29 for i < 8 (LO = 0: non-negative) and |value| < V r; l[8] is whatever is left.
// Every operation states its result bound in its return type and static_asserts what it needs, so a butterfly network that
// compiles cannot overflow:
//   add(a, b)      -> Lz<LOa + LOb, HIa + HIb, Va + Vb>     limbs must stay inside int32: LO, HI <= 4
//   sub(a, b)      -> Lz<LOa + HIb, HIa + LOb, Va + Vb>
//   mul(a, w)      -> Lz<0, 1, 2>               needs LO, HI <= 2 (limbs below 2^30 in magnitude), V <= 160
//   mul2(a,w,b,v)  -> Lz<0, 1, 2>               (a w + b v) / 2^261 with one reduction; needs tight a, b
//   norm(a)        -> Lz<0, 1, V>               carry propagation: limbs 0..7 back in [0, 2^29)
//   weak(a)        -> Lz<0, 1, 2>               value into [0, 2 r): needs V <= 16
// A twiddle is an Lw: nine canonical limbs of w * 2^261 mod r, kept UNPACKED in memory (48-byte entries: three dwordx4 loads
// instead of two loads and 27 shift/mask operations per product).
#pragma once
#include "fr29.hip.hpp"

namespace zk {

template <int LO, int HI, int V>
struct Lz {
  int l[9];
};
using LzT = Lz<0, 1, 2>;   // what a product returns: tight limbs, value in (-r, 2 r)
struct LzW : LzT {};       // what lz_weak returns: the same limbs, value in [0, 2 r) -- the only thing lz_store_weak accepts
struct Lw {   // canonical constant operand (twiddle), limbs in [0, 2^29)
  u32 l[9];
};

template <int V>
ZK_HD Lz<0, 1, V> lz_from_f29(const F29 &a) {   // caller's promise: a has tight limbs and a value below V r
  Lz<0, 1, V> r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = (int)a.l[i];
  return r;
}
// a canonical column value (x * 2^256 < r, packed)
ZK_HD Lz<0, 1, 1> lz_load(const Fr &w) { return lz_from_f29<1>(fr29_unpack(w)); }
ZK_HD Lz<0, 1, 1> lz_zero() { return lz_from_f29<1>(f29_zero()); }

template <int L1, int H1, int V1, int L2, int H2, int V2>
ZK_HD Lz<L1 + L2, H1 + H2, V1 + V2> lz_add(const Lz<L1, H1, V1> &a, const Lz<L2, H2, V2> &b) {
  static_assert(L1 + L2 <= 4 && H1 + H2 <= 4, "limb overflow");
  Lz<L1 + L2, H1 + H2, V1 + V2> r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
  return r;
}
template <int L1, int H1, int V1, int L2, int H2, int V2>
ZK_HD Lz<L1 + H2, H1 + L2, V1 + V2> lz_sub(const Lz<L1, H1, V1> &a, const Lz<L2, H2, V2> &b) {
  static_assert(L1 + H2 <= 4 && H1 + L2 <= 4, "limb overflow");
  Lz<L1 + H2, H1 + L2, V1 + V2> r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] - b.l[i];
  return r;
}

// A limb typed HI = h is a sum of h terms each at most 2^29 - 1, so l + carry (carry <= 3) stays inside int32 for h = 4;
// likewise on the negative side.
template <int LO, int HI, int V>
ZK_HD Lz<0, 1, V> lz_norm(const Lz<LO, HI, V> &a) {
  Lz<0, 1, V> r;
  int c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int v = a.l[i] + c;
    r.l[i] = v & (int)q29::MASK;
    c = v >> 29;                // arithmetic: floor
  }
  r.l[8] = a.l[8] + c;
  return r;
}

// value (either sign, |v| < 16 r) -> the same residue in [0, 2 r) (in fact below 1.04 r), limbs normalised.
// q = floor(t m / 2^16) with t = floor((l[8] - LO) / 2^13) and m = 169 for t >= 0, 170 for t < 0.  The lower limbs sum to more than
// -LO 2^232 (1 + 2^-29), so t 2^245 <= v + LO 2^203: not quite t 2^245 <= v, but the sliver (below 2^205 = 2^-48 r) is covered
// by the rounding of the multiplier -- 169 < 2^261 / r = 169.29.. < 170, i.e. the estimate is 0.17 % (t > 0) resp. 0.42 %
// (t < 0) of |t| 2^245 >= 2^245 on the safe side, and t = 0 gives q = 0 -- so q never exceeds v / r and falls short of it by
// less than 1.1.
template <int LO, int HI, int V>
ZK_HD LzW lz_weak(const Lz<LO, HI, V> &a) {
  static_assert(V <= 16, "weak reduction: |value| below 16 r");
  constexpr u32 P[9] = ZK_R29_P;
  const int t = (a.l[8] - LO) >> 13;
  const int q = (t * (169 - (t >> 31))) >> 16;
  LzW r;
  long long c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const long long v = (long long)a.l[i] - (long long)q * (long long)P[i] + c;
    r.l[i] = (int)((u32)v & q29::MASK);
    c = v >> 29;
  }
  r.l[8] = (int)((long long)a.l[8] - (long long)q * (long long)P[8] + c);
  return r;
}

#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_MAD_C)
#include "lz29_tied.inc"
#endif
// Montgomery product a * w / 2^261 mod r for |a w| < 2^261 r: value in (-r, 2 r).  Column k: nine |a_j| w_(k-j) < 2^30 2^29,
// nine m_j r_(k-j) < 2^58 and the carry: magnitude below 9 2^59 + 9 2^58 + 2^35 < 2^63.
// UNIFORM: the constant is the same for every lane of the wave (a butterfly constant, n^-1 -- loaded through a uniform address): its
// limbs are taken from SGPRs, as the compiler did on its own before the multiply-adds became inline assembly (fq29.hip.hpp zk_madi).
template <bool UNIFORM = false, int LO, int HI, int V>
ZK_HD LzT lz_mul(const Lz<LO, HI, V> &a, const Lw &b) {
  static_assert(LO <= 2 && HI <= 2, "product: limbs below 2^30 in magnitude");
  static_assert(V <= 160, "product: |a w| < 2^261 r");
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_MAD_C)
  if (UNIFORM) return lz_mul_tied_s(a, b);
  return lz_mul_tied_v(a, b);
#endif
  constexpr u32 P[9] = ZK_R29_P;
  int m[9];
  LzT r;
  long long acc = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int j = 0; j < k; ++j) {
      if (UNIFORM) zk_madi_s(acc, a.l[j], (int)b.l[k - j]); else zk_madi(acc, a.l[j], (int)b.l[k - j]);
      zk_madi_s(acc, m[j], (int)P[k - j]);
    }
    if (UNIFORM) zk_madi_s(acc, a.l[k], (int)b.l[0]); else zk_madi(acc, a.l[k], (int)b.l[0]);
    m[k] = (int)(((u32)acc * r29::INV) & q29::MASK);
    zk_madi_s(acc, m[k], (int)P[0]);
    acc >>= 29;   // exact: the low 29 bits are zero
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int j = k - 8; j < 9; ++j) {
      if (UNIFORM) zk_madi_s(acc, a.l[j], (int)b.l[k - j]); else zk_madi(acc, a.l[j], (int)b.l[k - j]);
      zk_madi_s(acc, m[j], (int)P[k - j]);
    }
    r.l[k - 9] = (int)((u32)acc & q29::MASK);
    acc >>= 29;
  }
  r.l[8] = (int)acc;
  return r;
}

// (a w + b v) / 2^261 mod r, one reduction: eighteen products below 2^29 2^29 in magnitude per column (limbs of either sign below
// 2^29: LO, HI <= 1) and nine m_j r_(k-j) below 2^58: 27 2^58 < 2^63
template <int L1, int H1, int V, int L2, int H2, int V2>
ZK_HD LzT lz_mul2(const Lz<L1, H1, V> &a, const Lw &w, const Lz<L2, H2, V2> &b, const Lw &v) {
  static_assert(L1 <= 1 && H1 <= 1 && L2 <= 1 && H2 <= 1, "two-product form: limbs below 2^29 in magnitude");
  static_assert(V + V2 <= 160, "two-product form: |a w + b v| < 2^261 r");
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_MAD_C)
  return lz_mul2_tied(a, w, b, v);
#endif
  constexpr u32 P[9] = ZK_R29_P;
  int m[9];
  LzT r;
  long long acc = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int j = 0; j < k; ++j) {
      zk_madi(acc, a.l[j], (int)w.l[k - j]);
      zk_madi(acc, b.l[j], (int)v.l[k - j]);
      zk_madi_s(acc, m[j], (int)P[k - j]);
    }
    zk_madi(acc, a.l[k], (int)w.l[0]);
    zk_madi(acc, b.l[k], (int)v.l[0]);
    m[k] = (int)(((u32)acc * r29::INV) & q29::MASK);
    zk_madi_s(acc, m[k], (int)P[0]);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int j = k - 8; j < 9; ++j) {
      zk_madi(acc, a.l[j], (int)w.l[k - j]);
      zk_madi(acc, b.l[j], (int)v.l[k - j]);
      zk_madi_s(acc, m[j], (int)P[k - j]);
    }
    r.l[k - 9] = (int)((u32)acc & q29::MASK);
    acc >>= 29;
  }
  r.l[8] = (int)acc;
  return r;
}

// (a0 w0 + a1 w1 + a2 w2 + a3 w3) / 2^261 mod r with ONE reduction, for canonical data (tight non-negative limbs, values below r)
// against canonical constants: a column is 36 products below 2^58 and nine m_j r_(k-j) below 2^58 -- 45 2^58 < 2^64, an UNSIGNED
// accumulator (v_mad_u64_u32).  The value is below 4 r r / 2^261 + r < 2 r.  (The radix-4 first stage of the quarter-column 2^13
// tile with a coset pre-multiplier: four table products per output for the price of 2.5.)
ZK_HD LzT lz_mul4u(const Lz<0, 1, 1> &a0, const Lw &w0, const Lz<0, 1, 1> &a1, const Lw &w1, const Lz<0, 1, 1> &a2, const Lw &w2, const Lz<0, 1, 1> &a3, const Lw &w3) {
  constexpr u32 P[9] = ZK_R29_P;
  u32 m[9];
  LzT r;
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int j = 0; j < k; ++j) {
      zk_madu(acc, (u32)a0.l[j], w0.l[k - j]);
      zk_madu(acc, (u32)a1.l[j], w1.l[k - j]);
      zk_madu(acc, (u32)a2.l[j], w2.l[k - j]);
      zk_madu(acc, (u32)a3.l[j], w3.l[k - j]);
      zk_madu_s(acc, m[j], P[k - j]);
    }
    zk_madu(acc, (u32)a0.l[k], w0.l[0]);
    zk_madu(acc, (u32)a1.l[k], w1.l[0]);
    zk_madu(acc, (u32)a2.l[k], w2.l[0]);
    zk_madu(acc, (u32)a3.l[k], w3.l[0]);
    m[k] = ((u32)acc * r29::INV) & q29::MASK;
    zk_madu_s(acc, m[k], P[0]);
    acc >>= 29;   // exact: the low 29 bits are zero
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int j = k - 8; j < 9; ++j) {
      zk_madu(acc, (u32)a0.l[j], w0.l[k - j]);
      zk_madu(acc, (u32)a1.l[j], w1.l[k - j]);
      zk_madu(acc, (u32)a2.l[j], w2.l[k - j]);
      zk_madu(acc, (u32)a3.l[j], w3.l[k - j]);
      zk_madu_s(acc, m[j], P[k - j]);
    }
    r.l[k - 9] = (int)((u32)acc & q29::MASK);
    acc >>= 29;
  }
  r.l[8] = (int)acc;
  return r;
}

// canonical packed column value from a weakly reduced one (value in [0, 2 r), normalised limbs): takes lz_weak's own type, so a
// product's result (which may be negative) cannot be passed by mistake
ZK_HD Fr lz_store_weak(const LzW &a) {
  constexpr u32 P[9] = ZK_R29_P;
  int t[9], c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int v = a.l[i] - (int)P[i] + c;
    t[i] = v & (int)q29::MASK;
    c = v >> 29;
  }
  t[8] = a.l[8] - (int)P[8] + c;
  F29 o;
#pragma unroll
  for (int i = 0; i < 9; ++i) o.l[i] = (u32)(t[8] >= 0 ? t[i] : a.l[i]);   // a - r when that is not negative
  return fr29_pack(o);
}
// any bounded value -> canonical packed
template <int LO, int HI, int V>
ZK_HD Fr lz_store(const Lz<LO, HI, V> &a) {
  return lz_store_weak(lz_weak(a));
}

// twiddle entry in memory: 12 dwords (nine limbs + padding), 16-byte aligned
struct alignas(16) LwMem {
  u32 l[12];
};
ZK_HD Lw lw_load(const LwMem &e) {
  Lw w;
#pragma unroll
  for (int i = 0; i < 9; ++i) w.l[i] = e.l[i];
  return w;
}
ZK_HD Lw lw_unpack(const Fr &w29 /* canonical, 2^261 form */) {
  const F29 u = fr29_unpack(w29);
  Lw w;
#pragma unroll
  for (int i = 0; i < 9; ++i) w.l[i] = u.l[i];
  return w;
}
ZK_HD LwMem lw_from_packed(const Fr &w29 /* canonical, 2^261 form */) {
  const F29 u = fr29_unpack(w29);
  LwMem e;
#pragma unroll
  for (int i = 0; i < 9; ++i) e.l[i] = u.l[i];
  e.l[9] = e.l[10] = e.l[11] = 0;
  return e;
}

}  // namespace zk
