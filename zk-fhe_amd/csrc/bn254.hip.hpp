// BN254 field and curve arithmetic for gfx950 (host + device).
//
// Replaces, for the GPU hot path, what the reference gets from the un-vendored
// halo2curves crate (`bn256::{Fr,Fq,G1Affine,G1}`; reference call sites
// src/poly_chip.rs:5-10,90 and examples/bfv.rs:1-6 -- SURVEY.md section 8a/8b):
//   * Fr / Fq: 256-bit Montgomery residues, R = 2^256, canonical (< p) at rest.
//     ABI layout = 4 x uint64 little-endian limbs == 8 x uint32 little-endian limbs.
//   * G1: y^2 = x^3 + 3 over Fq.  Affine {x,y} (identity = (0,0)), and the XYZZ
//     extended-Jacobian form used for accumulation (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2).
//
// Device arithmetic is on 32-bit limbs: gfx950 has no 64x64 multiplier; the widest
// integer multiply is v_mad_u64_u32 (32x32+64 -> 64), which fp_mul below is built from.
#pragma once
#include <stdint.h>

#include <vector>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZK_HD __host__ __device__ __forceinline__
#else
#define ZK_HD inline
#endif

namespace zk {

typedef uint32_t u32;
typedef uint64_t u64;

struct FrP {
  static constexpr u32 MOD[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr u32 ONE[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};  // R mod p
  static constexpr u32 R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
  static constexpr u32 R3[8] = {0xb4bf0040u, 0x5e94d8e1u, 0x1cfbb6b8u, 0x2a489cbeu,
                                0xa19fcfedu, 0x893cc664u, 0x7fcc657cu, 0x0cf8594bu};  // R^3 mod p
  static constexpr u32 INV = 0xefffffffu;  // -p^-1 mod 2^32
};

struct FqP {
  static constexpr u32 MOD[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr u32 ONE[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr u32 R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
  static constexpr u32 R3[8] = {0xda1530dfu, 0xb1cd6dafu, 0xa7283db6u, 0x62f210e6u,
                                0x0ada0afbu, 0xef7f0b0cu, 0x2d592544u, 0x20fd6e90u};
  static constexpr u32 INV = 0xe4866389u;
};

template <class P>
struct alignas(16) Fp {
  u32 l[8];

  static ZK_HD Fp zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = 0;
    return r;
  }
  static ZK_HD Fp one() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = P::ONE[i];
    return r;
  }
  static ZK_HD Fp r2() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = P::R2[i];
    return r;
  }
  ZK_HD bool is_zero() const {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= l[i];
    return o == 0;
  }
  ZK_HD bool operator==(const Fp& b) const {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= l[i] ^ b.l[i];
    return o == 0;
  }
  ZK_HD bool operator!=(const Fp& b) const { return !(*this == b); }
};

// Carry-chain helpers.  __builtin_addc/__builtin_subc lower to v_add_co_u32 / v_addc_co_u32
// (v_sub_co / v_subb_co) chains on gfx950 -- one VALU op per limb, carries in VCC -- instead of the
// 64-bit-per-limb arithmetic a (u64) formulation produces.
ZK_HD u32 zk_addc(u32 a, u32 b, u32 cin, u32 *cout) { return __builtin_addc(a, b, cin, cout); }
ZK_HD u32 zk_subc(u32 a, u32 b, u32 bin, u32 *bout) { return __builtin_subc(a, b, bin, bout); }

// r = a - p if a >= p (a < 2p).
template <class P>
ZK_HD void fp_reduce_once(u32 (&a)[8]) {
  u32 t[8];
  u32 br = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = zk_subc(a[i], P::MOD[i], br, &br);
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = br ? a[i] : t[i];
}

template <class P>
ZK_HD Fp<P> fp_add(const Fp<P>& a, const Fp<P>& b) {
  Fp<P> r;
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.l[i] = zk_addc(a.l[i], b.l[i], c, &c);
  // p < 2^254 so a+b < 2^255: no carry out of limb 7.
  fp_reduce_once<P>(r.l);
  return r;
}

template <class P>
ZK_HD Fp<P> fp_sub(const Fp<P>& a, const Fp<P>& b) {
  Fp<P> r;
  u32 br = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.l[i] = zk_subc(a.l[i], b.l[i], br, &br);
  const u32 mask = (u32)0 - br;  // all ones when a < b: add p back
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.l[i] = zk_addc(r.l[i], P::MOD[i] & mask, c, &c);
  return r;
}

template <class P>
ZK_HD Fp<P> fp_neg(const Fp<P>& a) {
  if (a.is_zero()) return a;
  Fp<P> r;
  u32 br = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.l[i] = zk_subc(P::MOD[i], a.l[i], br, &br);
  return r;
}

template <class P>
ZK_HD Fp<P> fp_dbl(const Fp<P>& a) {
  return fp_add<P>(a, a);
}

// Montgomery product a*b*R^-1 mod p.
//
// Device: finely-integrated product scanning (FIPS) over 32-bit limbs.  Column k of a*b + m*p is
// summed into a 96-bit accumulator {acc (64 bit), top (32 bit)}: every partial product is ONE
// v_mad_u64_u32 (32x32+64 -> 64, carry-out in VCC) followed by ONE v_addc_co_u32 into `top`.
// 128 multiply-adds + 8 v_mul_lo_u32 per product, ~30 live VGPRs.  The modulus limbs are SGPR
// operands (VOP3 on gfx9-family cannot encode a 32-bit literal).
// Host: portable CIOS with the same result (both are exact, result canonical < p).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void zk_mac(u64 &acc, u32 &top, u32 a, u32 b) {
  // gfx950: a VALU write of VCC needs 2 wait states before a VALU reads it as carry-in; hipcc does
  // not pad inside an asm string, so the s_nop is ours.
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(acc), "+v"(top)
      : "v"(a), "v"(b)
      : "vcc");
}
__device__ __forceinline__ void zk_mac_s(u64 &acc, u32 &top, u32 a, u32 b_sgpr) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(acc), "+v"(top)
      : "v"(a), "s"(b_sgpr)
      : "vcc");
}

#endif

template <class P>
ZK_HD Fp<P> fp_mul(const Fp<P>& a, const Fp<P>& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  u32 m[8];
  u32 r[8];
  u64 acc = 0;
  u32 top = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
#pragma unroll
    for (int j = 0; j < k; ++j) {
      zk_mac(acc, top, a.l[j], b.l[k - j]);
      zk_mac_s(acc, top, m[j], P::MOD[k - j]);
    }
    zk_mac(acc, top, a.l[k], b.l[0]);
    m[k] = (u32)acc * P::INV;
    zk_mac_s(acc, top, m[k], P::MOD[0]);  // low word becomes 0
    acc = (acc >> 32) | ((u64)top << 32);
    top = 0;
  }
#pragma unroll
  for (int k = 8; k < 16; ++k) {
#pragma unroll
    for (int j = k - 7; j < 8; ++j) {
      zk_mac(acc, top, a.l[j], b.l[k - j]);
      zk_mac_s(acc, top, m[j], P::MOD[k - j]);
    }
    r[k - 8] = (u32)acc;
    acc = (acc >> 32) | ((u64)top << 32);
    top = 0;
  }
  // a, b < p < 2^254  =>  result < 2p < 2^255: nothing left in acc
  Fp<P> o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o.l[i] = r[i];
  fp_reduce_once<P>(o.l);
  return o;
#else
  // host: CIOS on four 64-bit limbs (the 32-bit limb array read as little-endian u64 pairs), fully unrolled by the compiler;
  // p < 2^254, so the running value fits four limbs and the carry word
  typedef unsigned __int128 u128;
  u64 A[4], B[4], Pm[4];
  for (int i = 0; i < 4; ++i) {
    A[i] = (u64)a.l[2 * i] | ((u64)a.l[2 * i + 1] << 32);
    B[i] = (u64)b.l[2 * i] | ((u64)b.l[2 * i + 1] << 32);
    Pm[i] = (u64)P::MOD[2 * i] | ((u64)P::MOD[2 * i + 1] << 32);
  }
  // -p^-1 mod 2^64 from the 32-bit constant: one Newton step (x <- x (2 + p x), x = -p^-1 mod 2^32 doubles its valid bits)
  const u64 inv32 = (u64)P::INV;
  const u64 inv = inv32 * (2 + Pm[0] * inv32);
  u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0;
  for (int i = 0; i < 4; ++i) {
    u128 c;
    const u64 bi = B[i];
    c = (u128)A[0] * bi + t0; t0 = (u64)c;
    c = (u128)A[1] * bi + t1 + (u64)(c >> 64); t1 = (u64)c;
    c = (u128)A[2] * bi + t2 + (u64)(c >> 64); t2 = (u64)c;
    c = (u128)A[3] * bi + t3 + (u64)(c >> 64); t3 = (u64)c;
    const u64 t4 = (u64)(c >> 64);
    const u64 m = t0 * inv;
    c = (u128)m * Pm[0] + t0;
    c = (u128)m * Pm[1] + t1 + (u64)(c >> 64); t0 = (u64)c;
    c = (u128)m * Pm[2] + t2 + (u64)(c >> 64); t1 = (u64)c;
    c = (u128)m * Pm[3] + t3 + (u64)(c >> 64); t2 = (u64)c;
    t3 = t4 + (u64)(c >> 64);
  }
  Fp<P> r;
  r.l[0] = (u32)t0, r.l[1] = (u32)(t0 >> 32), r.l[2] = (u32)t1, r.l[3] = (u32)(t1 >> 32);
  r.l[4] = (u32)t2, r.l[5] = (u32)(t2 >> 32), r.l[6] = (u32)t3, r.l[7] = (u32)(t3 >> 32);
  fp_reduce_once<P>(r.l);
  return r;
#endif
}

template <class P>
ZK_HD Fp<P> fp_sqr(const Fp<P>& a) {
  return fp_mul<P>(a, a);
}

// (a*b + c*d) * R^-1 mod p with ONE Montgomery reduction: 128 product multiply-adds + 64 for the reduction instead of the
// 256 of two products (the Y coordinate of every point addition / doubling is such a sum).  (ab + cd)/R + p < 1.4 p.
template <class P>
ZK_HD Fp<P> fp_mul2(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d) {
#if defined(__HIP_DEVICE_COMPILE__)
  u32 m[8];
  u32 r[8];
  u64 acc = 0;
  u32 top = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
#pragma unroll
    for (int j = 0; j < k; ++j) {
      zk_mac(acc, top, a.l[j], b.l[k - j]);
      zk_mac(acc, top, c.l[j], d.l[k - j]);
      zk_mac_s(acc, top, m[j], P::MOD[k - j]);
    }
    zk_mac(acc, top, a.l[k], b.l[0]);
    zk_mac(acc, top, c.l[k], d.l[0]);
    m[k] = (u32)acc * P::INV;
    zk_mac_s(acc, top, m[k], P::MOD[0]);  // low word becomes 0
    acc = (acc >> 32) | ((u64)top << 32);
    top = 0;
  }
#pragma unroll
  for (int k = 8; k < 16; ++k) {
#pragma unroll
    for (int j = k - 7; j < 8; ++j) {
      zk_mac(acc, top, a.l[j], b.l[k - j]);
      zk_mac(acc, top, c.l[j], d.l[k - j]);
      zk_mac_s(acc, top, m[j], P::MOD[k - j]);
    }
    r[k - 8] = (u32)acc;
    acc = (acc >> 32) | ((u64)top << 32);
    top = 0;
  }
  Fp<P> o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o.l[i] = r[i];
  fp_reduce_once<P>(o.l);
  return o;
#else
  return fp_add<P>(fp_mul<P>(a, b), fp_mul<P>(c, d));
#endif
}

// Montgomery form <-> canonical integer.
template <class P>
ZK_HD Fp<P> fp_to_mont(const Fp<P>& a) {
  return fp_mul<P>(a, Fp<P>::r2());
}
template <class P>
ZK_HD Fp<P> fp_from_mont(const Fp<P>& a) {
  Fp<P> o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o.l[i] = (i == 0) ? 1u : 0u;
  return fp_mul<P>(a, o);
}

// a^e for a 256-bit exponent given as 8 little-endian u32 words (not secret: vartime).
template <class P>
ZK_HD Fp<P> fp_pow(const Fp<P>& a, const u32 (&e)[8]) {
  Fp<P> r = Fp<P>::one();
  bool started = false;
  for (int i = 255; i >= 0; --i) {
    if (started) r = fp_sqr<P>(r);
    if ((e[i >> 5] >> (i & 31)) & 1) {
      r = started ? fp_mul<P>(r, a) : a;
      started = true;
    }
  }
  return r;
}

// a^-1, 0 -> 0 (the halo2 `invert().unwrap_or(0)` convention used by batch_invert).
// Bernstein-Yang "safegcd" division steps (half-delta variant: 590 steps suffice for a 256-bit modulus; 20 batches
// of 30 are run), on signed 30-bit limbs: each batch runs 30 branch-free steps on the low 32 bits of (f, g) only,
// collecting a 2x2 transition matrix with |entries| <= 2^30, and then applies the matrix once to the full (f, g)
// (exact division by 2^30) and to (d, e) (division by 2^30 modulo p).  ~10 k instructions in a straight line for all
// lanes -- the binary Euclid below needs ~4x that on random inputs and diverges per lane.  Applied to the Montgomery
// representative x = aR it yields a^-1 R^-1; one Montgomery product with R^3 brings that back to a^-1 R.
template <class P>
ZK_HD Fp<P> fp_inv(const Fp<P>& a) {
  if (a.is_zero()) return a;
  typedef int i32;
  typedef long long i64;
  const i32 M30 = 0x3fffffff;
  i32 m[9], f[9], g[9], d[9], e[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bit = 30 * i, w = bit >> 5, sh = bit & 31;
    u32 vm = P::MOD[w] >> sh, va = a.l[w] >> sh;
    if (sh > 2 && w + 1 < 8) {
      vm |= P::MOD[w + 1] << (32 - sh);
      va |= a.l[w + 1] << (32 - sh);
    }
    m[i] = (i32)(vm & (u32)M30);
    f[i] = m[i];
    g[i] = (i32)(va & (u32)M30);
    d[i] = 0;
    e[i] = i == 0 ? 1 : 0;
  }
  const u32 minv30 = (0u - P::INV) & (u32)M30;  // p^-1 mod 2^30   (P::INV = -p^-1 mod 2^32)
  i32 zeta = -1;                                 // -(delta + 1/2), delta = 1/2
  for (int batch = 0; batch < 20; ++batch) {
    // 30 division steps on the low limbs; (u v; q r) * (f, g) = 2^30 * (f', g')
    u32 u = 1, v = 0, q = 0, r = 1;
    u32 fl = (u32)f[0] | ((u32)f[1] << 30), gl = (u32)g[0] | ((u32)g[1] << 30);
#pragma unroll 6
    for (int i = 0; i < 30; ++i) {
      u32 c1 = (u32)(zeta >> 31);        // delta > 0
      const u32 c2 = 0u - (gl & 1u);     // g odd
      const u32 x = (fl ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;
      gl += x & c2;
      q += y & c2;
      r += z & c2;
      c1 &= c2;
      zeta = (i32)((u32)zeta ^ c1) - 1;
      fl += gl & c1;
      u += q & c1;
      v += r & c1;
      gl >>= 1;
      u <<= 1;
      v <<= 1;
    }
    const i32 tu = (i32)u, tv = (i32)v, tq = (i32)q, tr = (i32)r;
    {  // (d, e) <- (t * (d, e) + p * (md, me)) / 2^30, with md, me chosen to clear the low 30 bits
      const i32 sd = d[8] >> 31, se = e[8] >> 31;
      i32 md = (tu & sd) + (tv & se), me = (tq & sd) + (tr & se);
      i64 cd = (i64)tu * d[0] + (i64)tv * e[0];
      i64 ce = (i64)tq * d[0] + (i64)tr * e[0];
      md -= (i32)((minv30 * (u32)cd + (u32)md) & (u32)M30);
      me -= (i32)((minv30 * (u32)ce + (u32)me) & (u32)M30);
      cd += (i64)m[0] * md;
      ce += (i64)m[0] * me;
      cd >>= 30;
      ce >>= 30;
#pragma unroll
      for (int i = 1; i < 9; ++i) {
        cd += (i64)tu * d[i] + (i64)tv * e[i] + (i64)m[i] * md;
        ce += (i64)tq * d[i] + (i64)tr * e[i] + (i64)m[i] * me;
        d[i - 1] = (i32)cd & M30;
        e[i - 1] = (i32)ce & M30;
        cd >>= 30;
        ce >>= 30;
      }
      d[8] = (i32)cd;
      e[8] = (i32)ce;
    }
    {  // (f, g) <- t * (f, g) / 2^30 (exact)
      i64 cf = (i64)tu * f[0] + (i64)tv * g[0];
      i64 cg = (i64)tq * f[0] + (i64)tr * g[0];
      cf >>= 30;
      cg >>= 30;
#pragma unroll
      for (int i = 1; i < 9; ++i) {
        cf += (i64)tu * f[i] + (i64)tv * g[i];
        cg += (i64)tq * f[i] + (i64)tr * g[i];
        f[i - 1] = (i32)cf & M30;
        g[i - 1] = (i32)cg & M30;
        cf >>= 30;
        cg >>= 30;
      }
      f[8] = (i32)cf;
      g[8] = (i32)cg;
    }
  }
  // now g = 0, f = +-1 and d = +-x^-1 in (-2p, p): fold the sign of f in and bring d to [0, p)
  {
    const i32 sf = f[8] >> 31;
    i32 neg = d[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      d[i] += m[i] & neg;
      d[i] = (d[i] ^ sf) - sf;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      d[i + 1] += d[i] >> 30;
      d[i] &= M30;
    }
    neg = d[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] += m[i] & neg;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      d[i + 1] += d[i] >> 30;
      d[i] &= M30;
    }
  }
  Fp<P> res, r3;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int bit = 32 * j, w = bit / 30, sh = bit % 30;
    u32 val = (u32)d[w] >> sh;
    if (w + 1 < 9) val |= (u32)d[w + 1] << (30 - sh);
    res.l[j] = val;
    r3.l[j] = P::R3[j];
  }
  return fp_mul<P>(res, r3);
}

// Reference inversion kept for cross-checks (tests compare fp_inv against it): a^-1, 0 -> 0.
// Binary extended Euclid on the Montgomery representative x = aR: it yields x^-1 = a^-1 R^-1, and one
// Montgomery product with R^3 brings that back to a^-1 R.  ~750 shift/add/sub rounds on 8 limbs -- an order
// of magnitude cheaper than the 381 Montgomery products of a Fermat ladder, which is what the latency of
// every normalisation (bucket reduction -> affine, batch inversion) used to be made of.
template <class P>
ZK_HD Fp<P> fp_inv_euclid(const Fp<P>& a) {
  if (a.is_zero()) return a;
  u32 u[8], v[8], x1[8], x2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    u[i] = a.l[i];
    v[i] = P::MOD[i];
    x1[i] = i == 0 ? 1u : 0u;
    x2[i] = 0u;
  }
  auto is_one = [](const u32(&t)[8]) {
    u32 o = t[0] ^ 1u;
#pragma unroll
    for (int i = 1; i < 8; ++i) o |= t[i];
    return o == 0;
  };
  auto shr1 = [](u32(&t)[8]) {
#pragma unroll
    for (int i = 0; i < 7; ++i) t[i] = (t[i] >> 1) | (t[i + 1] << 31);
    t[7] >>= 1;
  };
  auto halve_mod = [&](u32(&t)[8]) {  // t/2 mod p for t < p
    if (t[0] & 1u) {
      u32 c = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = zk_addc(t[i], P::MOD[i], c, &c);  // < 2^255: no carry out
    }
    shr1(t);
  };
  auto geq = [](const u32(&s)[8], const u32(&t)[8]) {
    u32 br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) (void)zk_subc(s[i], t[i], br, &br);
    return br == 0;
  };
  auto sub_raw = [](u32(&s)[8], const u32(&t)[8]) {
    u32 br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = zk_subc(s[i], t[i], br, &br);
  };
  auto sub_mod = [&](u32(&s)[8], const u32(&t)[8]) {
    u32 br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = zk_subc(s[i], t[i], br, &br);
    if (br) {
      u32 c = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] = zk_addc(s[i], P::MOD[i], c, &c);
    }
  };
  while (!is_one(u) && !is_one(v)) {
    while (!(u[0] & 1u)) {
      shr1(u);
      halve_mod(x1);
    }
    while (!(v[0] & 1u)) {
      shr1(v);
      halve_mod(x2);
    }
    if (geq(u, v)) {
      sub_raw(u, v);
      sub_mod(x1, x2);
    } else {
      sub_raw(v, u);
      sub_mod(x2, x1);
    }
  }
  Fp<P> r, r3;
  const bool pick_u = is_one(u);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    r.l[i] = pick_u ? x1[i] : x2[i];
    r3.l[i] = P::R3[i];
  }
  return fp_mul<P>(r, r3);
}

typedef Fp<FrP> Fr;
typedef Fp<FqP> Fq;

ZK_HD Fr operator+(const Fr& a, const Fr& b) { return fp_add<FrP>(a, b); }
ZK_HD Fr operator-(const Fr& a, const Fr& b) { return fp_sub<FrP>(a, b); }
ZK_HD Fr operator*(const Fr& a, const Fr& b) { return fp_mul<FrP>(a, b); }
ZK_HD Fq operator+(const Fq& a, const Fq& b) { return fp_add<FqP>(a, b); }
ZK_HD Fq operator-(const Fq& a, const Fq& b) { return fp_sub<FqP>(a, b); }
ZK_HD Fq operator*(const Fq& a, const Fq& b) { return fp_mul<FqP>(a, b); }

// ---------------------------------------------------------------------------
// G1
// ---------------------------------------------------------------------------
struct G1Affine {  // 64 B; identity encoded as (0,0) (halo2curves convention)
  Fq x, y;
  ZK_HD bool is_identity() const { return x.is_zero() && y.is_zero(); }
};

struct G1Jac {  // 96 B, ABI output form {X,Y,Z}: x = X/Z^2, y = Y/Z^3, identity Z = 0
  Fq x, y, z;
};

struct G1X {  // XYZZ accumulator, 128 B; identity zz = zzz = 0
  Fq x, y, zz, zzz;
  static ZK_HD G1X identity() {
    G1X r;
    r.x = Fq::zero();
    r.y = Fq::zero();
    r.zz = Fq::zero();
    r.zzz = Fq::zero();
    return r;
  }
  ZK_HD bool is_identity() const { return zz.is_zero(); }
};

// 2*P for affine P (not identity) -> XYZZ   (EFD "mdbl-2008-s-1")
ZK_HD G1X g1x_from_affine_dbl(const G1Affine& p) {
  G1X r;
  Fq u = fp_dbl<FqP>(p.y);
  Fq v = fp_sqr<FqP>(u);
  Fq w = u * v;
  Fq s = p.x * v;
  Fq xx = fp_sqr<FqP>(p.x);
  Fq m = fp_add<FqP>(fp_dbl<FqP>(xx), xx);  // a = 0
  r.x = fp_sqr<FqP>(m) - fp_dbl<FqP>(s);
  r.y = fp_mul2<FqP>(m, s - r.x, fp_neg<FqP>(w), p.y);
  r.zz = v;
  r.zzz = w;
  return r;
}

ZK_HD G1X g1x_from_affine(const G1Affine& p) {
  G1X r;
  if (p.is_identity()) return G1X::identity();
  r.x = p.x;
  r.y = p.y;
  r.zz = Fq::one();
  r.zzz = Fq::one();
  return r;
}

// 2*P in XYZZ  (EFD "dbl-2008-s-1")
ZK_HD G1X g1x_dbl(const G1X& p) {
  if (p.is_identity()) return p;
  G1X r;
  Fq u = fp_dbl<FqP>(p.y);
  Fq v = fp_sqr<FqP>(u);
  Fq w = u * v;
  Fq s = p.x * v;
  Fq xx = fp_sqr<FqP>(p.x);
  Fq m = fp_add<FqP>(fp_dbl<FqP>(xx), xx);
  r.x = fp_sqr<FqP>(m) - fp_dbl<FqP>(s);
  r.y = fp_mul2<FqP>(m, s - r.x, fp_neg<FqP>(w), p.y);
  r.zz = v * p.zz;
  r.zzz = w * p.zzz;
  return r;
}

// acc += (neg ? -q : q), q affine  (EFD "madd-2008-s", with the doubling / cancel cases)
ZK_HD void g1x_add_affine(G1X& acc, const G1Affine& q, bool neg) {
  if (q.is_identity()) return;
  Fq qy = neg ? fp_neg<FqP>(q.y) : q.y;
  if (acc.is_identity()) {
    acc.x = q.x;
    acc.y = qy;
    acc.zz = Fq::one();
    acc.zzz = Fq::one();
    return;
  }
  Fq u2 = q.x * acc.zz;
  Fq s2 = qy * acc.zzz;
  Fq p = u2 - acc.x;
  Fq r = s2 - acc.y;
  if (p.is_zero()) {
    if (r.is_zero()) {
      G1Affine t;
      t.x = q.x;
      t.y = qy;
      acc = g1x_from_affine_dbl(t);
    } else {
      acc = G1X::identity();
    }
    return;
  }
  Fq pp = fp_sqr<FqP>(p);
  Fq ppp = p * pp;
  Fq qq = acc.x * pp;
  Fq x3 = fp_sqr<FqP>(r) - ppp - fp_dbl<FqP>(qq);
  acc.y = fp_mul2<FqP>(r, qq - x3, fp_neg<FqP>(acc.y), ppp);
  acc.x = x3;
  acc.zz = acc.zz * pp;
  acc.zzz = acc.zzz * ppp;
}

// acc += q, both XYZZ  (EFD "add-2008-s")
ZK_HD void g1x_add(G1X& acc, const G1X& q) {
  if (q.is_identity()) return;
  if (acc.is_identity()) {
    acc = q;
    return;
  }
  Fq u1 = acc.x * q.zz;
  Fq u2 = q.x * acc.zz;
  Fq s1 = acc.y * q.zzz;
  Fq s2 = q.y * acc.zzz;
  Fq p = u2 - u1;
  Fq r = s2 - s1;
  if (p.is_zero()) {
    if (r.is_zero()) {
      acc = g1x_dbl(acc);
    } else {
      acc = G1X::identity();
    }
    return;
  }
  Fq pp = fp_sqr<FqP>(p);
  Fq ppp = p * pp;
  Fq qq = u1 * pp;
  Fq x3 = fp_sqr<FqP>(r) - ppp - fp_dbl<FqP>(qq);
  acc.y = fp_mul2<FqP>(r, qq - x3, fp_neg<FqP>(s1), ppp);
  acc.x = x3;
  acc.zz = acc.zz * q.zz * pp;
  acc.zzz = acc.zzz * q.zzz * ppp;
}

ZK_HD G1X g1x_neg(const G1X& p) {
  G1X r = p;
  r.y = fp_neg<FqP>(p.y);
  return r;
}

// XYZZ -> affine (one field inversion); identity -> (0,0)
ZK_HD G1Affine g1x_to_affine(const G1X& p) {
  G1Affine r;
  if (p.is_identity()) {
    r.x = Fq::zero();
    r.y = Fq::zero();
    return r;
  }
  Fq izzz = fp_inv<FqP>(p.zzz);       // 1/ZZZ
  Fq izz = fp_sqr<FqP>(izzz * p.zz);  // (ZZ/ZZZ)^2 = 1/ZZ  (since ZZ^3 = ZZZ^2)
  r.x = p.x * izz;
  r.y = p.y * izzz;
  return r;
}

// Host: n accumulator-form sums -> affine points with ONE field inversion (Montgomery's trick over the non-zero ZZZ: three
// products per point), x = X (ZZ / ZZZ)^2, y = Y / ZZZ as in g1x_to_affine; identity (ZZ = 0) -> (0, 0).  The prover normalises
// the commitments of a Fiat-Shamir round this way (the MSM's last kernel stores the XYZZ sum and skips its own inversion).
// A slot with ZZ != 0 and ZZZ == 0 is not a point any kernel writes (ZZ^3 = ZZZ^2); were one ever there (a stale or damaged
// slot) it must stay ITS fault: it is left out of the shared product -- a zero factor would turn every point of the round into
// garbage -- and comes out as the identity, which the Poseidon transcript refuses loudly.
inline void g1x_normalize_batch(const G1X *in, size_t n, G1Affine *out) {
  std::vector<Fq> pre(n);
  Fq acc = Fq::one();
  auto skip = [](const G1X &p) { return p.is_identity() || p.zzz.is_zero(); };
  for (size_t i = 0; i < n; ++i) {
    pre[i] = acc;
    if (!skip(in[i])) acc = acc * in[i].zzz;
  }
  Fq inv = fp_inv<FqP>(acc);   // 1 / (product of all ZZZ)
  for (size_t i = n; i-- > 0;) {
    if (skip(in[i])) {
      out[i].x = Fq::zero();
      out[i].y = Fq::zero();
      continue;
    }
    const Fq izzz = inv * pre[i];
    inv = inv * in[i].zzz;
    const Fq izz = fp_sqr<FqP>(izzz * in[i].zz);
    out[i].x = in[i].x * izz;
    out[i].y = in[i].y * izzz;
  }
}

// XYZZ -> Jacobian {X,Y,Z} with Z = ZZZ/ZZ:  X_j = x Z^2 = X*ZZZ^2/ZZ^3 ... use affine-free map:
// take Z = ZZ (then Z^2 = ZZ^2, Z^3 = ZZ^3 = ZZZ^2):  X_j = X*ZZ, Y_j = Y*ZZZ.
ZK_HD G1Jac g1x_to_jac(const G1X& p) {
  G1Jac r;
  if (p.is_identity()) {
    r.x = Fq::zero();
    r.y = Fq::one();
    r.z = Fq::zero();
    return r;
  }
  r.x = p.x * p.zz;
  r.y = p.y * p.zzz;
  r.z = p.zz;
  return r;
}

}  // namespace zk
