// BN254 Fq in radix 2^29 (nine u32 limbs, Montgomery with R' = 2^261) for the EC kernels of the MSM.
//
// Why: the 8 x 32-bit product-scanning multiply (bn254.hip.hpp fp_mul) spends half of its issue slots on carries -- every
// v_mad_u64_u32 is followed by s_nop + v_addc_co_u32 (9.9 cycles per pair against 5.3 for the multiply-add alone,
// profiles/r1_microbench.md).  With 29-bit limbs a column of a*b + m*p is at most 18 (27 for the fused two-product form)
// terms below 2^58: they add up in ONE 64-bit accumulator with no carry-out, so a product is 162 plain multiply-adds and
// 18 shift/mask pairs.  The seven spare bits of R' also allow lazy reduction: a product only needs a*b < 2^261 p, i.e.
// operands below 11 p, and returns a value below 2 p -- additions and subtractions in the point formulas do not reduce.
//
// Scope: registers only.  In memory a value stays one packed 256-bit word (the layout of zk::Fq, so tables, partials and
// buckets keep their size); what changes is the Montgomery constant: the MSM tables and accumulators hold x * 2^261 mod p.
// g1x29_to_std() converts a result back to the library's standard form (x * 2^256 mod p, canonical) before normalisation.
//
// Invariants: every F29 has limbs < 2^29 (the top limb holds whatever is left: values stay < 2^261); the comment of each
// function states the bound on its VALUE (as a multiple of p) it needs and gives.
#pragma once
#include "bn254.hip.hpp"

namespace zk {

struct F29 {
  u32 l[9];
};

// acc += a * b, the multiply-add of the nine-limb products' C bodies (the host pass, the native CPU checks, and -DZK_MAD_C).
// ON THE DEVICE the products are generated inline assembly since round 6 (f29_tied.inc, lz29_tied.inc; tools/gen_tied_products.py): one
// asm statement per COLUMN on the running accumulator.  Left to the compiler, `acc += (u64)a * b` after `acc >>= 29` is reassociated so
// that the previous column's carry is added last: every column starts in a register pair of its own and a v_lshl_add_u64 joins it to the
// carry -- 17 of a product's ~240 instructions, up to eleven accumulator pairs in flight.  Tied to one pair the joins go (k_msm_table's
// addition loop: 277 -> 39; k_ntt13: 246 -> 220 VGPRs).  One statement per multiply-add was tried first and lost: the compiler cannot see
// into an asm statement and puts a wait state before every VALU read of a register one defines (~150 s_nop per product); inside ONE
// statement it inserts nothing, and the step m_k = (low word * inv) mod 2^29 between two statements costs one wait state per column.
// Measured (profiles/r6_probes.md section 2): k_msm_table -6 %, k_msm_accumulate -3..4 %, k_ntt13 unchanged, the driver's wave +4.5 %,
// 96 steps +4 %, one proof alone -2 % (faster), bit-exact.  -DZK_MAD_C (ZKFHE_EXTRA_FLAGS) restores the compiler's form.
ZK_HD void zk_madu(u64 &acc, u32 a, u32 b) { acc += (u64)a * b; }
ZK_HD void zk_madu_s(u64 &acc, u32 a, u32 s) { acc += (u64)a * s; }
ZK_HD void zk_madi(long long &acc, int a, int b) { acc += (long long)a * (long long)b; }
ZK_HD void zk_madi_s(long long &acc, int a, int s) { acc += (long long)a * (long long)s; }

namespace q29 {
constexpr u32 MASK = (1u << 29) - 1;
constexpr u32 INV = 0x04866389u;  // -q^-1 mod 2^29
#define ZK_Q29_P \
  { 0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu }
#define ZK_Q29_2P \
  { 0x10f9fa8eu, 0x0208c16du, 0x18e5469eu, 0x05aa45a1u, 0x0b0bb2f0u, 0x05b68181u, 0x014dc282u, 0x1cb84c68u, 0x0060c89cu }
#define ZK_Q29_3P \
  { 0x0976f7d5u, 0x030d2224u, 0x1557e9edu, 0x087f6872u, 0x00918c68u, 0x0891c242u, 0x01f4a3c3u, 0x0b14729cu, 0x00912cebu }
#define ZK_Q29_4P \
  { 0x01f3f51cu, 0x041182dbu, 0x11ca8d3cu, 0x0b548b43u, 0x161765e0u, 0x0b6d0302u, 0x029b8504u, 0x197098d0u, 0x00c19139u }
#define ZK_Q29_8P \
  { 0x03e7ea38u, 0x082305b6u, 0x03951a78u, 0x16a91687u, 0x0c2ecbc0u, 0x16da0605u, 0x05370a08u, 0x12e131a0u, 0x01832273u }
#define ZK_Q29_ONE /* 2^261 mod p: the Montgomery form of 1 */ \
  { 0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u }
#define ZK_Q29_R256 /* 2^256 mod p as a plain integer: multiplying by it turns x 2^261 into x 2^256 */ \
  { 0x058f0d9du, 0x1aea1c6eu, 0x11c2cf74u, 0x11d651ebu, 0x1462c0a7u, 0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u }
}  // namespace q29

ZK_HD F29 f29_zero() {
  F29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = 0;
  return r;
}
ZK_HD F29 f29_const(const u32 (&c)[9]) {
  F29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = c[i];
  return r;
}
ZK_HD bool f29_is_literal_zero(const F29 &a) {
  u32 o = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) o |= a.l[i];
  return o == 0;
}
ZK_HD bool f29_eq(const F29 &a, const u32 (&c)[9]) {
  u32 o = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) o |= a.l[i] ^ c[i];
  return o == 0;
}

// packed 256-bit word (8 x u32) <-> nine 29-bit limbs.  Packing needs value < 2^256.
ZK_HD F29 f29_unpack(const Fq &w) {
  F29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bit = 29 * i, k = bit >> 5, sh = bit & 31;
    u32 v = w.l[k] >> sh;
    if (sh > 3 && k + 1 < 8) v |= w.l[k + 1] << (32 - sh);
    r.l[i] = i < 8 ? (v & q29::MASK) : v;
  }
  return r;
}
ZK_HD Fq f29_pack(const F29 &a) {
  Fq w;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    // word k holds bits 32k .. 32k+31: limb i = floor(32k / 29) from bit 32k - 29i, then the next limb(s)
    const int i = (32 * k) / 29, sh = 32 * k - 29 * i;
    u32 v = a.l[i] >> sh;
    v |= a.l[i + 1] << (29 - sh);
    if (29 - sh + 29 < 32 && i + 2 < 9) v |= a.l[i + 2] << (58 - sh);
    w.l[k] = v;
  }
  return w;
}

// signed carry propagation: t[i] in (-2^31, 2^31), total value >= 0  ->  limbs < 2^29 (top limb: the rest)
ZK_HD F29 f29_normalise(const int (&t)[9]) {
  F29 r;
  int c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int v = t[i] + c;
    r.l[i] = (u32)v & q29::MASK;
    c = v >> 29;  // arithmetic shift: floor division
  }
  r.l[8] = (u32)(t[8] + c);
  return r;
}

// a + b: value = a + b
ZK_HD F29 f29_add(const F29 &a, const F29 &b) {
  int t[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) t[i] = (int)(a.l[i] + b.l[i]);
  return f29_normalise(t);
}
// a - b + k p for a constant multiple kp >= b: value = a + kp - b > 0
ZK_HD F29 f29_sub(const F29 &a, const F29 &b, const u32 (&kp)[9]) {
  int t[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) t[i] = (int)a.l[i] - (int)b.l[i] + (int)kp[i];
  return f29_normalise(t);
}
ZK_HD F29 f29_neg(const F29 &b, const u32 (&kp)[9]) {  // kp - b
  int t[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) t[i] = (int)kp[i] - (int)b.l[i];
  return f29_normalise(t);
}
ZK_HD F29 f29_dbl(const F29 &a) { return f29_add(a, a); }

#define F29_FN(name) f29_##name
#define F29_P ZK_Q29_P
#define F29_2P ZK_Q29_2P
#define F29_3P ZK_Q29_3P
#define F29_INV q29::INV
#include "f29_field.inc"
#undef F29_FN
#undef F29_P
#undef F29_2P
#undef F29_3P
#undef F29_INV

// ---- G1 in XYZZ coordinates over F29.  Stored coordinates are below 2 p; identity: zz = zzz = literal 0 --------------
struct G1A29 {  // affine, canonical coordinates (table entries); identity (0, 0)
  F29 x, y;
  ZK_HD bool is_identity() const { return f29_is_literal_zero(x) && f29_is_literal_zero(y); }
};
struct G1X29 {
  F29 x, y, zz, zzz;
  static ZK_HD G1X29 identity() {
    G1X29 r;
    r.x = r.y = r.zz = r.zzz = f29_zero();
    return r;
  }
  ZK_HD bool is_identity() const { return f29_is_literal_zero(zz); }
};

// memory forms: zk::G1Affine / zk::G1X hold packed 256-bit words in the 2^261 Montgomery form
ZK_HD G1A29 g1a29_load(const G1Affine &p) {
  G1A29 r;
  r.x = f29_unpack(p.x);
  r.y = f29_unpack(p.y);
  return r;
}
ZK_HD G1X29 g1x29_load(const G1X &p) {
  G1X29 r;
  r.x = f29_unpack(p.x);
  r.y = f29_unpack(p.y);
  r.zz = f29_unpack(p.zz);
  r.zzz = f29_unpack(p.zzz);
  return r;
}
ZK_HD G1X g1x29_store(const G1X29 &p) {
  G1X r;
  r.x = f29_pack(p.x);
  r.y = f29_pack(p.y);
  r.zz = f29_pack(p.zz);
  r.zzz = f29_pack(p.zzz);
  return r;
}

// 2 P for affine P (not the identity)
ZK_HD G1X29 g1x29_from_affine_dbl(const F29 &px, const F29 &py) {  // px, py < 2 p
  const u32 P2[9] = ZK_Q29_2P, P4[9] = ZK_Q29_4P;
  G1X29 r;
  const F29 u = f29_dbl(py);                       // < 4p
  const F29 v = f29_sqr(u);                        // < 2p
  const F29 w = f29_mul(u, v);
  const F29 s = f29_mul(px, v);
  const F29 xx = f29_sqr(px);
  const F29 m = f29_add(f29_dbl(xx), xx);          // < 6p
  const F29 x3 = f29_weak_reduce(f29_sub(f29_sqr(m), f29_dbl(s), P4));   // m^2 + 4p - 2s < 6p -> < 2p
  r.y = f29_mul2(m, f29_sub(s, x3, P2), f29_neg(w, P2), py);             // 6p*4p + 2p*2p = 28 p^2
  r.x = x3;
  r.zz = v;
  r.zzz = w;
  return r;
}
ZK_HD G1X29 g1x29_dbl(const G1X29 &p) {
  if (p.is_identity()) return p;
  G1X29 r = g1x29_from_affine_dbl(p.x, p.y);   // the same formulas, then the Z factors
  r.zz = f29_mul(r.zz, p.zz);
  r.zzz = f29_mul(r.zzz, p.zzz);
  return r;
}

// acc += (neg ? -q : q), q affine canonical
ZK_HD void g1x29_add_affine(G1X29 &acc, const G1A29 &q, bool neg) {
  const u32 P1[9] = ZK_Q29_P, P2[9] = ZK_Q29_2P, P4[9] = ZK_Q29_4P, ONE[9] = ZK_Q29_ONE;
  if (q.is_identity()) return;
  const F29 qy = neg ? f29_neg(q.y, P1) : q.y;     // <= p
  if (acc.is_identity()) {
    acc.x = q.x;
    acc.y = qy;
    acc.zz = f29_const(ONE);
    acc.zzz = f29_const(ONE);
    return;
  }
  const F29 u2 = f29_mul(q.x, acc.zz);
  const F29 s2 = f29_mul(qy, acc.zzz);
  const F29 p = f29_sub(u2, acc.x, P2);            // in (0, 4p)
  const F29 r = f29_sub(s2, acc.y, P2);
  if (f29_is_zero_mod_p(p)) {
    if (f29_is_zero_mod_p(r)) acc = g1x29_from_affine_dbl(q.x, qy);
    else acc = G1X29::identity();
    return;
  }
  const F29 pp = f29_sqr(p);                        // 16 p^2
  const F29 ppp = f29_mul(p, pp);
  const F29 qq = f29_mul(acc.x, pp);
  // r^2 - ppp - 2 qq  ->  r^2 + (2p - ppp) + (4p - 2 qq) < 8p -> < 2p
  const F29 x3 = f29_weak_reduce(f29_add(f29_sub(f29_sqr(r), ppp, P2), f29_neg(f29_dbl(qq), P4)));
  acc.y = f29_mul2(r, f29_sub(qq, x3, P2), f29_neg(acc.y, P2), ppp);   // 4p*4p + 2p*2p = 20 p^2
  acc.x = x3;
  acc.zz = f29_mul(acc.zz, pp);
  acc.zzz = f29_mul(acc.zzz, ppp);
}

// acc += q, both XYZZ
ZK_HD void g1x29_add(G1X29 &acc, const G1X29 &q) {
  const u32 P2[9] = ZK_Q29_2P, P4[9] = ZK_Q29_4P;
  if (q.is_identity()) return;
  if (acc.is_identity()) {
    acc = q;
    return;
  }
  const F29 u1 = f29_mul(acc.x, q.zz);
  const F29 u2 = f29_mul(q.x, acc.zz);
  const F29 s1 = f29_mul(acc.y, q.zzz);
  const F29 s2 = f29_mul(q.y, acc.zzz);
  const F29 p = f29_sub(u2, u1, P2);
  const F29 r = f29_sub(s2, s1, P2);
  if (f29_is_zero_mod_p(p)) {
    if (f29_is_zero_mod_p(r)) acc = g1x29_dbl(acc);
    else acc = G1X29::identity();
    return;
  }
  const F29 pp = f29_sqr(p);
  const F29 ppp = f29_mul(p, pp);
  const F29 qq = f29_mul(u1, pp);
  const F29 x3 = f29_weak_reduce(f29_add(f29_sub(f29_sqr(r), ppp, P2), f29_neg(f29_dbl(qq), P4)));
  acc.y = f29_mul2(r, f29_sub(qq, x3, P2), f29_neg(s1, P2), ppp);
  acc.x = x3;
  acc.zz = f29_mul(f29_mul(acc.zz, q.zz), pp);
  acc.zzz = f29_mul(f29_mul(acc.zzz, q.zzz), ppp);
}

// XYZZ in the 2^261 form -> the library's standard XYZZ (canonical coordinates, Montgomery with 2^256)
ZK_HD G1X g1x29_to_std(const G1X29 &p) {
  const u32 K[9] = ZK_Q29_R256;
  if (p.is_identity()) return G1X::identity();
  const F29 k = f29_const(K);
  G1X r;
  r.x = f29_pack(f29_canonical(f29_mul(p.x, k)));
  r.y = f29_pack(f29_canonical(f29_mul(p.y, k)));
  r.zz = f29_pack(f29_canonical(f29_mul(p.zz, k)));
  r.zzz = f29_pack(f29_canonical(f29_mul(p.zzz, k)));
  return r;
}
// standard affine point (canonical, 2^256 form) -> packed affine in the 2^261 form (what the MSM tables hold): x * 32
ZK_HD G1Affine g1_affine_to_29(const G1Affine &p) {
  if (p.is_identity()) return p;
  Fq c = Fq::zero();
  c.l[0] = 32;
  const Fq k = fp_to_mont<FqP>(c);
  G1Affine r;
  r.x = p.x * k;
  r.y = p.y * k;
  return r;
}

}  // namespace zk
