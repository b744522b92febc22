// BN254 Fr in radix 2^29 (nine limbs, Montgomery constant 2^261): the scalar-field twin of fq29.hip.hpp, used where a kernel
// multiplies data by table constants (the NTT twiddles, coset shifts, n^-1).
//
// A column value stays in the library's standard form x * 2^256 in memory; the CONSTANT is stored as w * 2^261, so that
// fr29_mul(unpack(x 2^256), unpack(w 2^261)) = x w 2^256: the product of the nine-limb multiply is again a standard value,
// with no conversion of the data beyond regrouping its bits into 29-bit limbs (27 shift/mask operations).
// zk_fr_to_29() turns a standard constant into that form (times 32).
#pragma once
#include "fq29.hip.hpp"

namespace zk {
namespace r29 {
constexpr u32 INV = 0x0fffffffu;  // -r^-1 mod 2^29
#define ZK_R29_P \
  { 0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu }
#define ZK_R29_2P \
  { 0x00000002u, 0x1e1f593fu, 0x1cb848a1u, 0x0fa121e6u, 0x0b0ba506u, 0x05b68181u, 0x014dc282u, 0x1cb84c68u, 0x0060c89cu }
#define ZK_R29_3P \
  { 0x10000003u, 0x1d2f05deu, 0x0b146cf2u, 0x1771b2dau, 0x00917789u, 0x0891c242u, 0x01f4a3c3u, 0x0b14729cu, 0x00912cebu }
#define ZK_R29_4P \
  { 0x00000004u, 0x1c3eb27eu, 0x19709143u, 0x1f4243cdu, 0x16174a0cu, 0x0b6d0302u, 0x029b8504u, 0x197098d0u, 0x00c19139u }
#define ZK_R29_8P \
  { 0x00000008u, 0x187d64fcu, 0x12e12287u, 0x1e84879bu, 0x0c2e9419u, 0x16da0605u, 0x05370a08u, 0x12e131a0u, 0x01832273u }
}  // namespace r29

#define F29_FN(name) fr29_##name
#define F29_P ZK_R29_P
#define F29_2P ZK_R29_2P
#define F29_3P ZK_R29_3P
#define F29_INV r29::INV
#include "f29_field.inc"
#undef F29_FN
#undef F29_P
#undef F29_2P
#undef F29_3P
#undef F29_INV

ZK_HD F29 fr29_unpack(const Fr &w) {
  Fq t;
#pragma unroll
  for (int i = 0; i < 8; ++i) t.l[i] = w.l[i];
  return f29_unpack(t);
}
ZK_HD Fr fr29_pack(const F29 &a) {
  const Fq t = f29_pack(a);
  Fr w;
#pragma unroll
  for (int i = 0; i < 8; ++i) w.l[i] = t.l[i];
  return w;
}
// standard constant (w 2^256, canonical) -> the packed 2^261 form the nine-limb multiply wants for its constant operand
ZK_HD Fr zk_fr_to_29(const Fr &w) {
  Fr c = Fr::zero();
  c.l[0] = 32;
  return w * fp_to_mont<FrP>(c);
}
// x * w for a standard x (any representative below 2^256 and below 11 r) and a 2^261-form constant: canonical standard result
ZK_HD Fr fr29_mul_const(const Fr &x, const Fr &w29) {
#if defined(__HIP_DEVICE_COMPILE__)
  // keep one product's nine-limb temporaries live at a time: without the fences the scheduler interleaves all the
  // independent products of a butterfly stage and spills (408 bytes of scratch per lane in the 2^13 NTT tile)
  __builtin_amdgcn_sched_barrier(0);
#endif
  const Fr r = fr29_pack(fr29_canonical(fr29_mul(fr29_unpack(x), fr29_unpack(w29))));
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_sched_barrier(0);
#endif
  return r;
}

// canonical integer -> standard form (x 2^256) through the nine-limb product: the constant operand is 2^517 mod r.  A dependent
// chain of these issues three times as fast as the 8 x 32-bit fp_to_mont (0.37 against 1.2 us per product in a lone wave): the
// witness gadgets store ~100 cells per thread one after the other.
ZK_HD Fr fr29_to_mont(const Fr &x) {
  const u32 c[8] = {0xd42db4dfu, 0x333ad321u, 0xf1d1cbd2u, 0x57936df3u, 0xf5eeb84du, 0xd0e021f3u, 0x0896f487u, 0x1275c7bdu};
  Fr w;
#pragma unroll
  for (int i = 0; i < 8; ++i) w.l[i] = c[i];
  return fr29_mul_const(x, w);
}

}  // namespace zk
