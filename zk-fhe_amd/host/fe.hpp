// Host-side BN254 Fr values for witness generation: canonical 256-bit integers in [0, r).
// (halo2-base keeps cell values as `F`; here they stay canonical on the host and are converted to
// Montgomery form in bulk on the GPU -- zkfhe_fr_to_mont -- when the columns are uploaded.)
#pragma once
#include <cstdint>
#include <cstring>

#include "../csrc/bn254.hip.hpp"
#include "bigint.hpp"

namespace zkhost {

struct U256 {
  uint64_t l[4];
  bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
  bool operator==(const U256 &o) const { return l[0] == o.l[0] && l[1] == o.l[1] && l[2] == o.l[2] && l[3] == o.l[3]; }
  bool operator!=(const U256 &o) const { return !(*this == o); }
  bool operator<(const U256 &o) const {
    for (int i = 3; i >= 0; --i)
      if (l[i] != o.l[i]) return l[i] < o.l[i];
    return false;
  }
  unsigned bits() const {
    for (int i = 3; i >= 0; --i)
      if (l[i]) return 64 * i + (64 - __builtin_clzll(l[i]));
    return 0;
  }
  bool fits_u128() const { return (l[2] | l[3]) == 0; }
};

namespace fe {

static const U256 MOD = {{0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}};
static const unsigned MOD_BITS = 254;

inline U256 zero() { return U256{{0, 0, 0, 0}}; }
inline U256 from_u64(uint64_t v) { return U256{{v, 0, 0, 0}}; }
inline U256 one() { return from_u64(1); }

inline uint64_t add_raw(U256 &r, const U256 &a, const U256 &b) {
  unsigned __int128 c = 0;
  for (int i = 0; i < 4; ++i) {
    c += (unsigned __int128)a.l[i] + b.l[i];
    r.l[i] = (uint64_t)c;
    c >>= 64;
  }
  return (uint64_t)c;
}
inline uint64_t sub_raw(U256 &r, const U256 &a, const U256 &b) {
  uint64_t borrow = 0;
  for (int i = 0; i < 4; ++i) {
    unsigned __int128 d = (unsigned __int128)a.l[i] - b.l[i] - borrow;
    r.l[i] = (uint64_t)d;
    borrow = (uint64_t)(d >> 64) & 1;
  }
  return borrow;
}
inline U256 add(const U256 &a, const U256 &b) {
  U256 r;
  add_raw(r, a, b);  // < 2^255
  if (!(r < MOD)) sub_raw(r, r, MOD);
  return r;
}
inline U256 sub(const U256 &a, const U256 &b) {
  U256 r;
  if (sub_raw(r, a, b)) add_raw(r, r, MOD);
  return r;
}
inline U256 neg(const U256 &a) { return a.is_zero() ? a : sub(MOD, a); }

inline zk::Fr to_fr_raw(const U256 &a) {
  zk::Fr f;
  memcpy(f.l, a.l, 32);
  return f;
}
inline U256 from_fr_raw(const zk::Fr &f) {
  U256 a;
  memcpy(a.l, f.l, 32);
  return a;
}
inline zk::Fr to_mont(const U256 &a) { return zk::fp_to_mont<zk::FrP>(to_fr_raw(a)); }
inline U256 from_mont(const zk::Fr &f) { return from_fr_raw(zk::fp_from_mont<zk::FrP>(f)); }

inline U256 mul(const U256 &a, const U256 &b) {
  // (a R)(b) R^-1 = a b
  return from_fr_raw(zk::fp_mul<zk::FrP>(to_mont(a), to_fr_raw(b)));
}

// x mod r for a (possibly negative) BigInt of at most 256 bits
inline U256 from_bigint(const BigInt &x) {
  if (x.mag.size() > 8) throw std::overflow_error("BigInt does not fit 256 bits");
  U256 v = zero();
  for (size_t i = 0; i < x.mag.size(); ++i) v.l[i / 2] |= (uint64_t)x.mag[i] << (32 * (i & 1));
  while (!(v < MOD)) sub_raw(v, v, MOD);
  return x.neg ? neg(v) : v;
}
inline BigInt to_bigint(const U256 &a) {
  BigInt r;
  for (int i = 0; i < 8; ++i) r.mag.push_back((uint32_t)(a.l[i / 2] >> (32 * (i & 1))));
  while (!r.mag.empty() && r.mag.back() == 0) r.mag.pop_back();
  return r;
}

inline U256 pow2(unsigned bits) {
  U256 r = zero();
  r.l[bits / 64] = (uint64_t)1 << (bits % 64);
  return r;
}
// (a >> shift) & 0xff for canonical a
inline uint32_t byte_at(const U256 &a, unsigned shift) {
  unsigned limb = shift / 64, off = shift % 64;
  if (limb >= 4) return 0;
  uint64_t v = a.l[limb] >> off;
  if (off > 56 && limb + 1 < 4) v |= a.l[limb + 1] << (64 - off);
  return (uint32_t)(v & 0xff);
}

}  // namespace fe
}  // namespace zkhost
