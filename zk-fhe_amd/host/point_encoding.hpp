// The 32-byte compressed form of a BN254 G1 point in the proof byte stream -- ONE definition, used by the prover's transcript
// (transcript.hpp), the verifier's reader (verifier.cpp) and mirrored by oracle/point_encoding.py (which both oracle
// transcripts import).  tests/test_point_encoding.py pins the three against tests/golden/point_encoding.json.
//
// Layout = halo2curves `new_curve_impl!` `GroupEncoding for G1Affine` as of the 0.3.2 .. 0.5 line that halo2-axiom
// (halo2curves-axiom) builds on -- the crate behind `Snark.proof` of reference examples/bfv.rs:311:
//     to_bytes:   identity      -> all zero except  bytes[31] |= 0b1000_0000
//                 otherwise     -> x.to_bytes() (little-endian, canonical), bytes[31] |= (y.to_bytes()[0] & 1) << 6
//     from_bytes: is_inf = bytes[31] >> 7, ysign = (bytes[31] >> 6) & 1, bytes[31] &= 0b0011_1111
// BN254's q has 254 bits, so bits 6 and 7 of the last byte are free.  (Round 2 had the two flags the other way round: sign in
// bit 7, identity in bit 6; the pasta-style 0.3.1 form is sign in bit 7 and an all-zero identity.  None of the three can be
// checked against a reference-made proof here -- no Rust toolchain, no committed .snark -- see DESIGN.md section 4.)
//
// ONE switch for all of it: the environment variable ZKFHE_POINT_ENCODING, read here and by oracle/point_encoding.py alike --
//     "halo2curves-0.3.2" (default; also the 0.4 / 0.5 / -axiom layout above)
//     "halo2curves-0.3.1" (sign in bit 7, the identity as 32 zero bytes: the pasta-style form, should the -ce tags resolve to it)
// so that the day a reference-made point is available the layout is flipped in one place and every test follows.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace zkhost {
namespace ptenc {

struct Layout {
  uint8_t sign_bit;       // in byte 31: y is odd
  uint8_t identity_bit;   // in byte 31: the point at infinity (every other bit zero); 0 = the identity is the all-zero string
  uint8_t x_mask;         // what is left of byte 31 for x
};
inline const Layout &layout() {
  static const Layout L = [] {
    const char *e = getenv("ZKFHE_POINT_ENCODING");
    if (e && strcmp(e, "halo2curves-0.3.1") == 0) return Layout{0x80, 0x00, 0x7f};
    return Layout{0x40, 0x80, 0x3f};
  }();
  return L;
}
// b: 32 bytes.  The identity's encoding under the layout in force.
inline bool is_identity_encoding(const uint8_t *b) {
  const Layout &L = layout();
  if (L.identity_bit) return (b[31] & L.identity_bit) != 0;
  for (int i = 0; i < 32; ++i)
    if (b[i]) return false;
  return true;
}

}  // namespace ptenc
}  // namespace zkhost
