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
#pragma once
#include <cstdint>

namespace zkhost {
namespace ptenc {

constexpr uint8_t SIGN_BIT = 0x40;       // bit 6 of byte 31: y is odd
constexpr uint8_t IDENTITY_BIT = 0x80;   // bit 7 of byte 31: the point at infinity (every other bit zero)
constexpr uint8_t X_MASK = 0x3f;         // what is left of byte 31 for x

}  // namespace ptenc
}  // namespace zkhost
