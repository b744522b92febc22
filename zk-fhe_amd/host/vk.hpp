// The verifying key's transcript representation: one Fr that commits to the configuration (including the transcript
// kind) and to every fixed / sigma commitment.  halo2 derives its `transcript_repr` from a Blake2b hash of the pinned
// verifying key's Debug text (plonk.rs VerifyingKey::from_parts); that text cannot be reproduced without the Rust types
// (and the reference's vk carries an unused Keccak sub-circuit), so this is the library's own digest:
// Blake2b-512(person "zkfhe-vk", 8 x u32 config || fixed commitments || sigma commitments) mod r.
#pragma once
#include <vector>

#include "bfv_circuit.hpp"
#include "transcript.hpp"

namespace zkhost {

inline U256 vk_digest(const CircuitConfig &cfg, const std::vector<AffinePoint> &fixed_commit, const std::vector<AffinePoint> &sigma_commit) {
  Blake2b h(64, "zkfhe-vk");
  const uint32_t hdr[8] = {cfg.k, cfg.n_gate0, cfg.n_gate1, cfg.n_lookup, cfg.n_rlc, cfg.unusable_rows, cfg.lookup_bits, cfg.transcript};
  h.update(hdr, sizeof(hdr));
  for (const auto *v : {&fixed_commit, &sigma_commit})
    for (const auto &p : *v) {
      h.update(p.x.l, 32);
      h.update(p.y.l, 32);
    }
  uint8_t d[64];
  h.digest(d);
  return from_bytes_wide(d);
}

}  // namespace zkhost
