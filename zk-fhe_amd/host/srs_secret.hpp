// The toxic waste of the (unsafe, test-only) structured reference strings this library can derive itself -- README.md:34:
// "generate a random universal trusted setup for testing purposes ... unsafe ... stored in params/".
//
//  * ZKFHE_SRS_HALO2_UNSAFE: the reference's own.  halo2-scaffold `gen_srs(k)` (third-party, reached from examples/bfv.rs:311)
//    runs `ParamsKZG::<Bn256>::setup(k, ChaCha20Rng::from_seed(Default::default()))`; setup draws `s = Fr::random(rng)`, which for
//    halo2curves' bn256 Fr is `from_u512` of eight `next_u64()` words -- the first 64 bytes of the ChaCha20 keystream of the
//    all-zero key, counter 0, stream 0 (the published zero-key vector of RFC 7539 / draft-agl-tls-chacha20poly1305), read as a
//    512-bit little-endian integer and reduced mod r.
//  * any other seed: Blake2b-512(person "zkfhe-srs", seed) reduced the same way (the suite's seeds, oracle/halo2_ref.py make_srs).
#pragma once
#include <cstdint>
#include <cstring>

#include "../../include/zkfhe.h"
#include "transcript.hpp"

namespace zkhost {

// ChaCha20 block function (RFC 7539 section 2.3): state = constants | key | words 12..15 (counter and nonce as the caller lays
// them out: RFC 7539 uses a 32-bit counter and a 96-bit nonce, rand_chacha a 64-bit counter and a 64-bit stream id)
inline void chacha20_block(const uint8_t key[32], const uint32_t counter_nonce[4], uint8_t out[64]) {
  uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
  for (int i = 0; i < 8; ++i) s[4 + i] = (uint32_t)key[4 * i] | (uint32_t)key[4 * i + 1] << 8 | (uint32_t)key[4 * i + 2] << 16 | (uint32_t)key[4 * i + 3] << 24;
  for (int i = 0; i < 4; ++i) s[12 + i] = counter_nonce[i];
  uint32_t x[16];
  memcpy(x, s, sizeof(x));
  auto rotl = [](uint32_t v, int n) { return (v << n) | (v >> (32 - n)); };
  auto qr = [&](int a, int b, int c, int d) {
    x[a] += x[b], x[d] = rotl(x[d] ^ x[a], 16);
    x[c] += x[d], x[b] = rotl(x[b] ^ x[c], 12);
    x[a] += x[b], x[d] = rotl(x[d] ^ x[a], 8);
    x[c] += x[d], x[b] = rotl(x[b] ^ x[c], 7);
  };
  for (int r = 0; r < 10; ++r) {
    qr(0, 4, 8, 12), qr(1, 5, 9, 13), qr(2, 6, 10, 14), qr(3, 7, 11, 15);
    qr(0, 5, 10, 15), qr(1, 6, 11, 12), qr(2, 7, 8, 13), qr(3, 4, 9, 14);
  }
  for (int i = 0; i < 16; ++i) {
    const uint32_t v = x[i] + s[i];
    out[4 * i] = (uint8_t)v, out[4 * i + 1] = (uint8_t)(v >> 8), out[4 * i + 2] = (uint8_t)(v >> 16), out[4 * i + 3] = (uint8_t)(v >> 24);
  }
}

inline bool is_halo2_unsafe_seed(const uint8_t *seed, size_t seed_len) {
  return seed && seed_len == strlen(ZKFHE_SRS_HALO2_UNSAFE) && memcmp(seed, ZKFHE_SRS_HALO2_UNSAFE, seed_len) == 0;
}

inline U256 srs_secret(const uint8_t *seed, size_t seed_len) {
  uint8_t d[64];
  if (is_halo2_unsafe_seed(seed, seed_len)) {
    const uint8_t key[32] = {0};
    const uint32_t zero[4] = {0, 0, 0, 0};
    chacha20_block(key, zero, d);
  } else {
    Blake2b h(64, "zkfhe-srs");
    h.update(seed, seed_len);
    h.digest(d);
  }
  return from_bytes_wide(d);
}

}  // namespace zkhost
