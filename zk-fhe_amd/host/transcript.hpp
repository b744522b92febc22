// Blake2b (RFC 7693) and the Fiat-Shamir transcript of the prover.
//
// The reference's scaffold proves and verifies with snark-verifier's PoseidonTranscript (examples/bfv.rs:311 ->
// gen_snark_shplonk); that is the default here (poseidon.hpp).  halo2_proofs' own Blake2b transcript stays selectable
// (zkfhe_bfv_config.transcript).  Mirrors oracle/halo2_ref.py `TRANSCRIPTS` / `Rng`.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "fe.hpp"
#include "point_encoding.hpp"
#include "poseidon.hpp"

namespace zkhost {

class Blake2b {
 public:
  Blake2b(size_t outlen, const char *personal) {
    static const uint64_t IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                   0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    for (int i = 0; i < 8; ++i) h[i] = IV[i];
    h[0] ^= 0x01010000ULL ^ (uint64_t)outlen;
    uint8_t p[16] = {0};
    if (personal) memcpy(p, personal, std::min<size_t>(16, strlen(personal)));
    uint64_t p0, p1;
    memcpy(&p0, p, 8);
    memcpy(&p1, p + 8, 8);
    h[6] ^= p0;
    h[7] ^= p1;
    out_len = outlen;
  }
  void update(const void *data, size_t len) {
    const uint8_t *in = (const uint8_t *)data;
    while (len) {
      if (buf_len == 128) {
        t += 128;
        compress(false);
        buf_len = 0;
      }
      size_t take = std::min(len, (size_t)128 - buf_len);
      memcpy(buf + buf_len, in, take);
      buf_len += take;
      in += take;
      len -= take;
    }
  }
  // finalises a COPY: the running state can keep absorbing (halo2's squeeze clones the hasher)
  void digest(uint8_t *out) const {
    Blake2b c = *this;
    c.t += c.buf_len;
    memset(c.buf + c.buf_len, 0, 128 - c.buf_len);
    c.compress(true);
    uint8_t full[64];
    memcpy(full, c.h, 64);
    memcpy(out, full, out_len);
  }

 private:
  uint64_t h[8];
  uint64_t t = 0;
  uint8_t buf[128];
  size_t buf_len = 0;
  size_t out_len;
  static uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
  void compress(bool last) {
    static const uint64_t IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                   0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    static const uint8_t S[12][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
    uint64_t m[16], v[16];
    memcpy(m, buf, 128);
    for (int i = 0; i < 8; ++i) {
      v[i] = h[i];
      v[i + 8] = IV[i];
    }
    v[12] ^= t;
    if (last) v[14] = ~v[14];
    auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
      v[a] = v[a] + v[b] + x;
      v[d] = rotr(v[d] ^ v[a], 32);
      v[c] = v[c] + v[d];
      v[b] = rotr(v[b] ^ v[c], 24);
      v[a] = v[a] + v[b] + y;
      v[d] = rotr(v[d] ^ v[a], 16);
      v[c] = v[c] + v[d];
      v[b] = rotr(v[b] ^ v[c], 63);
    };
    for (int r = 0; r < 12; ++r) {
      const uint8_t *s = S[r];
      G(0, 4, 8, 12, m[s[0]], m[s[1]]);
      G(1, 5, 9, 13, m[s[2]], m[s[3]]);
      G(2, 6, 10, 14, m[s[4]], m[s[5]]);
      G(3, 7, 11, 15, m[s[6]], m[s[7]]);
      G(0, 5, 10, 15, m[s[8]], m[s[9]]);
      G(1, 6, 11, 12, m[s[10]], m[s[11]]);
      G(2, 7, 8, 13, m[s[12]], m[s[13]]);
      G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
    for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
  }
};

// 64 little-endian bytes -> value mod r  (Fr::from_bytes_wide): lo + hi * 2^256, each half reduced first
inline U256 from_bytes_wide(const uint8_t b[64]) {
  U256 lo, hi;
  memcpy(lo.l, b, 32);
  memcpy(hi.l, b + 32, 32);
  auto reduce = [](U256 v) {   // 2^256 < 6 r
    while (!(v < fe::MOD)) fe::sub_raw(v, v, fe::MOD);
    return v;
  };
  lo = reduce(lo);
  hi = reduce(hi);
  // Montgomery product of hi with R^2 is hi * R mod r = hi * 2^256 mod r
  pos::F h;
  memcpy(h.l, hi.l, 32);
  const pos::F hr = pos::mul(h, pos::R2);
  U256 hi_r;
  memcpy(hi_r.l, hr.l, 32);
  return fe::add(lo, hi_r);
}

class Rng {  // blinding stream: Blake2b-512(person "zkfhe-rng", seed[32] || counter_le64) mod r
 public:
  explicit Rng(const uint8_t seed32[32]) { memcpy(seed, seed32, 32); }
  U256 next() {
    Blake2b h(64, "zkfhe-rng");
    uint8_t msg[40];
    memcpy(msg, seed, 32);
    for (int i = 0; i < 8; ++i) msg[32 + i] = (uint8_t)(ctr >> (8 * i));
    ++ctr;
    h.update(msg, 40);
    uint8_t d[64];
    h.digest(d);
    return from_bytes_wide(d);
  }
  uint64_t ctr = 0;

 private:
  uint8_t seed[32];
};

struct AffinePoint {  // canonical coordinates; identity = (0,0)
  U256 x, y;
  bool is_identity() const { return x.is_zero() && y.is_zero(); }
};

enum TranscriptKind : uint32_t {
  TR_POSEIDON = 0,  // snark-verifier PoseidonTranscript<NativeLoader> -- what the reference's prove / verify use
  TR_BLAKE2B = 1,   // halo2_proofs Blake2bWrite / Blake2bRead with Challenge255
};

// Fiat-Shamir transcript, writer and reader side (the reader feeds decoded values through common_*).
//  * Poseidon: a scalar is absorbed as itself, a point as its affine coordinates mapped Fq -> Fr (value mod r); the
//    identity cannot be absorbed (snark-verifier: "Cannot write points at infinity to the transcript").
//  * Blake2b: personalisation "Halo2-Transcript", prefix bytes 0 = challenge, 1 = point, 2 = scalar; a challenge is the
//    64-byte digest of a copy of the state, reduced mod r.
// The byte stream is the same for both: 32-byte compressed points (x little-endian, bit 7 of byte 31 = parity of y) and
// 32-byte little-endian scalars.
class Transcript {
 public:
  explicit Transcript(uint32_t kind_) : kind(kind_), h(64, "Halo2-Transcript") {
    if (kind != TR_POSEIDON && kind != TR_BLAKE2B) throw std::runtime_error("unknown transcript kind");
  }
  ~Transcript() { join(); }
  Transcript(const Transcript &) = delete;
  Transcript &operator=(const Transcript &) = delete;
  std::vector<uint8_t> out;

  void common_point(const AffinePoint &p) {
    join();
    if (kind == TR_POSEIDON) {
      if (p.is_identity()) throw std::runtime_error("Cannot write points at infinity to the transcript");
      sp.update(fq_to_fr(p.x));
      sp.update(fq_to_fr(p.y));
      return;
    }
    uint8_t b[65];
    b[0] = 1;
    memcpy(b + 1, p.x.l, 32);
    memcpy(b + 33, p.y.l, 32);
    h.update(b, 65);
  }
  void common_scalar(const U256 &s) {
    join();
    absorb_scalar(s);
  }
  // Absorbs a long run of scalars (the public inputs) beside the caller; every later call waits for it first.  With several
  // provers in flight the run goes to the eight-lane hash service (poseidon_x8.cpp), where it shares a core with the other
  // proofs' runs; a lone prover hashes it on a helper thread of its own.
  void common_scalars_async(std::vector<U256> v) {
    join();
    pending = std::move(v);
    if (bulk_ok(pending.size())) {
      sp.begin_bulk(pending.data(), pending.size(), job);
      in_bulk = true;
      return;
    }
    worker = std::thread([this] {
      for (const U256 &s : pending) absorb_scalar(s);
    });
  }
  // The hash state between two absorptions (both hashers; the byte stream is not part of it): what common_scalars_async_marked
  // hands to its callback and restore() takes back.  A transcript that restores the state another one had after absorbing the
  // same values continues exactly as if it had absorbed them itself.
  struct State {
    Blake2b h{64, "Halo2-Transcript"};
    pos::Sponge sp;
  };
  State snapshot() {
    join();
    return State{h, sp};
  }
  void restore(const State &s) {
    join();
    h = s.h;
    sp = s.sp;
  }
  // common_scalars_async, with a callback on the helper thread after the first `mark` values: the state at that point (the
  // prover caches it per public key -- prefix_cache.hpp).  Both halves, [0, mark) and [mark, end), go to the hash service when
  // common_scalars_async would have sent the run there (shared mode: a batch of cold keys keeps the eight-lane saving), with the
  // state taken between the two jobs; otherwise through the single sponge.  A callback that throws (it allocates) loses the
  // cache entry, not the process.
  void common_scalars_async_marked(std::vector<U256> v, size_t mark, std::function<void(const State &)> on_mark) {
    join();
    pending = std::move(v);
    if (mark > pending.size()) mark = pending.size();
    worker = std::thread([this, mark, on_mark] {
      if (bulk_ok(mark)) {
        sp.begin_bulk(pending.data(), mark, job);
        sp.end_bulk(job);
      } else {
        for (size_t i = 0; i < mark; ++i) absorb_scalar(pending[i]);
      }
      try {
        on_mark(State{h, sp});
      } catch (...) {
      }
      const size_t rest = pending.size() - mark;
      if (bulk_ok(rest)) {
        sp.begin_bulk(pending.data() + mark, rest, job);
        sp.end_bulk(job);
      } else {
        for (size_t i = mark; i < pending.size(); ++i) absorb_scalar(pending[i]);
      }
    });
  }
  // a run of scalars / points written at once (the evaluations, a round's commitments): same bytes and the same sponge state as
  // one write_* call each
  void write_scalars(const std::vector<U256> &v) {
    join();
    for (const U256 &s : v) {
      const uint8_t *b = (const uint8_t *)s.l;
      out.insert(out.end(), b, b + 32);
    }
    if (!bulk_ok(v.size())) {
      for (const U256 &s : v) absorb_scalar(s);
      return;
    }
    sp.begin_bulk(v.data(), v.size(), job);
    sp.end_bulk(job);
  }
  void write_points(const std::vector<AffinePoint> &v) {
    join();
    if (kind != TR_POSEIDON || !bulk_ok(2 * v.size())) {
      for (const AffinePoint &p : v) write_point(p);
      return;
    }
    std::vector<U256> xy(2 * v.size());
    for (size_t i = 0; i < v.size(); ++i) {
      if (v[i].is_identity()) throw std::runtime_error("Cannot write points at infinity to the transcript");
      xy[2 * i] = fq_to_fr(v[i].x);
      xy[2 * i + 1] = fq_to_fr(v[i].y);
      uint8_t b[32];
      compress(v[i], b);
      out.insert(out.end(), b, b + 32);
    }
    sp.begin_bulk(xy.data(), xy.size(), job);
    sp.end_bulk(job);
  }
  void write_point(const AffinePoint &p) {
    common_point(p);
    uint8_t b[32];
    compress(p, b);
    out.insert(out.end(), b, b + 32);
  }
  void write_scalar(const U256 &s) {
    common_scalar(s);
    const uint8_t *b = (const uint8_t *)s.l;
    out.insert(out.end(), b, b + 32);
  }
  U256 squeeze() {
    join();
    if (kind == TR_POSEIDON) return sp.squeeze();
    const uint8_t z = 0;
    h.update(&z, 1);
    uint8_t d[64];
    h.digest(d);
    return from_bytes_wide(d);
  }
  static void compress(const AffinePoint &p, uint8_t b[32]) {
    const ptenc::Layout &L = ptenc::layout();
    if (p.is_identity()) {
      memset(b, 0, 32);
      b[31] |= L.identity_bit;
    } else {
      memcpy(b, p.x.l, 32);
      if (p.y.l[0] & 1) b[31] |= L.sign_bit;
    }
  }
  size_t poseidon_permutations() const { return sp.n_perm; }

 private:
  uint32_t kind;
  Blake2b h;
  pos::Sponge sp;
  std::thread worker;
  std::vector<U256> pending;
  pos::AbsorbJob job;
  bool in_bulk = false;
  void join() {
    if (worker.joinable()) worker.join();
    if (in_bulk) {
      in_bulk = false;
      sp.end_bulk(job);
    }
  }
  // the hash service: only in the "shared" mode (poseidon.hpp hash_mode), only when somebody can share the lanes
  // (ZKFHE_X8_MIN provers in flight, default 2), only for runs of 16 values or more
  bool bulk_ok(size_t n_values) const {
    static const int min_clients = [] {
      const char *e = getenv("ZKFHE_X8_MIN");
      return e ? atoi(e) : 2;
    }();
    return kind == TR_POSEIDON && n_values >= 16 && pos::hash_mode().load(std::memory_order_relaxed) == 1 && pos::x8_available() &&
           pos::bulk_clients().load(std::memory_order_relaxed) >= min_clients;
  }
  void absorb_scalar(const U256 &s) {
    if (kind == TR_POSEIDON) {
      sp.update(s);
      return;
    }
    uint8_t b[33];
    b[0] = 2;
    memcpy(b + 1, s.l, 32);
    h.update(b, 33);
  }
  static U256 fq_to_fr(U256 v) {  // fe_to_fe::<Fq, Fr>: the integer value mod r (q < 2 r)
    if (!(v < fe::MOD)) fe::sub_raw(v, v, fe::MOD);
    return v;
  }
};

}  // namespace zkhost
