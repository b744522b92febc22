// The transcript mid-state of the per-public-key prefix cache (host/prefix_cache.hpp, host/transcript.hpp State / restore /
// common_scalars_async_marked): a transcript that restores the state behind `digest | key values` and absorbs the rest must
// squeeze what a transcript that absorbed everything squeezes -- both hashers, odd and even prefix lengths (the Poseidon sponge
// carries a half-filled chunk across the cut) -- and the cache must hit on equal keys only, evict the least recently used entry and
// keep entries of two interleaved keys apart.  usage: transcript_prefix_check
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <stdexcept>
#include <thread>
#include <vector>

#include "prefix_cache.hpp"
using namespace zkhost;

static U256 val(uint64_t seed, uint64_t i) {   // some canonical field element
  U256 v;
  uint64_t x = seed * 0x9e3779b97f4a7c15ULL + i * 0xbf58476d1ce4e5b9ULL + 1;
  for (int j = 0; j < 4; ++j) {
    x ^= x >> 31;
    x *= 0x94d049bb133111ebULL;
    v.l[j] = x;
  }
  v.l[3] &= 0x0fffffffffffffffULL;   // < 2^252 < r
  return v;
}

static int fails = 0;
#define CHECK(c)                                                   \
  do {                                                             \
    if (!(c)) {                                                    \
      printf("FAIL line %d: %s\n", __LINE__, #c);                  \
      ++fails;                                                     \
    }                                                              \
  } while (0)

static bool same(const U256 &a, const U256 &b) { return memcmp(a.l, b.l, 32) == 0; }

int main() {
  for (uint32_t kind : {(uint32_t)TR_POSEIDON, (uint32_t)TR_BLAKE2B})
    for (size_t n_key : {(size_t)0, (size_t)1, (size_t)16, (size_t)17, (size_t)2048}) {
      const size_t n_all = n_key + 3 * 1024 + 1;
      std::vector<U256> inst(n_all);
      for (size_t i = 0; i < n_all; ++i) inst[i] = val(kind + 7, i);
      const U256 digest = val(99, 0);
      // (a) everything absorbed by one transcript
      Transcript a(kind);
      a.common_scalar(digest);
      a.common_scalars_async(inst);
      const U256 ca = a.squeeze(), ca2 = a.squeeze();
      // (b) the marked run: same challenges, and the callback sees the state behind the prefix
      Transcript b(kind);
      Transcript::State mid;
      bool called = false;
      b.common_scalar(digest);
      b.common_scalars_async_marked(inst, n_key, [&](const Transcript::State &s) {
        mid = s;
        called = true;
      });
      const U256 cb = b.squeeze(), cb2 = b.squeeze();
      CHECK(called && same(ca, cb) && same(ca2, cb2));
      // (c) a fresh transcript restores it and absorbs the rest
      Transcript c(kind);
      c.common_scalar(val(1234, 5));   // whatever it held is replaced
      c.restore(mid);
      c.common_scalars_async(std::vector<U256>(inst.begin() + (long)n_key, inst.end()));
      const U256 cc = c.squeeze(), cc2 = c.squeeze();
      CHECK(same(ca, cc) && same(ca2, cc2));
      // (d) snapshot() of a transcript fed by hand equals the callback's state: restoring it gives the same challenges
      Transcript d(kind), e(kind);
      d.common_scalar(digest);
      for (size_t i = 0; i < n_key; ++i) d.common_scalar(inst[i]);
      e.restore(d.snapshot());
      for (size_t i = n_key; i < n_all; ++i) e.common_scalar(inst[i]);
      CHECK(same(ca, e.squeeze()));
      // a different prefix gives a different challenge (the check is not vacuous)
      if (n_key) {
        Transcript f(kind);
        f.common_scalar(digest);
        std::vector<U256> other = inst;
        other[n_key - 1] = val(5, 5);
        f.common_scalars_async(other);
        CHECK(!same(ca, f.squeeze()));
      }
    }
  // the cache: equal keys hit, one changed word misses, least recently used goes first, capacity 0 = off
  {
    PrefixCache pc;
    pc.set_capacity(2);
    std::vector<U256> k1(2048), k2(2048), k3(2048);
    for (size_t i = 0; i < 2048; ++i) k1[i] = val(1, i), k2[i] = val(2, i), k3[i] = val(3, i);
    Transcript t1(TR_POSEIDON), t2(TR_POSEIDON), t3(TR_POSEIDON);
    t1.common_scalar(val(1, 0)), t2.common_scalar(val(2, 0)), t3.common_scalar(val(3, 0));
    Transcript::State got;
    CHECK(!pc.lookup(k1.data(), 2048, got));
    pc.insert(k1.data(), 2048, t1.snapshot());
    pc.insert(k1.data(), 2048, t2.snapshot());   // a second insert of the same key is ignored
    pc.insert(k2.data(), 2048, t2.snapshot());
    CHECK(pc.lookup(k1.data(), 2048, got));
    {
      Transcript r(TR_POSEIDON);
      r.restore(got);
      CHECK(same(r.squeeze(), t1.squeeze()));
    }
    CHECK(pc.lookup(k2.data(), 2048, got));
    CHECK(pc.lookup(k1.data(), 2048, got));       // k1 is now the most recently used
    pc.insert(k3.data(), 2048, t3.snapshot());    // evicts k2
    CHECK(!pc.lookup(k2.data(), 2048, got));
    CHECK(pc.lookup(k1.data(), 2048, got) && pc.lookup(k3.data(), 2048, got));
    std::vector<U256> k1b = k1;
    k1b[2047].l[0] ^= 1;
    CHECK(!pc.lookup(k1b.data(), 2048, got));
    CHECK(!pc.lookup(k1.data(), 2047, got));      // a shorter key is another key
    uint64_t h, m, e;
    pc.stats(&h, &m, &e);
    CHECK(h == 5 && m == 4 && e == 2);
    pc.set_capacity(0);
    CHECK(!pc.lookup(k1.data(), 2048, got));
    pc.stats(&h, &m, &e);
    CHECK(e == 0 && h == 5 && m == 4);            // switched off: neither a hit nor a miss
  }
  // announced proofs (PreHash): the parked state continues like a transcript that absorbed digest | inputs itself; an entry serves one
  // proof, the oldest announcement of a text first; another text misses; public inputs that differ from the helper's are refused; a
  // failing helper is a miss and its entry is dropped; the queue is bounded
  for (uint32_t kind : {(uint32_t)TR_POSEIDON, (uint32_t)TR_BLAKE2B}) {
    std::vector<U256> in1(5121), in2(5121);   // before `ph`: its helper threads read them until its destructor has joined them
    for (size_t i = 0; i < 5121; ++i) in1[i] = val(11 + kind, i), in2[i] = val(12 + kind, i);
    PreHash ph;
    const U256 digest = val(77, 1);
    // "parsing": text "1..." -> in1, "2..." -> in2, anything else does not parse
    const PreHash::Work work = [&, kind, digest](const std::string &text, std::vector<U256> &inst, Transcript::State &out) {
      if (text[0] == '1') inst = in1;
      else if (text[0] == '2') inst = in2;
      else throw std::runtime_error("does not parse");
      Transcript t(kind);
      t.common_scalar(digest);
      t.common_scalars_async(inst);
      out = t.snapshot();
    };
    Transcript ref(kind);
    ref.common_scalar(digest);
    ref.common_scalars_async(in1);
    ref.common_scalar(val(5, 9));
    const U256 want = ref.squeeze();
    Transcript::State st;
    CHECK(!ph.take("1a", 2, in1.data(), in1.size(), st));
    CHECK(ph.start("1a", 2, work) && ph.start("1a", 2, work) && ph.start("2a", 2, work));
    CHECK(!ph.take("1b", 2, in1.data(), in1.size(), st) && !ph.take("1", 1, in1.data(), in1.size(), st));   // other texts
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(ph.take("1a", 2, in1.data(), in1.size(), st));
      Transcript t(kind);
      t.common_scalar(val(3, 3));   // replaced by the restore
      t.restore(st);
      t.common_scalar(val(5, 9));
      CHECK(same(t.squeeze(), want));
    }
    CHECK(!ph.take("1a", 2, in1.data(), in1.size(), st));   // both announcements are used up
    CHECK(!ph.take("2a", 2, in1.data(), in1.size(), st));   // the helper derived other public inputs than the proof: refused (and consumed)
    uint64_t s0, t0, p0;
    ph.stats(&s0, &t0, &p0);
    CHECK(s0 == 3 && t0 == 2 && p0 == 0);
    CHECK(ph.start("xx", 2, work));
    CHECK(!ph.take("xx", 2, in1.data(), in1.size(), st));   // a failed helper: the proof hashes for itself
    CHECK(ph.start("yy", 2, work));                         // fails in the background ...
    std::this_thread::sleep_for(std::chrono::milliseconds(200));
    for (size_t i = 0; i < PreHash::MAX_PENDING; ++i) CHECK(ph.start("2a", 2, work));   // ... and is dropped by the next announcement
    CHECK(!ph.start("2a", 2, work));                         // MAX_PENDING entries are waiting
  }
  printf("transcript prefix check: %s\n", fails ? "FAILED" : "ok");
  return fails ? 1 : 0;
}
