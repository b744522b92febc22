// Per-key mid-state of the Fiat-Shamir transcript for the public-input prefix.
//
// The instance column of the BFV circuit is pk0 | pk1 | c0 | c1 | cyclo (examples/bfv.rs:118-122: the order of the five
// to_public calls), absorbed scalar by scalar right after the verifying-key digest.  pk0 and pk1 -- the first 2 N of the
// 5 N + 1 values, 40 % -- are the encryptor's PUBLIC KEY: every encryption under one key starts its transcript with the same
// 1 + 2 N absorptions (k = 13: 1 024 of the 2 561 sequential Poseidon permutations that stand between a proof and its first
// challenge).  The state of the transcript after `vk digest | pk0 | pk1` is therefore kept per (proving key, public key): a
// proof whose pk0 | pk1 equal a cached entry's restores that state and absorbs c0 | c1 | cyclo from there.  Same state, hence
// the same challenges and the same proof bytes, by construction; tests/test_gpu_prover.py proves hit == miss byte for byte and
// interleaves two keys.  Nothing beyond the public key is cached: c0, c1 differ with every encryption.
//
// The cache belongs to one zkfhe_bfv_pk (fixed vk digest, fixed transcript kind): the key of an entry is the 2 N canonical
// values themselves (compared in full -- a fingerprint only selects the candidate), a few entries, least recently used out.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <thread>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "transcript.hpp"

namespace zkhost {

class PrefixCache {
 public:
  PrefixCache() {
    if (const char *e = getenv("ZKFHE_PREFIX_CACHE")) {
      // a number in [0, 64]; anything else (negative, garbage, empty) leaves the default -- never "-1 = the maximum"
      char *end = nullptr;
      const long v = strtol(e, &end, 10);
      if (end != e && *end == 0 && v >= 0) capacity_ = v > 64 ? 64 : (size_t)v;
    }
  }
  static uint64_t fingerprint(const U256 *v, size_t n) {
    uint64_t h = 0x9e3779b97f4a7c15ULL ^ (uint64_t)n;
    for (size_t i = 0; i < n; ++i)
      for (int j = 0; j < 4; ++j) {
        h ^= v[i].l[j];
        h *= 0xff51afd7ed558ccdULL;
        h ^= h >> 29;
      }
    return h;
  }
  // the state after vk digest | key values, if this key was seen
  bool lookup(const U256 *key, size_t n, Transcript::State &out) {
    if (!capacity()) return false;
    const uint64_t fp = fingerprint(key, n);
    std::shared_ptr<const Entry> hit;
    {
      std::lock_guard<std::mutex> l(mu_);
      for (auto &e : entries_)
        if (e->fp == fp && e->key.size() == n) {
          hit = e;
          e->stamp = ++clock_;
          break;
        }
    }
    if (hit && memcmp(hit->key.data(), key, n * sizeof(U256)) == 0) {   // compared outside the lock: entries are immutable
      out = hit->st;
      std::lock_guard<std::mutex> l(mu_);
      ++hits_;
      return true;
    }
    std::lock_guard<std::mutex> l(mu_);
    ++misses_;
    return false;
  }
  void insert(const U256 *key, size_t n, const Transcript::State &st) {
    if (!capacity()) return;
    auto e = std::make_shared<Entry>();
    e->fp = fingerprint(key, n);
    e->key.assign(key, key + n);
    e->st = st;
    std::lock_guard<std::mutex> l(mu_);
    for (auto &o : entries_)
      if (o->fp == e->fp && o->key.size() == n && memcmp(o->key.data(), key, n * sizeof(U256)) == 0) return;   // two proofs missed on the same key at once
    e->stamp = ++clock_;
    if (entries_.size() >= capacity_) {
      size_t old = 0;
      for (size_t i = 1; i < entries_.size(); ++i)
        if (entries_[i]->stamp < entries_[old]->stamp) old = i;
      entries_[old] = e;
    } else {
      entries_.push_back(e);
    }
  }
  size_t capacity() {
    std::lock_guard<std::mutex> l(mu_);
    return capacity_;
  }
  void set_capacity(size_t c) {
    std::lock_guard<std::mutex> l(mu_);
    capacity_ = c > 64 ? 64 : c;
    while (entries_.size() > capacity_) {
      size_t old = 0;
      for (size_t i = 1; i < entries_.size(); ++i)
        if (entries_[i]->stamp < entries_[old]->stamp) old = i;
      entries_.erase(entries_.begin() + (long)old);
    }
  }
  void stats(uint64_t *hits, uint64_t *misses, uint64_t *entries) {
    std::lock_guard<std::mutex> l(mu_);
    if (hits) *hits = hits_;
    if (misses) *misses = misses_;
    if (entries) *entries = entries_.size();
  }

 private:
  struct Entry {
    uint64_t fp = 0, stamp = 0;
    std::vector<U256> key;
    Transcript::State st;
  };
  std::mutex mu_;
  std::vector<std::shared_ptr<Entry>> entries_;
  size_t capacity_ = 8;   // public keys remembered per proving key (ZKFHE_PREFIX_CACHE, zkfhe_bfv_pk_prefix_cache; 0 = off)
  uint64_t clock_ = 0, hits_ = 0, misses_ = 0;
};

// Transcript states of proofs ANNOUNCED ahead of time (zkfhe_bfv_pk_prehash).  The sponge over the 5 N + 1 public inputs is one
// sequential chain that depends on nothing but the public inputs -- k = 16: 10 241 Poseidon permutations, 30 ms on a core; k = 19:
// 40 961, 120 ms: as long as the proof's whole GPU work -- so a caller that knows its next input (a queue of encryptions to prove)
// announces it while the previous proof is on the GPU: a helper thread absorbs `vk digest | public inputs` (through the per-key
// prefix above) and parks the state here; the proof of that input picks it up (waiting for it if need be) instead of hashing.
// ONE-SHOT: an entry serves one proof and is gone -- every proof's public inputs are absorbed exactly once, just earlier.  Same
// state as absorbing inside the proof, hence the same bytes (tests/test_gpu_prover.py::test_announced_proofs_*).
class PreHash {
 public:
  static constexpr size_t MAX_PENDING = 16;
  // the helper's work: input text -> public inputs and the state behind `vk digest | public inputs`; throws on an input that does not parse
  typedef std::function<void(const std::string &, std::vector<U256> &, Transcript::State &)> Work;
  ~PreHash() {
    for (auto &e : entries_)
      if (e->th.joinable()) e->th.join();
  }
  // Returns at once: parsing and absorbing both happen on the helper thread.  false: MAX_PENDING announced proofs are waiting.
  bool start(const char *text, size_t len, Work work) {
    auto e = std::make_shared<Entry>();
    e->text.assign(text, len);
    {
      std::lock_guard<std::mutex> l(mu_);
      for (size_t i = 0; i < entries_.size();) {   // announcements whose input did not parse serve nobody: drop them
        bool dead;
        {
          std::lock_guard<std::mutex> le(entries_[i]->m);
          dead = entries_[i]->ready && entries_[i]->failed;
        }
        if (dead) {
          if (entries_[i]->th.joinable()) entries_[i]->th.join();
          entries_.erase(entries_.begin() + (long)i);
        } else {
          ++i;
        }
      }
      if (entries_.size() >= MAX_PENDING) return false;
      entries_.push_back(e);
      ++started_;
    }
    Entry *p = e.get();
    e->th = std::thread([p, work] {
      bool ok = true;
      try {
        work(p->text, p->inst, p->st);
      } catch (...) {
        ok = false;
      }
      std::lock_guard<std::mutex> l(p->m);
      p->failed = !ok;
      p->ready = true;
      p->cv.notify_all();
    });
    return true;
  }
  // The parked state for this input text, if one was announced: the OLDEST such announcement (two announcements of one text serve two
  // proofs, in order); waits until it is complete; the entry is consumed.  inst / n: the proof's own public inputs -- an entry whose
  // helper derived different ones (it cannot, but the state would be wrong) is not used.
  bool take(const char *text, size_t len, const U256 *inst, size_t n, Transcript::State &out) {
    std::shared_ptr<Entry> hit;
    {
      std::lock_guard<std::mutex> l(mu_);
      for (size_t i = 0; i < entries_.size(); ++i)
        if (entries_[i]->text.size() == len && memcmp(entries_[i]->text.data(), text, len) == 0) {
          hit = entries_[i];
          entries_.erase(entries_.begin() + (long)i);
          break;
        }
    }
    if (!hit) return false;
    {
      std::unique_lock<std::mutex> l(hit->m);
      hit->cv.wait(l, [&] { return hit->ready; });
    }
    if (hit->th.joinable()) hit->th.join();
    if (hit->failed || hit->inst.size() != n || memcmp(hit->inst.data(), inst, n * sizeof(U256)) != 0) return false;
    out = hit->st;
    std::lock_guard<std::mutex> l(mu_);
    ++taken_;
    return true;
  }
  void stats(uint64_t *started, uint64_t *taken, uint64_t *pending) {
    std::lock_guard<std::mutex> l(mu_);
    if (started) *started = started_;
    if (taken) *taken = taken_;
    if (pending) *pending = entries_.size();
  }

 private:
  struct Entry {
    std::string text;
    std::vector<U256> inst;
    Transcript::State st;
    bool ready = false, failed = false;
    std::mutex m;
    std::condition_variable cv;
    std::thread th;
  };
  std::mutex mu_;
  std::vector<std::shared_ptr<Entry>> entries_;
  uint64_t started_ = 0, taken_ = 0;
};

}  // namespace zkhost
