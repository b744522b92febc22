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
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
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

}  // namespace zkhost
