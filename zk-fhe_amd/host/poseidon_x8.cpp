// Eight Poseidon sponges in lockstep, one per 64-bit lane of an AVX-512 register (poseidon.hpp has the permutation, the sponge
// framing and the references; oracle/poseidon_ref.py is the checker).
//
// Why: the Fiat-Shamir sponge of ONE proof is a sequential chain (2 561 permutations for the 5 121 public inputs of a k = 13
// proof, examples/bfv.rs:118-122, before the first challenge), but a prover that keeps 16-20 proofs in flight has that many
// independent chains.  poseidon_ifma.cpp spreads one permutation over the lanes (4.7 us, bounded by the S-box chain); here a
// lane IS a sponge: every product of the permutation -- S-boxes included -- is one eight-lane Montgomery product in radix 2^52
// (vpmadd52luq / vpmadd52huq), so eight permutations cost about what 1.5 cost there.  Lanes are independent: a sponge joins a
// free lane between two permutations and leaves when its input is used up (HashService below), nobody waits for a full group.
//
// Arithmetic.  A value is five 52-bit limbs, Montgomery form with M = 2^260: lanes hold x * 2^260 mod r, not reduced below r --
// mont(a, b) = a b / M mod r < a b / M + r, and M / r > 84, so with operands below 35 r the result is below 16 r and nothing is ever
// compared with r.  Limbs are "normalised" (below 2^52, what vpmadd52 reads) after every product or sum that feeds a product.
// The bounds are stated at each step; tests/test_poseidon.py runs ragged batches against the scalar sponge and the oracle.
#include <immintrin.h>

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <thread>

#include "poseidon.hpp"

#define ZK_IFMA __attribute__((target("avx512f,avx512ifma,avx512vl,avx512dq,avx512bw")))
#define ZK_INL __attribute__((always_inline)) inline

namespace zkhost {
namespace pos {

namespace {

const uint64_t M52 = ((uint64_t)1 << 52) - 1;

struct C5 {   // a constant: five normalised limbs of c * 2^260 mod r (or of a plain integer)
  uint64_t l[5];
};
inline C5 limbs_of(const uint64_t v[4]) {
  C5 r;
  r.l[0] = v[0] & M52;
  r.l[1] = ((v[0] >> 52) | (v[1] << 12)) & M52;
  r.l[2] = ((v[1] >> 40) | (v[2] << 24)) & M52;
  r.l[3] = ((v[2] >> 28) | (v[3] << 36)) & M52;
  r.l[4] = v[3] >> 16;
  return r;
}
inline void pack_limbs(const uint64_t l[5], uint64_t v[4]) {   // normalised limbs, value below 2^256
  v[0] = l[0] | (l[1] << 52);
  v[1] = (l[1] >> 12) | (l[2] << 40);
  v[2] = (l[2] >> 24) | (l[3] << 28);
  v[3] = (l[3] >> 36) | (l[4] << 16);
}

struct X8Tables {
  C5 rc[ROUNDS][T];      // full rounds: plain round constants, c * M
  C5 mds[T][T], pre[T][T];
  C5 pc[R_P + 1][T];     // partial rounds: D_r c_r; pc[R_P] = 0
  C5 s_row[R_P][T], s_col[R_P][T - 1];
  C5 p, one_m;           // r; M mod r (mont(v, one_m) = v: the refresh of the running Y, Z)
  C5 in_canon;           // M^2 mod r: canonical integer x -> x M
  C5 in_256;             // 2^264 mod r: x 2^256 -> x M
  C5 out_256;            // 2^256 mod r: x M -> x 2^256
  uint64_t inv;          // -r^-1 mod 2^52
};

const X8Tables &x8_tables() {
  static const X8Tables Tb = [] {
    X8Tables t;
    memset(&t, 0, sizeof(t));
    const Constants &c = constants();
    const F sixteen = from_canon(U256{{16, 0, 0, 0}});
    auto cm = [&](const F &v) {   // v = c 2^256 (canonical) -> limbs of c 2^260 mod r
      const F w = mul(v, sixteen);
      return limbs_of(w.l);
    };
    for (int r = 0; r < ROUNDS; ++r)
      for (int i = 0; i < T; ++i) t.rc[r][i] = cm(c.rc[r][i]);
    for (int i = 0; i < T; ++i)
      for (int j = 0; j < T; ++j) t.mds[i][j] = cm(c.mds[i][j]), t.pre[i][j] = cm(c.pre[i][j]);
    for (int r = 0; r < R_P; ++r) {
      for (int i = 0; i < T; ++i) t.pc[r][i] = cm(c.pc[r][i]), t.s_row[r][i] = cm(c.s_row[r][i]);
      for (int i = 0; i < T - 1; ++i) t.s_col[r][i] = cm(c.s_col[r][i]);
    }
    t.p = limbs_of(P);
    // powers of two mod r as plain integers: to_canon(from_canon(2^a) * from_canon(2^b)) = 2^(a+b) mod r
    auto pow2 = [&](int e) {
      F acc = ONE;   // 1 in the 2^256 form
      const F two = from_canon(U256{{2, 0, 0, 0}});
      for (int i = 0; i < e; ++i) acc = mul(acc, two);
      const U256 v = to_canon(acc);
      uint64_t w[4];
      memcpy(w, v.l, 32);
      return limbs_of(w);
    };
    t.one_m = pow2(260);
    t.in_canon = pow2(520);
    t.in_256 = pow2(264);
    t.out_256 = pow2(256);
    uint64_t x = 1;
    for (int i = 0; i < 6; ++i) x *= 2 - P[0] * x;   // r^-1 mod 2^64
    t.inv = (0 - x) & M52;
    return t;
  }();
  return Tb;
}

struct V {   // eight values, limb j of all of them in l[j]
  __m512i l[5];
};

struct K {   // loop invariants
  __m512i zero, mask, inv, p[5];
};

ZK_IFMA ZK_INL __m512i bc(uint64_t v) { return _mm512_set1_epi64((long long)v); }

// carry propagation: limbs below 2^63 in, below 2^52 out (value below 2^260)
ZK_IFMA ZK_INL void normalise(V &v, const K &k) {
  __m512i c = _mm512_srli_epi64(v.l[0], 52);
  v.l[0] = _mm512_and_si512(v.l[0], k.mask);
  v.l[1] = _mm512_add_epi64(v.l[1], c);
  c = _mm512_srli_epi64(v.l[1], 52);
  v.l[1] = _mm512_and_si512(v.l[1], k.mask);
  v.l[2] = _mm512_add_epi64(v.l[2], c);
  c = _mm512_srli_epi64(v.l[2], 52);
  v.l[2] = _mm512_and_si512(v.l[2], k.mask);
  v.l[3] = _mm512_add_epi64(v.l[3], c);
  c = _mm512_srli_epi64(v.l[3], 52);
  v.l[3] = _mm512_and_si512(v.l[3], k.mask);
  v.l[4] = _mm512_add_epi64(v.l[4], c);
}

struct Acc {
  __m512i t0, t1, t2, t3, t4, t5;
};

// acc += a * b_i (a normalised, b_i one normalised limb in every lane)
ZK_IFMA ZK_INL void acc_mul(Acc &t, const V &a, __m512i bi) {
  t.t0 = _mm512_madd52lo_epu64(t.t0, a.l[0], bi);
  t.t1 = _mm512_madd52lo_epu64(t.t1, a.l[1], bi);
  t.t2 = _mm512_madd52lo_epu64(t.t2, a.l[2], bi);
  t.t3 = _mm512_madd52lo_epu64(t.t3, a.l[3], bi);
  t.t4 = _mm512_madd52lo_epu64(t.t4, a.l[4], bi);
  t.t1 = _mm512_madd52hi_epu64(t.t1, a.l[0], bi);
  t.t2 = _mm512_madd52hi_epu64(t.t2, a.l[1], bi);
  t.t3 = _mm512_madd52hi_epu64(t.t3, a.l[2], bi);
  t.t4 = _mm512_madd52hi_epu64(t.t4, a.l[3], bi);
  t.t5 = _mm512_madd52hi_epu64(t.t5, a.l[4], bi);
}
// one Montgomery reduction step: acc = (acc + m r) / 2^52.  An accumulator limb collects at most eight 52-bit terms per step
// (dot3) and lives five steps: below 2^58.
ZK_IFMA ZK_INL void acc_reduce(Acc &t, const K &k) {
  const __m512i m = _mm512_madd52lo_epu64(k.zero, t.t0, k.inv);
  t.t0 = _mm512_madd52lo_epu64(t.t0, m, k.p[0]);
  t.t1 = _mm512_madd52lo_epu64(t.t1, m, k.p[1]);
  t.t2 = _mm512_madd52lo_epu64(t.t2, m, k.p[2]);
  t.t3 = _mm512_madd52lo_epu64(t.t3, m, k.p[3]);
  t.t4 = _mm512_madd52lo_epu64(t.t4, m, k.p[4]);
  t.t1 = _mm512_madd52hi_epu64(t.t1, m, k.p[0]);
  t.t2 = _mm512_madd52hi_epu64(t.t2, m, k.p[1]);
  t.t3 = _mm512_madd52hi_epu64(t.t3, m, k.p[2]);
  t.t4 = _mm512_madd52hi_epu64(t.t4, m, k.p[3]);
  t.t5 = _mm512_madd52hi_epu64(t.t5, m, k.p[4]);
  t.t0 = _mm512_add_epi64(t.t1, _mm512_srli_epi64(t.t0, 52));
  t.t1 = t.t2, t.t2 = t.t3, t.t3 = t.t4, t.t4 = t.t5, t.t5 = k.zero;
}
ZK_IFMA ZK_INL Acc acc_zero(const K &k) { return Acc{k.zero, k.zero, k.zero, k.zero, k.zero, k.zero}; }
ZK_IFMA ZK_INL V acc_out(const Acc &t) {
  V v;
  v.l[0] = t.t0, v.l[1] = t.t1, v.l[2] = t.t2, v.l[3] = t.t3, v.l[4] = t.t4;
  return v;
}

// a * b / M, limbs NOT normalised (callers add to it first, or normalise)
ZK_IFMA ZK_INL V mont_raw(const V &a, const V &b, const K &k) {
  Acc t = acc_zero(k);
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    acc_mul(t, a, b.l[i]);
    acc_reduce(t, k);
  }
  return acc_out(t);
}
ZK_IFMA ZK_INL V mont(const V &a, const V &b, const K &k) {
  V v = mont_raw(a, b, k);
  normalise(v, k);
  return v;
}
ZK_IFMA ZK_INL V mont_c_raw(const V &a, const C5 &c, const K &k) {   // by a constant
  Acc t = acc_zero(k);
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    acc_mul(t, a, bc(c.l[i]));
    acc_reduce(t, k);
  }
  return acc_out(t);
}
ZK_IFMA ZK_INL V mont_c(const V &a, const C5 &c, const K &k) {
  V v = mont_c_raw(a, c, k);
  normalise(v, k);
  return v;
}
// (a0 c0 + a1 c1 + a2 c2) / M with ONE reduction, normalised
ZK_IFMA ZK_INL V dot3_c(const V &a0, const C5 &c0, const V &a1, const C5 &c1, const V &a2, const C5 &c2, const K &k) {
  Acc t = acc_zero(k);
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    acc_mul(t, a0, bc(c0.l[i]));
    acc_mul(t, a1, bc(c1.l[i]));
    acc_mul(t, a2, bc(c2.l[i]));
    acc_reduce(t, k);
  }
  V v = acc_out(t);
  normalise(v, k);
  return v;
}
ZK_IFMA ZK_INL V add_c(const V &a, const C5 &c) {   // limb-wise, not normalised
  V v;
#pragma unroll
  for (int j = 0; j < 5; ++j) v.l[j] = _mm512_add_epi64(a.l[j], bc(c.l[j]));
  return v;
}
ZK_IFMA ZK_INL V add_v(const V &a, const V &b) {
  V v;
#pragma unroll
  for (int j = 0; j < 5; ++j) v.l[j] = _mm512_add_epi64(a.l[j], b.l[j]);
  return v;
}
// u (normalised, below 36 r) -> u^5: below 15.5 r, below 3.9 r, below 2.7 r
ZK_IFMA ZK_INL V pow5(const V &u, const K &k) {
  const V u2 = mont(u, u, k);
  const V u4 = mont(u2, u2, k);
  return mont(u4, u, k);
}

ZK_IFMA ZK_INL void full_round_x8(V s[T], const C5 rc[T], const C5 (*m)[T], const K &k) {
  V v[T];
#pragma unroll
  for (int i = 0; i < T; ++i) {
    V u = add_c(s[i], rc[i]);
    normalise(u, k);
    v[i] = pow5(u, k);
  }
#pragma unroll
  for (int i = 0; i < T; ++i) s[i] = dot3_c(v[0], m[i][0], v[1], m[i][1], v[2], m[i][2], k);   // below (3 * 2.7 / 84 + 1) r
}

// One permutation of the eight states.  In: normalised, below 4 r.  Out: normalised, below 1.2 r.
ZK_IFMA void permute_x8(V s[T], const X8Tables &Tb, const K &k) {
  const int half = R_F / 2;
  for (int r = 0; r < half; ++r) full_round_x8(s, Tb.rc[r], r == half - 1 ? Tb.pre : Tb.mds, k);
  // partial rounds, sparse form: x = (s0 + pc0)^5; s0' = row . (x, Y, Z); Y' = col0 x + Y + pc1' (same for Z), where the running
  // Y, Z already carry the NEXT round's constant.  They grow by about 2 r a round and are refreshed (times 1) every 16 rounds:
  // below 1.5 r + 16 * 2.1 r < 36 r, so s0' stays below (1.1 + 2 * 36) / 84 r + r < 2 r.
  V Y = add_c(s[1], Tb.pc[0][1]), Z = add_c(s[2], Tb.pc[0][2]);
  normalise(Y, k);
  normalise(Z, k);
  V s0 = s[0];
  for (int t = 0; t < R_P; ++t) {
    V u = add_c(s0, Tb.pc[t][0]);
    normalise(u, k);
    const V x = pow5(u, k);
    s0 = dot3_c(x, Tb.s_row[t][0], Y, Tb.s_row[t][1], Z, Tb.s_row[t][2], k);
    V y2 = add_v(mont_c_raw(x, Tb.s_col[t][0], k), add_c(Y, Tb.pc[t + 1][1]));
    V z2 = add_v(mont_c_raw(x, Tb.s_col[t][1], k), add_c(Z, Tb.pc[t + 1][2]));
    normalise(y2, k);
    normalise(z2, k);
    if ((t & 15) == 15) {
      y2 = mont_c(y2, Tb.one_m, k);
      z2 = mont_c(z2, Tb.one_m, k);
    }
    Y = y2, Z = z2;
  }
  s[0] = s0, s[1] = Y, s[2] = Z;   // pc[R_P] = 0; Y, Z below 20 r: fine for the S-box of the next full round
  for (int r = half + R_P; r < ROUNDS; ++r) full_round_x8(s, Tb.rc[r], Tb.mds, k);
}

ZK_IFMA K make_k(const X8Tables &Tb) {
  K k;
  k.zero = _mm512_setzero_si512();
  k.mask = bc(M52);
  k.inv = bc(Tb.inv);
  for (int j = 0; j < 5; ++j) k.p[j] = bc(Tb.p.l[j]);
  return k;
}

struct alignas(64) Buf8 {
  uint64_t l[5][8];
};
ZK_IFMA ZK_INL V load_v(const Buf8 &b) {
  V v;
  for (int j = 0; j < 5; ++j) v.l[j] = _mm512_load_si512((const void *)b.l[j]);
  return v;
}
ZK_IFMA ZK_INL void store_v(Buf8 &b, const V &v) {
  for (int j = 0; j < 5; ++j) _mm512_store_si512((void *)b.l[j], v.l[j]);
}
ZK_IFMA ZK_INL V blend_v(__mmask8 m, const V &a, const V &b) {   // lanes of m from b
  V v;
  for (int j = 0; j < 5; ++j) v.l[j] = _mm512_mask_blend_epi64(m, a.l[j], b.l[j]);
  return v;
}

bool cpu_has_ifma() {
  static const bool ok = [] {
    __builtin_cpu_init();
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512ifma") && __builtin_cpu_supports("avx512vl") &&
           __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512bw");
  }();
  return ok;
}

// ---- the lockstep engine: up to eight jobs, each in its lane, every one at its own position -----------------------------------
struct Lanes {
  const X8Tables &Tb;
  K k;
  V s[T];
  AbsorbJob *job[8];
  size_t pos[8];   // next pair of the lane's job
  int n_active = 0;

  ZK_IFMA Lanes() : Tb(x8_tables()), k(make_k(Tb)) {
    for (int i = 0; i < T; ++i)
      for (int j = 0; j < 5; ++j) s[i].l[j] = k.zero;
    for (int l = 0; l < 8; ++l) job[l] = nullptr, pos[l] = 0;
  }
  bool has_free() const { return n_active < 8; }

  // puts jobs into free lanes (all of them at once: one product per state word for any number of joiners)
  ZK_IFMA void join(AbsorbJob *const *jobs, int n) {
    Buf8 b[T];
    memset(b, 0, sizeof(b));
    __mmask8 m = 0;
    int l = 0;
    for (int q = 0; q < n; ++q) {
      while (job[l]) ++l;
      job[l] = jobs[q];
      pos[l] = 0;
      ++n_active;
      m |= (__mmask8)(1u << l);
      for (int i = 0; i < T; ++i) {
        const C5 c = limbs_of(jobs[q]->st[i].l);
        for (int j = 0; j < 5; ++j) b[i].l[j][l] = c.l[j];
      }
    }
    for (int i = 0; i < T; ++i) s[i] = blend_v(m, s[i], mont_c(load_v(b[i]), Tb.in_256, k));   // x 2^256 (below 2 r) -> x M, below 1.1 r
  }

  // one permutation for every active lane; returns the lanes' jobs that are finished (their state written back)
  ZK_IFMA int step(AbsorbJob *done[8]) {
    Buf8 in[2];
    memset(in, 0, sizeof(in));
    for (int l = 0; l < 8; ++l) {
      if (!job[l]) continue;
      const U256 *d = job[l]->data + 2 * pos[l];
      for (int e = 0; e < 2; ++e) {
        const C5 c = limbs_of(d[e].l);
        for (int j = 0; j < 5; ++j) in[e].l[j][l] = c.l[j];
      }
    }
    // canonical x (below r) -> x M, below 1.02 r; idle lanes absorb zeros
    V a = add_v(s[1], mont_c_raw(load_v(in[0]), Tb.in_canon, k)), b = add_v(s[2], mont_c_raw(load_v(in[1]), Tb.in_canon, k));
    normalise(a, k);
    normalise(b, k);
    s[1] = a, s[2] = b;   // below 1.2 r + 1.02 r
    permute_x8(s, Tb, k);
    int nd = 0;
    __mmask8 fin = 0;
    for (int l = 0; l < 8; ++l)
      if (job[l] && ++pos[l] == job[l]->n_pairs) fin |= (__mmask8)(1u << l);
    if (fin) {
      Buf8 o[T];
      for (int i = 0; i < T; ++i) store_v(o[i], mont_c(s[i], Tb.out_256, k));   // x M -> x 2^256, below 1.02 r: the scalar code's weak form
      for (int l = 0; l < 8; ++l) {
        if (!(fin & (1u << l))) continue;
        for (int i = 0; i < T; ++i) {
          const uint64_t q[5] = {o[i].l[0][l], o[i].l[1][l], o[i].l[2][l], o[i].l[3][l], o[i].l[4][l]};
          pack_limbs(q, job[l]->st[i].l);
        }
        done[nd++] = job[l];
        job[l] = nullptr;
        --n_active;
      }
    }
    return nd;
  }
};

// ---- the service: worker threads that own a Lanes each ---------------------------------------------------------------------------
class HashService {
 public:
  static HashService &get() {
    static HashService *s = new HashService();   // never destroyed: detached workers may outlive static destruction
    return *s;
  }
  // Packing first: a job waits (at most one permutation, ~10 us) for a running worker with a free lane; an idle worker is woken,
  // or a new one started, only when the running ones cannot take the whole queue.
  void submit(AbsorbJob *j) {
    bool wake = false;
    {
      std::lock_guard<std::mutex> g(mu);
      queue.push_back(j);
      n_queued.store(queue.size(), std::memory_order_release);
      size_t room = 0;
      for (int w = 0; w < n_threads; ++w) room += (size_t)free_of[w].load(std::memory_order_relaxed);
      if (room < queue.size()) {
        if (idle > 0) {
          wake = true;
        } else if (n_threads < max_threads) {
          const int w = n_threads++;
          std::thread([this, w] { run(w); }).detach();
        }
      }
    }
    if (wake) cv.notify_one();
  }

 private:
  static const int MAX_WORKERS = 16;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<AbsorbJob *> queue;
  std::atomic<size_t> n_queued{0};
  std::atomic<int> free_of[MAX_WORKERS];   // free lanes of a worker that is stepping (0 while it sleeps)
  int n_threads = 0, idle = 0, max_threads = 3;

  HashService() {
    for (auto &f : free_of) f.store(0);
    if (const char *e = getenv("ZKFHE_HASH_THREADS")) max_threads = std::min(MAX_WORKERS, std::max(1, atoi(e)));
  }
  int take(AbsorbJob **out, int max_n) {   // mu held
    int n = 0;
    while (n < max_n && !queue.empty()) {
      out[n++] = queue.front();
      queue.pop_front();
    }
    n_queued.store(queue.size(), std::memory_order_release);
    return n;
  }
  ZK_IFMA void run(int w) {
    Lanes L;
    AbsorbJob *got[8], *done[8];
    for (;;) {
      int n = 0;
      if (L.n_active == 0) {
        std::unique_lock<std::mutex> g(mu);
        free_of[w].store(0, std::memory_order_relaxed);
        ++idle;
        cv.wait(g, [&] { return !queue.empty(); });
        --idle;
        n = take(got, 8);
      } else if (L.has_free() && n_queued.load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> g(mu);
        n = take(got, 8 - L.n_active);
      }
      if (n) L.join(got, n);
      free_of[w].store(8 - L.n_active, std::memory_order_relaxed);
      const int nd = L.step(done);
      for (int i = 0; i < nd; ++i) done[i]->finish();
    }
  }
};

}  // namespace

bool x8_available() {
  static const bool ok = cpu_has_ifma() && !getenv("ZKFHE_POSEIDON_SCALAR") && !(getenv("ZKFHE_POSEIDON_X8") && getenv("ZKFHE_POSEIDON_X8")[0] == '0');
  return ok;
}

void AbsorbJob::finish() {
  std::lock_guard<std::mutex> g(mu);   // notified under the lock: the waiter may destroy the job as soon as it sees `done`
  done = true;
  cv.notify_all();
}
void AbsorbJob::wait() {
  std::unique_lock<std::mutex> g(mu);
  cv.wait(g, [&] { return done; });
  done = false;   // the job object is reused by the transcript's next run
}

void x8_submit(AbsorbJob *j) {
  if (j->n_pairs == 0) {
    j->finish();
    return;
  }
  HashService::get().submit(j);
}

// all jobs on the calling thread, eight at a time, lanes refilled as they run dry (tests, and callers that batch themselves)
ZK_IFMA static void absorb_inline(AbsorbJob *const *jobs, size_t n) {
  Lanes L;
  AbsorbJob *done[8];
  size_t next = 0;
  while (next < n && jobs[next]->n_pairs == 0) jobs[next++]->finish();
  while (next < n || L.n_active) {
    AbsorbJob *got[8];
    int g = 0;
    while (L.n_active + g < 8 && next < n) {
      if (jobs[next]->n_pairs) got[g++] = jobs[next];
      else jobs[next]->finish();
      ++next;
    }
    if (g) L.join(got, g);
    if (!L.n_active) break;
    const int nd = L.step(done);
    for (int i = 0; i < nd; ++i) done[i]->finish();
  }
}
bool x8_absorb_now(AbsorbJob *const *jobs, size_t n) {
  if (!cpu_has_ifma()) return false;
  absorb_inline(jobs, n);
  return true;
}

}  // namespace pos
}  // namespace zkhost
