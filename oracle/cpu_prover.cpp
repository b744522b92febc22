/* CPU PROVER -- test infrastructure only (bench.py's `cpu_baseline` leg and tests/test_cpu_prover.py).
 * The product (zk-fhe_amd/) never includes, links or calls this file; it is the other way round: this is the build's
 * own host code (witness generation zk-fhe_amd/host/bfv_circuit.hpp, transcript zk-fhe_amd/host/transcript.hpp,
 * opening bookkeeping zk-fhe_amd/host/shplonk.hpp) driven through the whole proof on the host cores, with the field and
 * curve arithmetic of oracle/bn254_ref.h under OpenMP where the GPU prover launches kernels.
 *
 * What it restates: oracle/halo2_ref.py `prove` step for step (halo2_proofs `create_proof` + `ProverSHPLONK`, reached from
 * reference examples/bfv.rs:311), so that the CPU leg of bench.py is a multithreaded native prover and not Python
 * orchestration.  Same seed => the same proof bytes as oracle/halo2_ref.py and as the GPU prover
 * (tests/test_cpu_prover.py, bench.py).  The proving key (fixed / sigma polynomials, break points, vk digest) and the SRS
 * are handed in by the caller (oracle/halo2_ref.py keygen + make_srs): key generation is not part of the timed path.
 */
#include <omp.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "bn254_ref.h"

#include "../zk-fhe_amd/host/shplonk.hpp"
#include "../zk-fhe_amd/host/transcript.hpp"

using namespace zkhost;

namespace {

typedef fe_t F;

// Montgomery product, CIOS on four 64-bit limbs, fully unrolled (the modulus is below 2^254, so the running value fits
// four limbs plus the carry word).  oracle/bn254_ref.h's fe_mul (product, then a separate reduction) stays the reference
// formulation; this one is checked against it by the byte-identity of whole proofs.
typedef unsigned __int128 u128;
template <const field_t *Fd>
inline void mul_cios(fe_t *r, const fe_t *a, const fe_t *b) {
  const uint64_t *p = Fd->p;
  const uint64_t inv = Fd->inv;
  uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
  for (int i = 0; i < 4; ++i) {
    u128 c;
    const uint64_t bi = b->l[i];
    c = (u128)a->l[0] * bi + t0; t0 = (uint64_t)c;
    c = (u128)a->l[1] * bi + t1 + (uint64_t)(c >> 64); t1 = (uint64_t)c;
    c = (u128)a->l[2] * bi + t2 + (uint64_t)(c >> 64); t2 = (uint64_t)c;
    c = (u128)a->l[3] * bi + t3 + (uint64_t)(c >> 64); t3 = (uint64_t)c;
    t4 = (uint64_t)(c >> 64);
    const uint64_t m = t0 * inv;
    c = (u128)m * p[0] + t0;
    c = (u128)m * p[1] + t1 + (uint64_t)(c >> 64); t0 = (uint64_t)c;
    c = (u128)m * p[2] + t2 + (uint64_t)(c >> 64); t1 = (uint64_t)c;
    c = (u128)m * p[3] + t3 + (uint64_t)(c >> 64); t2 = (uint64_t)c;
    c = (u128)t4 + (uint64_t)(c >> 64); t3 = (uint64_t)c;
  }
  uint64_t t[4] = {t0, t1, t2, t3};
  if (limbs_geq(t, p)) limbs_sub(t, t, p);
  memcpy(r->l, t, 32);
}
inline void qmul(fe_t *r, const fe_t *a, const fe_t *b) { mul_cios<&FQ>(r, a, b); }
inline void qsqr(fe_t *r, const fe_t *a) { mul_cios<&FQ>(r, a, a); }

inline F fmul(const F &a, const F &b) { F r; mul_cios<&FR>(&r, &a, &b); return r; }
inline F fadd(const F &a, const F &b) { F r; fe_add(&FR, &r, &a, &b); return r; }
inline F fsub(const F &a, const F &b) { F r; fe_sub(&FR, &r, &a, &b); return r; }
inline F fneg(const F &a) { F r; fe_neg(&FR, &r, &a); return r; }
inline F fone() { F r; fe_one(&FR, &r); return r; }
inline F fzero() { F r; fe_zero(&r); return r; }
inline F finv(const F &a) { F r; fe_inv(&FR, &r, &a); return r; }
inline F f_u64(uint64_t v) { F r; fe_from_u64(&FR, &r, v); return r; }
inline F to_m(const U256 &c) { F t, r; memcpy(t.l, c.l, 32); fe_to_mont(&FR, &r, &t); return r; }
inline U256 from_m(const F &a) { F t; fe_from_mont(&FR, &t, &a); U256 u; memcpy(u.l, t.l, 32); return u; }
inline F fpow(F a, uint64_t e) {
  F r = fone();
  for (; e; e >>= 1) {
    if (e & 1) r = fmul(r, a);
    a = fmul(a, a);
  }
  return r;
}

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ------------------------------------------------------------------------------------------------ NTT
// Iterative radix-2 with one shared table of the 2^log_n-th roots; columns in parallel.
struct Domain {
  int log_n;
  size_t n;
  std::vector<F> tw, tw_inv;  // w^i, w^-i for i < n/2
  F n_inv;
  std::vector<uint32_t> rev;
  explicit Domain(int lg) : log_n(lg), n((size_t)1 << lg) {
    static const uint64_t ROOT28[4] = {0xd34f1ed960c37c9cULL, 0x3215cf6dd39329c8ULL, 0x98865ea93dd31f74ULL, 0x03ddb9f5166d18b7ULL};  // 7^((r-1)/2^28)
    F c, w;
    memcpy(c.l, ROOT28, 32);
    fe_to_mont(&FR, &w, &c);
    for (int i = 0; i < 28 - lg; ++i) w = fmul(w, w);
    const F wi = finv(w);
    tw.resize(n / 2 ? n / 2 : 1);
    tw_inv.resize(tw.size());
    F a = fone(), b = fone();
    for (size_t i = 0; i < tw.size(); ++i) {
      tw[i] = a;
      tw_inv[i] = b;
      a = fmul(a, w);
      b = fmul(b, wi);
    }
    n_inv = finv(f_u64(n));
    rev.resize(n);
    for (size_t i = 0; i < n; ++i) {
      uint32_t r = 0;
      for (int b2 = 0; b2 < lg; ++b2) r |= ((i >> b2) & 1) << (lg - 1 - b2);
      rev[i] = r;
    }
  }
  F omega() const { return n > 1 ? tw[1] : fone(); }
  void fft(F *a, bool inverse) const {
    for (size_t i = 0; i < n; ++i)
      if (i < rev[i]) std::swap(a[i], a[rev[i]]);
    const std::vector<F> &t = inverse ? tw_inv : tw;
    for (size_t m = 1; m < n; m <<= 1) {
      const size_t stride = n / (2 * m);
      for (size_t k = 0; k < n; k += 2 * m)
        for (size_t j = 0; j < m; ++j) {
          const F x = fmul(a[k + j + m], t[j * stride]);
          a[k + j + m] = fsub(a[k + j], x);
          a[k + j] = fadd(a[k + j], x);
        }
    }
    if (inverse)
      for (size_t i = 0; i < n; ++i) a[i] = fmul(a[i], n_inv);
  }
};

// Big per-proof arrays live in the key object and are reused from proof to proof (no page faults after the first one);
// nothing here is value-initialised on one thread: every element is written by the parallel loop that produces it.
struct Buf {
  F *p = nullptr;
  size_t cap = 0;
  F *get(size_t count) {
    if (count > cap) {
      free(p);
      p = (F *)aligned_alloc(64, (count * sizeof(F) + 63) / 64 * 64);
      if (!p) throw std::bad_alloc();
      cap = count;
    }
    return p;
  }
  ~Buf() { free(p); }
};
struct Workspace {
  Buf adv, la, ls, pz, lz, adv_e, la_e, ls_e, pz_e, lz_e, inst_e, h_e, msm_k, gp;
  std::vector<U256> advice_table;  // the Assigner's canonical cells: the fixed layout rewrites every used cell, the rest stays zero
  std::vector<g1j_t> msm_win;
  size_t cells[3] = {0, 0, 0}, sels[3] = {0, 0, 0}, lookups = 0;  // stream lengths of the previous proof (same circuit): reserved up front
};

// ------------------------------------------------------------------------------------------------ MSM
// Jacobian x = X/Z^2, y = Y/Z^3 as in bn254_ref.h (whose g1j_* are the reference formulation); same formulas on the CIOS product
inline void jdbl(g1j_t *r, const g1j_t *p) {  // dbl-2009-l
  if (g1j_is_identity(p)) { *r = *p; return; }
  fe_t a, b, c, d, e, f, t, x3, y3, z3;
  qsqr(&a, &p->x);
  qsqr(&b, &p->y);
  qsqr(&c, &b);
  fe_add(&FQ, &t, &p->x, &b); qsqr(&t, &t); fe_sub(&FQ, &t, &t, &a); fe_sub(&FQ, &t, &t, &c);
  fe_add(&FQ, &d, &t, &t);
  fe_add(&FQ, &e, &a, &a); fe_add(&FQ, &e, &e, &a);
  qsqr(&f, &e);
  fe_sub(&FQ, &x3, &f, &d); fe_sub(&FQ, &x3, &x3, &d);
  qmul(&z3, &p->y, &p->z); fe_add(&FQ, &z3, &z3, &z3);
  fe_sub(&FQ, &t, &d, &x3); qmul(&y3, &e, &t);
  fe_add(&FQ, &c, &c, &c); fe_add(&FQ, &c, &c, &c); fe_add(&FQ, &c, &c, &c);
  fe_sub(&FQ, &y3, &y3, &c);
  r->x = x3; r->y = y3; r->z = z3;
}
inline void jadd(g1j_t *r, const g1j_t *p, const g1j_t *q) {  // add-2007-bl
  if (g1j_is_identity(p)) { *r = *q; return; }
  if (g1j_is_identity(q)) { *r = *p; return; }
  fe_t z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t, x3, y3, z3;
  qsqr(&z1z1, &p->z);
  qsqr(&z2z2, &q->z);
  qmul(&u1, &p->x, &z2z2);
  qmul(&u2, &q->x, &z1z1);
  qmul(&s1, &p->y, &q->z); qmul(&s1, &s1, &z2z2);
  qmul(&s2, &q->y, &p->z); qmul(&s2, &s2, &z1z1);
  if (fe_eq(&u1, &u2)) {
    if (fe_eq(&s1, &s2)) { jdbl(r, p); return; }
    g1j_set_identity(r); return;
  }
  fe_sub(&FQ, &h, &u2, &u1);
  fe_add(&FQ, &i, &h, &h); qsqr(&i, &i);
  qmul(&j, &h, &i);
  fe_sub(&FQ, &rr, &s2, &s1); fe_add(&FQ, &rr, &rr, &rr);
  qmul(&v, &u1, &i);
  qsqr(&x3, &rr); fe_sub(&FQ, &x3, &x3, &j); fe_sub(&FQ, &x3, &x3, &v); fe_sub(&FQ, &x3, &x3, &v);
  fe_sub(&FQ, &t, &v, &x3); qmul(&y3, &rr, &t);
  qmul(&t, &s1, &j); fe_add(&FQ, &t, &t, &t); fe_sub(&FQ, &y3, &y3, &t);
  fe_add(&FQ, &z3, &p->z, &q->z); qsqr(&z3, &z3); fe_sub(&FQ, &z3, &z3, &z1z1); fe_sub(&FQ, &z3, &z3, &z2z2);
  qmul(&z3, &z3, &h);
  r->x = x3; r->y = y3; r->z = z3;
}
inline void g1j_madd(g1j_t *r, const g1j_t *p, const g1a_t *q) {  // madd-2007-bl
  if (g1j_is_identity(p)) { g1j_from_affine(r, q); return; }
  fe_t z1z1, u2, s2, h, hh, i, j, rr, v, t, x3, y3, z3;
  qsqr(&z1z1, &p->z);
  qmul(&u2, &q->x, &z1z1);
  qmul(&s2, &q->y, &p->z); qmul(&s2, &s2, &z1z1);
  if (fe_eq(&u2, &p->x)) {
    if (fe_eq(&s2, &p->y)) { jdbl(r, p); return; }
    g1j_set_identity(r); return;
  }
  fe_sub(&FQ, &h, &u2, &p->x);
  qsqr(&hh, &h);
  fe_add(&FQ, &i, &hh, &hh); fe_add(&FQ, &i, &i, &i);
  qmul(&j, &h, &i);
  fe_sub(&FQ, &rr, &s2, &p->y); fe_add(&FQ, &rr, &rr, &rr);
  qmul(&v, &p->x, &i);
  qsqr(&x3, &rr); fe_sub(&FQ, &x3, &x3, &j); fe_sub(&FQ, &x3, &x3, &v); fe_sub(&FQ, &x3, &x3, &v);
  fe_sub(&FQ, &t, &v, &x3); qmul(&y3, &rr, &t);
  qmul(&t, &p->y, &j); fe_add(&FQ, &t, &t, &t); fe_sub(&FQ, &y3, &y3, &t);
  fe_add(&FQ, &z3, &p->z, &h); qsqr(&z3, &z3); fe_sub(&FQ, &z3, &z3, &z1z1); fe_sub(&FQ, &z3, &z3, &hh);
  r->x = x3; r->y = y3; r->z = z3;
}

inline unsigned window_of(const uint64_t k[4], unsigned lo, unsigned c) {
  const unsigned limb = lo >> 6, off = lo & 63;
  if (limb >= 4) return 0;
  uint64_t v = k[limb] >> off;
  if (off + c > 64 && limb + 1 < 4) v |= k[limb + 1] << (64 - off);
  return (unsigned)(v & ((1u << c) - 1));
}

// n_cols MSMs over one basis (halo2 `best_multiexp` per column): Pippenger with unsigned c-bit windows; the
// (column, window) bucket passes run in parallel, so a call of ONE column still uses every core.
void msm_cols(Workspace &ws, const F *scalars, size_t n_cols, const g1a_t *bases, size_t n, AffinePoint *out) {
  if (!n_cols) return;
  const unsigned c = n < 32 ? 3 : n < 1024 ? 7 : n < 65536 ? 10 : 13;
  const unsigned nwin = (254 + c - 1) / c;
  const size_t nb = ((size_t)1 << c) - 1;
  static_assert(sizeof(U256) == sizeof(F), "canonical and Montgomery values share a buffer type");
  U256 *k = (U256 *)ws.msm_k.get(n_cols * n);
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n_cols * n; ++i) k[i] = from_m(scalars[i]);
  if (ws.msm_win.size() < n_cols * nwin) ws.msm_win.resize(n_cols * nwin);
  g1j_t *win = ws.msm_win.data();
#pragma omp parallel
  {
    std::vector<g1j_t> buckets(nb);
#pragma omp for schedule(dynamic, 1)
    for (size_t task = 0; task < n_cols * nwin; ++task) {
      const size_t col = task / nwin;
      const unsigned w = (unsigned)(task % nwin);
      const U256 *kc = k + col * n;
      bool any = false;
      for (size_t b = 0; b < nb; ++b) g1j_set_identity(&buckets[b]);
      for (size_t i = 0; i < n; ++i) {
        const unsigned d = window_of(kc[i].l, w * c, c);
        if (d) {
          g1j_madd(&buckets[d - 1], &buckets[d - 1], &bases[i]);
          any = true;
        }
      }
      g1j_t run, sum;
      g1j_set_identity(&run);
      g1j_set_identity(&sum);
      if (any)
        for (size_t b = nb; b-- > 0;) {
          jadd(&run, &run, &buckets[b]);
          jadd(&sum, &sum, &run);
        }
      win[task] = sum;
    }
  }
#pragma omp parallel for schedule(dynamic, 1)
  for (size_t col = 0; col < n_cols; ++col) {
    g1j_t total;
    g1j_set_identity(&total);
    for (unsigned w = nwin; w-- > 0;) {
      for (unsigned d = 0; d < c; ++d) jdbl(&total, &total);
      jadd(&total, &total, &win[col * nwin + w]);
    }
    g1a_t a;
    g1j_to_affine(&a, &total);
    fe_t cx, cy;
    fe_from_mont(&FQ, &cx, &a.x);
    fe_from_mont(&FQ, &cy, &a.y);
    memcpy(out[col].x.l, cx.l, 32);
    memcpy(out[col].y.l, cy.l, 32);
  }
}

// Montgomery-trick batch inversion of one vector (zeros stay zero)
void batch_inv(F *a, size_t n) {
  std::vector<F> pre(n);
  F acc = fone();
  for (size_t i = 0; i < n; ++i) {
    pre[i] = acc;
    if (!fe_is_zero(&a[i])) acc = fmul(acc, a[i]);
  }
  acc = finv(acc);
  for (size_t i = n; i-- > 0;) {
    if (fe_is_zero(&a[i])) continue;
    const F t = fmul(acc, pre[i]);
    acc = fmul(acc, a[i]);
    a[i] = t;
  }
}

F horner(const F *p, size_t n, const F &x) {
  F acc = fzero();
  for (size_t i = n; i-- > 0;) acc = fadd(fmul(acc, x), p[i]);
  return acc;
}

// q(X) = (p(X) - p(root)) / (X - root), in place; the top coefficient becomes zero
void div_linear(F *p, size_t n, const F &root) {
  F carry = fzero();
  for (size_t i = n - 1; i >= 1; --i) {
    const F cur = fadd(p[i], fmul(carry, root));
    p[i] = carry;
    carry = cur;
  }
  p[0] = carry;
}

// coefficients (ascending) of the polynomial through (pts[i], vals[i])
std::vector<F> interpolate(const std::vector<F> &pts, const std::vector<F> &vals) {
  const size_t m = pts.size();
  std::vector<F> res(m, fzero());
  for (size_t i = 0; i < m; ++i) {
    std::vector<F> num(1, fone());
    F den = fone();
    for (size_t j = 0; j < m; ++j) {
      if (j == i) continue;
      std::vector<F> nx(num.size() + 1, fzero());
      for (size_t t = 0; t < num.size(); ++t) {
        nx[t + 1] = fadd(nx[t + 1], num[t]);
        nx[t] = fsub(nx[t], fmul(pts[j], num[t]));
      }
      num.swap(nx);
      den = fmul(den, fsub(pts[i], pts[j]));
    }
    const F sc = fmul(vals[i], finv(den));
    for (size_t t = 0; t < num.size(); ++t) res[t] = fadd(res[t], fmul(num[t], sc));
  }
  return res;
}

// halo2 `permute_expression_pair` on the usable rows (oracle/halo2_ref.py permute_lookup)
void permute_lookup(const U256 *a_vals, const std::vector<U256> &table, size_t u, std::vector<U256> &a_sorted, std::vector<U256> &s_perm) {
  a_sorted.assign(a_vals, a_vals + u);
  std::sort(a_sorted.begin(), a_sorted.end());
  std::map<U256, size_t> left;
  for (size_t i = 0; i < u; ++i) ++left[table[i]];
  s_perm.assign(u, fe::zero());
  std::vector<size_t> holes;
  for (size_t i = 0; i < u; ++i) {
    if (i == 0 || a_sorted[i] != a_sorted[i - 1]) {
      auto it = left.find(a_sorted[i]);
      if (it == left.end() || it->second == 0) throw std::runtime_error("lookup input not in table");
      --it->second;
      s_perm[i] = a_sorted[i];
    } else {
      holes.push_back(i);
    }
  }
  size_t h = 0;
  for (const auto &kv : left)
    for (size_t r = 0; r < kv.second; ++r) {
      if (h >= holes.size()) throw std::runtime_error("lookup permutation does not fit");
      s_perm[holes[h++]] = kv.first;
    }
  if (h != holes.size()) throw std::runtime_error("lookup permutation does not fit");
}


}  // namespace

struct cpu_pk {
  mutable Workspace ws;
  CircuitConfig cfg;
  BfvParams prm;
  U256 vk_digest;
  F delta;
  std::vector<F> fixed_l, sigma_l, fixed_c, sigma_c, l_c;   // [cols][n], Montgomery
  std::vector<F> fixed_e, sigma_e, l_e, x_e;                 // extended coset (4n), the halo2 pk's `*_cosets`
  std::vector<g1a_t> g_lag, g_mon;
  Domain *dom = nullptr, *dom_e = nullptr;
  ~cpu_pk() {
    delete dom;
    delete dom_e;
  }
};

namespace {

const int LOG_EXT = 2;
const uint64_t COSET_G = 7;

// coefficient vectors [cols][n] -> natural-order evaluations on g * <w_ext>, [cols][4n]
void to_ext(const cpu_pk *pk, const F *coeff, size_t cols, F *out) {
  const size_t n = pk->cfg.n(), ne = n << LOG_EXT;
  std::vector<F> gp(n);
  const F g = f_u64(COSET_G);
  F a = fone();
  for (size_t i = 0; i < n; ++i) {
    gp[i] = a;
    a = fmul(a, g);
  }
#pragma omp parallel for schedule(dynamic, 1)
  for (size_t c = 0; c < cols; ++c) {
    F *o = out + c * ne;
    const F *in = coeff + c * n;
    for (size_t i = 0; i < n; ++i) o[i] = fmul(in[i], gp[i]);
    for (size_t i = n; i < ne; ++i) o[i] = fzero();
    pk->dom_e->fft(o, false);
  }
}

void ntt_cols(const Domain *d, F *a, size_t cols, bool inverse) {
#pragma omp parallel for schedule(dynamic, 1)
  for (size_t c = 0; c < cols; ++c) d->fft(a + c * d->n, inverse);
}

struct Timer {
  double t0, *slots;
  int idx = 0;
  explicit Timer(double *s) : t0(now_ms()), slots(s) {}
  void lap() {
    const double t = now_ms();
    if (slots) slots[idx] = t - t0;
    ++idx;
    t0 = t;
  }
};

void prove_impl(const cpu_pk *pk, const char *input_json, const uint8_t seed32[32], std::vector<uint8_t> &proof, double *ms) {
  const CircuitConfig &cfg = pk->cfg;
  const size_t n = cfg.n(), u = cfg.u(), ne = n << LOG_EXT, step = (size_t)1 << LOG_EXT;
  const unsigned n_adv = cfg.n_advice(), n_lk = cfg.n_lookup, n_chunks = cfg.n_chunks(), n_perm = cfg.n_perm();
  const F w = pk->dom->omega();
  Rng rng(seed32);
  Transcript tr(cfg.transcript);
  Timer tm(ms);
  auto take = [&](F *dst, size_t cnt) {  // the blinding stream is sequential by construction
    for (size_t i = 0; i < cnt; ++i) dst[i] = to_m(rng.next());
  };
  Workspace &ws = pk->ws;
  auto commit = [&](const F *cols, size_t count, bool lagrange, std::vector<AffinePoint> &out) {
    out.resize(count);
    msm_cols(ws, cols, count, lagrange ? pk->g_lag.data() : pk->g_mon.data(), n, out.data());
  };

  // ---- witness, phase 0 (examples/bfv.rs:63-165)
  tr.common_scalar(pk->vk_digest);
  const CircuitInput in = CircuitInput::parse_json(input_json);
  Context ctx0(CTX_PHASE0, false, false), ctx_gate(CTX_GATE1, false, false), ctx_rlc(CTX_RLC1, true, false);
  std::vector<Cell> make_public;
  {
    Context *cs[3] = {&ctx0, &ctx_gate, &ctx_rlc};
    for (int i = 0; i < 3; ++i) {
      cs[i]->advice.reserve(ws.cells[i]);
      cs[i]->selector.reserve(ws.sels[i]);
    }
    ctx_gate.lookup.reserve(ws.lookups);
  }
  const BfvState st = bfv_phase0(ctx0, in, pk->prm, make_public);
  std::vector<U256> inst;
  for (const Cell &c : make_public) inst.push_back(c.value);
  for (const U256 &v : inst) tr.common_scalar(v);
  if (ws.advice_table.size() != (size_t)n_adv * n) ws.advice_table.assign((size_t)n_adv * n, fe::zero());
  Assigner as(cfg, false, ws.advice_table.data());
  as.place(ctx0, true);
  F *adv = ws.adv.get((size_t)n_adv * n);
  auto load_advice = [&](unsigned c0, unsigned c1) {
    for (unsigned c = c0; c < c1; ++c) take(adv + (size_t)c * n + u, n - u);
#pragma omp parallel for schedule(static)
    for (size_t idx = (size_t)c0 * u; idx < (size_t)c1 * u; ++idx) {
      const size_t c = idx / u, r = idx % u;
      adv[c * n + r] = to_m(as.t.advice[c][r]);
    }
  };
  load_advice(0, cfg.n_gate0);
  std::vector<AffinePoint> adv_commit(n_adv), cm;
  commit(adv, cfg.n_gate0, true, cm);
  for (unsigned c = 0; c < cfg.n_gate0; ++c) tr.write_point(adv_commit[c] = cm[c]);
  const U256 gamma_rlc = tr.squeeze();
  tm.lap();  // 0: phase 0

  // ---- witness, phase 1 (examples/bfv.rs:171-301)
  bfv_phase1(st, pk->prm, ctx_gate, ctx_rlc, gamma_rlc);
  as.place(ctx_gate, true);
  as.place(ctx_rlc, true);
  as.place_lookups(ctx_gate);
  {
    const Context *cs[3] = {&ctx0, &ctx_gate, &ctx_rlc};
    for (int i = 0; i < 3; ++i) {
      ws.cells[i] = cs[i]->advice.size();
      ws.sels[i] = cs[i]->selector.size();
    }
    ws.lookups = ctx_gate.lookup.size();
  }
  tm.lap();  // 1: phase-1 witness
  load_advice(cfg.n_gate0, n_adv);
  commit(adv + (size_t)cfg.n_gate0 * n, n_adv - cfg.n_gate0, true, cm);
  for (unsigned c = cfg.n_gate0; c < n_adv; ++c) tr.write_point(adv_commit[c] = cm[c - cfg.n_gate0]);
  (void)tr.squeeze();  // theta: squeezed in protocol order, unused by single-expression lookups
  tm.lap();  // 2: advice commitments

  // ---- lookups: permuted input / table
  std::vector<U256> table(n);
  for (size_t r = 0; r < n; ++r) table[r] = from_m(pk->fixed_l[(size_t)cfg.fix_table() * n + r]);
  F *la = ws.la.get((size_t)n_lk * n), *ls = ws.ls.get((size_t)n_lk * n);
  for (unsigned i = 0; i < n_lk; ++i) {
    take(la + (size_t)i * n + u, n - u);
    take(ls + (size_t)i * n + u, n - u);
  }
  std::string lookup_err;
#pragma omp parallel for schedule(dynamic, 1)
  for (unsigned i = 0; i < n_lk; ++i) {
    try {
      std::vector<U256> ap, sp;
      permute_lookup(as.t.advice[cfg.adv_lookup0() + i], table, u, ap, sp);
      for (size_t r = 0; r < u; ++r) {
        la[(size_t)i * n + r] = to_m(ap[r]);
        ls[(size_t)i * n + r] = to_m(sp[r]);
      }
    } catch (const std::exception &e) {
#pragma omp critical
      lookup_err = e.what();
    }
  }
  if (!lookup_err.empty()) throw std::runtime_error(lookup_err);
  std::vector<AffinePoint> la_commit, ls_commit;
  commit(la, n_lk, true, la_commit);
  commit(ls, n_lk, true, ls_commit);
  for (unsigned i = 0; i < n_lk; ++i) {
    tr.write_point(la_commit[i]);
    tr.write_point(ls_commit[i]);
  }
  const U256 beta_c = tr.squeeze(), gamma_c = tr.squeeze();
  const F beta = to_m(beta_c), gamma = to_m(gamma_c);
  tm.lap();  // 3: lookup permutation + commitments

  // ---- permutation grand products
  std::vector<F> inst_col(n, fzero());
  for (size_t i = 0; i < inst.size(); ++i) inst_col[i] = to_m(inst[i]);
  auto permcol_l = [&](unsigned c) -> const F * {
    if (c < n_adv) return adv + (size_t)c * n;
    return c == cfg.perm_const() ? pk->fixed_l.data() + (size_t)cfg.fix_const() * n : inst_col.data();
  };
  std::vector<F> wpow(n), bdelta(n_perm);
  {
    F a = fone();
    for (size_t i = 0; i < n; ++i) {
      wpow[i] = a;
      a = fmul(a, w);
    }
    F d = beta;
    for (unsigned c = 0; c < n_perm; ++c) {
      bdelta[c] = d;
      d = fmul(d, pk->delta);
    }
  }
  F *pz = ws.pz.get((size_t)n_chunks * n), *lz = ws.lz.get((size_t)n_lk * n);
#pragma omp parallel for schedule(dynamic, 1)
  for (unsigned j = 0; j < n_chunks; ++j) {
    std::vector<F> num(u, fone()), den(u, fone());
    for (unsigned c = j * cfg.chunk(); c < std::min((j + 1) * cfg.chunk(), n_perm); ++c) {
      const F *v = permcol_l(c), *sg = pk->sigma_l.data() + (size_t)c * n;
      for (size_t r = 0; r < u; ++r) {
        den[r] = fmul(den[r], fadd(fadd(v[r], fmul(sg[r], beta)), gamma));
        num[r] = fmul(num[r], fadd(fadd(v[r], fmul(wpow[r], bdelta[c])), gamma));
      }
    }
    batch_inv(den.data(), u);
    F *z = pz + (size_t)j * n;  // local prefix products (start 1); the carry over the chunks is applied below
    z[0] = fone();
    for (size_t r = 0; r < u; ++r) z[r + 1] = fmul(z[r], fmul(num[r], den[r]));
  }
  {
    // z_j(row) = (product of the chunks before j) * local prefix product; the last chunk must close at 1
    std::vector<F> carry(n_chunks + 1);
    carry[0] = fone();
    for (unsigned j = 0; j < n_chunks; ++j) carry[j + 1] = fmul(carry[j], pz[(size_t)j * n + u]);
    const F one = fone();
    if (!fe_eq(&carry[n_chunks], &one)) throw std::runtime_error("permutation argument does not close: copy constraints violated");
#pragma omp parallel for schedule(static)
    for (size_t idx = (size_t)(u + 1); idx < (size_t)n_chunks * (u + 1); ++idx) {
      const size_t j = idx / (u + 1), r = idx % (u + 1);
      pz[j * n + r] = fmul(pz[j * n + r], carry[j]);
    }
    for (unsigned j = 0; j < n_chunks; ++j) take(pz + (size_t)j * n + u + 1, n - u - 1);
  }
  // ---- lookup grand products
  const F *tab_l = pk->fixed_l.data() + (size_t)cfg.fix_table() * n;
  int lookup_open = 0;
#pragma omp parallel for schedule(dynamic, 1)
  for (unsigned i = 0; i < n_lk; ++i) {
    const F *a_l = adv + (size_t)(cfg.adv_lookup0() + i) * n, *ap = la + (size_t)i * n, *sp = ls + (size_t)i * n;
    std::vector<F> den(u);
    for (size_t r = 0; r < u; ++r) den[r] = fmul(fadd(ap[r], beta), fadd(sp[r], gamma));
    batch_inv(den.data(), u);
    F *z = lz + (size_t)i * n;
    z[0] = fone();
    for (size_t r = 0; r < u; ++r) z[r + 1] = fmul(z[r], fmul(fmul(fadd(a_l[r], beta), fadd(tab_l[r], gamma)), den[r]));
    const F one = fone();
    if (!fe_eq(&z[u], &one)) {
#pragma omp atomic write
      lookup_open = 1;
    }
  }
  if (lookup_open) throw std::runtime_error("lookup argument does not close");
  for (unsigned i = 0; i < n_lk; ++i) take(lz + (size_t)i * n + u + 1, n - u - 1);
  std::vector<AffinePoint> pz_commit, lz_commit;
  commit(pz, n_chunks, true, pz_commit);
  for (const AffinePoint &p : pz_commit) tr.write_point(p);
  commit(lz, n_lk, true, lz_commit);
  for (const AffinePoint &p : lz_commit) tr.write_point(p);
  // ---- vanishing argument: random polynomial
  std::vector<F> rand_c(n);
  take(rand_c.data(), n);
  std::vector<AffinePoint> rand_commit;
  commit(rand_c.data(), 1, false, rand_commit);
  tr.write_point(rand_commit[0]);
  const F y = to_m(tr.squeeze());
  tm.lap();  // 4: grand products + commitments

  // ---- coefficient forms and the extended coset
  ntt_cols(pk->dom, adv, n_adv, true);
  ntt_cols(pk->dom, la, n_lk, true);
  ntt_cols(pk->dom, ls, n_lk, true);
  ntt_cols(pk->dom, pz, n_chunks, true);
  ntt_cols(pk->dom, lz, n_lk, true);
  pk->dom->fft(inst_col.data(), true);
  F *adv_e = ws.adv_e.get((size_t)n_adv * ne), *la_e = ws.la_e.get((size_t)n_lk * ne), *ls_e = ws.ls_e.get((size_t)n_lk * ne);
  F *pz_e = ws.pz_e.get((size_t)n_chunks * ne), *lz_e = ws.lz_e.get((size_t)n_lk * ne), *inst_e = ws.inst_e.get(ne);
  to_ext(pk, adv, n_adv, adv_e);
  to_ext(pk, la, n_lk, la_e);
  to_ext(pk, ls, n_lk, ls_e);
  to_ext(pk, pz, n_chunks, pz_e);
  to_ext(pk, lz, n_lk, lz_e);
  to_ext(pk, inst_col.data(), 1, inst_e);
  tm.lap();  // 5: NTTs

  // ---- quotient numerator, expression by expression in halo2's folding order (oracle/halo2_ref.py expressions_at)
  F *h_e = ws.h_e.get(ne);
  {
    const F g_rlc = to_m(gamma_rlc), one = fone();
    const F gn = fpow(f_u64(COSET_G), n), i4 = fpow(pk->dom_e->omega(), n);
    F zinv[4];
    for (size_t t = 0; t < step; ++t) zinv[t] = finv(fsub(fmul(gn, fpow(i4, t)), one));
    const F *fx = pk->fixed_e.data(), *sg = pk->sigma_e.data(), *l0 = pk->l_e.data(), *ll = l0 + ne, *lact = ll + ne;
    const unsigned m = n_chunks - 1;
    auto permcol_e = [&](unsigned c) -> const F * {
      if (c < n_adv) return adv_e + (size_t)c * ne;
      return c == cfg.perm_const() ? fx + (size_t)cfg.fix_const() * ne : inst_e;
    };
    std::vector<const F *> pcol(n_perm);
    for (unsigned c = 0; c < n_perm; ++c) pcol[c] = permcol_e(c);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < ne; ++i) {
      const size_t i1 = (i + step) % ne, i2 = (i + 2 * step) % ne, i3 = (i + 3 * step) % ne, il = (i + u * step) % ne, ip = (i + ne - step) % ne;
      F acc = fzero();
      auto fold = [&](const F &e) { acc = fadd(fmul(acc, y), e); };
      for (unsigned j = 0; j < cfg.n_gate(); ++j) {
        const F *a = adv_e + (size_t)j * ne;
        fold(fmul(fx[(size_t)j * ne + i], fsub(fadd(a[i], fmul(a[i1], a[i2])), a[i3])));
      }
      for (unsigned j = 0; j < cfg.n_rlc; ++j) {
        const F *a = adv_e + (size_t)(cfg.adv_rlc0() + j) * ne;
        fold(fmul(fx[(size_t)(cfg.fix_qrlc0() + j) * ne + i], fsub(fadd(fmul(a[i], g_rlc), a[i1]), a[i2])));
      }
      fold(fmul(l0[i], fsub(one, pz_e[i])));
      const F zl = pz_e[(size_t)m * ne + i];
      fold(fmul(ll[i], fsub(fmul(zl, zl), zl)));
      for (unsigned j = 1; j <= m; ++j) fold(fmul(l0[i], fsub(pz_e[(size_t)j * ne + i], pz_e[(size_t)(j - 1) * ne + il])));
      const F xi = pk->x_e[i];
      for (unsigned j = 0; j <= m; ++j) {
        F left = pz_e[(size_t)j * ne + i1], right = pz_e[(size_t)j * ne + i];
        for (unsigned c = j * cfg.chunk(); c < std::min((j + 1) * cfg.chunk(), n_perm); ++c) {
          const F v = pcol[c][i];
          left = fmul(left, fadd(fadd(v, fmul(sg[(size_t)c * ne + i], beta)), gamma));
          right = fmul(right, fadd(fadd(v, fmul(xi, bdelta[c])), gamma));
        }
        fold(fmul(lact[i], fsub(left, right)));
      }
      for (unsigned k2 = 0; k2 < n_lk; ++k2) {
        const F z0 = lz_e[(size_t)k2 * ne + i], z1 = lz_e[(size_t)k2 * ne + i1];
        const F a = adv_e[(size_t)(cfg.adv_lookup0() + k2) * ne + i], s = fx[(size_t)cfg.fix_table() * ne + i];
        const F ap = la_e[(size_t)k2 * ne + i], apm = la_e[(size_t)k2 * ne + ip], sp = ls_e[(size_t)k2 * ne + i];
        fold(fmul(l0[i], fsub(one, z0)));
        fold(fmul(ll[i], fsub(fmul(z0, z0), z0)));
        fold(fmul(lact[i], fsub(fmul(z1, fmul(fadd(ap, beta), fadd(sp, gamma))), fmul(z0, fmul(fadd(a, beta), fadd(s, gamma))))));
        fold(fmul(l0[i], fsub(ap, sp)));
        fold(fmul(lact[i], fmul(fsub(ap, sp), fsub(ap, apm))));
      }
      h_e[i] = fmul(acc, zinv[i % step]);
    }
  }
  // extended_to_coeff: inverse transform, then a[i] *= g^-i
  pk->dom_e->fft(h_e, true);
  {
    const F ginv = finv(f_u64(COSET_G));
    std::vector<F> gp(ne);
    F a = fone();
    for (size_t i = 0; i < ne; ++i) {
      gp[i] = a;
      a = fmul(a, ginv);
    }
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < ne; ++i) h_e[i] = fmul(h_e[i], gp[i]);
  }
  for (size_t i = 3 * n; i < ne; ++i)
    if (!fe_is_zero(&h_e[i])) throw std::runtime_error("quotient degree too high: some constraint is violated");
  std::vector<AffinePoint> h_commit;
  commit(h_e, 3, false, h_commit);
  for (const AffinePoint &p : h_commit) tr.write_point(p);
  const F x = to_m(tr.squeeze());
  tm.lap();  // 6: quotient

  // ---- evaluations, in the write order of create_proof (OpenLayout's item order)
  const OpenLayout L(cfg);
  std::vector<F> Hpoly(n);
  {
    const F xn = fpow(x, n), xn2 = fmul(xn, xn);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) Hpoly[i] = fadd(h_e[i], fadd(fmul(h_e[n + i], xn), fmul(h_e[2 * n + i], xn2)));
  }
  std::vector<const F *> poly(L.count);
  for (unsigned c = 0; c < n_adv; ++c) poly[L.adv0 + c] = adv + (size_t)c * n;
  for (unsigned c = 0; c < cfg.n_fixed(); ++c) poly[L.fixed0 + c] = pk->fixed_c.data() + (size_t)c * n;
  poly[L.H] = Hpoly.data();
  poly[L.rand] = rand_c.data();
  for (unsigned c = 0; c < n_perm; ++c) poly[L.sigma0 + c] = pk->sigma_c.data() + (size_t)c * n;
  for (unsigned j = 0; j < n_chunks; ++j) poly[L.pz0 + j] = pz + (size_t)j * n;
  for (unsigned i = 0; i < n_lk; ++i) {
    poly[L.lk0 + 3 * i] = lz + (size_t)i * n;
    poly[L.lk0 + 3 * i + 1] = la + (size_t)i * n;
    poly[L.lk0 + 3 * i + 2] = ls + (size_t)i * n;
  }
  F pt[N_ROT_IDS];
  pt[ROT_0] = x;
  pt[ROT_1] = fmul(x, w);
  pt[ROT_2] = fmul(pt[ROT_1], w);
  pt[ROT_3] = fmul(pt[ROT_2], w);
  pt[ROT_LAST] = fmul(x, fpow(w, u));
  pt[ROT_PREV] = fmul(x, finv(w));
  std::vector<size_t> ev_off(L.count + 1, 0);
  for (size_t it = 0; it < L.count; ++it) ev_off[it + 1] = ev_off[it] + L.rots[it].size();
  std::vector<F> evs(ev_off[L.count]);
  std::vector<std::pair<size_t, int>> jobs;
  for (size_t it = 0; it < L.count; ++it)
    for (size_t t = 0; t < L.rots[it].size(); ++t) jobs.push_back({it, (int)t});
#pragma omp parallel for schedule(dynamic, 4)
  for (size_t j = 0; j < jobs.size(); ++j) {
    const size_t it = jobs[j].first;
    evs[ev_off[it] + jobs[j].second] = horner(poly[it], n, pt[L.rots[it][jobs[j].second]]);
  }
  for (size_t it = 0; it < L.count; ++it) {
    if (it == L.H) continue;  // implied by the identity, not written
    for (size_t t = 0; t < L.rots[it].size(); ++t) tr.write_scalar(from_m(evs[ev_off[it] + t]));
  }
  tm.lap();  // 7: evaluations

  // ---- SHPLONK (halo2 ProverSHPLONK::create_proof)
  const F yq = to_m(tr.squeeze());
  std::vector<int> super;
  const std::vector<OpenSet> sets = intermediate_sets(L, open_queries(cfg, L), super);
  std::vector<std::vector<F>> f_polys(sets.size()), r_coeffs(sets.size());
  for (size_t s = 0; s < sets.size(); ++s) {
    const OpenSet &os = sets[s];
    std::vector<F> pw(os.members.size());
    F a = fone();
    for (size_t i = 0; i < pw.size(); ++i) {
      pw[i] = a;
      a = fmul(a, yq);
    }
    std::vector<F> &f = f_polys[s];
    f.assign(n, fzero());
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < n; ++r) {
      F acc = fzero();
      for (size_t i = 0; i < pw.size(); ++i) acc = fadd(acc, fmul(poly[os.members[i]][r], pw[i]));
      f[r] = acc;
    }
    std::vector<F> pts, comb;
    for (int rid : os.rots) {
      F acc = fzero();
      for (size_t i = 0; i < pw.size(); ++i) {
        const int slot = L.eval_slot(os.members[i], rid);
        if (slot < 0) throw std::logic_error("opening set member lacks a rotation of its set");
        acc = fadd(acc, fmul(pw[i], evs[ev_off[os.members[i]] + slot]));
      }
      pts.push_back(pt[rid]);
      comb.push_back(acc);
    }
    r_coeffs[s] = interpolate(pts, comb);
  }
  const F v = to_m(tr.squeeze());
  std::vector<F> hq(n, fzero());
  {
    std::vector<std::vector<F>> parts(sets.size());
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t s = 0; s < sets.size(); ++s) {
      std::vector<F> num = f_polys[s];
      for (size_t t = 0; t < r_coeffs[s].size(); ++t) num[t] = fsub(num[t], r_coeffs[s][t]);
      for (int rid : sets[s].rots) div_linear(num.data(), n, pt[rid]);
      parts[s].swap(num);
    }
    F vj = fone();
    for (size_t s = 0; s < sets.size(); ++s) {
#pragma omp parallel for schedule(static)
      for (size_t r = 0; r < n; ++r) hq[r] = fadd(hq[r], fmul(parts[s][r], vj));
      vj = fmul(vj, v);
    }
  }
  std::vector<AffinePoint> one_commit;
  commit(hq.data(), 1, false, one_commit);
  tr.write_point(one_commit[0]);
  const F uu = to_m(tr.squeeze());
  F zt_u = fone();
  for (int rid : super) zt_u = fmul(zt_u, fsub(uu, pt[rid]));
  std::vector<F> Lp(n);
  {
    const F mz = fneg(zt_u);
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < n; ++r) Lp[r] = fmul(hq[r], mz);
  }
  F z_diff_0 = fone(), vj = fone();
  for (size_t s = 0; s < sets.size(); ++s) {
    F zdiff = fone();
    for (int rid : super)
      if (std::find(sets[s].rots.begin(), sets[s].rots.end(), rid) == sets[s].rots.end()) zdiff = fmul(zdiff, fsub(uu, pt[rid]));
    if (s == 0) z_diff_0 = zdiff;
    const F coef = fmul(vj, zdiff);
    const std::vector<F> &f = f_polys[s];
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < n; ++r) Lp[r] = fadd(Lp[r], fmul(f[r], coef));
    const F r_u = horner(r_coeffs[s].data(), r_coeffs[s].size(), uu);
    Lp[0] = fsub(Lp[0], fmul(coef, r_u));
    vj = fmul(vj, v);
  }
  {
    const F chk = horner(Lp.data(), n, uu);
    if (!fe_is_zero(&chk)) throw std::logic_error("SHPLONK linearisation does not vanish at u");
  }
  div_linear(Lp.data(), n, uu);
  {
    const F zi = finv(z_diff_0);
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < n; ++r) Lp[r] = fmul(Lp[r], zi);
  }
  commit(Lp.data(), 1, false, one_commit);
  tr.write_point(one_commit[0]);
  tm.lap();  // 8: multi-open
  proof = tr.out;
}

}  // namespace

extern "C" {

#define CPU_EXPORT __attribute__((visibility("default")))

struct cpu_cfg_c {
  uint32_t k, n_gate0, n_gate1, n_lookup, n_rlc, unusable_rows, lookup_bits, transcript;
  const uint32_t *bp_gate0; uint32_t n_bp_gate0;
  const uint32_t *bp_gate1; uint32_t n_bp_gate1;
  const uint32_t *bp_rlc; uint32_t n_bp_rlc;
  uint64_t bfv_n, bfv_q, bfv_t, bfv_b;
};

// Arrays are Montgomery Fr, 4 x u64 per value ([columns][n] row-major); points affine Montgomery Fq (x||y), n of each.
// fixed_l / fixed_c: n_fixed columns; sigma_l / sigma_c: n_perm columns; l_c: l_0, l_last, l_active (coefficients).
// delta_canon: the permutation argument's coset multiplier (canonical). Returns NULL and fills err on failure.
CPU_EXPORT cpu_pk *cpu_pk_create(const cpu_cfg_c *c, const uint8_t vk_digest_le[32], const uint8_t delta_canon_le[32], const uint64_t *fixed_l,
                                 const uint64_t *sigma_l, const uint64_t *fixed_c, const uint64_t *sigma_c, const uint64_t *l_c, const uint64_t *g_lagrange,
                                 const uint64_t *g_monomial, char *err, size_t err_len) {
  try {
    cpu_pk *pk = new cpu_pk();
    CircuitConfig &cfg = pk->cfg;
    cfg.k = c->k;
    cfg.n_gate0 = c->n_gate0;
    cfg.n_gate1 = c->n_gate1;
    cfg.n_lookup = c->n_lookup;
    cfg.n_rlc = c->n_rlc;
    cfg.unusable_rows = c->unusable_rows;
    cfg.lookup_bits = c->lookup_bits;
    cfg.transcript = c->transcript;
    cfg.bp_gate0.assign(c->bp_gate0, c->bp_gate0 + c->n_bp_gate0);
    cfg.bp_gate1.assign(c->bp_gate1, c->bp_gate1 + c->n_bp_gate1);
    cfg.bp_rlc.assign(c->bp_rlc, c->bp_rlc + c->n_bp_rlc);
    pk->prm.N = (size_t)c->bfv_n;
    pk->prm.Q = c->bfv_q;
    pk->prm.T = c->bfv_t;
    pk->prm.B = c->bfv_b;
    memcpy(pk->vk_digest.l, vk_digest_le, 32);
    U256 d;
    memcpy(d.l, delta_canon_le, 32);
    pk->delta = to_m(d);
    const size_t n = cfg.n(), ne = n << LOG_EXT;
    auto copy = [&](std::vector<F> &dst, const uint64_t *src, size_t cols) {
      dst.resize(cols * n);
      memcpy(dst.data(), src, cols * n * sizeof(F));
    };
    copy(pk->fixed_l, fixed_l, cfg.n_fixed());
    copy(pk->sigma_l, sigma_l, cfg.n_perm());
    copy(pk->fixed_c, fixed_c, cfg.n_fixed());
    copy(pk->sigma_c, sigma_c, cfg.n_perm());
    copy(pk->l_c, l_c, 3);
    pk->g_lag.resize(n);
    pk->g_mon.resize(n);
    memcpy(pk->g_lag.data(), g_lagrange, n * sizeof(g1a_t));
    memcpy(pk->g_mon.data(), g_monomial, n * sizeof(g1a_t));
    pk->dom = new Domain((int)cfg.k);
    pk->dom_e = new Domain((int)cfg.k + LOG_EXT);
    pk->fixed_e.resize((size_t)cfg.n_fixed() * ne);
    pk->sigma_e.resize((size_t)cfg.n_perm() * ne);
    pk->l_e.resize(3 * ne);
    to_ext(pk, pk->fixed_c.data(), cfg.n_fixed(), pk->fixed_e.data());
    to_ext(pk, pk->sigma_c.data(), cfg.n_perm(), pk->sigma_e.data());
    to_ext(pk, pk->l_c.data(), 3, pk->l_e.data());
    pk->x_e.resize(ne);
    F a = f_u64(COSET_G);
    const F we = pk->dom_e->omega();
    for (size_t i = 0; i < ne; ++i) {
      pk->x_e[i] = a;
      a = fmul(a, we);
    }
    return pk;
  } catch (const std::exception &e) {
    if (err && err_len) snprintf(err, err_len, "%s", e.what());
    return nullptr;
  }
}

CPU_EXPORT void cpu_pk_destroy(cpu_pk *pk) { delete pk; }

// One proof. phase_ms (optional, 9 doubles): phase 0, phase-1 witness, advice commitments, lookups, grand products,
// NTTs, quotient, evaluations, multi-open.  Returns 0, or -1 with err filled; *len is the proof length (cap too small => -2).
CPU_EXPORT int cpu_prove(const cpu_pk *pk, const char *input_json, const uint8_t seed32[32], uint8_t *out, size_t cap, size_t *len, double *phase_ms,
                         char *err, size_t err_len) {
  try {
    std::vector<uint8_t> proof;
    prove_impl(pk, input_json, seed32, proof, phase_ms);
    *len = proof.size();
    if (proof.size() > cap) return -2;
    memcpy(out, proof.data(), proof.size());
    return 0;
  } catch (const std::exception &e) {
    if (err && err_len) snprintf(err, err_len, "%s", e.what());
    return -1;
  }
}

CPU_EXPORT int cpu_prover_threads(void) { return omp_get_max_threads(); }
CPU_EXPORT void cpu_prover_set_threads(int n) {
  if (n > 0) omp_set_num_threads(n);
}

}  // extern "C"
