/* CPU ORACLE -- test infrastructure only (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
 * The product (zk-fhe_amd/) must never include, link or call anything in oracle/.
 *
 * Plain-C restatement of the BN254 arithmetic the BFV prover hot path uses.  The arithmetic is
 * NOT in /root/reference: it lives in the un-vendored crates halo2curves (bn256::{Fr,Fq,G1}) and
 * halo2_proofs (arithmetic::{best_fft,best_multiexp}), pulled in by reference Cargo.toml:9-11 and
 * reached through examples/bfv.rs:311.  Their published algorithms are restated
 * here (Montgomery residues R = 2^256 on 4 x u64 little-endian limbs; short-Weierstrass
 * y^2 = x^3 + 3) and pinned against exact big-integer vectors in tests/golden/ (made by
 * tests/golden/gen_vectors.py from oracle/pyref.py) plus the reference's own constants
 * (F::MODULUS src/poly_chip.rs:90, SURVEY.md section 4 KAT 4).
 *
 * Deliberately a different formulation from the device code (64-bit limbs, full 512-bit product
 * followed by a separate Montgomery reduction) so the two do not share mistakes.
 */
#ifndef ZKFHE_ORACLE_BN254_REF_H
#define ZKFHE_ORACLE_BN254_REF_H
#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe_t;          /* field element, Montgomery form unless noted */
typedef struct { fe_t x, y; } g1a_t;             /* affine, identity = (0,0) */
typedef struct { fe_t x, y, z; } g1j_t;          /* Jacobian, identity z = 0 */

typedef struct {
  uint64_t p[4];
  uint64_t r1[4];   /* R mod p   */
  uint64_t r2[4];   /* R^2 mod p */
  uint64_t inv;     /* -p^-1 mod 2^64 */
} field_t;

static const field_t FR = {
  {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
  {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL},
  {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL},
  0xc2e1f593efffffffULL};

static const field_t FQ = {
  {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
  {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL},
  {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL},
  0x87d20782e4866389ULL};

static inline int fe_is_zero(const fe_t *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_eq(const fe_t *a, const fe_t *b) { return memcmp(a, b, sizeof(fe_t)) == 0; }

/* a >= b ? */
static inline int limbs_geq(const uint64_t *a, const uint64_t *b) {
  for (int i = 3; i >= 0; --i) {
    if (a[i] > b[i]) return 1;
    if (a[i] < b[i]) return 0;
  }
  return 1;
}

static inline uint64_t limbs_sub(uint64_t *r, const uint64_t *a, const uint64_t *b) {
  uint64_t borrow = 0;
  for (int i = 0; i < 4; ++i) {
    u128 d = (u128)a[i] - b[i] - borrow;
    r[i] = (uint64_t)d;
    borrow = (uint64_t)(d >> 64) & 1;
  }
  return borrow;
}

static inline uint64_t limbs_add(uint64_t *r, const uint64_t *a, const uint64_t *b) {
  uint64_t carry = 0;
  for (int i = 0; i < 4; ++i) {
    u128 s = (u128)a[i] + b[i] + carry;
    r[i] = (uint64_t)s;
    carry = (uint64_t)(s >> 64);
  }
  return carry;
}

static inline void fe_add(const field_t *F, fe_t *r, const fe_t *a, const fe_t *b) {
  uint64_t t[4];
  limbs_add(t, a->l, b->l);               /* < 2^255, no carry */
  if (limbs_geq(t, F->p)) limbs_sub(t, t, F->p);
  memcpy(r->l, t, 32);
}

static inline void fe_sub(const field_t *F, fe_t *r, const fe_t *a, const fe_t *b) {
  uint64_t t[4];
  if (limbs_sub(t, a->l, b->l)) limbs_add(t, t, F->p);
  memcpy(r->l, t, 32);
}

static inline void fe_neg(const field_t *F, fe_t *r, const fe_t *a) {
  if (fe_is_zero(a)) { *r = *a; return; }
  uint64_t t[4];
  limbs_sub(t, F->p, a->l);
  memcpy(r->l, t, 32);
}

/* Montgomery product: full 4x4 schoolbook product, then 4 reduction rounds. */
static inline void fe_mul(const field_t *F, fe_t *r, const fe_t *a, const fe_t *b) {
  uint64_t w[9] = {0};
  for (int i = 0; i < 4; ++i) {
    uint64_t carry = 0;
    for (int j = 0; j < 4; ++j) {
      u128 s = (u128)a->l[i] * b->l[j] + w[i + j] + carry;
      w[i + j] = (uint64_t)s;
      carry = (uint64_t)(s >> 64);
    }
    w[i + 4] = carry;
  }
  for (int i = 0; i < 4; ++i) {
    uint64_t m = w[i] * F->inv;
    uint64_t carry = 0;
    for (int j = 0; j < 4; ++j) {
      u128 s = (u128)m * F->p[j] + w[i + j] + carry;
      w[i + j] = (uint64_t)s;
      carry = (uint64_t)(s >> 64);
    }
    for (int k = i + 4; carry && k < 9; ++k) {
      u128 s = (u128)w[k] + carry;
      w[k] = (uint64_t)s;
      carry = (uint64_t)(s >> 64);
    }
  }
  uint64_t t[4] = {w[4], w[5], w[6], w[7]};   /* < 2p */
  if (w[8] || limbs_geq(t, F->p)) limbs_sub(t, t, F->p);
  memcpy(r->l, t, 32);
}

static inline void fe_sqr(const field_t *F, fe_t *r, const fe_t *a) { fe_mul(F, r, a, a); }

static inline void fe_one(const field_t *F, fe_t *r) { memcpy(r->l, F->r1, 32); }
static inline void fe_zero(fe_t *r) { memset(r, 0, sizeof(*r)); }

static inline void fe_to_mont(const field_t *F, fe_t *r, const fe_t *a) {
  fe_t r2; memcpy(r2.l, F->r2, 32);
  fe_mul(F, r, a, &r2);
}
static inline void fe_from_mont(const field_t *F, fe_t *r, const fe_t *a) {
  fe_t one = {{1, 0, 0, 0}};
  fe_mul(F, r, a, &one);
}

/* a^e, e = 4 LE limbs */
static inline void fe_pow(const field_t *F, fe_t *r, const fe_t *a, const uint64_t e[4]) {
  fe_t acc; fe_one(F, &acc);
  for (int i = 255; i >= 0; --i) {
    fe_sqr(F, &acc, &acc);
    if ((e[i >> 6] >> (i & 63)) & 1) fe_mul(F, &acc, &acc, a);
  }
  *r = acc;
}

/* a^-1 (0 -> 0), Fermat */
static inline void fe_inv(const field_t *F, fe_t *r, const fe_t *a) {
  uint64_t e[4] = {F->p[0] - 2, F->p[1], F->p[2], F->p[3]};
  fe_pow(F, r, a, e);
}

static inline void fe_from_u64(const field_t *F, fe_t *r, uint64_t v) {
  fe_t t = {{v, 0, 0, 0}};
  fe_to_mont(F, r, &t);
}

/* ------------------------------------------------------------------ G1 (Jacobian, a = 0, b = 3) */
static inline void g1j_set_identity(g1j_t *p) { fe_zero(&p->x); fe_one(&FQ, &p->y); fe_zero(&p->z); }
static inline int g1j_is_identity(const g1j_t *p) { return fe_is_zero(&p->z); }
static inline int g1a_is_identity(const g1a_t *p) { return fe_is_zero(&p->x) && fe_is_zero(&p->y); }

static inline void g1j_from_affine(g1j_t *r, const g1a_t *p) {
  if (g1a_is_identity(p)) { g1j_set_identity(r); return; }
  r->x = p->x; r->y = p->y; fe_one(&FQ, &r->z);
}

/* dbl-2009-l */
static inline void g1j_dbl(g1j_t *r, const g1j_t *p) {
  if (g1j_is_identity(p)) { *r = *p; return; }
  fe_t a, b, c, d, e, f, t, x3, y3, z3;
  fe_sqr(&FQ, &a, &p->x);
  fe_sqr(&FQ, &b, &p->y);
  fe_sqr(&FQ, &c, &b);
  fe_add(&FQ, &t, &p->x, &b); fe_sqr(&FQ, &t, &t); fe_sub(&FQ, &t, &t, &a); fe_sub(&FQ, &t, &t, &c);
  fe_add(&FQ, &d, &t, &t);
  fe_add(&FQ, &e, &a, &a); fe_add(&FQ, &e, &e, &a);
  fe_sqr(&FQ, &f, &e);
  fe_sub(&FQ, &x3, &f, &d); fe_sub(&FQ, &x3, &x3, &d);
  fe_mul(&FQ, &z3, &p->y, &p->z); fe_add(&FQ, &z3, &z3, &z3);
  fe_sub(&FQ, &t, &d, &x3); fe_mul(&FQ, &y3, &e, &t);
  fe_add(&FQ, &c, &c, &c); fe_add(&FQ, &c, &c, &c); fe_add(&FQ, &c, &c, &c);
  fe_sub(&FQ, &y3, &y3, &c);
  r->x = x3; r->y = y3; r->z = z3;
}

/* add-2007-bl with the exceptional cases handled */
static inline void g1j_add(g1j_t *r, const g1j_t *p, const g1j_t *q) {
  if (g1j_is_identity(p)) { *r = *q; return; }
  if (g1j_is_identity(q)) { *r = *p; return; }
  fe_t z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t, x3, y3, z3;
  fe_sqr(&FQ, &z1z1, &p->z);
  fe_sqr(&FQ, &z2z2, &q->z);
  fe_mul(&FQ, &u1, &p->x, &z2z2);
  fe_mul(&FQ, &u2, &q->x, &z1z1);
  fe_mul(&FQ, &s1, &p->y, &q->z); fe_mul(&FQ, &s1, &s1, &z2z2);
  fe_mul(&FQ, &s2, &q->y, &p->z); fe_mul(&FQ, &s2, &s2, &z1z1);
  if (fe_eq(&u1, &u2)) {
    if (fe_eq(&s1, &s2)) { g1j_dbl(r, p); return; }
    g1j_set_identity(r); return;
  }
  fe_sub(&FQ, &h, &u2, &u1);
  fe_add(&FQ, &i, &h, &h); fe_sqr(&FQ, &i, &i);
  fe_mul(&FQ, &j, &h, &i);
  fe_sub(&FQ, &rr, &s2, &s1); fe_add(&FQ, &rr, &rr, &rr);
  fe_mul(&FQ, &v, &u1, &i);
  fe_sqr(&FQ, &x3, &rr); fe_sub(&FQ, &x3, &x3, &j); fe_sub(&FQ, &x3, &x3, &v); fe_sub(&FQ, &x3, &x3, &v);
  fe_sub(&FQ, &t, &v, &x3); fe_mul(&FQ, &y3, &rr, &t);
  fe_mul(&FQ, &t, &s1, &j); fe_add(&FQ, &t, &t, &t); fe_sub(&FQ, &y3, &y3, &t);
  fe_add(&FQ, &z3, &p->z, &q->z); fe_sqr(&FQ, &z3, &z3); fe_sub(&FQ, &z3, &z3, &z1z1); fe_sub(&FQ, &z3, &z3, &z2z2);
  fe_mul(&FQ, &z3, &z3, &h);
  r->x = x3; r->y = y3; r->z = z3;
}

static inline void g1j_add_affine(g1j_t *r, const g1j_t *p, const g1a_t *q) {
  g1j_t qj; g1j_from_affine(&qj, q);
  g1j_add(r, p, &qj);
}

static inline void g1j_neg(g1j_t *r, const g1j_t *p) { r->x = p->x; fe_neg(&FQ, &r->y, &p->y); r->z = p->z; }

static inline void g1j_to_affine(g1a_t *r, const g1j_t *p) {
  if (g1j_is_identity(p)) { fe_zero(&r->x); fe_zero(&r->y); return; }
  fe_t zi, zi2, zi3;
  fe_inv(&FQ, &zi, &p->z);
  fe_sqr(&FQ, &zi2, &zi);
  fe_mul(&FQ, &zi3, &zi2, &zi);
  fe_mul(&FQ, &r->x, &p->x, &zi2);
  fe_mul(&FQ, &r->y, &p->y, &zi3);
}

/* k * P, k canonical (non-Montgomery) 4-limb integer; double-and-add MSB first */
static inline void g1j_mul(g1j_t *r, const g1a_t *p, const uint64_t k[4]) {
  g1j_t acc; g1j_set_identity(&acc);
  for (int i = 255; i >= 0; --i) {
    g1j_dbl(&acc, &acc);
    if ((k[i >> 6] >> (i & 63)) & 1) g1j_add_affine(&acc, &acc, p);
  }
  *r = acc;
}

#endif
