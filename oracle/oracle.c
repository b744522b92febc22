/* CPU ORACLE entry points (ctypes) -- test infrastructure only; see bn254_ref.h header.
 *
 * Restates, in plain C, the halo2_proofs routines on the BFV prover hot path (third-party, not
 * under /root/reference; call site examples/bfv.rs:311; SURVEY.md Appendix B "Core numeric routines"):
 *   best_fft(a, omega, log_n)    -> orc_fft            bit-reversal + radix-2 DIT, in place
 *   EvaluationDomain::ifft       -> orc_ntt(inverse)   best_fft with omega^-1, then * n^-1
 *   coeff_to_extended            -> orc_coset_ntt      a[i] *= g^i on the zero-padded vector, then fft
 *   best_multiexp(coeffs, bases) -> orc_msm            exact sum_i s_i*P_i (plain windowed Pippenger)
 *   batch_invert                 -> orc_fr_batch_inv   Montgomery trick, zero stays zero
 *   eval_polynomial              -> orc_fr_horner
 * All Fr/Fq buffers are arrays of 4 x u64 little-endian limbs in Montgomery form.
 */
#include <stdlib.h>
#include "bn254_ref.h"

#define EXPORT __attribute__((visibility("default")))

static const field_t *pick(int which) { return which ? &FQ : &FR; }

/* ---- element-wise field ops (which: 0 = Fr, 1 = Fq) ---- */
EXPORT void orc_fe_mul(int which, const fe_t *a, const fe_t *b, fe_t *out, size_t n) {
  const field_t *F = pick(which);
  for (size_t i = 0; i < n; ++i) fe_mul(F, &out[i], &a[i], &b[i]);
}
EXPORT void orc_fe_add(int which, const fe_t *a, const fe_t *b, fe_t *out, size_t n) {
  const field_t *F = pick(which);
  for (size_t i = 0; i < n; ++i) fe_add(F, &out[i], &a[i], &b[i]);
}
EXPORT void orc_fe_sub(int which, const fe_t *a, const fe_t *b, fe_t *out, size_t n) {
  const field_t *F = pick(which);
  for (size_t i = 0; i < n; ++i) fe_sub(F, &out[i], &a[i], &b[i]);
}
EXPORT void orc_fe_inv(int which, const fe_t *a, fe_t *out, size_t n) {
  const field_t *F = pick(which);
  for (size_t i = 0; i < n; ++i) fe_inv(F, &out[i], &a[i]);
}
EXPORT void orc_fe_to_mont(int which, const fe_t *a, fe_t *out, size_t n) {
  const field_t *F = pick(which);
  for (size_t i = 0; i < n; ++i) fe_to_mont(F, &out[i], &a[i]);
}
EXPORT void orc_fe_from_mont(int which, const fe_t *a, fe_t *out, size_t n) {
  const field_t *F = pick(which);
  for (size_t i = 0; i < n; ++i) fe_from_mont(F, &out[i], &a[i]);
}

/* batch_invert: Montgomery trick, zeros skipped (stay zero) */
EXPORT void orc_fr_batch_inv(fe_t *a, size_t n) {
  fe_t *pre = (fe_t *)malloc(sizeof(fe_t) * (n ? n : 1));
  fe_t acc; fe_one(&FR, &acc);
  for (size_t i = 0; i < n; ++i) {
    pre[i] = acc;
    if (!fe_is_zero(&a[i])) fe_mul(&FR, &acc, &acc, &a[i]);
  }
  fe_inv(&FR, &acc, &acc);
  for (size_t i = n; i-- > 0;) {
    if (fe_is_zero(&a[i])) continue;
    fe_t t; fe_mul(&FR, &t, &acc, &pre[i]);
    fe_mul(&FR, &acc, &acc, &a[i]);
    a[i] = t;
  }
  free(pre);
}

/* Horner: out = sum_i poly[i] x^i */
EXPORT void orc_fr_horner(const fe_t *poly, size_t n, const fe_t *x, fe_t *out) {
  fe_t acc; fe_zero(&acc);
  for (size_t i = n; i-- > 0;) {
    fe_mul(&FR, &acc, &acc, x);
    fe_add(&FR, &acc, &acc, &poly[i]);
  }
  *out = acc;
}


/* ---- vector helpers used by the oracle prover (oracle/halo2_ref.py) ---- */
EXPORT void orc_fr_scale(const fe_t *a, const fe_t *s, fe_t *out, size_t n) {
#pragma omp parallel for if (n > 4096)
  for (size_t i = 0; i < n; ++i) fe_mul(&FR, &out[i], &a[i], s);
}
EXPORT void orc_fr_add_scalar(const fe_t *a, const fe_t *s, fe_t *out, size_t n) {
  for (size_t i = 0; i < n; ++i) fe_add(&FR, &out[i], &a[i], s);
}
/* acc[i] += s * x[i] */
EXPORT void orc_fr_axpy(fe_t *acc, const fe_t *x, const fe_t *s, size_t n) {
#pragma omp parallel for if (n > 4096)
  for (size_t i = 0; i < n; ++i) { fe_t t; fe_mul(&FR, &t, &x[i], s); fe_add(&FR, &acc[i], &acc[i], &t); }
}
EXPORT void orc_fr_mul_vec(const fe_t *a, const fe_t *b, fe_t *out, size_t n) {
#pragma omp parallel for if (n > 4096)
  for (size_t i = 0; i < n; ++i) fe_mul(&FR, &out[i], &a[i], &b[i]);
}
/* out[0] = init, out[i+1] = out[i] * a[i]  (out has n+1 entries) */
EXPORT void orc_fr_prefix_prod(const fe_t *a, const fe_t *init, fe_t *out, size_t n) {
  out[0] = *init;
  for (size_t i = 0; i < n; ++i) fe_mul(&FR, &out[i + 1], &out[i], &a[i]);
}
/* out[i] = start * base^i */
EXPORT void orc_fr_powers(const fe_t *start, const fe_t *base, fe_t *out, size_t n) {
  fe_t cur = *start;
  for (size_t i = 0; i < n; ++i) { out[i] = cur; fe_mul(&FR, &cur, &cur, base); }
}
/* q(X) = (p(X) - p(root)) / (X - root);  q has n-1 coefficients, out[n-1] = 0 */
EXPORT void orc_fr_div_linear(const fe_t *p, size_t n, const fe_t *root, fe_t *out) {
  fe_t carry; fe_zero(&carry);
  fe_zero(&out[n - 1]);
  for (size_t i = n - 1; i >= 1; --i) {
    fe_t t; fe_mul(&FR, &t, &carry, root);
    fe_add(&FR, &carry, &p[i], &t);
    out[i - 1] = carry;
  }
}
/* out = sum_j s[j] * cols[j]  (cols: n_cols x n contiguous) */
EXPORT void orc_fr_lincomb(const fe_t *cols, size_t n_cols, size_t n, const fe_t *s, fe_t *out) {
#pragma omp parallel for if (n > 1024)
  for (size_t i = 0; i < n; ++i) {
    fe_t acc; fe_zero(&acc);
    for (size_t j = 0; j < n_cols; ++j) { fe_t t; fe_mul(&FR, &t, &cols[j * n + i], &s[j]); fe_add(&FR, &acc, &acc, &t); }
    out[i] = acc;
  }
}
/* many Horner evaluations: out[j] = cols[j](x[j]) */
EXPORT void orc_fr_horner_batch(const fe_t *cols, size_t n_cols, size_t n, const fe_t *x, fe_t *out) {
#pragma omp parallel for schedule(dynamic)
  for (size_t j = 0; j < n_cols; ++j) orc_fr_horner(cols + j * n, n, &x[j], &out[j]);
}

/* ---- NTT ---- */
static const uint64_t ROOT_OF_UNITY_CANON[4] = {  /* 7^((r-1)/2^28), canonical form; SURVEY section 4 KAT 4 */
  0xd34f1ed960c37c9cULL, 0x3215cf6dd39329c8ULL, 0x98865ea93dd31f74ULL, 0x03ddb9f5166d18b7ULL};

static void root_of_unity(int log_n, fe_t *w) {
  fe_t c; memcpy(c.l, ROOT_OF_UNITY_CANON, 32);
  fe_to_mont(&FR, w, &c);
  for (int i = 0; i < 28 - log_n; ++i) fe_sqr(&FR, w, w);
}

EXPORT void orc_root_of_unity(int log_n, fe_t *out) { root_of_unity(log_n, out); }

static uint32_t bitrev32(uint32_t x, int bits) {
  uint32_t r = 0;
  for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
}

/* best_fft on one vector of length 2^log_n with the given omega */
EXPORT void orc_fft(fe_t *a, int log_n, const fe_t *omega) {
  size_t n = (size_t)1 << log_n;
  for (size_t k = 0; k < n; ++k) {
    size_t rk = bitrev32((uint32_t)k, log_n);
    if (k < rk) { fe_t t = a[k]; a[k] = a[rk]; a[rk] = t; }
  }
  size_t m = 1;
  for (int s = 0; s < log_n; ++s) {
    fe_t wm = *omega;                       /* omega^(n/(2m)) */
    for (size_t e = n / (2 * m); e > 1; e >>= 1) fe_sqr(&FR, &wm, &wm);
    for (size_t k = 0; k < n; k += 2 * m) {
      fe_t w; fe_one(&FR, &w);
      for (size_t j = 0; j < m; ++j) {
        fe_t t; fe_mul(&FR, &t, &a[k + j + m], &w);
        fe_sub(&FR, &a[k + j + m], &a[k + j], &t);
        fe_add(&FR, &a[k + j], &a[k + j], &t);
        fe_mul(&FR, &w, &w, &wm);
      }
    }
    m *= 2;
  }
}

/* n_cols independent vectors, contiguous (column c at a + c*n). inverse: omega^-1 and * n^-1. */
EXPORT void orc_ntt(fe_t *a, size_t n_cols, int log_n, int inverse) {
  size_t n = (size_t)1 << log_n;
  fe_t w; root_of_unity(log_n, &w);
  fe_t ninv; fe_zero(&ninv);
  if (inverse) {
    fe_inv(&FR, &w, &w);
    fe_t nn; fe_from_u64(&FR, &nn, (uint64_t)n);
    fe_inv(&FR, &ninv, &nn);
  }
#pragma omp parallel for schedule(dynamic)
  for (size_t c = 0; c < n_cols; ++c) {
    fe_t *v = a + c * n;
    orc_fft(v, log_n, &w);
    if (inverse) for (size_t i = 0; i < n; ++i) fe_mul(&FR, &v[i], &v[i], &ninv);
  }
}

/* coeff_to_extended: in (n coeffs) -> out (2^log_ext evaluations on the coset g*<omega_ext>).
 * inverse = extended_to_coeff: ifft on the extended domain, then a[i] *= g^-i. */
EXPORT void orc_coset_ntt(const fe_t *in, size_t n_in, fe_t *out, int log_ext, const fe_t *g, int inverse) {
  size_t ne = (size_t)1 << log_ext;
  fe_t w; root_of_unity(log_ext, &w);
  if (!inverse) {
    fe_t gi; fe_one(&FR, &gi);
    for (size_t i = 0; i < ne; ++i) {
      if (i < n_in) { fe_mul(&FR, &out[i], &in[i], &gi); fe_mul(&FR, &gi, &gi, g); }
      else fe_zero(&out[i]);
    }
    orc_fft(out, log_ext, &w);
  } else {
    for (size_t i = 0; i < ne; ++i) out[i] = in[i];
    fe_inv(&FR, &w, &w);
    orc_fft(out, log_ext, &w);
    fe_t nn, ninv, ginv, gi;
    fe_from_u64(&FR, &nn, (uint64_t)ne); fe_inv(&FR, &ninv, &nn);
    fe_inv(&FR, &ginv, g);
    gi = ninv;
    for (size_t i = 0; i < ne; ++i) { fe_mul(&FR, &out[i], &out[i], &gi); fe_mul(&FR, &gi, &gi, &ginv); }
  }
}

/* batched forward coset extension: n_cols coefficient vectors of n_in -> n_cols x 2^log_ext */
EXPORT void orc_coset_ntt_cols(const fe_t *in, size_t n_cols, size_t n_in, fe_t *out, int log_ext, const fe_t *g) {
  size_t ne = (size_t)1 << log_ext;
#pragma omp parallel for schedule(dynamic)
  for (size_t c = 0; c < n_cols; ++c) orc_coset_ntt(in + c * n_in, n_in, out + c * ne, log_ext, g, 0);
}

/* ---- G1 ---- */
EXPORT void orc_g1_add(const g1a_t *a, const g1a_t *b, g1a_t *out, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    g1j_t pa, r; g1j_from_affine(&pa, &a[i]);
    g1j_add_affine(&r, &pa, &b[i]);
    g1j_to_affine(&out[i], &r);
  }
}

/* out[i] = k[i] * p[i]; k in Montgomery Fr form */
EXPORT void orc_g1_mul(const g1a_t *p, const fe_t *k, g1a_t *out, size_t n) {
#pragma omp parallel for schedule(dynamic, 16) if (n > 64)
  for (size_t i = 0; i < n; ++i) {
    fe_t kc; fe_from_mont(&FR, &kc, &k[i]);
    g1j_t r; g1j_mul(&r, &p[i], kc.l);
    g1j_to_affine(&out[i], &r);
  }
}

EXPORT int orc_g1_on_curve(const g1a_t *p) {
  if (g1a_is_identity(p)) return 1;
  fe_t y2, x3, b;
  fe_sqr(&FQ, &y2, &p->y);
  fe_sqr(&FQ, &x3, &p->x); fe_mul(&FQ, &x3, &x3, &p->x);
  fe_from_u64(&FQ, &b, 3);
  fe_add(&FQ, &x3, &x3, &b);
  return fe_eq(&y2, &x3);
}

static unsigned get_window(const uint64_t k[4], int lo, int c) {
  unsigned v = 0;
  for (int b = 0; b < c; ++b) {
    int bit = lo + b;
    if (bit < 256) v |= (unsigned)((k[bit >> 6] >> (bit & 63)) & 1) << b;
  }
  return v;
}

/* one MSM: scalars (Montgomery Fr), bases affine, n terms -> affine result. Unsigned-window Pippenger. */
static void msm_one(const fe_t *scalars, const g1a_t *bases, size_t n, g1a_t *out) {
  int c = n < 32 ? 3 : n < 1024 ? 7 : n < 65536 ? 10 : 13;
  int nwin = (254 + c - 1) / c;
  uint64_t(*k)[4] = malloc(sizeof(uint64_t[4]) * (n ? n : 1));
  for (size_t i = 0; i < n; ++i) { fe_t t; fe_from_mont(&FR, &t, &scalars[i]); memcpy(k[i], t.l, 32); }
  size_t nb = ((size_t)1 << c) - 1;
  g1j_t *buckets = malloc(sizeof(g1j_t) * nb);
  g1j_t total; g1j_set_identity(&total);
  for (int w = nwin - 1; w >= 0; --w) {
    for (int d = 0; d < c; ++d) g1j_dbl(&total, &total);
    for (size_t b = 0; b < nb; ++b) g1j_set_identity(&buckets[b]);
    for (size_t i = 0; i < n; ++i) {
      unsigned d = get_window(k[i], w * c, c);
      if (d) g1j_add_affine(&buckets[d - 1], &buckets[d - 1], &bases[i]);
    }
    g1j_t run, sum; g1j_set_identity(&run); g1j_set_identity(&sum);
    for (size_t b = nb; b-- > 0;) {
      g1j_add(&run, &run, &buckets[b]);
      g1j_add(&sum, &sum, &run);
    }
    g1j_add(&total, &total, &sum);
  }
  g1j_to_affine(out, &total);
  free(buckets); free(k);
}

/* n_cols MSMs sharing one basis: scalars[c*n + i], out[c] */
EXPORT void orc_msm(const fe_t *scalars, size_t n_cols, const g1a_t *bases, size_t n, g1a_t *out) {
#pragma omp parallel for schedule(dynamic)
  for (size_t c = 0; c < n_cols; ++c) msm_one(scalars + c * n, bases, n, &out[c]);
}

/* naive sum_i s_i * P_i by double-and-add (cross-check of the Pippenger above; small n) */
EXPORT void orc_msm_naive(const fe_t *scalars, const g1a_t *bases, size_t n, g1a_t *out) {
  g1j_t acc; g1j_set_identity(&acc);
  for (size_t i = 0; i < n; ++i) {
    fe_t kc; fe_from_mont(&FR, &kc, &scalars[i]);
    g1j_t t; g1j_mul(&t, &bases[i], kc.l);
    g1j_add(&acc, &acc, &t);
  }
  g1j_to_affine(out, &acc);
}

/* bases[i] = (start * step^i) * G for G = (1,2): deterministic test/bench basis (not an SRS).
 * Built by repeated scalar multiplication of the previous point -- n small scalar muls. */
EXPORT void orc_g1_powers(const fe_t *start, const fe_t *step, g1a_t *out, size_t n) {
  g1a_t g; fe_from_u64(&FQ, &g.x, 1); fe_from_u64(&FQ, &g.y, 2);
  fe_t sc; fe_from_mont(&FR, &sc, start);
  fe_t stc; fe_from_mont(&FR, &stc, step);
  g1j_t cur; g1j_mul(&cur, &g, sc.l);
  for (size_t i = 0; i < n; ++i) {
    g1j_to_affine(&out[i], &cur);
    g1j_mul(&cur, &out[i], stc.l);
  }
}

EXPORT int orc_num_threads(void) {
#ifdef _OPENMP
  extern int omp_get_max_threads(void);
  return omp_get_max_threads();
#else
  return 1;
#endif
}
