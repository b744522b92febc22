// The structured reference string on one MI355X (include/zkfhe.h "SRS"): replaces halo2-scaffold `gen_srs` /
// ParamsKZG::setup (third-party, reached from reference examples/bfv.rs:311; README.md:34 "unsafe" seeded setup).
#include "prover_internal.hpp"

extern "C" {

static int srs_create_impl(zkfhe_ctx *ctx, zkfhe_comm *comm, uint32_t k, const uint8_t *seed, size_t seed_len, zkfhe_srs **out) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, out != nullptr && k >= 3 && k <= 20 && (seed != nullptr || seed_len == 0));
  const size_t n = (size_t)1 << k;
  size_t lo = 0, hi = n;
  if (comm) zkfhe_comm_point_range(comm, n, &lo, &hi);   // this rank's bases
  const size_t nl = hi - lo;
  ZK_ARG(ctx, nl > 0);
  Blake2b h(64, "zkfhe-srs");
  h.update(seed, seed_len);
  uint8_t d[64];
  h.digest(d);
  const Fr s = mont(from_bytes_wide(d));
  const NttDomain *dom;
  CK(zk_domain(ctx, (int)k, &dom));
  // everything this function owns until it succeeds: released on every early return (an error code of a later step used
  // to leak the two buffers and the half-built SRS with its bases)
  struct Guard {
    zkfhe_ctx *ctx;
    DevBuf sc, pts;
    zkfhe_srs *srs = nullptr;
    ~Guard() {
      sc.release();
      pts.release();
      if (srs) zkfhe_srs_destroy(ctx, srs);
    }
  } guard{ctx};
  DevBuf &sc = guard.sc, &pts = guard.pts;
  CK(sc.alloc(ctx, nl * 32));
  CK(pts.alloc(ctx, nl * 64));
  G1Affine gen;
  gen.x = zk::fp_to_mont<zk::FqP>([] { zk::Fq t = zk::Fq::zero(); t.l[0] = 1; return t; }());
  gen.y = zk::fp_to_mont<zk::FqP>([] { zk::Fq t = zk::Fq::zero(); t.l[0] = 2; return t; }());
  std::vector<G1Affine> host(nl);
  zkfhe_srs *srs = guard.srs = new zkfhe_srs();
  srs->k = k;
  srs->comm = comm;
  srs->lo = lo;
  srs->hi = hi;
  const unsigned gr = (unsigned)((nl + 255) / 256);
  for (int which = 0; which < 2; ++which) {
    if (which == 0) {
      zkp::k_powers<<<gr, 256, 0, ctx->stream>>>(fr_pow(s, lo), s, sc.fr(), nl);   // s^(lo + i)
      ZK_LAUNCH_CHECK(ctx);
    } else {
      zkp::k_srs_den<<<gr, 256, 0, ctx->stream>>>(dom->fwd + lo, s, mont_u64(n), sc.fr(), nl);
      ZK_LAUNCH_CHECK(ctx);
      CK(zkfhe_fr_batch_invert(ctx, (zkfhe_fr *)sc.p, nl));
      zkp::k_srs_li<<<gr, 256, 0, ctx->stream>>>(dom->fwd + lo, fr_pow(s, n) - Fr::one(), sc.fr(), nl);
      ZK_LAUNCH_CHECK(ctx);
    }
    zkp::k_fill_point<<<gr, 256, 0, ctx->stream>>>(gen, (G1Affine *)pts.p, nl);
    ZK_LAUNCH_CHECK(ctx);
    CK(zkfhe_g1_mul(ctx, (const zkfhe_g1_affine *)pts.p, (const zkfhe_fr *)sc.p, (zkfhe_g1_affine *)pts.p, nl));
    CK(zkfhe_download(ctx, host.data(), pts.p, nl * 64));
    CK(zkfhe_basis_create(ctx, (const zkfhe_g1_affine *)host.data(), nl, 0, which == 0 ? &srs->g : &srs->g_lagrange));
    if (which == 1) {
      static int small_c = -1;
      if (small_c < 0) {
        const char *e = getenv("ZKFHE_SMALL_C");
        small_c = e ? atoi(e) : 10;
      }
      if (small_c > 0 && k >= 12 && k <= 14 && !zkfhe_basis_has_multiples(srs->g_lagrange)) CK(zkfhe_basis_create(ctx, (const zkfhe_g1_affine *)host.data(), nl, small_c, &srs->g_lagrange_small));
    }
  }
  guard.srs = nullptr;   // success: the caller owns it (the buffers go with the guard)
  *out = srs;
  return ZKFHE_OK;
}

int zkfhe_srs_create(zkfhe_ctx *ctx, uint32_t k, const uint8_t *seed, size_t seed_len, zkfhe_srs **out) {
  return srs_create_impl(ctx, nullptr, k, seed, seed_len, out);
}

int zkfhe_srs_create_sharded(zkfhe_ctx *ctx, zkfhe_comm *comm, uint32_t k, const uint8_t *seed, size_t seed_len, zkfhe_srs **out) {
  if (!comm) return zk_fail_msg(ctx, ZKFHE_EINVAL, "comm is NULL");
  return srs_create_impl(ctx, comm, k, seed, seed_len, out);
}

int zkfhe_srs_from_points(zkfhe_ctx *ctx, uint32_t k, const zkfhe_g1_affine *g_host, const zkfhe_g1_affine *g_lagrange_host, zkfhe_srs **out) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, out != nullptr && g_host != nullptr && g_lagrange_host != nullptr && k >= 3 && k <= 20);
  const size_t n = (size_t)1 << k;
  zkfhe_srs *srs = new zkfhe_srs();
  srs->k = k;
  srs->hi = n;
  int rc = zkfhe_basis_create(ctx, g_host, n, 0, &srs->g);
  if (!rc) rc = zkfhe_basis_create(ctx, g_lagrange_host, n, 0, &srs->g_lagrange);
  if (!rc && k >= 12 && k <= 14 && !zkfhe_basis_has_multiples(srs->g_lagrange)) rc = zkfhe_basis_create(ctx, g_lagrange_host, n, 10, &srs->g_lagrange_small);
  if (rc) {
    zkfhe_srs_destroy(ctx, srs);
    return rc;
  }
  *out = srs;
  return ZKFHE_OK;
}

int zkfhe_srs_table_bits(const zkfhe_srs *srs, int *wide_calls) {
  if (wide_calls) *wide_calls = 0;
  return srs ? zkfhe_basis_table_bits(srs->g_lagrange, wide_calls) : 0;
}

int zkfhe_srs_destroy(zkfhe_ctx *ctx, zkfhe_srs *srs) {
  ZK_ENTER(ctx);
  if (!srs) return ZKFHE_OK;
  zkfhe_basis_destroy(ctx, srs->g);
  zkfhe_basis_destroy(ctx, srs->g_lagrange);
  zkfhe_basis_destroy(ctx, srs->g_lagrange_small);
  delete srs;
  return ZKFHE_OK;
}

}  // extern "C"
