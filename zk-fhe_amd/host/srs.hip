// The structured reference string on one MI355X (include/zkfhe.h "SRS"): replaces halo2-scaffold `gen_srs` /
// ParamsKZG::setup (third-party, reached from reference examples/bfv.rs:311; README.md:34 "unsafe" seeded setup).
#include <unistd.h>

#include "prover_internal.hpp"
#include "srs_secret.hpp"

// The monomial half of an SRS (g: the quotient pieces, the random polynomial -- calls of one to three columns) gets this share of
// the table budget of the Lagrange half (every wide commitment): 13-bit digits (43 GB) under the 160 GB service profile, 11 (13 GB)
// under the library default of 48 GB.  A wider table there buys nothing a proof can measure and costs tens of GB.
static constexpr double ZK_MONOMIAL_TABLE_SHARE = 0.3;

extern "C" {

static int srs_create_impl(zkfhe_ctx *ctx, zkfhe_comm *comm, uint32_t k, const uint8_t *seed, size_t seed_len, zkfhe_srs **out) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, out != nullptr && k >= 3 && k <= 20 && (seed != nullptr || seed_len == 0));
  const size_t n = (size_t)1 << k;
  size_t lo = 0, hi = n;
  if (comm) zkfhe_comm_point_range(comm, n, &lo, &hi);   // this rank's bases
  const size_t nl = hi - lo;
  ZK_ARG(ctx, nl > 0);
  const U256 s_canon = srs_secret(seed, seed_len);   // srs_secret.hpp: the reference's ChaCha20 derivation or the seeded one
  const Fr s = mont(s_canon);
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
    // the monomial half serves the calls of 1-3 columns only (random polynomial, quotient pieces): half the table budget
    CK(zk_basis_create_scaled(ctx, (const zkfhe_g1_affine *)host.data(), nl, 0, which == 0 ? ZK_MONOMIAL_TABLE_SHARE : 1.0, which == 0 ? &srs->g : &srs->g_lagrange));
    if (!comm) (which == 0 ? srs->g_host : srs->gl_host) = host;   // zkfhe_srs_save writes them
    if (which == 1) {
      static int small_c = -1;
      if (small_c < 0) {
        const char *e = getenv("ZKFHE_SMALL_C");
        small_c = e ? atoi(e) : 10;
      }
      if (small_c > 0 && k >= 12 && k <= 14 && !zkfhe_basis_has_multiples(srs->g_lagrange)) CK(zkfhe_basis_create(ctx, (const zkfhe_g1_affine *)host.data(), nl, small_c, &srs->g_lagrange_small));
    }
  }
  zk_srs_g2_from_secret(s_canon, srs->g2_raw, srs->sg2_raw);
  srs->have_g2 = true;
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
  srs->g_host.assign((const G1Affine *)g_host, (const G1Affine *)g_host + n);
  srs->gl_host.assign((const G1Affine *)g_lagrange_host, (const G1Affine *)g_lagrange_host + n);
  int rc = zk_basis_create_scaled(ctx, g_host, n, 0, ZK_MONOMIAL_TABLE_SHARE, &srs->g);
  if (!rc) rc = zk_basis_create_scaled(ctx, g_lagrange_host, n, 0, 1.0, &srs->g_lagrange);
  if (!rc && k >= 12 && k <= 14 && !zkfhe_basis_has_multiples(srs->g_lagrange)) rc = zkfhe_basis_create(ctx, g_lagrange_host, n, 10, &srs->g_lagrange_small);
  if (rc) {
    zkfhe_srs_destroy(ctx, srs);
    return rc;
  }
  *out = srs;
  return ZKFHE_OK;
}

int zkfhe_srs_g2(const zkfhe_srs *srs, uint8_t g2_le[128], uint8_t s_g2_le[128]) {
  if (!srs || !g2_le || !s_g2_le || !srs->have_g2) return ZKFHE_EINVAL;
  return zk_g2_raw_to_canon(srs->g2_raw, g2_le) && zk_g2_raw_to_canon(srs->sg2_raw, s_g2_le) ? ZKFHE_OK : ZKFHE_EINVAL;
}

int zkfhe_srs_set_g2(zkfhe_srs *srs, const uint8_t g2_le[128], const uint8_t s_g2_le[128]) {
  if (!srs || !g2_le || !s_g2_le) return ZKFHE_EINVAL;
  uint8_t a[128], b[128];
  if (!zk_g2_canon_to_raw(g2_le, a) || !zk_g2_canon_to_raw(s_g2_le, b)) return ZKFHE_EINVAL;
  memcpy(srs->g2_raw, a, 128);
  memcpy(srs->sg2_raw, b, 128);
  srs->have_g2 = true;
  return ZKFHE_OK;
}

// halo2 ParamsKZG::write (SerdeFormat::RawBytes): u32 k | g | g_lagrange | g2 | s_g2, raw Montgomery coordinates
int zkfhe_srs_save(zkfhe_ctx *ctx, const zkfhe_srs *srs, const char *path) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, srs && path);
  const size_t n = (size_t)1 << srs->k;
  if (srs->sharded()) return zk_fail_msg(ctx, ZKFHE_EINVAL, "zkfhe_srs_save: a sharded SRS holds only a slice of the points");
  if (srs->g_host.size() != n || srs->gl_host.size() != n)
    return zk_fail_msg(ctx, ZKFHE_EINVAL, "zkfhe_srs_save: this SRS keeps no host copy of its points (made over a communicator, or zkfhe_srs_drop_host_copy was called)");
  if (!srs->have_g2) return zk_fail_msg(ctx, ZKFHE_EINVAL, "zkfhe_srs_save: the SRS has no G2 half (zkfhe_srs_set_g2)");
  // written aside under a name of this process' own and renamed into place: two ranks (or two CLI runs) that both find the file
  // missing and derive it at once never write into each other's temporary, and a reader sees the old file, none, or a whole one
  static std::atomic<unsigned> serial{0};
  const std::string tmp = std::string(path) + ".tmp." + std::to_string((long)getpid()) + "." + std::to_string(serial.fetch_add(1));
  FILE *f = fopen(tmp.c_str(), "wb");
  if (!f) return zk_fail_msg(ctx, ZKFHE_EINVAL, std::string("cannot write ") + tmp);
  const uint32_t k = srs->k;
  bool ok = fwrite(&k, 4, 1, f) == 1 && fwrite(srs->g_host.data(), 64, n, f) == n && fwrite(srs->gl_host.data(), 64, n, f) == n &&
            fwrite(srs->g2_raw, 128, 1, f) == 1 && fwrite(srs->sg2_raw, 128, 1, f) == 1;
  ok = (fclose(f) == 0) && ok;
  if (!ok || rename(tmp.c_str(), path) != 0) {
    remove(tmp.c_str());
    return zk_fail_msg(ctx, ZKFHE_EINVAL, std::string("writing ") + path + " failed");
  }
  return ZKFHE_OK;
}

// An unsharded SRS keeps host copies of both point vectors (64 B x 2^k each: 128 MB at k = 20) only so that zkfhe_srs_save can write
// them; a prover that has saved (or never will) gives them back with this.  Saving afterwards is refused with a message that says so.
int zkfhe_srs_drop_host_copy(zkfhe_srs *srs) {
  if (!srs) return ZKFHE_EINVAL;
  std::vector<G1Affine>().swap(srs->g_host);
  std::vector<G1Affine>().swap(srs->gl_host);
  return ZKFHE_OK;
}

int zkfhe_srs_load(zkfhe_ctx *ctx, const char *path, zkfhe_srs **out) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, path && out);
  *out = nullptr;
  FILE *f = fopen(path, "rb");
  if (!f) return zk_fail_msg(ctx, ZKFHE_EINVAL, std::string("cannot open ") + path);
  uint32_t k = 0;
  std::vector<G1Affine> g, gl;
  uint8_t raw[256], canon[128];
  bool ok = fread(&k, 4, 1, f) == 1 && k >= 3 && k <= 20;
  if (ok) {
    const size_t n = (size_t)1 << k;
    g.resize(n), gl.resize(n);
    ok = fread(g.data(), 64, n, f) == n && fread(gl.data(), 64, n, f) == n && fread(raw, 256, 1, f) == 1 && fgetc(f) == EOF;
  }
  fclose(f);
  if (!ok) return zk_fail_msg(ctx, ZKFHE_EINVAL, std::string(path) + ": not a params file (u32 k | 2^k g | 2^k g_lagrange | g2 | s_g2, RawBytes), or k outside 3..20");
  // what halo2's RawBytes read checks: reduced coordinates, y^2 = x^3 + 3 (the identity (0, 0) cannot occur in an SRS)
  const zk::Fq three = zk::fp_to_mont<zk::FqP>([] { zk::Fq t = zk::Fq::zero(); t.l[0] = 3; return t; }());
  static const U256 QM = {{0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}};
  for (const auto *v : {&g, &gl})
    for (const G1Affine &p : *v) {
      U256 x, y;
      memcpy(x.l, p.x.l, 32);
      memcpy(y.l, p.y.l, 32);
      if (!(x < QM) || !(y < QM) || !(p.y * p.y == p.x * p.x * p.x + three)) return zk_fail_msg(ctx, ZKFHE_EINVAL, std::string(path) + ": a G1 point is not on the curve");
    }
  if (!zk_g2_raw_to_canon(raw, canon) || !zk_g2_raw_to_canon(raw + 128, canon)) return zk_fail_msg(ctx, ZKFHE_EINVAL, std::string(path) + ": a G2 point is not on the curve");
  zkfhe_srs *srs = nullptr;
  CK(zkfhe_srs_from_points(ctx, k, (const zkfhe_g1_affine *)g.data(), (const zkfhe_g1_affine *)gl.data(), &srs));
  memcpy(srs->g2_raw, raw, 128);
  memcpy(srs->sg2_raw, raw + 128, 128);
  srs->have_g2 = true;
  *out = srs;
  return ZKFHE_OK;
}

int zkfhe_srs_table_bits(const zkfhe_srs *srs, int *wide_calls) {
  if (wide_calls) *wide_calls = 0;
  return srs ? zkfhe_basis_table_bits(srs->g_lagrange, wide_calls) : 0;
}

// What the digit-multiple tables of this SRS look like: bits[0] / bits[1] = digit width of the monomial (g) and Lagrange halves
// (0: none), *bytes = resident bytes of both, *narrowed = 1 when a half got a narrower table than its budget allowed because the
// device did not have the room (another SRS alive, other tenants) -- the calls are then slower, never wrong.  All outputs optional.
int zkfhe_srs_table_info(const zkfhe_srs *srs, int bits[2], uint64_t *bytes, int *narrowed) {
  if (!srs) return ZKFHE_EINVAL;
  int n0 = 0, n1 = 0;
  const size_t b0 = zkfhe_basis_table_bytes(srs->g, &n0), b1 = zkfhe_basis_table_bytes(srs->g_lagrange, &n1);
  if (bits) {
    bits[0] = zkfhe_basis_table_bits(srs->g, nullptr);
    bits[1] = zkfhe_basis_table_bits(srs->g_lagrange, nullptr);
  }
  if (bytes) *bytes = (uint64_t)b0 + (uint64_t)b1;
  if (narrowed) *narrowed = (n0 | n1) ? 1 : 0;
  return ZKFHE_OK;
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
