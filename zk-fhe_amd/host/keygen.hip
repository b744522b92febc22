// keygen (README.md:28-38; halo2 keygen_vk / keygen_pk as driven by halo2-scaffold `run_eth`, reached from reference
// examples/bfv.rs:311), the per-context prover workspace, and the proving / verifying key files.
#include "prover_internal.hpp"

std::vector<BigInt> GpuPolyMul::mul_u64(const std::vector<uint64_t> &a, const std::vector<uint64_t> &b) {
  const U256 *host = mul_u64_raw(a, b);
  std::vector<BigInt> out(2 * a.size() - 1);
  for (size_t i = 0; i < out.size(); ++i) out[i] = fe::to_bigint(host[i]);
  return out;
}

const U256 *GpuPolyMul::mul_u64_raw(const std::vector<uint64_t> &a, const std::vector<uint64_t> &b) {
  const size_t n = a.size();
  if (ws->polyio.bytes < 2 * n * 8 + 2 * n * 32) throw std::runtime_error("polynomial too long for the prover workspace");
  uint64_t *da = (uint64_t *)ws->polyio.p, *db = da + n;
  Fr *dout = (Fr *)(db + n);
  // operands through the pinned ring, the product back into pinned memory: pageable copies take a process-wide staging
  // path in the runtime, and twenty proofs starting together queued on it (5 ms per product instead of 0.25)
  const U256 *host = ws->host_poly;
  int rc = up(ctx, ws, da, a.data(), n * 8);
  if (!rc) rc = up(ctx, ws, db, b.data(), n * 8);
  if (!rc) rc = zkfhe_witness_poly_mul_u64(ctx, da, db, n, (zkfhe_fr *)dout);
  if (!rc) rc = zkfhe_fr_from_mont(ctx, (const zkfhe_fr *)dout, (zkfhe_fr *)dout, 2 * n - 1);
  if (!rc && ws->host_poly && 2 * n - 1 <= ws->host_poly_len) {
    if (hipMemcpyAsync(ws->host_poly, dout, (2 * n - 1) * 32, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = ZKFHE_EHIP;
    if (!rc) rc = zkfhe_sync(ctx);
  } else if (!rc) {
    pageable.resize(2 * n - 1);
    host = pageable.data();
    rc = zkfhe_download(ctx, pageable.data(), dout, (2 * n - 1) * 32);
  }
  if (rc) throw std::runtime_error(std::string("GPU poly mul failed: ") + zkfhe_last_error(ctx));
  return host;
}

int up(zkfhe_ctx *ctx, Workspace *ws, void *dst, const void *src, size_t bytes) {
  if (!bytes) return ZKFHE_OK;
  const size_t need = (bytes + 63) & ~(size_t)63;
  if (!ws->ring || ws->ring_off + need > Workspace::RING_BYTES) return zkfhe_upload(ctx, dst, src, bytes);
  uint8_t *slot = ws->ring + ws->ring_off;
  ws->ring_off += need;
  memcpy(slot, src, bytes);
  ZK_HIP(ctx, hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, ctx->stream));
  return ZKFHE_OK;
}

int alloc_workspace(zkfhe_ctx *ctx, const CircuitConfig &c, Workspace *ws, int ext_rows) {
  const size_t n = c.n(), ne = (size_t)ext_rows * n, col = n * 32, ecol = ne * 32;
  const size_t n_all = (size_t)c.n_advice() + 3 * c.n_lookup + c.n_chunks() + 1;
  ws->n_all = n_all;
  CK(ws->all_l.alloc(ctx, n_all * col));
  CK(ws->all_ext.alloc(ctx, n_all * ecol));
  {
    size_t o = 0;
    auto take = [&](View &l, View &e, size_t cols) {
      l.p = (char *)ws->all_l.p + o * col;
      e.p = (char *)ws->all_ext.p + o * ecol;
      o += cols;
    };
    take(ws->adv_l, ws->adv_ext, c.n_advice());
    take(ws->la_l, ws->la_ext, c.n_lookup);
    take(ws->ls_l, ws->ls_ext, c.n_lookup);
    take(ws->pz_l, ws->pz_ext, c.n_chunks());
    take(ws->lz_l, ws->lz_ext, c.n_lookup);
    take(ws->inst_l, ws->inst_ext, 1);
  }
  ZK_HIP(ctx, hipMemsetAsync(ws->inst_l.p, 0, col, ctx->stream));   // the prover only ever writes the public-input rows
  ws->inst_count = 0;
  CK(ws->tmp_c.alloc(ctx, std::max<size_t>(n_all, c.n_perm()) * col));
  {
    // quotient partials: one row set (ext_rows * n values) per expression group -- the groups prove.hip cuts (8 gates, all RLC
    // gates, 32 chaining terms, 4 permutation chunks, 3 lookups per group; a rank of a sharded proof evaluates a contiguous share:
    // at most one more group per kind), not a flat 96 x 4 columns (6 GB at k = 19).  The buffer doubles as the receive side of the
    // sharded SHPLONK gather (ranks x rotation sets columns) and as scratch of the evaluation round: at least 96 columns.
    const size_t groups = ((size_t)c.n_gate() + 7) / 8 + 1 + 2 + ((size_t)c.n_chunks() + 31) / 32 + ((size_t)c.n_chunks() + 3) / 4 + ((size_t)c.n_lookup + 2) / 3 + 8;
    CK(ws->partials.alloc(ctx, std::max<size_t>(groups * (size_t)ext_rows, 96) * col));
  }
  CK(ws->h_ext.alloc(ctx, 4 * col));
  CK(ws->h_c.alloc(ctx, 4 * col));
  CK(ws->misc.alloc(ctx, 32 * col));
  CK(ws->points.alloc(ctx, std::max<size_t>(n_all, c.n_perm()) * 64 + 64));
  CK(ws->num.alloc(ctx, std::max<size_t>(c.n_chunks(), c.n_lookup) * col));
  CK(ws->den.alloc(ctx, std::max<size_t>(c.n_chunks(), c.n_lookup) * col));
  // ws->small is carved up at fixed offsets (prove.hip): [0, 128 K) beta * delta^i per permutation column, [128 K, 256 K) chunk /
  // lookup totals, 128 KB each for the quotient groups, y powers and inverses, then pointer / scalar / point lists, the SHPLONK
  // sets (700 K), gadget arguments (768 K) and the early-commitment patch lists (900 K, 920 K): the widest lists must fit
  if ((size_t)c.n_perm() > 4096 || (size_t)c.n_chunks() + c.n_lookup > 4096)
    return zk_fail_msg(ctx, ZKFHE_EINVAL, "configuration too wide for the prover workspace (more than 4096 permutation columns)");
  CK(ws->small.alloc(ctx, 1 << 20));
  const size_t max_items = (size_t)c.n_advice() + c.n_fixed() + 2 + c.n_perm() + c.n_chunks() + 3 * c.n_lookup;
  CK(ws->jobs.alloc(ctx, max_items * sizeof(zkp::EvalJob) + max_items * (sizeof(void *) + 32) + 256));
  CK(ws->evout.alloc(ctx, (max_items + 64) * 4 * 32));   // + the padding of the sharded evaluations' all-gather (equal slices per rank)
  CK(ws->polyio.alloc(ctx, 4 * c.n() * 32));
  ZK_HIP(ctx, hipHostMalloc((void **)&ws->host_adv, (size_t)c.n_advice() * col, hipHostMallocDefault));
  memset(ws->host_adv, 0, (size_t)c.n_advice() * col);
  ZK_HIP(ctx, hipHostMalloc((void **)&ws->host_blind, std::max<size_t>(2 * c.n_lookup, 2) * col, hipHostMallocDefault));
  return ZKFHE_OK;
}

void free_workspace(Workspace *ws) {
  (void)hipSetDevice(ws->all_l.device);
  for (DevBuf *b : ws->all()) b->release();
  for (void *h : {(void *)ws->host_adv, (void *)ws->host_blind, (void *)ws->host_pool, (void *)ws->host_poly, (void *)ws->host_pts, (void *)ws->ring, (void *)ws->host_rand_pt,
                  (void *)ws->host_early, (void *)ws->host_early_err, (void *)ws->host_out})
    if (h) (void)hipHostFree(h);
  if (ws->ev_pts) (void)hipEventDestroy(ws->ev_pts);
  if (ws->ev_rand) (void)hipEventDestroy(ws->ev_rand);
  if (ws->ev_early) (void)hipEventDestroy(ws->ev_early);
  if (ws->aux) (void)zkfhe_ctx_destroy(ws->aux);
  delete ws;
}

int get_workspace(zkfhe_ctx *ctx, zkfhe_bfv_pk *pk, Workspace **out) {
  std::lock_guard<std::mutex> lock(pk->mu);
  auto it = pk->workspaces.find(ctx->uid);
  if (it == pk->workspaces.end()) {
    Workspace *ws = new Workspace();
    int rc = alloc_workspace(ctx, pk->cfg, ws, pk->ext_rows);
    if (rc) {
      for (DevBuf *b : ws->all()) b->release();
      delete ws;
      return rc;
    }
    it = pk->workspaces.emplace(ctx->uid, ws).first;
  }
  if (it->second->all_l.device != ctx->device) return zk_fail_msg(ctx, ZKFHE_EINVAL, "prover workspace belongs to another device");
  *out = it->second;
  return ZKFHE_OK;
}

// coefficient form of `count` Lagrange columns (copy into tmp, iNTT), then coset-extend into ext
int extend_cols(zkfhe_ctx *ctx, zkfhe_bfv_pk *pk, Workspace *ws, const Fr *lagr, size_t count, Fr *ext) {
  if (!count) return ZKFHE_OK;
  const int rows = pk->ext_rows;
  const size_t n = pk->cfg.n();
  const Fr g = mont_u64(COSET_G);
  (void)n;
  return zk_extend_lagrange(ctx, lagr, ws->tmp_c.fr(), ext, count, (int)pk->cfg.k, 2, g, rows);
}

// extended-coset evaluations of the fixed / sigma columns, l_0 / l_last / l_active and X on the coset: derived from the
// Lagrange-form columns, rebuilt on load instead of stored (4x the size of the columns themselves)
static int build_resident_tables(zkfhe_ctx *ctx, zkfhe_bfv_pk *pk, Workspace *ws) {
  const CircuitConfig &cfg = pk->cfg;
  const size_t n = cfg.n(), cells = (size_t)cfg.n_perm() * n;
  const NttDomain *dom;
  CK(zk_domain(ctx, (int)cfg.k, &dom));
  const size_t R = (size_t)pk->ext_rows;
  CK(pk->fixed_ext.alloc(ctx, (size_t)cfg.n_fixed() * R * n * 32));
  CK(pk->sigma_ext.alloc(ctx, cells * R * 32));
  CK(pk->l_ext.alloc(ctx, 3 * R * n * 32));
  CK(pk->xs_ext.alloc(ctx, R * n * 32));
  CK(extend_cols(ctx, pk, ws, pk->fixed_l.fr(), cfg.n_fixed(), pk->fixed_ext.fr()));
  CK(extend_cols(ctx, pk, ws, pk->sigma_l.fr(), cfg.n_perm(), pk->sigma_ext.fr()));
  {
    std::vector<U256> l(3 * n, fe::zero());
    const size_t u = cfg.u();
    l[0] = fe::one();
    l[n + u] = fe::one();
    for (size_t i = 0; i < u; ++i) l[2 * n + i] = fe::one();
    CK(upload_canon(ctx, ws->misc.fr(), l.data(), 3 * n));
    CK(extend_cols(ctx, pk, ws, ws->misc.fr(), 3, pk->l_ext.fr()));
  }
  {
    const Fr wext = zk_fr_root_of_unity((int)cfg.k + 2);
    Fr shift = mont_u64(COSET_G);
    for (int k1 = 0; k1 < pk->ext_rows; ++k1) {
      zkp::k_powers<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(shift, dom->omega, pk->xs_ext.fr() + (size_t)k1 * n, n);
      ZK_LAUNCH_CHECK(ctx);
      shift = shift * wext;
    }
    if (pk->ext_rows == 3) {
      // the three-coset quotient's interpolation back to coefficients (prove.hip, k_ext3_combine) multiplies row t by
      // (g w_ext^t)^-i: constant per key, so the powers are resident instead of three launches per proof
      CK(pk->ext3_pw.alloc(ctx, 3 * n * 32));
      Fr gk = mont_u64(COSET_G);
      for (int t = 0; t < 3; ++t) {
        zkp::k_powers<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(Fr::one(), fr_inv(gk), pk->ext3_pw.fr() + (size_t)t * n, n);
        ZK_LAUNCH_CHECK(ctx);
        gk = gk * wext;
      }
    }
  }
  return ZKFHE_OK;
}

static U256 vk_digest_of(const zkfhe_bfv_pk *pk) { return vk_digest(pk->cfg, pk->fixed_commit, pk->sigma_commit); }

static int keygen_impl(zkfhe_ctx *ctx, const zkfhe_srs *srs, const char *input_json, const BfvParams &prm, CircuitConfig cfg, bool replay,
                zkfhe_bfv_pk **out) {
  const size_t n = cfg.n();
  ZK_ARG(ctx, srs->k == cfg.k);
  // ---- circuit structure (host)
  const CircuitInput in = CircuitInput::parse_json(input_json);
  Context ctx0(CTX_PHASE0, false, true), ctx_gate(CTX_GATE1, false, true), ctx_rlc(CTX_RLC1, true, true);
  std::vector<Cell> make_public;
  BfvState st = bfv_phase0(ctx0, in, prm, make_public);
  bfv_phase1(st, prm, ctx_gate, ctx_rlc, fe::zero());
  Assigner as(cfg, true);
  as.place(ctx0, replay);
  as.place(ctx_gate, replay);
  as.place(ctx_rlc, replay);
  as.place_lookups(ctx_gate);
  as.finish_structure(ctx0, ctx_gate, ctx_rlc, make_public);
  cfg.bp_gate0 = as.t.bp_gate0;
  cfg.bp_gate1 = as.t.bp_gate1;
  cfg.bp_rlc = as.t.bp_rlc;
  zkfhe_bfv_pk *pk = new zkfhe_bfv_pk();
  pk->cfg = cfg;
  pk->prm = prm;
  pk->ext_rows = getenv("ZKFHE_CHECK_QUOTIENT") ? 4 : 3;
  {
    // structure of the phase-1 gate stream for the GPU witness generator: where looked-up cells and deferred
    // inverses sit in the stream, and which stream range each gate column holds (break points)
    pk->gate1_cells = ctx_gate.advice.size();
    pk->n_lookup_cells = ctx_gate.lookup.size();
    pk->n_inv_slots = ctx_gate.inv_slots.size();
    std::vector<uint32_t> lsrc(ctx_gate.lookup.size());
    for (size_t i = 0; i < lsrc.size(); ++i) {
      ZK_ASSERT(ctx_gate.lookup[i].ctx == CTX_GATE1, "lookup cell outside the phase-1 gate context");
      lsrc[i] = ctx_gate.lookup[i].off;
    }
    std::vector<uint32_t> start(cfg.n_gate1, 0), len(cfg.n_gate1, 0);
    size_t s = 0;
    for (unsigned c = 0; c < cfg.n_gate1 && s < pk->gate1_cells; ++c) {
      start[c] = (uint32_t)s;
      if (c < cfg.bp_gate1.size()) {
        len[c] = cfg.bp_gate1[c] + 1;   // rows 0..bp: the last one is the duplicate of the next column's first cell
        s += cfg.bp_gate1[c];
      } else {
        len[c] = (uint32_t)(pk->gate1_cells - s);
        s = pk->gate1_cells;
      }
    }
    CK(pk->lookup_src.alloc(ctx, (lsrc.size() + 1) * 4));
    CK(pk->inv_slots.alloc(ctx, (ctx_gate.inv_slots.size() + 1) * 4));
    CK(pk->place_start.alloc(ctx, (start.size() + 1) * 4));
    CK(pk->place_len.alloc(ctx, (len.size() + 1) * 4));
    if (!lsrc.empty()) CK(zkfhe_upload(ctx, pk->lookup_src.p, lsrc.data(), lsrc.size() * 4));
    if (!ctx_gate.inv_slots.empty()) CK(zkfhe_upload(ctx, pk->inv_slots.p, ctx_gate.inv_slots.data(), ctx_gate.inv_slots.size() * 4));
    CK(zkfhe_upload(ctx, pk->place_start.p, start.data(), start.size() * 4));
    CK(zkfhe_upload(ctx, pk->place_len.p, len.data(), len.size() * 4));
  }
  // ---- sigma: union-find over cell ids, each class sorted by id is one cycle
  const size_t cells = (size_t)cfg.n_perm() * n;
  std::vector<uint32_t> parent(cells);
  for (size_t i = 0; i < cells; ++i) parent[i] = (uint32_t)i;
  auto find = [&](uint32_t x) {
    uint32_t root = x;
    while (parent[root] != root) root = parent[root];
    while (parent[x] != root) {
      uint32_t nx = parent[x];
      parent[x] = root;
      x = nx;
    }
    return root;
  };
  for (const auto &cp : as.t.copies) {
    uint32_t ra = find((uint32_t)cp.first), rb = find((uint32_t)cp.second);
    if (ra != rb) {
      if (ra < rb) parent[rb] = ra;
      else parent[ra] = rb;
    }
  }
  // the root of a class is its smallest id; walk ids in ascending order and chain each member to the previous one
  std::vector<uint32_t> target(cells), last_of(cells, 0xffffffffu), first_of(cells);
  for (size_t i = 0; i < cells; ++i) target[i] = (uint32_t)i;
  for (size_t i = 0; i < cells; ++i) {
    const uint32_t r = find((uint32_t)i);
    if (last_of[r] == 0xffffffffu) first_of[r] = (uint32_t)i;
    else target[last_of[r]] = (uint32_t)i;
    last_of[r] = (uint32_t)i;
  }
  for (size_t i = 0; i < cells; ++i)
    if (parent[i] == i && last_of[i] != 0xffffffffu) target[last_of[i]] = first_of[i];
  // ---- device: fixed + sigma in Lagrange form
  const NttDomain *dom;
  CK(zk_domain(ctx, (int)cfg.k, &dom));
  CK(pk->fixed_l.alloc(ctx, (size_t)cfg.n_fixed() * n * 32));
  CK(pk->sigma_l.alloc(ctx, cells * 32));
  CK(pk->dpow.alloc(ctx, (size_t)cfg.n_perm() * 32));
  for (unsigned c = 0; c < cfg.n_fixed(); ++c) {
    ZK_HIP(ctx, hipMemcpyAsync(pk->fixed_l.fr() + (size_t)c * n, as.t.fixed[c].data(), n * 32, hipMemcpyHostToDevice, ctx->stream));
  }
  CK(zkfhe_fr_to_mont(ctx, (const zkfhe_fr *)pk->fixed_l.p, (zkfhe_fr *)pk->fixed_l.p, (size_t)cfg.n_fixed() * n));
  U256 dcan;
  memcpy(dcan.l, DELTA_CANON, 32);
  zkp::k_powers<<<1, 256, 0, ctx->stream>>>(Fr::one(), mont(dcan), pk->dpow.fr(), cfg.n_perm());
  ZK_LAUNCH_CHECK(ctx);
  DevBuf tgt;
  CK(tgt.alloc(ctx, cells * 4));
  CK(zkfhe_upload(ctx, tgt.p, target.data(), cells * 4));
  zkp::k_sigma_values<<<(unsigned)((cells + 255) / 256), 256, 0, ctx->stream>>>((const uint32_t *)tgt.p, pk->dpow.fr(), dom->fwd, pk->sigma_l.fr(),
                                                                                 cells, (int)cfg.k);
  ZK_LAUNCH_CHECK(ctx);
  // ---- commitments
  Workspace *ws;
  CK(get_workspace(ctx, pk, &ws));
  CK(commit_cols(ctx, srs, srs->g_lagrange, pk->fixed_l.fr(), cfg.n_fixed(), (G1Affine *)ws->points.p, pk->fixed_commit));
  CK(commit_cols(ctx, srs, srs->g_lagrange, pk->sigma_l.fr(), cfg.n_perm(), (G1Affine *)ws->points.p, pk->sigma_commit));
  CK(build_resident_tables(ctx, pk, ws));
  CK(zkfhe_sync(ctx));
  tgt.release();
  pk->vk_digest = vk_digest_of(pk);
  *out = pk;
  return ZKFHE_OK;
}


extern "C" {

int zkfhe_bfv_keygen(zkfhe_ctx *ctx, const zkfhe_srs *srs, const char *input_json, const zkfhe_bfv_params *params,
                     const zkfhe_bfv_config *config, zkfhe_bfv_pk **out) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, srs && input_json && params && config && out);
  try {
    return keygen_impl(ctx, srs, input_json, params_from_c(params), config_from_c(config), config->replay != 0, out);
  } catch (const std::exception &e) {
    return zk_fail_msg(ctx, ZKFHE_EINVAL, e.what());
  }
}

int zkfhe_bfv_pk_destroy(zkfhe_ctx *ctx, zkfhe_bfv_pk *pk) {
  ZK_ENTER(ctx);
  if (!pk) return ZKFHE_OK;
  zkfhe_sync(ctx);
  DevBuf *bufs[] = {&pk->fixed_l, &pk->sigma_l, &pk->fixed_ext, &pk->sigma_ext, &pk->l_ext, &pk->xs_ext, &pk->dpow, &pk->ext3_pw,
                    &pk->lookup_src, &pk->inv_slots, &pk->place_start, &pk->place_len};
  for (DevBuf *b : bufs) b->release();
  for (auto &kv : pk->workspaces) free_workspace(kv.second);
  delete pk;
  return ZKFHE_OK;
}

// Frees the prover workspace (0.3 GB at k = 13, several GB at k = 19) this key holds for `ctx`.  Call it before destroying a
// context that proved against a key that lives on; zkfhe_bfv_pk_destroy frees whatever is left.
int zkfhe_bfv_pk_release_ctx(zkfhe_ctx *ctx, zkfhe_bfv_pk *pk) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, pk != nullptr);
  zkfhe_sync(ctx);
  std::lock_guard<std::mutex> lock(pk->mu);
  auto it = pk->workspaces.find(ctx->uid);
  if (it != pk->workspaces.end()) {
    free_workspace(it->second);
    pk->workspaces.erase(it);
  }
  return ZKFHE_OK;
}

// ---- proving key on disk ("ZKFHEPK2"): configuration, break points, the structure lists of the GPU witness generator,
// commitments and the fixed / sigma columns in Lagrange form (device limbs as they are).  The extended-coset tables are
// rebuilt on load.  The key is bound to the SRS it was generated with (its commitments are stored, not recomputed).
namespace {
struct FileW {
  FILE *f;
  bool ok = true;
  void raw(const void *p, size_t n) { ok = ok && fwrite(p, 1, n, f) == n; }
  void u32(uint32_t v) { raw(&v, 4); }
  void u64(uint64_t v) { raw(&v, 8); }
  void vec32(const std::vector<uint32_t> &v) {
    u64(v.size());
    if (!v.empty()) raw(v.data(), v.size() * 4);
  }
};
struct FileR {
  FILE *f;
  bool ok = true;
  void raw(void *p, size_t n) { ok = ok && fread(p, 1, n, f) == n; }
  uint32_t u32() {
    uint32_t v = 0;
    raw(&v, 4);
    return v;
  }
  uint64_t u64() {
    uint64_t v = 0;
    raw(&v, 8);
    return v;
  }
  std::vector<uint32_t> vec32(size_t limit) {
    const uint64_t n = u64();
    std::vector<uint32_t> v;
    if (!ok || n > limit) {
      ok = false;
      return v;
    }
    v.resize(n);
    if (n) raw(v.data(), n * 4);
    return v;
  }
};
int download_u32(zkfhe_ctx *ctx, const DevBuf &b, size_t count, std::vector<uint32_t> &out) {
  out.resize(count);
  if (count) CK(zkfhe_download(ctx, out.data(), b.p, count * 4));
  return ZKFHE_OK;
}
int upload_u32(zkfhe_ctx *ctx, DevBuf &b, const std::vector<uint32_t> &v) {
  CK(b.alloc(ctx, (v.size() + 1) * 4));
  if (!v.empty()) CK(zkfhe_upload(ctx, b.p, v.data(), v.size() * 4));
  return ZKFHE_OK;
}
}  // namespace

int zkfhe_bfv_pk_save(zkfhe_ctx *ctx, const zkfhe_bfv_pk *pk, const char *path) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, pk != nullptr && path != nullptr);
  const CircuitConfig &cfg = pk->cfg;
  const size_t n = cfg.n();
  std::vector<uint32_t> lookup_src, inv_slots, place_start, place_len;
  CK(download_u32(ctx, pk->lookup_src, pk->n_lookup_cells, lookup_src));
  CK(download_u32(ctx, pk->inv_slots, pk->n_inv_slots, inv_slots));
  CK(download_u32(ctx, pk->place_start, cfg.n_gate1, place_start));
  CK(download_u32(ctx, pk->place_len, cfg.n_gate1, place_len));
  std::vector<uint8_t> cols((size_t)(cfg.n_fixed() + cfg.n_perm()) * n * 32);
  CK(zkfhe_download(ctx, cols.data(), pk->fixed_l.p, (size_t)cfg.n_fixed() * n * 32));
  CK(zkfhe_download(ctx, cols.data() + (size_t)cfg.n_fixed() * n * 32, pk->sigma_l.p, (size_t)cfg.n_perm() * n * 32));
  FILE *f = fopen(path, "wb");
  if (!f) return zk_fail_msg(ctx, ZKFHE_EINVAL, std::string("cannot write ") + path);
  FileW w{f};
  w.raw("ZKFHEPK2", 8);
  const uint32_t hdr[8] = {cfg.k, cfg.n_gate0, cfg.n_gate1, cfg.n_lookup, cfg.n_rlc, cfg.unusable_rows, cfg.lookup_bits, cfg.transcript};
  w.raw(hdr, sizeof(hdr));
  w.vec32(cfg.bp_gate0), w.vec32(cfg.bp_gate1), w.vec32(cfg.bp_rlc);
  w.u64(pk->prm.N), w.u64(pk->prm.Q), w.u64(pk->prm.T), w.u64(pk->prm.B);
  w.u64(pk->gate1_cells);
  w.vec32(lookup_src), w.vec32(inv_slots), w.vec32(place_start), w.vec32(place_len);
  w.u64(pk->fixed_commit.size()), w.u64(pk->sigma_commit.size());
  for (const auto &pt : pk->fixed_commit) w.raw(&pt, sizeof(AffinePoint));
  for (const auto &pt : pk->sigma_commit) w.raw(&pt, sizeof(AffinePoint));
  w.raw(pk->vk_digest.l, 32);
  w.u64(cols.size());
  w.raw(cols.data(), cols.size());
  const bool ok = w.ok && fclose(f) == 0;
  if (!ok) return zk_fail_msg(ctx, ZKFHE_EINVAL, std::string("short write to ") + path);
  return ZKFHE_OK;
}

int zkfhe_bfv_pk_load(zkfhe_ctx *ctx, const zkfhe_srs *srs, const char *path, zkfhe_bfv_pk **out) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, srs != nullptr && path != nullptr && out != nullptr);
  FILE *f = fopen(path, "rb");
  if (!f) return zk_fail_msg(ctx, ZKFHE_EINVAL, std::string("cannot read ") + path);
  FileR r{f};
  char magic[8];
  r.raw(magic, 8);
  if (!r.ok || memcmp(magic, "ZKFHEPK2", 8) != 0) {
    fclose(f);
    return zk_fail_msg(ctx, ZKFHE_EINVAL, std::string(path) + " is not a ZKFHEPK2 proving key");
  }
  zkfhe_bfv_pk *pk = new zkfhe_bfv_pk();
  pk->ext_rows = getenv("ZKFHE_CHECK_QUOTIENT") ? 4 : 3;
  auto fail = [&](const std::string &why) {
    fclose(f);
    zkfhe_bfv_pk_destroy(ctx, pk);
    return zk_fail_msg(ctx, ZKFHE_EINVAL, std::string(path) + ": " + why);
  };
  try {
    uint32_t hdr[8];
    r.raw(hdr, sizeof(hdr));
    CircuitConfig &cfg = pk->cfg;
    cfg.k = hdr[0], cfg.n_gate0 = hdr[1], cfg.n_gate1 = hdr[2], cfg.n_lookup = hdr[3], cfg.n_rlc = hdr[4], cfg.unusable_rows = hdr[5], cfg.lookup_bits = hdr[6];
    cfg.transcript = hdr[7];
    if (!r.ok || cfg.transcript > TR_BLAKE2B || cfg.k < 3 || cfg.k > 20 || cfg.n_gate0 + cfg.n_gate1 > 4096 || cfg.n_lookup > 4096 || cfg.n_rlc > 4096) return fail("bad header");
    // the same bounds as the verifier's parse_vk, BEFORE anything derives a row count from them: u() and max_rows() subtract
    // unusable_rows from 2^k, and the lookup table has 2^lookup_bits rows
    if (cfg.n_gate() == 0 || cfg.unusable_rows < 4 || cfg.unusable_rows >= cfg.n()) return fail("bad header (unusable_rows)");
    if (cfg.lookup_bits == 0 || cfg.lookup_bits > 20 || ((size_t)1 << cfg.lookup_bits) > cfg.max_rows()) return fail("bad header (lookup_bits)");
    if (cfg.k != srs->k) return fail("proving key and SRS have different k");
    cfg.bp_gate0 = r.vec32(1 << 16), cfg.bp_gate1 = r.vec32(1 << 16), cfg.bp_rlc = r.vec32(1 << 16);
    pk->prm.N = r.u64(), pk->prm.Q = r.u64(), pk->prm.T = r.u64(), pk->prm.B = r.u64();
    pk->gate1_cells = r.u64();
    // the parameters against the layout they were keyed for: N a power of two whose 5 N + 1 public inputs fit the usable
    // rows, Q within the range the witness kernels divide by, T and B below Q
    if (!r.ok || pk->prm.N < 2 || (pk->prm.N & (pk->prm.N - 1)) || 5 * (uint64_t)pk->prm.N + 1 > cfg.u() || pk->prm.Q < 2 ||
        pk->prm.Q >= ((uint64_t)1 << 63) || pk->prm.T < 1 || pk->prm.T >= pk->prm.Q || pk->prm.B >= pk->prm.Q)
      return fail("BFV parameters do not fit the recorded layout");
    const size_t n = cfg.n(), max_cells = (size_t)cfg.n_gate1 * n;
    const std::vector<uint32_t> lookup_src = r.vec32(max_cells), inv_slots = r.vec32(max_cells), place_start = r.vec32(cfg.n_gate1), place_len = r.vec32(cfg.n_gate1);
    if (!r.ok || pk->gate1_cells > max_cells || place_start.size() != cfg.n_gate1 || place_len.size() != cfg.n_gate1) return fail("bad structure lists");
    for (uint32_t o : lookup_src)
      if (o >= pk->gate1_cells) return fail("lookup offset outside the gate stream");
    for (uint32_t o : inv_slots)
      if (o >= pk->gate1_cells) return fail("inverse slot outside the gate stream");
    for (unsigned c = 0; c < cfg.n_gate1; ++c)
      if ((size_t)place_start[c] + place_len[c] > pk->gate1_cells || place_len[c] > n) return fail("column range outside the gate stream");
    pk->n_lookup_cells = lookup_src.size();
    pk->n_inv_slots = inv_slots.size();
    const uint64_t nf = r.u64(), ns = r.u64();
    if (!r.ok || nf != cfg.n_fixed() || ns != cfg.n_perm()) return fail("commitment counts do not match the configuration");
    pk->fixed_commit.resize(nf), pk->sigma_commit.resize(ns);
    for (auto &pt : pk->fixed_commit) r.raw(&pt, sizeof(AffinePoint));
    for (auto &pt : pk->sigma_commit) r.raw(&pt, sizeof(AffinePoint));
    r.raw(pk->vk_digest.l, 32);
    const uint64_t col_bytes = r.u64();
    if (!r.ok || col_bytes != (uint64_t)(cfg.n_fixed() + cfg.n_perm()) * n * 32) return fail("column block has the wrong size");
    if (!(vk_digest_of(pk) == pk->vk_digest)) return fail("verifying-key digest does not match the stored commitments");
    std::vector<uint8_t> cols(col_bytes);
    r.raw(cols.data(), cols.size());
    if (!r.ok) return fail("truncated file");
    fclose(f);
    f = nullptr;
    int rc;
    if ((rc = upload_u32(ctx, pk->lookup_src, lookup_src)) || (rc = upload_u32(ctx, pk->inv_slots, inv_slots)) ||
        (rc = upload_u32(ctx, pk->place_start, place_start)) || (rc = upload_u32(ctx, pk->place_len, place_len)) ||
        (rc = pk->fixed_l.alloc(ctx, (size_t)cfg.n_fixed() * n * 32)) || (rc = pk->sigma_l.alloc(ctx, (size_t)cfg.n_perm() * n * 32)) ||
        (rc = pk->dpow.alloc(ctx, (size_t)cfg.n_perm() * 32)) ||
        (rc = zkfhe_upload(ctx, pk->fixed_l.p, cols.data(), (size_t)cfg.n_fixed() * n * 32)) ||
        (rc = zkfhe_upload(ctx, pk->sigma_l.p, cols.data() + (size_t)cfg.n_fixed() * n * 32, (size_t)cfg.n_perm() * n * 32))) {
      zkfhe_bfv_pk_destroy(ctx, pk);
      return rc;
    }
    U256 dcan;
    memcpy(dcan.l, DELTA_CANON, 32);
    zkp::k_powers<<<1, 256, 0, ctx->stream>>>(Fr::one(), mont(dcan), pk->dpow.fr(), cfg.n_perm());
    Workspace *ws;
    if ((rc = get_workspace(ctx, pk, &ws)) || (rc = build_resident_tables(ctx, pk, ws)) || (rc = zkfhe_sync(ctx))) {
      zkfhe_bfv_pk_destroy(ctx, pk);
      return rc;
    }
    // A key is bound to the SRS it was generated with: recommit the lookup-table column (the last fixed one, never zero) and
    // the first permutation column against THIS SRS and compare with the stored commitments -- a key loaded against another
    // seed / ceremony / world size would otherwise produce proofs that silently fail verification.
    std::vector<AffinePoint> chk_f, chk_s;
    if ((rc = commit_cols(ctx, srs, srs->g_lagrange, pk->fixed_l.fr() + (size_t)cfg.fix_table() * n, 1, (G1Affine *)ws->points.p, chk_f)) ||
        (rc = commit_cols(ctx, srs, srs->g_lagrange, pk->sigma_l.fr(), 1, (G1Affine *)ws->points.p, chk_s))) {
      zkfhe_bfv_pk_destroy(ctx, pk);
      return rc;
    }
    if (!(chk_f[0].x == pk->fixed_commit[cfg.fix_table()].x) || !(chk_f[0].y == pk->fixed_commit[cfg.fix_table()].y) ||
        !(chk_s[0].x == pk->sigma_commit[0].x) || !(chk_s[0].y == pk->sigma_commit[0].y)) {
      zkfhe_bfv_pk_destroy(ctx, pk);
      return zk_fail_msg(ctx, ZKFHE_EINVAL, std::string(path) + ": proving key was generated with a different SRS");
    }
  } catch (const std::exception &e) {
    if (f) fclose(f);
    zkfhe_bfv_pk_destroy(ctx, pk);
    return zk_fail_msg(ctx, ZKFHE_EINVAL, e.what());
  }
  *out = pk;
  return ZKFHE_OK;
}

int zkfhe_bfv_pk_info(const zkfhe_bfv_pk *pk, uint8_t vk_digest[32], uint32_t *n_fixed, uint32_t *n_sigma) {
  if (!pk) return ZKFHE_EINVAL;
  if (vk_digest) memcpy(vk_digest, pk->vk_digest.l, 32);
  if (n_fixed) *n_fixed = (uint32_t)pk->fixed_commit.size();
  if (n_sigma) *n_sigma = (uint32_t)pk->sigma_commit.size();
  return ZKFHE_OK;
}

// The per-public-key transcript cache of this proving key (prefix_cache.hpp): capacity >= 0 sets the number of public keys
// remembered (0 = off, entries beyond it are dropped, at most 64), negative leaves it; hits / misses / entries (each optional)
// receive the counters since the key was made.
int zkfhe_bfv_pk_prefix_cache(const zkfhe_bfv_pk *pk_c, int capacity, uint64_t *hits, uint64_t *misses, uint64_t *entries) {
  if (!pk_c) return ZKFHE_EINVAL;
  zkfhe_bfv_pk *pk = const_cast<zkfhe_bfv_pk *>(pk_c);
  if (capacity >= 0) pk->prefix.set_capacity((size_t)capacity);
  pk->prefix.stats(hits, misses, entries);
  return ZKFHE_OK;
}

int zkfhe_bfv_pk_commitments(const zkfhe_bfv_pk *pk, uint8_t *fixed_out, uint8_t *sigma_out) {
  if (!pk) return ZKFHE_EINVAL;
  for (size_t i = 0; i < pk->fixed_commit.size() && fixed_out; ++i) {
    memcpy(fixed_out + 64 * i, pk->fixed_commit[i].x.l, 32);
    memcpy(fixed_out + 64 * i + 32, pk->fixed_commit[i].y.l, 32);
  }
  for (size_t i = 0; i < pk->sigma_commit.size() && sigma_out; ++i) {
    memcpy(sigma_out + 64 * i, pk->sigma_commit[i].x.l, 32);
    memcpy(sigma_out + 64 * i + 32, pk->sigma_commit[i].y.l, 32);
  }
  return ZKFHE_OK;
}

int zkfhe_bfv_pk_export_vk(const zkfhe_bfv_pk *pk, uint8_t *out, size_t cap, size_t *len) {
  if (!pk || !len) return ZKFHE_EINVAL;
  const size_t nf = pk->fixed_commit.size(), ns = pk->sigma_commit.size();
  const size_t need = 8 + 40 + 32 + 64 * (nf + ns);
  *len = need;
  if (!out || cap < need) return out ? ZKFHE_EINVAL : ZKFHE_OK;
  memcpy(out, "ZKFHEVK2", 8);
  const uint32_t hdr[10] = {pk->cfg.k, pk->cfg.n_gate0, pk->cfg.n_gate1, pk->cfg.n_lookup, pk->cfg.n_rlc, pk->cfg.unusable_rows,
                            pk->cfg.lookup_bits, pk->cfg.transcript, (uint32_t)nf, (uint32_t)ns};
  memcpy(out + 8, hdr, 40);
  memcpy(out + 48, pk->vk_digest.l, 32);
  uint8_t *p = out + 80;
  for (const auto &c : pk->fixed_commit) {
    memcpy(p, c.x.l, 32);
    memcpy(p + 32, c.y.l, 32);
    p += 64;
  }
  for (const auto &c : pk->sigma_commit) {
    memcpy(p, c.x.l, 32);
    memcpy(p + 32, c.y.l, 32);
    p += 64;
  }
  return ZKFHE_OK;
}

int zkfhe_bfv_pk_break_points(const zkfhe_bfv_pk *pk, int which, uint32_t *out, uint32_t *count) {
  if (!pk || !count) return ZKFHE_EINVAL;
  const auto &v = which == 0 ? pk->cfg.bp_gate0 : which == 1 ? pk->cfg.bp_gate1 : pk->cfg.bp_rlc;
  if (out && *count >= v.size()) memcpy(out, v.data(), v.size() * 4);
  *count = (uint32_t)v.size();
  return ZKFHE_OK;
}

}  // extern "C"
