// create_proof for the BFV circuit on one MI355X (include/zkfhe.h "prove"): replaces halo2 `create_proof` with
// ProverSHPLONK as driven by snark-verifier-sdk `gen_snark_shplonk` (third-party, reached from reference
// examples/bfv.rs:311; protocol restated in oracle/halo2_ref.py, which this file must match byte for byte for the same
// seed).  Host side: phase-0 witness (bfv_circuit.hpp), Fiat-Shamir transcript.  Device side: everything that touches a
// full column -- the phase-1 witness, blinding, MSM commits, (i)NTT / coset NTT, grand products, quotient evaluation,
// evaluations at x, SHPLONK polynomials.
#include "prover_internal.hpp"

// buffers of the GPU witness generator (sizes known only after keygen)
int alloc_witness_buffers(zkfhe_ctx *ctx, const zkfhe_bfv_pk *pk, Workspace *ws) {
  if (ws->stream.p) return ZKFHE_OK;
  const size_t N = pk->prm.N;
  CK(ws->stream.alloc(ctx, (pk->gate1_cells + 8) * 32));
  CK(ws->pool.alloc(ctx, 40 * (2 * N + 8) * 32));
  CK(ws->invtmp.alloc(ctx, (pk->n_inv_slots + 8) * 32));
  ZK_HIP(ctx, hipHostMalloc((void **)&ws->host_pool, 24 * (2 * N + 8) * 32, hipHostMallocDefault));
  ws->host_poly_len = 2 * N + 8;
  ZK_HIP(ctx, hipHostMalloc((void **)&ws->host_poly, ws->host_poly_len * 32, hipHostMallocDefault));
  CK(ws->wblind.alloc(ctx, 256));   // the lookup-permutation error flag
  ZK_HIP(ctx, hipHostMalloc((void **)&ws->ring, Workspace::RING_BYTES, hipHostMallocDefault));
  CK(ws->dev_ring.alloc(ctx, Workspace::RING_BYTES));
  {
    const CircuitConfig &c = pk->cfg;
    const size_t max_items = (size_t)c.n_advice() + c.n_fixed() + 2 + c.n_perm() + c.n_chunks() + 3 * c.n_lookup;
    ws->out_pts_cap = std::max<size_t>(ws->n_all, c.n_perm()) + 16;
    ws->out_ev_off = ws->out_pts_cap * PT_SLOT;
    ws->out_ev_cap = max_items * 4;
    ws->out_flag_off = ws->out_ev_off + ws->out_ev_cap * 32;
    ZK_HIP(ctx, hipHostMalloc((void **)&ws->host_out, ws->out_flag_off + 256, hipHostMallocDefault));
    memset(ws->host_out, 0, ws->out_flag_off + 256);
  }
  ZK_HIP(ctx, hipHostMalloc((void **)&ws->host_pts, ((size_t)pk->cfg.n_gate0 + 1) * PT_SLOT, hipHostMallocDefault));
  ZK_HIP(ctx, hipEventCreateWithFlags(&ws->ev_pts, hipEventDisableTiming | hipEventBlockingSync));
  ZK_HIP(ctx, hipEventCreateWithFlags(&ws->ev_rand, hipEventDisableTiming | hipEventBlockingSync));
  ZK_HIP(ctx, hipEventCreateWithFlags(&ws->ev_early, hipEventDisableTiming | hipEventBlockingSync));
  ZK_HIP(ctx, hipHostMalloc((void **)&ws->host_early, ((size_t)pk->cfg.n_advice() + 2 * pk->cfg.n_lookup + 16) * PT_SLOT, hipHostMallocDefault));
  ZK_HIP(ctx, hipHostMalloc((void **)&ws->host_early_err, 64, hipHostMallocDefault));
  ZK_HIP(ctx, hipHostMalloc((void **)&ws->host_rand_pt, PT_SLOT, hipHostMallocDefault));
  if (zkfhe_ctx_create(ctx->device, nullptr, &ws->aux)) return zk_fail_msg(ctx, ZKFHE_EHIP, std::string("auxiliary context: ") + zkfhe_last_error(nullptr));
  return ZKFHE_OK;
}

namespace {

// counting-sort restatement of halo2's permute_expression_pair for 8-bit tables
void permute_lookup(const std::vector<U256> &input, size_t u, unsigned table_size, std::vector<U256> &a_perm, std::vector<U256> &s_perm) {
  std::vector<size_t> cnt(table_size, 0);
  for (size_t i = 0; i < u; ++i) {
    const U256 &v = input[i];
    if (v.l[1] | v.l[2] | v.l[3] || v.l[0] >= table_size) throw std::runtime_error("lookup input not in table");
    ++cnt[v.l[0]];
  }
  // table multiset over usable rows: values 0..table_size-1 once each, and 0 for the remaining rows
  std::vector<size_t> left(table_size, 1);
  left[0] = u - (table_size - 1);
  a_perm.assign(u, fe::zero());
  s_perm.assign(u, fe::zero());
  std::vector<size_t> holes;
  size_t pos = 0;
  for (unsigned v = 0; v < table_size; ++v)
    for (size_t k = 0; k < cnt[v]; ++k, ++pos) {
      a_perm[pos] = fe::from_u64(v);
      if (k == 0) {
        if (!left[v]) throw std::runtime_error("lookup input not in table");
        --left[v];
        s_perm[pos] = fe::from_u64(v);
      } else {
        holes.push_back(pos);
      }
    }
  size_t hi = 0;
  for (unsigned v = 0; v < table_size; ++v)
    for (size_t k = 0; k < left[v]; ++k) s_perm[holes[hi++]] = fe::from_u64(v);
}

// ---------------------------------------------------------------------------------------------------------------
// Phase-1 gate context on the GPU: the device twin of bfv_phase1 (examples/bfv.rs:167-304 through src/poly_chip.rs).
// One launch per PolyChip call, one thread per coefficient; the stream offset of every call is a pure function of
// the circuit shape and is cross-checked against the stream length recorded at keygen.
struct DevPoly {
  Fr *d = nullptr;
  size_t len = 0;
  uint64_t bits = 0;
};

class GpuPhase1 {
 public:
  GpuPhase1(zkfhe_ctx *c, const zkfhe_bfv_pk *k, Workspace *w) : ctx(c), pk(k), ws(w) {}

  // Everything that does not depend on the phase-1 challenge: the gadget launches (the four constrain_mul gates only
  // reserve their cells) and the deferred inversions.  Enqueued behind the phase-0 commitment, so the GPU works on it
  // while the host hashes that commitment and evaluates the RLC context.
  int launch(const BfvState &st) {
    const uint64_t Q = pk->prm.Q, T = pk->prm.T, B = pk->prm.B;
    CK(alloc_witness_buffers(ctx, pk, ws));
    stream = ws->stream.fr();
    pool_used = 0;
    off = 0;
    calls.clear();
    call_level.clear();
    produced.clear();
    // inputs: one pinned staging block.  While they fit (k = 13: 0.66 MB) the block is a piece of the proof's staging ring and
    // goes up with the gadget arguments in ONE copy; longer inputs keep their own block and their own DMA.
    h_used = 0;
    {
      const PolyChip *ins[14] = {&st.e0, &st.e1, &st.u, &st.m, &st.pk0_u, &st.quotient_0, &st.remainder_0, &st.quotient_0_times_cyclo, &st.expected_c0,
                                 &st.pk1_u, &st.quotient_1, &st.remainder_1, &st.quotient_1_times_cyclo, &st.expected_c1};
      size_t cells = 1 + 16;   // delta, the reserved gate cells
      for (const PolyChip *pc : ins) cells += pc->assigned_coefficients.size();
      in_host = ws->host_pool;
      in_dev = ws->pool.fr();
      in_ring = false;
      void *h = nullptr;
      if (cells * 32 <= ((size_t)1 << 20)) {
        void *d = stage_reserve(ws, cells * 32, &h);
        if (d) {
          in_host = (U256 *)h;
          in_dev = (Fr *)d;
          in_ring = true;
        }
      }
    }
    DevPoly e0 = stage(st.e0), e1 = stage(st.e1), u = stage(st.u), m = stage(st.m);
    DevPoly pk0_u = stage(st.pk0_u), q0 = stage(st.quotient_0), r0 = stage(st.remainder_0), q0c = stage(st.quotient_0_times_cyclo), xc0 = stage(st.expected_c0);
    DevPoly pk1_u = stage(st.pk1_u), q1 = stage(st.quotient_1), r1 = stage(st.remainder_1), q1c = stage(st.quotient_1_times_cyclo), xc1 = stage(st.expected_c1);
    in_host[h_used] = st.delta.value;
    const Fr *delta = in_dev + h_used;
    const uint64_t delta_bits = st.delta.value.bits();
    ++h_used;
    pool_used = h_used;
    // the four gate cells of each constrain_mul (filled in by finish()), staged after the inputs
    mg_slot = h_used;
    n_mg = 0;
    h_used += 16;
    pool_used = h_used;
    if (!in_ring) ZK_HIP(ctx, hipMemcpyAsync(ws->pool.p, ws->host_pool, mg_slot * 32, hipMemcpyHostToDevice, ctx->stream));

    CK(in_range(e0, B, Q));
    CK(in_range(e1, B, Q));
    CK(gadget(zkw::G_CHI_KEY, u, nullptr, nullptr, zkw::cpc_chi_key(), Q - 1, 0, Fr::zero()));
    CK(in_range(m, T / 2, Q));
    const int n_side = 2;
    for (int side = 0; side < n_side; ++side) {
      const DevPoly &pku = side ? pk1_u : pk0_u, &q = side ? q1 : q0, &r = side ? r1 : r0, &qc = side ? q1c : q0c;
      CK(mul_gate());                                                    // pk_i * u = pk_i_u
      DevPoly pku_r;
      CK(reduce_by_modulo(pku, Q, pku_r));
      CK(in_field(q, Q));
      CK(in_field(r, Q));
      // reduce_by_cyclo (src/poly_chip.rs:183-223)
      CK(mul_gate());                                                    // quotient * cyclo = quotient_times_cyclo
      DevPoly sum, sum_mod;
      CK(add(qc, r, sum));
      CK(reduce_by_modulo(sum, Q, sum_mod));
      if (sum_mod.len < pku_r.len) return zk_fail_msg(ctx, ZKFHE_EINVAL, "degree <= self.degree (src/poly_chip.rs:375)");
      DevPoly trimmed{sum_mod.d + (sum_mod.len - pku_r.len), pku_r.len, sum_mod.bits};
      CK(equal(trimmed, pku_r));
      const size_t N = pk->prm.N;
      if (r.len < N) return zk_fail_msg(ctx, ZKFHE_EINVAL, "degree <= self.degree (src/poly_chip.rs:375)");
      DevPoly acc{r.d + (r.len - N), N, r.bits};
      if (side == 0) {
        DevPoly m_delta, t;
        CK(scalar_mul(m, delta, delta_bits, m_delta));
        CK(add(acc, m_delta, t));
        CK(add(t, e0, acc));
      } else {
        DevPoly t;
        CK(add(acc, e1, t));
        acc = t;
      }
      DevPoly c_red;
      CK(reduce_by_modulo(acc, Q, c_red));
      CK(equal(c_red, side ? xc1 : xc0));
    }
    if (off != pk->gate1_cells) return zk_fail_msg(ctx, ZKFHE_EINVAL, "GPU witness stream length differs from the keygen circuit shape");
    CK(launch_levels());
    // deferred 1/x cells of is_zero: one batch inversion over the structural slot list
    if (pk->n_inv_slots) {
      const unsigned g = (unsigned)((pk->n_inv_slots + 255) / 256);
      zkw::k_gather<<<g, 256, 0, ctx->stream>>>(stream, (const unsigned *)pk->inv_slots.p, pk->n_inv_slots, ws->invtmp.fr());
      ZK_LAUNCH_CHECK(ctx);
      CK(zkfhe_fr_batch_invert(ctx, (zkfhe_fr *)ws->invtmp.p, pk->n_inv_slots));
      zkw::k_scatter<<<g, 256, 0, ctx->stream>>>(stream, (const unsigned *)pk->inv_slots.p, pk->n_inv_slots, ws->invtmp.fr());
      ZK_LAUNCH_CHECK(ctx);
    }
    return ZKFHE_OK;
  }

  // Everything of phase 1 except the cells that depend on the challenge: the reserved constrain_mul cells stay zero in the
  // stream, then stream -> gate columns (break points) and lookup columns.  The columns can be committed from here on;
  // patch() supplies the missing cells and the correction columns.
  int place_early() {
    zkw::ZeroRuns z{};
    z.count = n_mg;
    for (unsigned i = 0; i < n_mg; ++i) z.at[i] = stream + mg_off[i];
    zkw::k_zero_runs<<<1, 64, 0, ctx->stream>>>(z);
    ZK_LAUNCH_CHECK(ctx);
    return place();
  }
  // (gate column, row) of every reserved cell -- twice for a cell on a break point, whose value is repeated at the top of the
  // next column.  columns: the distinct affected gate columns, in ascending order.
  struct PatchPlan {
    std::vector<unsigned> columns;
    struct Cell {
      unsigned col_slot, row, value;
    };
    std::vector<Cell> cells;
  };
  PatchPlan patch_plan() const {
    PatchPlan p;
    const CircuitConfig &cfg = pk->cfg;
    std::vector<size_t> start(cfg.n_gate1, 0), len(cfg.n_gate1, 0);
    size_t s = 0;
    for (unsigned c = 0; c < cfg.n_gate1 && s < pk->gate1_cells; ++c) {
      start[c] = s;
      if (c < cfg.bp_gate1.size()) {
        len[c] = (size_t)cfg.bp_gate1[c] + 1;
        s += cfg.bp_gate1[c];
      } else {
        len[c] = pk->gate1_cells - s;
        s = pk->gate1_cells;
      }
    }
    for (unsigned i = 0; i < n_mg; ++i)
      for (unsigned j = 0; j < 4; ++j) {
        const size_t o = mg_off[i] + j;
        for (unsigned c = 0; c < cfg.n_gate1; ++c)
          if (len[c] && o >= start[c] && o < start[c] + len[c]) {
            size_t slot = std::find(p.columns.begin(), p.columns.end(), c) - p.columns.begin();
            if (slot == p.columns.size()) p.columns.push_back(c);
            p.cells.push_back({(unsigned)slot, (unsigned)(o - start[c]), 4 * i + j});
          }
      }
    return p;
  }
  // evals as in finish(): writes the reserved cells into the advice columns and into patch_cols[slot][row] (zeroed here);
  // terms (optional): the same cells as sparse MSM terms (row, slot, scalar)
  int patch(const U256 evals[12], const PatchPlan &plan, Fr *patch_cols, std::vector<zkfhe_sparse_term> *terms = nullptr) {
    const CircuitConfig &cfg = pk->cfg;
    const size_t n = cfg.n();
    Fr mg[16];
    for (int i = 0; i < 4; ++i) {
      mg[4 * i] = Fr::zero();
      for (int j = 0; j < 3; ++j) mg[4 * i + 1 + j] = mont(evals[3 * i + j]);
    }
    if (terms) {
      terms->clear();
      for (const auto &c : plan.cells) {
        if ((c.value & 3) == 0) continue;   // the first cell of a gate is the constant 0
        zkfhe_sparse_term t;
        memcpy(&t.scalar, &mg[c.value], 32);
        t.row = c.row;
        t.slot = c.col_slot;
        terms->push_back(t);
      }
    }
    std::vector<zkw::PatchCell> cells(plan.cells.size());
    for (size_t t = 0; t < cells.size(); ++t) {
      const auto &c = plan.cells[t];
      cells[t].dst_adv = ws->adv_l.fr() + ((size_t)cfg.n_gate0 + plan.columns[c.col_slot]) * n + c.row;
      cells[t].dst_patch = terms ? nullptr : patch_cols + (size_t)c.col_slot * n + c.row;   // dense correction columns only without the sparse MSM
      cells[t].value = c.value;
    }
    if (cells.size() > 64) return zk_fail_msg(ctx, ZKFHE_EINVAL, "too many challenge-dependent gate cells");
    STAGE(vals_dev, Fr, ws, mg, sizeof(mg));
    STAGE(cells_dev, zkw::PatchCell, ws, cells.data(), cells.size() * sizeof(zkw::PatchCell));
    CK(flush_staged(ctx, ws));
    if (!terms) ZK_HIP(ctx, hipMemsetAsync(patch_cols, 0, plan.columns.size() * n * 32, ctx->stream));
    zkw::k_patch_cells<<<1, 64, 0, ctx->stream>>>(cells_dev, (unsigned)cells.size(), vals_dev);
    ZK_LAUNCH_CHECK(ctx);
    return ZKFHE_OK;
  }

  // evals[3 i .. 3 i + 2] = a(gamma), b(gamma), c(gamma) of the i-th constrain_mul: fill the reserved gate cells, then
  // stream -> gate columns (break points) and lookup columns
  int finish(const U256 evals[12]) {
    // the 16 values and their destinations go up with the staging ring, one small kernel stores them (it used to be one
    // upload and four device copies)
    Fr mg[16];
    for (int i = 0; i < 4; ++i) {
      mg[4 * i] = Fr::zero();
      for (int j = 0; j < 3; ++j) mg[4 * i + 1 + j] = mont(evals[3 * i + j]);
    }
    zkw::PatchCell cells[16];
    unsigned nc = 0;
    for (unsigned i = 0; i < n_mg; ++i)
      for (unsigned j = 0; j < 4; ++j) {
        cells[nc].dst_adv = stream + mg_off[i] + j;
        cells[nc].dst_patch = nullptr;
        cells[nc].value = 4 * i + j;
        ++nc;
      }
    if (nc) {
      STAGE(vals_dev, Fr, ws, mg, sizeof(mg));
      STAGE(cells_dev, zkw::PatchCell, ws, cells, nc * sizeof(zkw::PatchCell));
      CK(flush_staged(ctx, ws));
      zkw::k_patch_cells<<<1, 64, 0, ctx->stream>>>(cells_dev, nc, vals_dev);
      ZK_LAUNCH_CHECK(ctx);
    }
    return place();
  }

 private:
  int place() {
    const CircuitConfig &cfg = pk->cfg;
    const size_t n = cfg.n();
    zkw::k_place<<<grid_for(ctx, (size_t)cfg.n_gate1 * n), 256, 0, ctx->stream>>>(stream, (const unsigned *)pk->place_start.p, (const unsigned *)pk->place_len.p,
                                                                                 cfg.n_gate1, n, ws->adv_l.fr() + (size_t)cfg.n_gate0 * n);
    ZK_LAUNCH_CHECK(ctx);
    if (cfg.n_lookup) {
      if (pk->n_lookup_cells > (size_t)cfg.n_lookup * cfg.max_rows()) return zk_fail_msg(ctx, ZKFHE_EINVAL, "lookup cells do not fit the configured lookup columns");
      zkw::k_place_lookups<<<grid_for(ctx, (size_t)cfg.n_lookup * n), 256, 0, ctx->stream>>>(stream, (const unsigned *)pk->lookup_src.p, pk->n_lookup_cells,
                                                                                           (unsigned)cfg.max_rows(), n, cfg.n_lookup,
                                                                                           ws->adv_l.fr() + (size_t)cfg.adv_lookup0() * n);
      ZK_LAUNCH_CHECK(ctx);
    }
    return ZKFHE_OK;
  }

  zkfhe_ctx *ctx;
  const zkfhe_bfv_pk *pk;
  Workspace *ws;
  Fr *stream = nullptr;
  size_t off = 0, pool_used = 0, h_used = 0, mg_slot = 0, mg_off[4] = {0, 0, 0, 0};
  unsigned n_mg = 0;
  U256 *in_host = nullptr;   // where the input polynomials are staged on the host ...
  Fr *in_dev = nullptr;      // ... and where the gadgets read them
  bool in_ring = false;

  DevPoly stage(const PolyChip &p) {
    DevPoly d;
    d.len = p.assigned_coefficients.size();
    d.bits = p.max_num_bits;
    d.d = in_dev + h_used;
    for (size_t i = 0; i < d.len; ++i) in_host[h_used + i] = p.assigned_coefficients[i].value;
    h_used += d.len;
    return d;
  }
  Fr *take(size_t len) {
    Fr *p = ws->pool.fr() + pool_used;
    pool_used += len;
    return p;
  }
  int gadget(int type, const DevPoly &a, const Fr *b, Fr *out, size_t cpc, uint64_t p0, uint64_t p1, const Fr &bound) {
    if (pool_used * 32 > ws->pool.bytes || (off + a.len * cpc) > pk->gate1_cells)
      return zk_fail_msg(ctx, ZKFHE_EINVAL, "GPU witness stream exceeds the keygen circuit shape");
    zkw::GadgetArgs g;
    g.type = type;
    g.a = a.d;
    g.b = b;
    g.out = out;
    g.stream = stream;
    g.base = off;
    g.cpc = cpc;
    g.count = a.len;
    g.p0 = p0;
    g.p1 = p1;
    g.bound = bound;
    // recorded in program order (the stream offsets follow examples/bfv.rs:171-301), launched level by level: a call's level
    // is one more than the level of the call that produced its inputs
    calls.push_back(g);
    call_level.push_back(level_of(a.d, a.len, b));
    if (out) produced.push_back({out, out + a.len, call_level.back()});
    off += a.len * cpc;
    return ZKFHE_OK;
  }
  struct Produced {
    const Fr *lo, *hi;
    int level;
  };
  std::vector<zkw::GadgetArgs> calls;
  std::vector<int> call_level;
  std::vector<Produced> produced;
  int level_of(const Fr *a, size_t len, const Fr *b) const {
    int lv = 0;
    for (const Produced &p : produced) {
      if (a < p.hi && a + len > p.lo) lv = std::max(lv, p.level + 1);
      if (b && b < p.hi && b + 1 > p.lo) lv = std::max(lv, p.level + 1);   // b: an array aligned with a, or one scalar
      if (b && b + len > p.lo && b < p.hi) lv = std::max(lv, p.level + 1);
    }
    return lv;
  }
  int launch_levels() {
    int max_level = 0;
    for (int l : call_level) max_level = std::max(max_level, l);
    std::vector<zkw::GadgetArgs> ordered;
    std::vector<size_t> first(max_level + 2, 0);
    for (int l = 0; l <= max_level; ++l) {
      first[l] = ordered.size();
      for (size_t i = 0; i < calls.size(); ++i)
        if (call_level[i] == l) ordered.push_back(calls[i]);
    }
    first[max_level + 1] = ordered.size();
    STAGE(dev, zkw::GadgetArgs, ws, ordered.data(), ordered.size() * sizeof(zkw::GadgetArgs));
    CK(flush_staged(ctx, ws));
    for (int l = 0; l <= max_level; ++l) {
      size_t longest = 0;
      for (size_t i = first[l]; i < first[l + 1]; ++i) longest = std::max(longest, ordered[i].count);
      if (first[l + 1] == first[l]) continue;
      zkw::k_gadget<<<dim3((unsigned)((longest + 255) / 256), (unsigned)(first[l + 1] - first[l])), 256, 0, ctx->stream>>>(dev + first[l]);
      ZK_LAUNCH_CHECK(ctx);
    }
    return ZKFHE_OK;
  }
  int in_range(const DevPoly &a, uint64_t z, uint64_t y) {
    if (!(z < y)) return zk_fail_msg(ctx, ZKFHE_EINVAL, "z < y (src/poly_chip.rs:278)");
    return gadget(zkw::G_IN_RANGE, a, nullptr, nullptr, zkw::cpc_in_range(z, y), z, y, Fr::zero());
  }
  int in_field(const DevPoly &a, uint64_t q) { return gadget(zkw::G_IN_FIELD, a, nullptr, nullptr, zkw::cpc_in_field(q), q, 0, Fr::zero()); }
  int mul_gate() {
    if (off + 4 > pk->gate1_cells || n_mg >= 4) return zk_fail_msg(ctx, ZKFHE_EINVAL, "GPU witness stream exceeds the keygen circuit shape");
    mg_off[n_mg++] = off;
    off += 4;
    return ZKFHE_OK;
  }
  int reduce_by_modulo(const DevPoly &a, uint64_t q, DevPoly &out) {
    if (a.bits >= 192 || (q >> 63)) return zk_fail_msg(ctx, ZKFHE_EINVAL, "reduce_by_modulo: operands too wide for the device divider (a < 2^192, q < 2^63)");
    // halo2-base div_mod: div < 2^num_bits / q + 1
    Fr bound;
    zkw::u64 rem;
    zkw::c_divmod(zkw::c_pow2((unsigned)a.bits), q, bound, rem);
    bound = zkw::c_add(bound, zkw::c_u64(1));
    out.d = take(a.len);
    out.len = a.len;
    out.bits = bits_u64(q);
    return gadget(zkw::G_DIV_MOD, a, nullptr, out.d, zkw::cpc_div_mod(q, bound), q, 0, bound);
  }
  int add(const DevPoly &a, const DevPoly &b, DevPoly &out) {
    if (b.len < a.len) return zk_fail_msg(ctx, ZKFHE_EINVAL, "add: operand lengths differ");
    out.d = take(a.len);
    out.len = a.len;
    out.bits = std::max(a.bits, b.bits) + 1;
    return gadget(zkw::G_ADD, a, b.d, out.d, 4, 0, 0, Fr::zero());
  }
  int scalar_mul(const DevPoly &a, const Fr *scalar, uint64_t scalar_bits, DevPoly &out) {
    out.d = take(a.len);
    out.len = a.len;
    out.bits = a.bits + scalar_bits;
    return gadget(zkw::G_SCALAR_MUL, a, scalar, out.d, 4, 0, 0, Fr::zero());
  }
  int equal(const DevPoly &a, const DevPoly &b) {
    if (b.len < a.len) return zk_fail_msg(ctx, ZKFHE_EINVAL, "constrain_equality: operand lengths differ");
    return gadget(zkw::G_EQUAL, a, b.d, nullptr, zkw::cpc_equal(), 0, 0, Fr::zero());
  }
};

bool witness_on_host() {
  const char *e = getenv("ZKFHE_WITNESS");
  return e && strcmp(e, "host") == 0;
}

struct OpenItem {
  const Fr *lagr;        // device pointer, Lagrange form
  int n_rot;
  int rot[4];            // rotation ids: 0,1,2,3, 4 = last (w^u), 5 = -1
  U256 ev[4];            // canonical evaluations
};

// Admission gate of the GPU-heavy middle of a proof (grand products, their commitment, coset extension, quotient).  Proofs
// that start together move through the Fiat-Shamir rounds together: they all hash on the host at the same moments and all
// reach the light, serial end of the proof (evaluations, 758 Poseidon permutations, two one-column commitments) together, with
// the chip nearly idle.  Letting only a few proofs into the heavy part at a time -- first come, first served -- costs no
// throughput (that part is bound by the chip, a handful of proofs fill it) and spreads the proofs out: the serial end of one
// overlaps the heavy part of the next.  ZKFHE_GATE = proofs admitted at once (0 = no gate).
class HeavyGate {
 public:
  static HeavyGate &get() {
    static HeavyGate g;
    return g;
  }
  // returns whether a slot was taken (the gate may be switched while proofs are in flight: leave() only gives back what enter() took)
  bool enter() {
    std::unique_lock<std::mutex> l(mu);
    if (slots <= 0) return false;
    const uint64_t my = next_ticket++;
    cv.wait(l, [&] { return slots <= 0 || my < serving + (uint64_t)slots; });
    return true;
  }
  void leave() {
    {
      std::lock_guard<std::mutex> l(mu);
      ++serving;
    }
    cv.notify_all();
  }
  int set(int n) {
    int old;
    {
      std::lock_guard<std::mutex> l(mu);
      old = slots;
      if (n >= 0) slots = n;
    }
    cv.notify_all();
    return old;
  }

 private:
  HeavyGate() {
    if (const char *e = getenv("ZKFHE_GATE")) slots = atoi(e);
  }
  std::mutex mu;
  std::condition_variable cv;
  int slots = 0;
  uint64_t next_ticket = 0, serving = 0;   // tickets below `serving` have left; FIFO admission of serving .. serving + slots - 1
};
struct GateHold {
  bool held = false;
  void enter() { held = HeavyGate::get().enter(); }
  void leave() {
    if (held) HeavyGate::get().leave();
    held = false;
  }
  ~GateHold() { leave(); }
};

// ZKFHE_TRACE=1: host-side phase times of one proof on stderr
struct Trace {
  bool on;
  double t0, last, cpu_last;
  unsigned long tid;   // proofs in flight on other threads: their lines interleave
  static double thread_cpu_ms() {   // CPU time of the calling thread: what of a phase was work on this core and what was waiting
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
  }
  Trace() : on(getenv("ZKFHE_TRACE") != nullptr), t0(now_ms()), last(t0), cpu_last(on ? thread_cpu_ms() : 0), tid((unsigned long)std::hash<std::thread::id>()(std::this_thread::get_id()) % 10000) {}
  void mark(const char *what) {
    if (!on) return;
    const double t = now_ms(), c = thread_cpu_ms();
    fprintf(stderr, "[zkfhe trace %04lu] %8.3f ms (+%7.3f, cpu %6.3f) %s @%.3f\n", tid, t - t0, t - last, c - cpu_last, what, t);
    last = t;
    cpu_last = c;
  }
};

int prove_impl(zkfhe_ctx *ctx, const zkfhe_srs *srs, zkfhe_bfv_pk *pk, const char *input_json, const uint8_t seed[32],
               std::vector<uint8_t> &proof, std::vector<U256> &instances, float *timings) {
  const CircuitConfig &cfg = pk->cfg;
  const size_t n = cfg.n(), u = cfg.u(), ne = 4 * n;
  const unsigned k = cfg.k;
  const double t_start = now_ms();
  Trace trace;
  // The blinding stream (draw i = Blake2b(seed || i) mod r) is evaluated on the device, where each value is needed
  // (prover_kernels.hip.hpp k_rng_fill).  Draw order = oracle/halo2_ref.py: the blinding rows of every advice column in
  // column order, then per lookup its permuted input and permuted table, then the permutation and lookup products, then
  // the n coefficients of the random polynomial.
  zkp::RngSeed rseed;
  memcpy(rseed.w, seed, 32);
  const size_t nbl_all = n - u;
  const uint64_t ctr_adv = 0, ctr_lk = (uint64_t)cfg.n_advice() * nbl_all, ctr_pz = ctr_lk + 2 * (uint64_t)cfg.n_lookup * nbl_all,
                 ctr_lz = ctr_pz + (uint64_t)cfg.n_chunks() * (nbl_all - 1), ctr_rand = ctr_lz + (uint64_t)cfg.n_lookup * (nbl_all - 1);
  auto rng_fill = [&](hipStream_t stream, uint64_t ctr0, uint64_t ctr_col_stride, Fr *dst, size_t per_col, size_t col_stride, size_t n_cols) -> int {
    if (!per_col || !n_cols) return ZKFHE_OK;
    zkp::k_rng_fill<<<grid_for(ctx, per_col * n_cols), 256, 0, stream>>>(rseed, ctr0, ctr_col_stride, dst, per_col, col_stride, n_cols);
    ZK_LAUNCH_CHECK(ctx);
    return ZKFHE_OK;
  };
  Transcript tr(cfg.transcript);
  const NttDomain *dom;
  CK(zk_domain(ctx, (int)k, &dom));
  Workspace *ws;
  CK(get_workspace(ctx, pk, &ws));
  GpuPolyMul gpu_mul(ctx, ws);
  struct BackendGuard {
    explicit BackendGuard(PolyMulBackend *b) { poly_mul_backend() = b; }
    ~BackendGuard() { poly_mul_backend() = nullptr; }
  } guard(&gpu_mul);
  // ------------------------------------------------------------ phase 0 witness
  const zkfhe_basis *small_basis = srs->g_lagrange_small ? srs->g_lagrange_small : srs->g_lagrange;
  // a previous proof on this context may have been abandoned on an error with copies from the pinned staging buffers still
  // queued: drain the stream before the buffers are rewritten (free when the stream is idle)
  CK(zkfhe_sync(ctx));
  CK(alloc_witness_buffers(ctx, pk, ws));
  ws->ring_off = ws->ring_flushed = 0;
  trace.mark("setup (workspace)");
  // random polynomial, early: its coefficients (the tail of the blinding stream) depend on no challenge, so they are drawn and
  // committed on the auxiliary stream beside the phase-0 / witness work.  The main stream is idle here (synchronised above)
  // and only touches this region again after waiting for ev_rand.  A sharded commitment is a collective: it stays on the
  // main stream, in protocol order.
  const bool early_rand = !srs->sharded();
  if (early_rand) {
    zkfhe_ctx *aux = ws->aux;
    Fr *rand_dev = ws->misc.fr() + 8 * n;
    G1Affine *pt_dev = (G1Affine *)ws->points.p + std::max<size_t>(ws->n_all, cfg.n_perm());
    CK(rng_fill(aux->stream, ctr_rand, 0, rand_dev, n, n, 1));
    (void)pt_dev;   // the commitment is stored by the MSM's last kernel into pinned memory: no copy command
    if (int rc = zkfhe_msm_batch_xyzz(aux, srs->g, (const zkfhe_fr *)rand_dev, 1, (zkfhe_g1_xyzz *)ws->host_rand_pt))
      return zk_fail_msg(ctx, rc, std::string("random polynomial commitment (auxiliary stream): ") + zkfhe_last_error(aux));
    // ... and its Lagrange form, which the evaluation round reads: one single-column transform less on the proof's critical path
    if (int rc = zkfhe_ntt_batch_to(aux, (const zkfhe_fr *)rand_dev, (zkfhe_fr *)(ws->misc.fr() + 9 * n), 1, (int)k, 0))
      return zk_fail_msg(ctx, rc, std::string("random polynomial, Lagrange form (auxiliary stream): ") + zkfhe_last_error(aux));
    ZK_HIP(ctx, hipEventRecord(ws->ev_rand, aux->stream));
  }
  Context ctx0(CTX_PHASE0, false, false), ctx_gate(CTX_GATE1, false, false), ctx_rlc(CTX_RLC1, true, false);
  std::vector<Cell> make_public;
  instances.clear();
  tr.common_scalar(pk->vk_digest);
  // the 5121 public inputs are hashed on a helper thread, started as soon as they are known: it runs beside the phase-0
  // precomputation, upload and commitment (a Poseidon sponge is one sequential chain of 2561 permutations here)
  // The first 2 N of them are pk0 | pk1 (examples/bfv.rs:118-119), the same for every encryption under one public key: the
  // transcript state behind them is kept per key (prefix_cache.hpp), and a proof that finds its key there starts from it --
  // 1 024 of the 2 561 permutations at k = 13.
  const auto on_public = [&](const std::vector<Cell> &pub) {
    instances.reserve(pub.size());
    for (const Cell &c : pub) instances.push_back(c.value);
    const size_t n_key = std::min<size_t>(instances.size(), 2 * (size_t)pk->prm.N);
    Transcript::State mid;
    if (pk->prehash.take(input_json, strlen(input_json), instances.data(), instances.size(), mid)) {
      tr.restore(mid);   // announced ahead of time (zkfhe_bfv_pk_prehash): `vk digest | public inputs` are absorbed already
    } else if (!n_key || !pk->prefix.capacity()) {
      tr.common_scalars_async(instances);
    } else if (pk->prefix.lookup(instances.data(), n_key, mid)) {
      tr.restore(mid);
      tr.common_scalars_async(std::vector<U256>(instances.begin() + (long)n_key, instances.end()));
    } else {
      PrefixCache *cache = &pk->prefix;
      std::vector<U256> key(instances.begin(), instances.begin() + (long)n_key);
      tr.common_scalars_async_marked(instances, n_key, [cache, key](const Transcript::State &st) { cache->insert(key.data(), key.size(), st); });
    }
  };
  // machine-word phase 0 (bfv_phase0_fast.hpp) for every input in its domain; anything else goes through the line-by-line
  // restatement of the reference, which also words the errors
  BfvState st;
  if (!bfv_phase0_fast(ctx0, input_json, strlen(input_json), pk->prm, make_public, st, on_public)) {
    const CircuitInput in = CircuitInput::parse_json(input_json);
    trace.mark("parse_json");
    st = bfv_phase0(ctx0, in, pk->prm, make_public, on_public);
  }
  trace.mark("bfv_phase0");
  Assigner as(cfg, false, ws->host_adv);
  as.place(ctx0, true);
  trace.mark("place phase 0");
  // the instance column's values ride up with the proof's first table flush (they are needed from the grand products on)
  if (instances.size() > n) return zk_fail_msg(ctx, ZKFHE_EINVAL, "more public inputs than rows");
  const bool inst_via_ring = instances.size() * 32 <= ((size_t)1 << 20);   // long instance columns (k >= 18) keep their own copy
  const U256 *inst_staged = nullptr;
  if (inst_via_ring) {
    STAGE(staged, U256, ws, instances.data(), instances.size() * 32);
    inst_staged = staged;
  }
  auto blind_and_upload = [&](unsigned c_lo, unsigned c_hi) -> int {
    // the table is pinned and column-contiguous; the blinding rows u .. n-1 are drawn on the device.  The conversion kernel reads the
    // pinned table itself (hipHostMalloc memory is mapped into the device's address space): no copy command on the stream -- with twenty
    // proofs starting together the runtime's copy path cost each of them 2-5 ms of CPU before their first kernel (ZKFHE_UPLOAD=copy: one
    // DMA for the whole phase, then the conversion in place)
    const char *upload_env = getenv("ZKFHE_UPLOAD");   // read per proof, like ZKFHE_WITNESS / ZKFHE_EARLY_P1 (the tests switch it)
    const bool upload_by_copy = upload_env && !strcmp(upload_env, "copy");
    if (upload_by_copy) {
      ZK_HIP(ctx, hipMemcpyAsync(ws->adv_l.fr() + (size_t)c_lo * n, as.t.advice[c_lo], (size_t)(c_hi - c_lo) * n * 32, hipMemcpyHostToDevice, ctx->stream));
      CK(zkfhe_fr_to_mont(ctx, (const zkfhe_fr *)(ws->adv_l.fr() + (size_t)c_lo * n), (zkfhe_fr *)(ws->adv_l.fr() + (size_t)c_lo * n),
                          (size_t)(c_hi - c_lo) * n));
    } else {
      CK(zkfhe_fr_to_mont(ctx, (const zkfhe_fr *)as.t.advice[c_lo], (zkfhe_fr *)(ws->adv_l.fr() + (size_t)c_lo * n), (size_t)(c_hi - c_lo) * n));
    }
    return rng_fill(ctx->stream, ctr_adv + (uint64_t)c_lo * nbl_all, nbl_all, ws->adv_l.fr() + (size_t)c_lo * n + u, nbl_all, n, c_hi - c_lo);
  };
  std::vector<AffinePoint> adv_commit(cfg.n_advice()), pts;
  trace.mark("transcript: instances");
  CK(blind_and_upload(0, cfg.n_gate0));
  trace.mark("blind + upload phase 0");
  const bool host_witness = witness_on_host();
  // see "Early phase-1 commitment" below.  On with the Poseidon transcript, whose challenge after the phase-0 commitment is
  // 2561 sequential permutations (the public inputs, ~20 ms of host time) away: measured on a wave of 20 concurrent proofs
  // +9 % (the GPU commits while every proof hashes), -1.5 % in steady state (two extra small commitments per proof).  With
  // Blake2b there is no hash to hide -- but on long rows (k >= 18) the RLC witness the host computes from that challenge is
  // (k = 19: 6 ms of Horner chains and placement with the GPU idle): 155.8 -> 151.5 ms per proof; at k = 16 it costs 0.7 ms.
  // ZKFHE_EARLY_P1=0 / 1 overrides.
  const char *early_env = getenv("ZKFHE_EARLY_P1");
  const bool early_want = early_env ? early_env[0] == '1' : (cfg.transcript == TR_POSEIDON || k >= 18);
  const bool early_p1 = early_want && !host_witness && cfg.n_lookup > 0 && cfg.lookup_bits == 8;
  GpuPhase1 g1(ctx, pk, ws);
  const zkfhe_basis *p0_basis = small_basis;
  if (host_witness) {
    CK(commit_cols(ctx, srs, p0_basis, ws->adv_l.fr(), cfg.n_gate0, (G1Affine *)ws->points.p, pts));
  } else {
    // the commitment's points come back through an event; the phase-1 gadget launches queue up behind the MSM
    CK(alloc_witness_buffers(ctx, pk, ws));
    CK(srs_msm_pts(ctx, srs, p0_basis, ws->adv_l.fr(), cfg.n_gate0, ws->host_pts));   // stored into pinned memory by the MSM itself
    CK(srs_record(ctx, srs, ws->ev_pts));   // sharded: behind the all-gather on the communicator's stream, which the gadget launches below overlap
    CK(g1.launch(st));
    if (early_p1) {
      // Early phase-1 commitment.  Of the phase-1 columns only the RLC columns and the 16 constrain_mul gate cells depend on
      // the challenge the transcript yields after the phase-0 commitment -- and with the Poseidon transcript that challenge
      // is 2561 sequential permutations away (the public inputs).  So the gate and lookup columns are placed, blinded,
      // permuted and committed NOW, with those cells zero and the RLC columns zero; afterwards the missing cells are
      // committed as sparse correction columns and added (a commitment is linear in its column).  Same points, same bytes.
      const size_t nbl0 = n - u, n_adv1 = cfg.n_advice() - cfg.n_gate0, n_early = n_adv1 + 2 * cfg.n_lookup;
      CK(g1.place_early());
      CK(rng_fill(ctx->stream, ctr_adv + (uint64_t)cfg.n_gate0 * nbl0, nbl0, ws->adv_l.fr() + (size_t)cfg.n_gate0 * n + u, nbl0, n, cfg.adv_rlc0() - cfg.n_gate0));
      ZK_HIP(ctx, hipMemsetAsync(ws->adv_l.fr() + (size_t)cfg.adv_rlc0() * n, 0, (size_t)cfg.n_rlc * n * 32, ctx->stream));
      int *err_dev = ws->host_early_err;   // pinned: written by the kernel, read by the host after ev_early
      *err_dev = 0;
      CK(zkw::lookup_permute(ctx, ws->adv_l.fr() + (size_t)cfg.adv_lookup0() * n, n, (unsigned)u, cfg.n_lookup, ws->la_l.fr(), ws->ls_l.fr(), err_dev));
      CK(rng_fill(ctx->stream, ctr_lk, 2 * nbl0, ws->la_l.fr() + u, nbl0, n, cfg.n_lookup));
      CK(rng_fill(ctx->stream, ctr_lk + nbl0, 2 * nbl0, ws->ls_l.fr() + u, nbl0, n, cfg.n_lookup));
      CK(srs_msm_pts(ctx, srs, small_basis, ws->adv_l.fr() + (size_t)cfg.n_gate0 * n, n_early, ws->host_early));
      CK(srs_record(ctx, srs, ws->ev_early));
    }
    ZK_HIP(ctx, hipEventSynchronize(ws->ev_pts));
    pts.resize(cfg.n_gate0);
    points_canon(srs, ws->host_pts, cfg.n_gate0, pts.data());
  }
  trace.mark("commit phase 0 (GPU)");
  ctx->proof_marks[0] = (float)(now_ms() - t_start);
  for (unsigned c = 0; c < cfg.n_gate0; ++c) tr.write_point(adv_commit[c] = pts[c]);
  const U256 gamma_rlc = tr.squeeze();
  ctx->proof_marks[1] = (float)(now_ms() - t_start);
  // ------------------------------------------------------------ phase 1 witness
  const size_t nbl = n - u;
  int *lookup_err = nullptr;
  if (host_witness) {
    bfv_phase1(st, pk->prm, ctx_gate, ctx_rlc, gamma_rlc);
    as.place(ctx_gate, true);
    as.place(ctx_rlc, true);
    as.place_lookups(ctx_gate);
  } else {
    // RLC context on the host (6 K cells), gate context (1.2 M cells) on the device
    U256 evals[12];
    bfv_phase1_rlc(st, ctx_rlc, gamma_rlc, evals);
    trace.mark("rlc context");
    as.place(ctx_rlc, true);
    trace.mark("place rlc");
    if (!early_p1) {
      CK(g1.finish(evals));
    } else {
      // the challenge-dependent rest: the reserved gate cells (into the columns and into sparse correction columns), the RLC
      // columns; two small commitments, then early + correction on the host
      const GpuPhase1::PatchPlan plan = g1.patch_plan();
      const size_t np = plan.columns.size(), n_adv1 = cfg.n_advice() - cfg.n_gate0;
      Fr *patch_cols = ws->tmp_c.fr();
      std::vector<zkfhe_sparse_term> terms;
      const bool sparse = !srs->sharded() && zkfhe_basis_has_multiples(srs->g_lagrange);
      CK(g1.patch(evals, plan, patch_cols, sparse ? &terms : nullptr));
      CK(blind_and_upload(cfg.adv_rlc0(), cfg.n_advice()));
      // the corrections sit behind the early commitments in ws->points; the last slot of that buffer belongs to the random
      // polynomial's commitment (auxiliary stream)
      if (np + cfg.n_rlc > ws->out_pts_cap)
        return zk_fail_msg(ctx, ZKFHE_EINVAL, "early phase-1 commitment: correction points do not fit the workspace (set ZKFHE_EARLY_P1=0)");
      (void)n_adv1;
      uint8_t *fix_dev = ws->out_pts();   // pinned: the correction points are read by the host right below
      const size_t ps = pt_stride(srs);
      if (np && sparse) {
        // a dozen non-zero cells: one wave per correction column over the digit-multiple table
        STAGE(terms_dev, zkfhe_sparse_term, ws, terms.data(), terms.size() * sizeof(zkfhe_sparse_term));
        CK(flush_staged(ctx, ws));
        CK(zkfhe_msm_sparse_xyzz(ctx, srs->g_lagrange, terms_dev, terms.size(), np, (zkfhe_g1_xyzz *)fix_dev));   // sparse => one GPU => accumulator-form slots
      } else if (np) {
        CK(srs_msm_pts(ctx, srs, srs->g_lagrange, patch_cols, np, fix_dev));
      }
      if (cfg.n_rlc) CK(srs_msm_pts(ctx, srs, srs->g_lagrange, ws->adv_l.fr() + (size_t)cfg.adv_rlc0() * n, cfg.n_rlc, fix_dev + np * ps));
      CK(srs_join(ctx, srs, false));
      ZK_HIP(ctx, zk_wait(ctx));   // everything queued so far, the early commitment (ev_early) included
      if (*ws->host_early_err) return zk_fail_msg(ctx, ZKFHE_EINVAL, "lookup input not in table: a range check of the witness fails");
      // early commitment + correction, on the host (at most a handful of additions), in the form the slots hold; the sums are
      // normalised with the rest of the round below
      for (size_t t = 0; t < np; ++t) {
        if (srs->sharded()) {
          G1Affine &e = ((G1Affine *)ws->host_early)[plan.columns[t]];
          zk::G1X acc = zk::g1x_from_affine(e);
          zk::g1x_add_affine(acc, ((const G1Affine *)fix_dev)[t], false);
          e = zk::g1x_to_affine(acc);
        } else {
          zk::g1x_add(((zk::G1X *)ws->host_early)[plan.columns[t]], ((const zk::G1X *)fix_dev)[t]);
        }
      }
      for (unsigned j = 0; j < cfg.n_rlc; ++j) memcpy(ws->host_early + (size_t)(cfg.adv_rlc0() - cfg.n_gate0 + j) * ps, fix_dev + (np + j) * ps, ps);
    }
    trace.mark("gpu phase 1 (enqueue)");
  }
  std::vector<std::vector<U256>> lookup_inputs(host_witness ? cfg.n_lookup : 0);
  for (unsigned i = 0; i < lookup_inputs.size(); ++i)
    lookup_inputs[i].assign(as.t.advice[cfg.adv_lookup0() + i], as.t.advice[cfg.adv_lookup0() + i] + u);
  const double t_wit = now_ms();
  if (host_witness) {
    CK(blind_and_upload(cfg.n_gate0, cfg.n_advice()));
  } else if (!early_p1) {
    // blinding rows of the device-generated columns (same draw order as the host path: column by column)
    const unsigned nc = cfg.adv_rlc0() - cfg.n_gate0;
    CK(rng_fill(ctx->stream, ctr_adv + (uint64_t)cfg.n_gate0 * nbl, nbl, ws->adv_l.fr() + (size_t)cfg.n_gate0 * n + u, nbl, n, nc));
    CK(blind_and_upload(cfg.adv_rlc0(), cfg.n_advice()));
  }
  trace.mark("blind + upload phase 1");
  std::vector<AffinePoint> la_commit, ls_commit;
  const bool merged = !host_witness && cfg.n_lookup > 0;   // permuted lookup columns committed together with the advice
  if (early_p1) {
    const size_t n_adv1 = cfg.n_advice() - cfg.n_gate0;
    pts.resize(n_adv1 + 2 * (size_t)cfg.n_lookup);
    points_canon(srs, ws->host_early, pts.size(), pts.data());   // one inversion for the round: advice | permuted inputs | permuted tables
    la_commit.assign(pts.begin() + (long)n_adv1, pts.begin() + (long)(n_adv1 + cfg.n_lookup));
    ls_commit.assign(pts.begin() + (long)(n_adv1 + cfg.n_lookup), pts.end());
    pts.resize(n_adv1);
  } else if (merged) {
    // Single-expression lookups do not use theta, so the permuted columns can be built before the advice commitment is
    // hashed: one MSM call over [phase-1 advice | la | ls] (contiguous in all_l) instead of two, same points, same
    // transcript order, same draw order of the blinding values.
    if (cfg.lookup_bits != 8) return zk_fail_msg(ctx, ZKFHE_EINVAL, "the device lookup permutation is built for lookup_bits = 8");
    lookup_err = ws->out_flags() + 2;   // pinned
    *lookup_err = 0;
    CK(zkw::lookup_permute(ctx, ws->adv_l.fr() + (size_t)cfg.adv_lookup0() * n, n, (unsigned)u, cfg.n_lookup, ws->la_l.fr(), ws->ls_l.fr(), lookup_err));
    // blinding rows: la_i then ls_i, lookup by lookup -- two interleaved runs of the stream
    CK(rng_fill(ctx->stream, ctr_lk, 2 * nbl, ws->la_l.fr() + u, nbl, n, cfg.n_lookup));
    CK(rng_fill(ctx->stream, ctr_lk + nbl, 2 * nbl, ws->ls_l.fr() + u, nbl, n, cfg.n_lookup));
    const size_t n_adv1 = cfg.n_advice() - cfg.n_gate0;
    CK(commit_cols_out(ctx, srs, small_basis, ws->adv_l.fr() + (size_t)cfg.n_gate0 * n, n_adv1 + 2 * cfg.n_lookup, ws, pts));
    if (*lookup_err) return zk_fail_msg(ctx, ZKFHE_EINVAL, "lookup input not in table: a range check of the witness fails");
    la_commit.assign(pts.begin() + n_adv1, pts.begin() + n_adv1 + cfg.n_lookup);
    ls_commit.assign(pts.begin() + n_adv1 + cfg.n_lookup, pts.end());
    pts.resize(n_adv1);
  } else {
    CK(commit_cols_out(ctx, srs, small_basis, ws->adv_l.fr() + (size_t)cfg.n_gate0 * n, cfg.n_advice() - cfg.n_gate0, ws, pts));
  }
  trace.mark("commit phase 1 (GPU)");
  for (unsigned c = cfg.n_gate0; c < cfg.n_advice(); ++c) adv_commit[c] = pts[c - cfg.n_gate0];
  tr.write_points(pts);
  tr.squeeze();  // theta: squeezed in protocol order, unused by single-expression lookups
  // ------------------------------------------------------------ lookups: permuted input / table
  if (cfg.n_lookup) {
    if (!merged) {
      std::vector<U256> ap, sp;
      U256 *stA = ws->host_blind, *stS = ws->host_blind + (size_t)cfg.n_lookup * n;  // la | ls, as on the device
      for (unsigned i = 0; i < cfg.n_lookup; ++i) {
        permute_lookup(lookup_inputs[i], u, 1u << cfg.lookup_bits, ap, sp);
        U256 *colA = stA + (size_t)i * n, *colS = stS + (size_t)i * n;
        std::copy(ap.begin(), ap.end(), colA);
        std::copy(sp.begin(), sp.end(), colS);
        for (size_t r = u; r < n; ++r) colA[r] = colS[r] = fe::zero();
      }
      ZK_HIP(ctx, hipMemcpyAsync(ws->la_l.p, ws->host_blind, 2 * (size_t)cfg.n_lookup * n * 32, hipMemcpyHostToDevice, ctx->stream));
      CK(zkfhe_fr_to_mont(ctx, (const zkfhe_fr *)ws->la_l.p, (zkfhe_fr *)ws->la_l.p, 2 * (size_t)cfg.n_lookup * n));
      CK(rng_fill(ctx->stream, ctr_lk, 2 * nbl, ws->la_l.fr() + u, nbl, n, cfg.n_lookup));
      CK(rng_fill(ctx->stream, ctr_lk + nbl, 2 * nbl, ws->ls_l.fr() + u, nbl, n, cfg.n_lookup));
      CK(commit_cols_out(ctx, srs, small_basis, ws->la_l.fr(), 2 * cfg.n_lookup, ws, la_commit));  // la | ls contiguous
      ls_commit.assign(la_commit.begin() + cfg.n_lookup, la_commit.end());
      la_commit.resize(cfg.n_lookup);
    }
    std::vector<AffinePoint> both(2 * (size_t)cfg.n_lookup);
    for (unsigned i = 0; i < cfg.n_lookup; ++i) both[2 * i] = la_commit[i], both[2 * i + 1] = ls_commit[i];
    tr.write_points(both);
  }
  trace.mark("lookup permute + commit");
  const U256 beta_c = tr.squeeze(), gamma_c = tr.squeeze();
  const Fr beta = mont(beta_c), gamma = mont(gamma_c);
  GateHold gate;
  if (!srs->sharded()) gate.enter();   // a sharded proof's collectives must not wait for another rank's admission order
  // ------------------------------------------------------------ permutation grand products
  {
    if (instances.size() < ws->inst_count)
      ZK_HIP(ctx, hipMemsetAsync(ws->inst_l.fr() + instances.size(), 0, (ws->inst_count - instances.size()) * 32, ctx->stream));
    if (inst_via_ring) {
      CK(flush_staged(ctx, ws));   // nothing to send when an earlier round's flush took the values along
      if (!instances.empty()) CK(zkfhe_fr_to_mont(ctx, (const zkfhe_fr *)inst_staged, (zkfhe_fr *)ws->inst_l.p, instances.size()));
    } else {
      CK(upload_canon(ctx, ws->inst_l.fr(), instances.data(), instances.size()));
    }
    ws->inst_count = instances.size();
  }
  U256 dcan;
  memcpy(dcan.l, DELTA_CANON, 32);
  Fr *beta_delta_dev = (Fr *)ws->small.p;  // [n_perm]
  zkp::k_powers<<<1, 256, 0, ctx->stream>>>(beta, mont(dcan), beta_delta_dev, cfg.n_perm());
  ZK_LAUNCH_CHECK(ctx);
  zkp::PermArgs pa;
  pa.adv = ws->adv_l.fr();
  pa.constcol = pk->fixed_l.fr() + (size_t)cfg.fix_const() * n;
  pa.inst = ws->inst_l.fr();
  pa.sigma = pk->sigma_l.fr();
  pa.wpow = dom->fwd;
  pa.beta_delta = beta_delta_dev;
  pa.beta = beta;
  pa.gamma = gamma;
  pa.n_advice = cfg.n_advice();
  pa.n_perm = cfg.n_perm();
  pa.chunk = cfg.chunk();
  pa.n_chunks = cfg.n_chunks();
  pa.n = n;
  const size_t nch = cfg.n_chunks();
  zkp::k_perm_num_den<<<grid_for(ctx, nch * n), 256, 0, ctx->stream>>>(pa, ws->num.fr(), ws->den.fr());
  ZK_LAUNCH_CHECK(ctx);
  CK(zkfhe_fr_batch_invert(ctx, (zkfhe_fr *)ws->den.p, nch * n));
  CK(zkfhe_fr_mul(ctx, (const zkfhe_fr *)ws->num.p, (const zkfhe_fr *)ws->den.p, (zkfhe_fr *)ws->num.p, nch * n));
  Fr *totals_dev = (Fr *)ws->small.p + 4096;
  // running products per column; long columns in segments (prover_kernels.hip.hpp), the segment products in the dead `den` buffer
  auto prefix_products = [&](const Fr *ratio, Fr *z, size_t n_cols) -> int {
    if (n <= 65536) {   // measured at k = 16 (two segments): no gain over one workgroup per column; k = 19: 6.5 -> 1 ms per proof
      zkp::k_prefix_product<<<(unsigned)n_cols, 1024, 0, ctx->stream>>>(ratio, z, totals_dev, n, (unsigned)u);
      ZK_LAUNCH_CHECK(ctx);
      return ZKFHE_OK;
    }
    const unsigned seg_len = 32768, segs = (unsigned)(n / seg_len);   // u < n: the output row z[u] lies inside the last segment
    Fr *seg = ws->den.fr();
    if ((size_t)n_cols * segs * 32 > ws->den.bytes) return zk_fail_msg(ctx, ZKFHE_EINVAL, "prefix products: segment buffer too small");
    zkp::k_prefix_seg_totals<<<dim3(segs, (unsigned)n_cols), 1024, 0, ctx->stream>>>(ratio, seg, n, (unsigned)u, seg_len);
    ZK_LAUNCH_CHECK(ctx);
    zkp::k_prefix_seg_scan<<<(unsigned)((n_cols + 63) / 64), 64, 0, ctx->stream>>>(seg, segs, (unsigned)n_cols, totals_dev);
    ZK_LAUNCH_CHECK(ctx);
    zkp::k_prefix_seg_apply<<<dim3(segs, (unsigned)n_cols), 1024, 0, ctx->stream>>>(ratio, seg, z, n, (unsigned)u, seg_len);
    ZK_LAUNCH_CHECK(ctx);
    return ZKFHE_OK;
  };
  CK(prefix_products(ws->num.fr(), ws->pz_l.fr(), nch));
  int *const closes = ws->out_flags();   // pinned: [0] permutation, [1] lookups -- read after the commitment of the products below
  closes[0] = closes[1] = 1;
  {
    // chunk j starts where chunk j - 1 ended: the carries are the running product of the chunk totals, formed on the device
    zkp::k_chunk_carry<<<1, 1024, 0, ctx->stream>>>(totals_dev, (unsigned)nch, 0, closes);
    ZK_LAUNCH_CHECK(ctx);
    zkp::k_scale_rows<<<grid_for(ctx, nch * (u + 1)), 256, 0, ctx->stream>>>(ws->pz_l.fr(), totals_dev, n, (unsigned)(u + 1), (unsigned)nch);
    ZK_LAUNCH_CHECK(ctx);
    // blinding rows u+1 .. n-1 (drawn per chunk, in order)
    const size_t nb = n - u - 1;
    CK(rng_fill(ctx->stream, ctr_pz, nb, ws->pz_l.fr() + (u + 1), nb, n, nch));
  }
  // ------------------------------------------------------------ lookup grand products
  if (cfg.n_lookup) {
    const size_t nl = cfg.n_lookup;
    zkp::k_lookup_num_den<<<grid_for(ctx, nl * n), 256, 0, ctx->stream>>>(ws->adv_l.fr() + (size_t)cfg.adv_lookup0() * n,
                                                                          pk->fixed_l.fr() + (size_t)cfg.fix_table() * n, ws->la_l.fr(), ws->ls_l.fr(),
                                                                          beta, gamma, (unsigned)nl, n, ws->num.fr(), ws->den.fr());
    ZK_LAUNCH_CHECK(ctx);
    CK(zkfhe_fr_batch_invert(ctx, (zkfhe_fr *)ws->den.p, nl * n));
    CK(zkfhe_fr_mul(ctx, (const zkfhe_fr *)ws->num.p, (const zkfhe_fr *)ws->den.p, (zkfhe_fr *)ws->num.p, nl * n));
    CK(prefix_products(ws->num.fr(), ws->lz_l.fr(), nl));
    zkp::k_chunk_carry<<<1, 1024, 0, ctx->stream>>>(totals_dev, (unsigned)nl, 1, closes + 1);
    ZK_LAUNCH_CHECK(ctx);
    const size_t nb = n - u - 1;
    CK(rng_fill(ctx->stream, ctr_lz, nb, ws->lz_l.fr() + (u + 1), nb, n, nl));
  }
  std::vector<AffinePoint> pz_commit, lz_commit;
  CK(commit_cols_out(ctx, srs, srs->g_lagrange, ws->pz_l.fr(), nch + cfg.n_lookup, ws, pz_commit));  // pz | lz contiguous
  if (!closes[0]) return zk_fail_msg(ctx, ZKFHE_EINVAL, "permutation argument does not close: a copy constraint is violated");
  if (!closes[1]) return zk_fail_msg(ctx, ZKFHE_EINVAL, "lookup argument does not close");
  tr.write_points(pz_commit);   // permutation products, then lookup products
  lz_commit.assign(pz_commit.begin() + nch, pz_commit.end());
  pz_commit.resize(nch);
  // ------------------------------------------------------------ vanishing: random polynomial (coefficient form)
  Fr *rand_c = ws->misc.fr() + 8 * n, *rand_l = ws->misc.fr() + 9 * n, *H_c = ws->misc.fr() + 10 * n, *H_l = ws->misc.fr() + 11 * n;
  // committed at the start of the proof on the auxiliary stream ("random polynomial, early"): the last n draws of the stream
  if (early_rand) {
    ZK_HIP(ctx, hipEventSynchronize(ws->ev_rand));
    ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ws->ev_rand, 0));   // later reads of rand_c on the main stream
    AffinePoint rp;
    points_canon(srs, ws->host_rand_pt, 1, &rp);
    tr.write_point(rp);
  } else {
    CK(rng_fill(ctx->stream, ctr_rand, 0, rand_c, n, n, 1));
    std::vector<AffinePoint> rand_commit;
    CK(commit_cols_out(ctx, srs, srs->g, rand_c, 1, ws, rand_commit));
    tr.write_point(rand_commit[0]);
  }
  const Fr y = mont(tr.squeeze());
  const double t_commit = now_ms();
  trace.mark("grand products + commits");
  // ------------------------------------------------------------ quotient
  // The quotient has degree < 3n, so three cosets determine it (one quarter less extension and evaluation work).  The
  // fourth coset is only needed for the degree check below (a violated gate shows up as a non-zero top quarter):
  // a key built with ZKFHE_CHECK_QUOTIENT=1 in the environment keeps it (pk->ext_rows).
  const int q_rows = pk->ext_rows;
  // (Extending the challenge-independent columns -- 264 of 402 -- right after the early commitment, on the main or on the
  // auxiliary stream, was measured and dropped: 165-172 proofs/s on a wave of 20 against 172-177 without, 185-192 in steady
  // state against 192-203; the smaller NTT batches and the extra launches cost more than the idle time they fill.)
  // One proof over several GPUs (srs->sharded()): besides the commitments (point ranges, comm.hip) the extension to the coset
  // domain and the quotient are sharded BY COLUMN.  Rank r owns the permutation chunks [nch r / W, nch (r+1) / W), hence the
  // advice columns of those chunks, their gates / RLC gates / lookups and the chunks' permutation terms; it extends only those
  // columns (plus the grand product of the chunk before its first one, which the chaining term reads), evaluates only the
  // expression groups that belong to them, and the W partial quotients (rows * n values each) are all-gathered and summed on
  // every rank.  With W = 1 everything is owned: the same code.
  const unsigned W_sh = srs->sharded() ? (unsigned)zkfhe_comm_world(srs->comm) : 1u, r_sh = srs->sharded() ? (unsigned)zkfhe_comm_rank(srs->comm) : 0u;
  const size_t c_lo = nch * r_sh / W_sh, c_hi = nch * (r_sh + 1) / W_sh;
  auto owns_chunk = [&](size_t j) { return j >= c_lo && j < c_hi; };
  auto owns_col = [&](size_t perm_col) { return owns_chunk(perm_col / cfg.chunk()); };
  if (W_sh == 1) {
    CK(extend_cols(ctx, pk, ws, ws->all_l.fr(), ws->n_all, ws->all_ext.fr()));
  } else {
    const size_t ecol = n * (size_t)q_rows;
    auto ext_range = [&](const View &l, const View &e, size_t lo, size_t hi) -> int {
      if (hi <= lo) return ZKFHE_OK;
      return extend_cols(ctx, pk, ws, l.fr() + lo * n, hi - lo, e.fr() + lo * ecol);
    };
    const size_t a_lo = std::min<size_t>(c_lo * cfg.chunk(), cfg.n_advice()), a_hi = std::min<size_t>(c_hi * cfg.chunk(), cfg.n_advice());
    CK(ext_range(ws->adv_l, ws->adv_ext, a_lo, a_hi));
    const size_t l_lo = a_lo > cfg.adv_lookup0() ? std::min<size_t>(a_lo - cfg.adv_lookup0(), cfg.n_lookup) : 0;
    const size_t l_hi = a_hi > cfg.adv_lookup0() ? std::min<size_t>(a_hi - cfg.adv_lookup0(), cfg.n_lookup) : 0;
    CK(ext_range(ws->la_l, ws->la_ext, l_lo, l_hi));
    CK(ext_range(ws->ls_l, ws->ls_ext, l_lo, l_hi));
    CK(ext_range(ws->lz_l, ws->lz_ext, l_lo, l_hi));
    CK(ext_range(ws->pz_l, ws->pz_ext, c_lo ? c_lo - 1 : 0, c_hi));
    if (owns_col(cfg.perm_inst())) CK(ext_range(ws->inst_l, ws->inst_ext, 0, 1));
  }
  {
    // expression groups, in the folding order of oracle/halo2_ref.py expressions_at: the expressions are numbered globally (the
    // power of y of a group is fixed by the number of its last expression); a group is a run of owned, consecutive units of one
    // kind, cut at the kernel's group sizes
    std::vector<zkp::QGroup> groups;
    std::vector<size_t> last_e;  // global index of the last expression of each group
    size_t e = 0;
    auto run = [&](int type, size_t first, size_t count, size_t max_group, size_t expr_per_unit, const std::function<bool(size_t)> &owned) {
      size_t j = first;
      while (j < first + count) {
        if (!owned(j)) {
          e += expr_per_unit;
          ++j;
          continue;
        }
        size_t len = 0;
        while (j + len < first + count && len < max_group && owned(j + len)) ++len;
        groups.push_back(zkp::QGroup{type, (int)j, (int)len, 0});
        e += len * expr_per_unit;
        last_e.push_back(e - 1);
        j += len;
      }
    };
    run(zkp::QG_GATE, 0, cfg.n_gate(), 8, 1, [&](size_t j) { return owns_col(j); });
    run(zkp::QG_RLC, 0, cfg.n_rlc, cfg.n_rlc ? cfg.n_rlc : 1, 1, [&](size_t j) { return owns_col(cfg.adv_rlc0() + j); });
    run(zkp::QG_PERM_FIRST, 0, 1, 1, 1, [&](size_t) { return owns_chunk(0); });
    run(zkp::QG_PERM_LAST, 0, 1, 1, 1, [&](size_t) { return owns_chunk(nch - 1); });
    run(zkp::QG_PERM_C, 1, nch - 1, 32, 1, [&](size_t j) { return owns_chunk(j); });
    run(zkp::QG_PERM_D, 0, nch, 4, 1, [&](size_t j) { return owns_chunk(j); });
    run(zkp::QG_LOOKUP, 0, cfg.n_lookup, 3, 5, [&](size_t i) { return owns_col(cfg.adv_lookup0() + i); });
    const size_t E = e;
    // (The same expressions cut by COLUMN BLOCK instead of by kind -- k_quotient_blocks, round 5 -- read a quarter less and were 29 %
    // slower: tools/exp/patches/quotient_blocks.patch, profiles/r5_probes.md section 2.)
    const size_t G = groups.size();
    if (G > 96 || G * n * (size_t)q_rows * 32 > ws->partials.bytes) return zk_fail_msg(ctx, ZKFHE_EINVAL, "too many quotient groups for the workspace");
    std::vector<Fr> ypow(G);
    for (size_t g = 0; g < G; ++g) ypow[g] = zk::zk_fr_to_29(fr_pow(y, E - 1 - last_e[g]));   // 2^261 form: constant operands of k_quotient_combine
    if (groups.empty()) groups.push_back(zkp::QGroup{});     // a rank that owns nothing still stages a (unread) entry
    if (ypow.empty()) ypow.push_back(Fr::zero());
    STAGE(groups_dev, zkp::QGroup, ws, groups.data(), groups.size() * sizeof(zkp::QGroup));
    STAGE(ypow_dev, Fr, ws, ypow.data(), ypow.size() * 32);
    const Fr wext = zk_fr_root_of_unity((int)k + 2);
    const Fr gn = fr_pow(mont_u64(COSET_G), n), i4 = fr_pow(wext, n);
    Fr zinv[4], cur = gn;
    for (int t = 0; t < 4; ++t) {
      zinv[t] = zk::zk_fr_to_29(fr_inv(cur - Fr::one()));
      cur = cur * i4;
    }
    STAGE(zinv_dev, Fr, ws, zinv, 4 * 32);
    CK(flush_staged(ctx, ws));
    zkp::QArgs qa;
    qa.adv = ws->adv_ext.fr();
    qa.fix = pk->fixed_ext.fr();
    qa.sig = pk->sigma_ext.fr();
    qa.pz = ws->pz_ext.fr();
    qa.lz = ws->lz_ext.fr();
    qa.la = ws->la_ext.fr();
    qa.ls = ws->ls_ext.fr();
    qa.inst = ws->inst_ext.fr();
    qa.lext = pk->l_ext.fr();
    qa.xs = pk->xs_ext.fr();
    qa.beta_delta = beta_delta_dev;
    qa.groups = groups_dev;
    qa.partials = ws->partials.fr();
    qa.y = y;
    qa.beta = beta;
    qa.gamma = gamma;
    qa.gamma_rlc = mont(gamma_rlc);
    qa.log_n = k;
    qa.u = (unsigned)u;
    qa.n_gate = cfg.n_gate();
    qa.n_rlc = cfg.n_rlc;
    qa.adv_rlc0 = cfg.adv_rlc0();
    qa.fix_qrlc0 = cfg.fix_qrlc0();
    qa.fix_const = cfg.fix_const();
    qa.fix_table = cfg.fix_table();
    qa.adv_lookup0 = cfg.adv_lookup0();
    qa.n_lookup = cfg.n_lookup;
    qa.n_advice = cfg.n_advice();
    qa.n_perm = cfg.n_perm();
    qa.chunk = cfg.chunk();
    qa.n_chunks = cfg.n_chunks();
    qa.rows = (unsigned)q_rows;
    const size_t npts = n * (size_t)q_rows;
    if (W_sh == 1) {
      qa.pt0 = 0;
      qa.pt_count = npts;
      dim3 grid((unsigned)((npts + 255) / 256), (unsigned)G);
      zkp::k_quotient_partials<<<grid, 256, 0, ctx->stream>>>(qa);
      ZK_LAUNCH_CHECK(ctx);
      zkp::k_quotient_combine<<<(unsigned)((npts + 255) / 256), 256, 0, ctx->stream>>>(ws->partials.fr(), ypow_dev, (unsigned)G, zinv_dev, k, (unsigned)q_rows, 0, npts, ws->h_ext.fr());
      ZK_LAUNCH_CHECK(ctx);
    } else {
      // h_ext receives this rank's share of the quotient (the sum over ITS expression groups); the W shares are gathered and added up
      // -- field addition is not an RCCL reduction either.  One coset row at a time: the share of row k1 (n values: 16 MB at k = 19)
      // goes onto the communicator's stream as soon as it is formed and travels over xGMI while the groups of row k1 + 1 are
      // evaluated; only the last row's gather is exposed.  (The callback transport of the tests completes each gather in place.)
      if (ws->qgather.bytes < (size_t)W_sh * npts * 32) {
        CK(zkfhe_sync(ctx));
        ws->qgather.release();
        CK(ws->qgather.alloc(ctx, (size_t)W_sh * npts * 32));
      }
      for (int k1 = 0; k1 < q_rows; ++k1) {
        qa.pt0 = (size_t)k1 * n;
        qa.pt_count = n;
        dim3 grid((unsigned)((n + 255) / 256), (unsigned)G);
        if (G) {
          zkp::k_quotient_partials<<<grid, 256, 0, ctx->stream>>>(qa);
          ZK_LAUNCH_CHECK(ctx);
        }
        zkp::k_quotient_combine<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(ws->partials.fr(), ypow_dev, (unsigned)G, zinv_dev, k, (unsigned)q_rows, qa.pt0, n, ws->h_ext.fr());
        ZK_LAUNCH_CHECK(ctx);
        CK(zkfhe_comm_all_gather_async(ctx, srs->comm, ws->h_ext.fr() + qa.pt0, ws->qgather.fr() + (size_t)k1 * W_sh * n, n * 32));
      }
      CK(zkfhe_comm_join(ctx, srs->comm, 0));
      for (int k1 = 0; k1 < q_rows; ++k1) {
        zkp::k_sum_rows<<<grid_for(ctx, n), 256, 0, ctx->stream>>>(ws->qgather.fr() + (size_t)k1 * W_sh * n, W_sh, n, ws->h_ext.fr() + (size_t)k1 * n);
        ZK_LAUNCH_CHECK(ctx);
      }
    }
    const Fr g = mont_u64(COSET_G);
    if (q_rows == 4) {
      CK(zkfhe_coset_ntt_batch(ctx, (const zkfhe_fr *)ws->h_ext.p, (zkfhe_fr *)ws->h_c.p, 1, (int)k, 2, (const zkfhe_fr *)&g, 1));
    } else {
      // three size-n inverse transforms, then the 3x3 Vandermonde solve per coefficient (prover_kernels.hip.hpp: k_ext3_combine)
      Fr *rows3 = ws->partials.fr();            // the partials are dead after the combine; 3n values
      const Fr *pw = pk->ext3_pw.fr();           // (g w_ext^t)^-i, resident in the key
      CK(zkfhe_ntt_batch_to(ctx, (const zkfhe_fr *)ws->h_ext.p, (zkfhe_fr *)rows3, 3, (int)k, 1));
      Fr gk[3], c[3];
      gk[0] = g;
      gk[1] = g * wext;
      gk[2] = gk[1] * wext;
      zkp::Mat3 vinv;
      {
        for (int t = 0; t < 3; ++t) c[t] = fr_pow(gk[t], n);
        // inverse of V[k1][m] = c_k1^m by Lagrange basis polynomials: column k1 of V^-1 holds the coefficients of
        // L_k1(X) = prod_{j != k1} (X - c_j) / (c_k1 - c_j)
        for (int k1 = 0; k1 < 3; ++k1) {
          const Fr &a = c[(k1 + 1) % 3], &b = c[(k1 + 2) % 3];
          const Fr den = fr_inv((c[k1] - a) * (c[k1] - b));
          vinv.v[0 * 3 + k1] = a * b * den;              // X^0
          vinv.v[1 * 3 + k1] = (Fr::zero() - (a + b)) * den;  // X^1
          vinv.v[2 * 3 + k1] = den;                      // X^2
        }
      }
      zkp::k_ext3_combine<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(rows3, pw, vinv, n, ws->h_c.fr());
      ZK_LAUNCH_CHECK(ctx);
    }
  }
  std::vector<AffinePoint> h_commit;
  CK(commit_cols_out(ctx, srs, srs->g, ws->h_c.fr(), 3, ws, h_commit));
  if (q_rows == 4) {
    // the quotient must have degree < 3n: a non-zero top quarter means a violated constraint
    std::vector<U256> top(8);
    CK(zkfhe_download(ctx, top.data(), ws->h_c.fr() + 3 * n, 8 * 32));
    for (const auto &v : top)
      if (!v.is_zero()) return zk_fail_msg(ctx, ZKFHE_EINVAL, "quotient degree too high: a constraint is violated");
  }
  gate.leave();
  for (const auto &p : h_commit) tr.write_point(p);
  const U256 x_c = tr.squeeze();
  const Fr x = mont(x_c);
  const double t_quot = now_ms();
  trace.mark("quotient");
  // ------------------------------------------------------------ evaluations at x * w^rot (barycentric, Lagrange form)
  const Fr xn = fr_pow(x, n);
  // rotation ids: 0,1,2,3 -> w^r ; 4 -> w^u ("last") ; 5 -> w^-1
  Fr pts_rot[6];
  {
    const Fr w = dom->omega;
    pts_rot[0] = x;
    pts_rot[1] = x * w;
    pts_rot[2] = pts_rot[1] * w;
    pts_rot[3] = pts_rot[2] * w;
    pts_rot[4] = x * fr_pow(w, u);
    pts_rot[5] = x * dom->omega_inv;
  }
  Fr *pts_dev = nullptr;
  {
    // H(X) = h0 + x^n h1 + x^2n h2, and the random polynomial, in Lagrange form
    const Fr sc[3] = {zk::zk_fr_to_29(Fr::one()), zk::zk_fr_to_29(xn), zk::zk_fr_to_29(xn * xn)};   // k_lincomb_ptrs takes its scalars in the 2^261 form
    const Fr *ptrs[3] = {ws->h_c.fr(), ws->h_c.fr() + n, ws->h_c.fr() + 2 * n};
    STAGE(ptrs_dev, const Fr *, ws, ptrs, sizeof(ptrs));
    STAGE(sc_dev, Fr, ws, sc, sizeof(sc));
    STAGE(pts_stage, Fr, ws, pts_rot, sizeof(pts_rot));   // the six evaluation points go up with the same copy
    pts_dev = pts_stage;
    CK(flush_staged(ctx, ws));
    zkp::k_lincomb_ptrs<<<grid_for(ctx, n), 256, 0, ctx->stream>>>(ptrs_dev, sc_dev, 3, n, H_c);
    ZK_LAUNCH_CHECK(ctx);
    CK(zkfhe_ntt_batch_to(ctx, (const zkfhe_fr *)H_c, (zkfhe_fr *)H_l, 1, (int)k, 0));
    if (!early_rand) CK(zkfhe_ntt_batch_to(ctx, (const zkfhe_fr *)rand_c, (zkfhe_fr *)rand_l, 1, (int)k, 0));   // early: transformed on the auxiliary stream at the start
  }
  Fr *bw = ws->misc.fr();  // [6][n] barycentric weights
  {
    zkp::k_bary_den<<<grid_for(ctx, 6 * n), 256, 0, ctx->stream>>>(dom->fwd, pts_dev, 6, n, bw);
    ZK_LAUNCH_CHECK(ctx);
    CK(zkfhe_fr_batch_invert(ctx, (zkfhe_fr *)bw, 6 * n));
    const Fr c = zk::zk_fr_to_29((xn - Fr::one()) * dom->n_inv);   // the weights in the 2^261 form: the constant operand of k_eval_jobs' nine-limb products
    zkp::k_bary_weights<<<grid_for(ctx, 6 * n), 256, 0, ctx->stream>>>(dom->fwd, c, 6 * n, n, bw);
    ZK_LAUNCH_CHECK(ctx);
  }
  std::vector<OpenItem> items;
  auto add_item = [&](const Fr *lagr, std::initializer_list<int> rots) {
    OpenItem it;
    it.lagr = lagr;
    it.n_rot = 0;
    for (int r : rots) it.rot[it.n_rot++] = r;
    items.push_back(it);
  };
  for (unsigned c = 0; c < cfg.n_advice(); ++c) {
    const Fr *p = ws->adv_l.fr() + (size_t)c * n;
    if (c < cfg.n_gate()) add_item(p, {0, 1, 2, 3});
    else if (c < cfg.adv_rlc0()) add_item(p, {0});
    else add_item(p, {0, 1, 2});
  }
  for (unsigned c = 0; c < cfg.n_fixed(); ++c) add_item(pk->fixed_l.fr() + (size_t)c * n, {0});
  const size_t idx_H = items.size();
  add_item(H_l, {0});
  add_item(rand_l, {0});
  for (unsigned c = 0; c < cfg.n_perm(); ++c) add_item(pk->sigma_l.fr() + (size_t)c * n, {0});
  for (unsigned j = 0; j < nch; ++j) {
    if (j + 1 != nch) add_item(ws->pz_l.fr() + (size_t)j * n, {0, 1, 4});
    else add_item(ws->pz_l.fr() + (size_t)j * n, {0, 1});
  }
  for (unsigned i = 0; i < cfg.n_lookup; ++i) {
    add_item(ws->lz_l.fr() + (size_t)i * n, {0, 1});
    add_item(ws->la_l.fr() + (size_t)i * n, {0, 5});
    add_item(ws->ls_l.fr() + (size_t)i * n, {0});
  }
  {
    std::vector<zkp::EvalJob> jobs(items.size());
    for (size_t i = 0; i < items.size(); ++i) {
      jobs[i].col = items[i].lagr;
      jobs[i].n_rot = items[i].n_rot;
      for (int r = 0; r < 4; ++r) jobs[i].rot[r] = r < items[i].n_rot ? items[i].rot[r] : 0;
    }
    DevBuf &od = ws->evout;
    STAGE(jobs_dev, zkp::EvalJob, ws, jobs.data(), jobs.size() * sizeof(zkp::EvalJob));
    CK(flush_staged(ctx, ws));
    struct { const void *p; } jd{jobs_dev};
    // One proof over several GPUs, long rows (k >= 14): the evaluation jobs are sharded by index -- rank r takes jobs
    // [per r, per (r + 1)) with per = ceil(J / W), every Lagrange column being present on every rank -- and the scalars are
    // all-gathered (per * 128 bytes per rank).  Short rows are not worth a collective.
    const bool shard_open = W_sh > 1 && k >= 14;
    const size_t per_rank = shard_open ? (jobs.size() + W_sh - 1) / W_sh : jobs.size();
    const size_t j_lo = shard_open ? std::min(jobs.size(), per_rank * r_sh) : 0, j_hi = shard_open ? std::min(jobs.size(), j_lo + per_rank) : jobs.size();
    if (shard_open && (per_rank * W_sh * 4 * 32 > od.bytes || per_rank * (W_sh + 1) * 4 * 32 > ws->h_ext.bytes))
      return zk_fail_msg(ctx, ZKFHE_EINVAL, "too many ranks for the evaluation gather buffers");
    Fr *ev_dst = shard_open ? ws->h_ext.fr() : od.fr();   // my slice first (the quotient's extended values are dead by now)
    // long columns: 16 row slices per job (partial sums in the dead quotient buffer), then one small reduction
    const unsigned slices = (n > 32768 && (size_t)16 * jobs.size() * 4 * 32 <= ws->partials.bytes) ? 16u : 1u;
    if (j_hi > j_lo) {
      const zkp::EvalJob *jp = (const zkp::EvalJob *)jd.p + j_lo;
      const unsigned nj = (unsigned)(j_hi - j_lo);
      if (slices == 1) {
        zkp::k_eval_jobs<<<nj, 256, 0, ctx->stream>>>(jp, bw, n, ev_dst + (shard_open ? 0 : j_lo * 4));
      } else {
        zkp::k_eval_jobs<<<dim3(nj, slices), 256, 0, ctx->stream>>>(jp, bw, n, ws->partials.fr());
        ZK_LAUNCH_CHECK(ctx);
        zkp::k_sum_rows<<<grid_for(ctx, (size_t)nj * 4), 256, 0, ctx->stream>>>(ws->partials.fr(), slices, (size_t)nj * 4, ev_dst + (shard_open ? 0 : j_lo * 4));
      }
      ZK_LAUNCH_CHECK(ctx);
    }
    if (shard_open) CK(zkfhe_comm_all_gather(ctx, srs->comm, ev_dst, od.p, per_rank * 4 * 32));   // rank-major = job order
    if (jobs.size() * 4 > ws->out_ev_cap) return zk_fail_msg(ctx, ZKFHE_EINVAL, "more evaluations than the result block holds");
    // canonical values straight into pinned memory (the conversion kernel stores them there): no copy command
    CK(zkfhe_fr_from_mont(ctx, (const zkfhe_fr *)od.p, (zkfhe_fr *)ws->out_ev(), jobs.size() * 4));
    ZK_HIP(ctx, zk_wait(ctx));
    trace.mark("evaluations (GPU)");
    const U256 *ev = ws->out_ev();
    for (size_t i = 0; i < items.size(); ++i)
      for (int r = 0; r < items[i].n_rot; ++r) items[i].ev[r] = ev[i * 4 + r];
  }
  const OpenLayout layout(cfg);
  if (items.size() != layout.count || idx_H != layout.H) return zk_fail_msg(ctx, ZKFHE_EINVAL, "opening layout mismatch");
  {
    std::vector<U256> evs;
    evs.reserve(items.size() * 2);
    for (size_t i = 0; i < items.size(); ++i) {
      if (i == idx_H) continue;  // implied by the identity, not written
      for (int r = 0; r < items[i].n_rot; ++r) evs.push_back(items[i].ev[r]);
    }
    tr.write_scalars(evs);
  }
  // ------------------------------------------------------------ SHPLONK (halo2 ProverSHPLONK; Lagrange form, commitments are basis independent)
  const Fr yq = mont(tr.squeeze());
  trace.mark("transcript: evaluations");
  std::vector<int> all_rots;
  const std::vector<OpenSet> sets = intermediate_sets(layout, open_queries(cfg, layout), all_rots);
  const size_t ns = sets.size();
  Fr *F = ws->misc.fr() + 12 * n;  // [ns][n]
  if (ns > 8) return zk_fail_msg(ctx, ZKFHE_EINVAL, "too many rotation sets");
  std::vector<zkp::ShSet> shsets(ns);
  {
    // pointer and scalar tables of every set are staged first and go up with one copy; then the launches
    std::vector<const Fr *const *> set_ptrs(ns);
    std::vector<const Fr *> set_pw(ns);
    for (size_t j = 0; j < ns; ++j) {
      const auto &mem = sets[j].members;
      std::vector<const Fr *> ptrs(mem.size());
      std::vector<Fr> pw(mem.size());
      Fr cur = Fr::one();
      const size_t np = sets[j].rots.size();
      std::vector<Fr> comb(np, Fr::zero());
      for (size_t m = 0; m < mem.size(); ++m) {
        ptrs[m] = items[mem[m]].lagr;
        pw[m] = zk::zk_fr_to_29(cur);   // 2^261 form: the constant operand of k_lincomb_ptrs
        for (size_t t = 0; t < np; ++t) comb[t] = comb[t] + cur * mont(items[mem[m]].ev[layout.eval_slot(mem[m], sets[j].rots[t])]);
        cur = cur * yq;
      }
      STAGE(pd, const Fr *, ws, ptrs.data(), mem.size() * sizeof(void *));
      STAGE(sd, Fr, ws, pw.data(), mem.size() * 32);
      set_ptrs[j] = pd;
      set_pw[j] = sd;
      // r_j: interpolation through (pts, comb), ascending coefficients
      zkp::ShSet &S = shsets[j];
      for (int t = 0; t < 4; ++t) S.rc[t] = S.pts[t] = Fr::zero();
      S.n_pts = (int)np;
      for (size_t t = 0; t < np; ++t) S.pts[t] = pts_rot[sets[j].rots[t]];
      for (size_t i = 0; i < np; ++i) {
        std::vector<Fr> num(1, Fr::one());
        Fr den = Fr::one();
        for (size_t t = 0; t < np; ++t) {
          if (t == i) continue;
          std::vector<Fr> nxt(num.size() + 1, Fr::zero());
          for (size_t q = 0; q < num.size(); ++q) {
            nxt[q + 1] = nxt[q + 1] + num[q];
            nxt[q] = nxt[q] - S.pts[t] * num[q];
          }
          num = nxt;
          den = den * (S.pts[i] - S.pts[t]);
        }
        const Fr scl = comb[i] * fr_inv(den);
        for (size_t q = 0; q < num.size(); ++q) S.rc[q] = S.rc[q] + num[q] * scl;
      }
    }
    CK(flush_staged(ctx, ws));
    // One proof over several GPUs, long rows: every rank sums its share of a set's members (a k_lincomb_ptrs over zero members
    // writes zeros), the W partial combinations of all sets are all-gathered (ns n values per rank) and added up.
    const bool shard_sets = W_sh > 1 && k >= 14;
    if (shard_sets && ws->qgather.bytes < (size_t)W_sh * ns * n * 32) {   // [set][rank][n]: the buffer of the quotient shares, grown if there are more sets than coset rows
      CK(zkfhe_sync(ctx));
      ws->qgather.release();
      CK(ws->qgather.alloc(ctx, (size_t)W_sh * ns * n * 32));
    }
    for (size_t j = 0; j < ns; ++j) {
      const size_t all = sets[j].members.size();
      const size_t m_lo = shard_sets ? all * r_sh / W_sh : 0, m_hi = shard_sets ? all * (r_sh + 1) / W_sh : all;
      const unsigned msz = (unsigned)(m_hi - m_lo);
      const Fr *const *pp = set_ptrs[j] + m_lo;
      const Fr *pw = set_pw[j] + m_lo;
      const unsigned per = 48, chunks = (msz + per - 1) / per;
      if (chunks > 1 && (size_t)chunks * n * 32 <= ws->partials.bytes) {
        dim3 lg(grid_for(ctx, n), chunks);
        zkp::k_lincomb_ptrs_chunked<<<lg, 256, 0, ctx->stream>>>(pp, pw, msz, per, n, ws->partials.fr());
        ZK_LAUNCH_CHECK(ctx);
        zkp::k_sum_rows<<<grid_for(ctx, n), 256, 0, ctx->stream>>>(ws->partials.fr(), chunks, n, F + j * n);
      } else {
        zkp::k_lincomb_ptrs<<<grid_for(ctx, n), 256, 0, ctx->stream>>>(pp, pw, msz, n, F + j * n);
      }
      ZK_LAUNCH_CHECK(ctx);
      // set j's share goes onto the communicator's stream as soon as it is formed (16 MB per rank at k = 19) and travels while the
      // next set is combined; the sums wait for the join below
      if (shard_sets) CK(zkfhe_comm_all_gather_async(ctx, srs->comm, F + j * n, ws->qgather.fr() + j * (size_t)W_sh * n, n * 32));
    }
    if (shard_sets) {
      CK(zkfhe_comm_join(ctx, srs->comm, 0));
      for (size_t j = 0; j < ns; ++j) {
        zkp::k_sum_rows<<<grid_for(ctx, n), 256, 0, ctx->stream>>>(ws->qgather.fr() + j * (size_t)W_sh * n, W_sh, n, F + j * n);
        ZK_LAUNCH_CHECK(ctx);
      }
    }
  }
  const Fr v = mont(tr.squeeze());
  {
    Fr cur = Fr::one();
    for (size_t j = 0; j < ns; ++j) {
      shsets[j].vj = cur;
      shsets[j].coef = Fr::zero();
      shsets[j].r_u = Fr::zero();
      cur = cur * v;
    }
  }

  Fr *zs = ws->misc.fr() + 20 * n;   // [ns][n]
  Fr *hq = ws->misc.fr() + 28 * n;   // [n]
  Fr *Wq = ws->misc.fr() + 29 * n;   // [n]
  Fr *dinv = ws->misc.fr() + 30 * n; // [n]
  STAGE(sets_dev, zkp::ShSet, ws, shsets.data(), ns * sizeof(zkp::ShSet));
  CK(flush_staged(ctx, ws));
  zkp::k_sh_zs<<<grid_for(ctx, ns * n), 256, 0, ctx->stream>>>(sets_dev, (unsigned)ns, dom->fwd, n, zs);
  ZK_LAUNCH_CHECK(ctx);
  CK(zkfhe_fr_batch_invert(ctx, (zkfhe_fr *)zs, ns * n));
  zkp::k_sh_h<<<grid_for(ctx, n), 256, 0, ctx->stream>>>(sets_dev, (unsigned)ns, F, zs, dom->fwd, n, hq);
  ZK_LAUNCH_CHECK(ctx);
  std::vector<AffinePoint> hq_commit, w_commit;
  CK(commit_cols_out(ctx, srs, srs->g_lagrange, hq, 1, ws, hq_commit));
  tr.write_point(hq_commit[0]);
  const Fr uu = mont(tr.squeeze());
  trace.mark("shplonk h (GPU)");
  {
    Fr ztu = Fr::one();
    for (int r : all_rots) ztu = ztu * (uu - pts_rot[r]);
    // halo2 normalises the final quotient by the difference vanishing polynomial of the first set ("z_0_diff_inv"): every
    // coefficient below carries that factor, so the kernel is unchanged
    Fr z0_diff_inv = Fr::one();
    for (size_t j = 0; j < ns; ++j) {
      Fr zdiff = Fr::one();
      for (int r : all_rots)
        if (std::find(sets[j].rots.begin(), sets[j].rots.end(), r) == sets[j].rots.end()) zdiff = zdiff * (uu - pts_rot[r]);
      if (j == 0) z0_diff_inv = fr_inv(zdiff);
      shsets[j].coef = shsets[j].vj * zdiff * z0_diff_inv;
      Fr ru = shsets[j].rc[3];
      ru = ru * uu + shsets[j].rc[2];
      ru = ru * uu + shsets[j].rc[1];
      ru = ru * uu + shsets[j].rc[0];
      shsets[j].r_u = ru;
    }
    ztu = ztu * z0_diff_inv;
    STAGE(sets2_dev, zkp::ShSet, ws, shsets.data(), ns * sizeof(zkp::ShSet));
    CK(flush_staged(ctx, ws));
    zkp::k_sh_den<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(dom->fwd, uu, n, dinv);
    ZK_LAUNCH_CHECK(ctx);
    CK(zkfhe_fr_batch_invert(ctx, (zkfhe_fr *)dinv, n));
    zkp::k_sh_w<<<grid_for(ctx, n), 256, 0, ctx->stream>>>(sets2_dev, (unsigned)ns, F, hq, ztu, dinv, n, Wq);
    ZK_LAUNCH_CHECK(ctx);
  }
  CK(commit_cols_out(ctx, srs, srs->g_lagrange, Wq, 1, ws, w_commit));
  tr.write_point(w_commit[0]);
  proof = tr.out;
  const double t_end = now_ms();
  trace.mark("evaluations + shplonk");
  ctx->proof_marks[2] = (float)(t_end - t_start);
  if (timings) {
    timings[0] = (float)(t_wit - t_start);
    timings[1] = (float)(t_commit - t_wit);
    timings[2] = (float)(t_quot - t_commit);
    timings[3] = (float)(t_end - t_quot);
    timings[4] = (float)(t_end - t_start);
  }
  return ZKFHE_OK;
}

}  // namespace

extern "C" {

int zkfhe_bfv_prove(zkfhe_ctx *ctx, const zkfhe_srs *srs, const zkfhe_bfv_pk *pk, const char *input_json, const uint8_t seed[32],
                    uint8_t *proof_out, size_t proof_cap, size_t *proof_len, uint8_t *instances_out, size_t *n_instances, float *timings_ms) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, srs && pk && input_json && seed && proof_out && proof_len);
  try {
    pos::BulkClientScope in_flight;   // transcript.hpp bulk_ok(): the hash service is shared by the provers in flight
    std::vector<uint8_t> proof;
    std::vector<U256> inst;
    int rc = prove_impl(ctx, srs, const_cast<zkfhe_bfv_pk *>(pk), input_json, seed, proof, inst, timings_ms);
    if (rc) return rc;
    if (proof.size() > proof_cap) return zk_fail_msg(ctx, ZKFHE_EINVAL, "proof buffer too small");
    memcpy(proof_out, proof.data(), proof.size());
    *proof_len = proof.size();
    if (n_instances) {
      const size_t have = *n_instances;
      *n_instances = inst.size();
      if (instances_out) {
        if (have < inst.size()) return zk_fail_msg(ctx, ZKFHE_EINVAL, "instance buffer too small (the required count is returned in *n_instances)");
        memcpy(instances_out, inst.data(), inst.size() * 32);
      }
    }
    return ZKFHE_OK;
  } catch (const std::exception &e) {
    return zk_fail_msg(ctx, ZKFHE_EINVAL, e.what());
  }
}

// Admission gate of the GPU-heavy middle of the proofs of this process (HeavyGate above): n > 0 = that many proofs inside at once,
// 0 = no gate, negative = query.  Returns the previous setting.  For a service that keeps its streams full (measured: 16 streams,
// 4 slots: 224 proofs/s against 209); a batch that starts and ends together gains nothing from it.
int zkfhe_prover_gate(int n) { return HeavyGate::get().set(n); }

// halo2's permute_expression_pair for the 8-bit range table, as the prover runs it (gpu_witness.hip.hpp k_lookup_permute): the
// parity hook of SURVEY.md section 8a row P4.
int zkfhe_lookup_permute(zkfhe_ctx *ctx, const zkfhe_fr *cols_dev, size_t n_cols, size_t n, uint32_t usable_rows, zkfhe_fr *a_dev, zkfhe_fr *s_dev, int *not_in_table) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, cols_dev && a_dev && s_dev && not_in_table && n_cols > 0 && n_cols < 65536 && usable_rows <= n && usable_rows >= 256);
  void *flag;   // the flag, then the histogram of the sliced kernels (long columns)
  CK(zk_scratch(ctx, 3, 64 + n_cols * 256 * sizeof(unsigned), &flag));
  ZK_HIP(ctx, hipMemsetAsync(flag, 0, 4, ctx->stream));
  CK(zkw::lookup_permute(ctx, (const Fr *)cols_dev, n, usable_rows, (unsigned)n_cols, (Fr *)a_dev, (Fr *)s_dev, (int *)flag, (unsigned *)((char *)flag + 64)));
  return zkfhe_download(ctx, not_in_table, flag, 4);
}

// Announce the input of a proof that will be made later with this key: its public inputs (examples/bfv.rs:118-122) are parsed now
// and `vk digest | public inputs` is absorbed on a helper thread into a transcript state that the zkfhe_bfv_prove of the SAME input
// picks up (prefix_cache.hpp PreHash; one-shot).  Host only: no context, no GPU work.  started / taken / pending (optional) receive
// the counters of this key; input_json == NULL only queries them.
int zkfhe_bfv_pk_prehash(const zkfhe_bfv_pk *pk_c, const char *input_json, uint64_t *started, uint64_t *taken, uint64_t *pending) {
  if (!pk_c) return ZKFHE_EINVAL;
  zkfhe_bfv_pk *pk = const_cast<zkfhe_bfv_pk *>(pk_c);
  int rc = ZKFHE_OK;
  if (input_json) {
    const uint32_t kind = pk->cfg.transcript;
    const U256 digest = pk->vk_digest;
    const BfvParams prm = pk->prm;
    PrefixCache *cache = &pk->prefix;
    // everything on the helper thread: the caller gets its thread back after one copy of the text (2.3 MB of JSON at N = 16384 take 15 ms to parse)
    const bool ok = pk->prehash.start(input_json, strlen(input_json), [kind, digest, prm, cache](const std::string &text, std::vector<U256> &inst, Transcript::State &out) {
      struct PublicInputsKnown {};   // thrown by the callback: phase 0 stops as soon as the public inputs are complete
      try {
        Context ctx0(CTX_PHASE0, false, false);
        std::vector<Cell> make_public;
        BfvState st;
        const auto on_public = [&](const std::vector<Cell> &pub) {
          inst.reserve(pub.size());
          for (const Cell &c : pub) inst.push_back(c.value);
          throw PublicInputsKnown{};
        };
        if (!bfv_phase0_fast(ctx0, text.c_str(), text.size(), prm, make_public, st, on_public))
          (void)bfv_phase0(ctx0, CircuitInput::parse_json(text.c_str()), prm, make_public, on_public);
        throw std::runtime_error("phase 0 ended without public inputs");
      } catch (const PublicInputsKnown &) {
      }
      const size_t n_key = std::min<size_t>(inst.size(), 2 * (size_t)prm.N);
      Transcript tr(kind);
      tr.common_scalar(digest);
      Transcript::State mid;
      size_t from = 0;
      if (n_key && cache->capacity()) {
        if (cache->lookup(inst.data(), n_key, mid)) {
          tr.restore(mid);
        } else {
          tr.common_scalars_async(std::vector<U256>(inst.begin(), inst.begin() + (long)n_key));
          mid = tr.snapshot();
          cache->insert(inst.data(), n_key, mid);
        }
        from = n_key;
      }
      tr.common_scalars_async(std::vector<U256>(inst.begin() + (long)from, inst.end()));
      out = tr.snapshot();
    });
    if (!ok) rc = ZKFHE_EINVAL;   // PreHash::MAX_PENDING announced proofs are waiting already
  }
  pk->prehash.stats(started, taken, pending);
  return rc;
}

// The phase-1 gate stream exactly as the GPU witness generator produces it (gpu_witness.hip.hpp), for a caller-chosen
// challenge: the cells of examples/bfv.rs:171-301 in halo2-base order, canonical values.  This is the parity hook of
// tests compare it cell for cell with the oracle's restatement of src/poly_chip.rs.
int zkfhe_bfv_witness_stream(zkfhe_ctx *ctx, const zkfhe_bfv_pk *pk_c, const char *input_json, const uint8_t gamma_le[32], uint8_t *cells_out,
                             size_t cap_cells, size_t *n_cells) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, pk_c && input_json && gamma_le && n_cells);
  zkfhe_bfv_pk *pk = const_cast<zkfhe_bfv_pk *>(pk_c);
  *n_cells = pk->gate1_cells;
  if (!cells_out) return ZKFHE_OK;
  ZK_ARG(ctx, cap_cells >= pk->gate1_cells);
  try {
    Workspace *ws;
    CK(get_workspace(ctx, pk, &ws));
    CK(zkfhe_sync(ctx));
    GpuPolyMul gpu_mul(ctx, ws);
    struct BackendGuard {
      explicit BackendGuard(PolyMulBackend *b) { poly_mul_backend() = b; }
      ~BackendGuard() { poly_mul_backend() = nullptr; }
    } guard(&gpu_mul);
    Context ctx0(CTX_PHASE0, false, false), ctx_rlc(CTX_RLC1, true, false);
    std::vector<Cell> make_public;
    BfvState st;
    if (!bfv_phase0_fast(ctx0, input_json, strlen(input_json), pk->prm, make_public, st)) st = bfv_phase0(ctx0, CircuitInput::parse_json(input_json), pk->prm, make_public);
    U256 gamma, evals[12];
    memcpy(gamma.l, gamma_le, 32);
    if (!(gamma < fe::MOD)) return zk_fail_msg(ctx, ZKFHE_EINVAL, "gamma is not a canonical Fr value");
    CK(alloc_witness_buffers(ctx, pk, ws));
    ws->ring_off = ws->ring_flushed = 0;   // the stream is idle (synchronised above): the staging ring starts over, as in a proof
    GpuPhase1 g1(ctx, pk, ws);
    CK(g1.launch(st));
    bfv_phase1_rlc(st, ctx_rlc, gamma, evals);
    CK(g1.finish(evals));
    CK(zkfhe_fr_from_mont(ctx, (const zkfhe_fr *)ws->stream.p, (zkfhe_fr *)ws->stream.p, pk->gate1_cells));
    ZK_HIP(ctx, hipMemcpyAsync(cells_out, ws->stream.p, pk->gate1_cells * 32, hipMemcpyDeviceToHost, ctx->stream));
    CK(zkfhe_sync(ctx));
    return ZKFHE_OK;
  } catch (const std::exception &e) {
    return zk_fail_msg(ctx, ZKFHE_EINVAL, e.what());
  }
}

}  // extern "C"
