// C ABI of the host-side circuit layer (include/zkfhe.h "BFV circuit"): witness tables without a GPU.
#include <array>
#include <cstdio>
#include <cstring>
#include <map>

#include "../../include/zkfhe.h"
#include "bfv_circuit.hpp"

using namespace zkhost;

struct zkfhe_bfv_tables {
  CircuitConfig cfg;
  Tables t;
  size_t cells0 = 0, cells1 = 0, cells_rlc = 0, lookups = 0;
};

namespace zkhost {

CircuitConfig config_from_c(const zkfhe_bfv_config *c) {
  CircuitConfig cfg;
  cfg.k = c->k;
  cfg.n_gate0 = c->n_gate0;
  cfg.n_gate1 = c->n_gate1;
  cfg.n_lookup = c->n_lookup;
  cfg.n_rlc = c->n_rlc;
  cfg.unusable_rows = c->unusable_rows;
  cfg.lookup_bits = c->lookup_bits;
  if (c->bp_gate0) cfg.bp_gate0.assign(c->bp_gate0, c->bp_gate0 + c->n_bp_gate0);
  if (c->bp_gate1) cfg.bp_gate1.assign(c->bp_gate1, c->bp_gate1 + c->n_bp_gate1);
  if (c->bp_rlc) cfg.bp_rlc.assign(c->bp_rlc, c->bp_rlc + c->n_bp_rlc);
  return cfg;
}

BfvParams params_from_c(const zkfhe_bfv_params *p) {
  BfvParams prm;
  prm.N = (size_t)p->n;
  prm.Q = p->q;
  prm.T = p->t;
  prm.B = p->b;
  return prm;
}

}  // namespace zkhost

extern "C" {

int zkfhe_bfv_build_tables(const char *input_json, const zkfhe_bfv_params *params, const zkfhe_bfv_config *config,
                           const uint8_t gamma[32], int keygen_mode, zkfhe_bfv_tables **out, char *err, size_t err_len) {
  if (!input_json || !params || !config || !gamma || !out) return ZKFHE_EINVAL;
  *out = nullptr;
  try {
    auto *res = new zkfhe_bfv_tables();
    res->cfg = config_from_c(config);
    const BfvParams prm = params_from_c(params);
    const CircuitInput in = CircuitInput::parse_json(input_json);
    const bool kg = keygen_mode != 0;
    Context ctx0(CTX_PHASE0, false, kg), ctx_gate(CTX_GATE1, false, kg), ctx_rlc(CTX_RLC1, true, kg);
    std::vector<Cell> make_public;
    BfvState st = bfv_phase0(ctx0, in, prm, make_public);
    U256 g;
    memcpy(g.l, gamma, 32);
    bfv_phase1(st, prm, ctx_gate, ctx_rlc, g);
    Assigner as(res->cfg, kg);
    const bool replay = config->replay != 0;
    as.place(ctx0, replay);
    as.place(ctx_gate, replay);
    as.place(ctx_rlc, replay);
    as.place_lookups(ctx_gate);
    if (kg) as.finish_structure(ctx0, ctx_gate, ctx_rlc, make_public);
    for (const Cell &c : make_public) as.t.instance.push_back(c.value);
    res->t = std::move(as.t);
    if (!res->t.advice.own.empty()) res->t.advice.data = res->t.advice.own.data();
    res->cells0 = ctx0.advice.size();
    res->cells1 = ctx_gate.advice.size();
    res->cells_rlc = ctx_rlc.advice.size();
    res->lookups = ctx_gate.lookup.size();
    *out = res;
    return ZKFHE_OK;
  } catch (const std::exception &e) {
    if (err && err_len) snprintf(err, err_len, "%s", e.what());
    return ZKFHE_EINVAL;
  }
}

// halo2-base's auto-configuration (what the reference's `keygen` does before it writes configs/<name>.json, README.md:28-38):
// run the circuit once, walk the three cell streams with the break-point rule and count the columns they need at 2^k rows.
int zkfhe_bfv_auto_config(const char *input_json, const zkfhe_bfv_params *params, uint32_t k, uint32_t unusable_rows, uint32_t lookup_bits,
                          uint32_t counts_out[4], char *err, size_t err_len) {
  if (!input_json || !params || !counts_out || k < 3 || k > 24) return ZKFHE_EINVAL;
  try {
    const BfvParams prm = params_from_c(params);
    const CircuitInput in = CircuitInput::parse_json(input_json);
    Context ctx0(CTX_PHASE0, false, false), ctx_gate(CTX_GATE1, false, false), ctx_rlc(CTX_RLC1, true, false);
    std::vector<Cell> make_public;
    BfvState st = bfv_phase0(ctx0, in, prm, make_public);
    bfv_phase1(st, prm, ctx_gate, ctx_rlc, fe::zero());
    const size_t n = (size_t)1 << k;
    if (unusable_rows + 8 >= n) throw std::runtime_error("k too small for the blinding rows");
    const size_t max_rows = n - unusable_rows;
    if (((size_t)1 << lookup_bits) > max_rows) throw std::runtime_error("lookup table does not fit the rows");
    if (make_public.size() > max_rows) throw std::runtime_error("too many instances for this k");
    counts_out[0] = ctx0.advice.empty() ? 0 : place_stream(ctx0.advice.size(), ctx0.selector, max_rows, false, nullptr).n_columns;
    counts_out[1] = ctx_gate.advice.empty() ? 0 : place_stream(ctx_gate.advice.size(), ctx_gate.selector, max_rows, false, nullptr).n_columns;
    counts_out[2] = (uint32_t)((ctx_gate.lookup.size() + max_rows - 1) / max_rows);
    counts_out[3] = ctx_rlc.advice.empty() ? 0 : place_stream(ctx_rlc.advice.size(), ctx_rlc.selector, max_rows, true, nullptr).n_columns;
    return ZKFHE_OK;
  } catch (const std::exception &e) {
    if (err && err_len) snprintf(err, err_len, "%s", e.what());
    return ZKFHE_EINVAL;
  }
}

void zkfhe_bfv_tables_free(zkfhe_bfv_tables *t) { delete t; }

size_t zkfhe_bfv_tables_count(const zkfhe_bfv_tables *t, int what) {
  switch (what) {
    case 0: return t->t.advice.size();
    case 1: return t->t.fixed.size();
    case 2: return t->cfg.n();
    case 3: return t->t.instance.size();
    case 4: return t->t.copies.size();
    case 5: return t->t.bp_gate0.size();
    case 6: return t->t.bp_gate1.size();
    case 7: return t->t.bp_rlc.size();
    case 8: return t->cells0;
    case 9: return t->cells1;
    case 10: return t->cells_rlc;
    case 11: return t->lookups;
  }
  return 0;
}

int zkfhe_bfv_tables_copy_advice(const zkfhe_bfv_tables *t, uint64_t *out) {
  const size_t n = t->cfg.n();
  for (size_t c = 0; c < t->t.advice.size(); ++c) memcpy(out + c * n * 4, t->t.advice[c], n * 32);
  return ZKFHE_OK;
}
int zkfhe_bfv_tables_copy_fixed(const zkfhe_bfv_tables *t, uint64_t *out) {
  const size_t n = t->cfg.n();
  for (size_t c = 0; c < t->t.fixed.size(); ++c) memcpy(out + c * n * 4, t->t.fixed[c].data(), n * 32);
  return ZKFHE_OK;
}
int zkfhe_bfv_tables_copy_instance(const zkfhe_bfv_tables *t, uint64_t *out) {
  memcpy(out, t->t.instance.data(), t->t.instance.size() * 32);
  return ZKFHE_OK;
}
int zkfhe_bfv_tables_copy_copies(const zkfhe_bfv_tables *t, uint64_t *out) {
  for (size_t i = 0; i < t->t.copies.size(); ++i) {
    out[2 * i] = t->t.copies[i].first;
    out[2 * i + 1] = t->t.copies[i].second;
  }
  return ZKFHE_OK;
}
int zkfhe_bfv_tables_copy_break_points(const zkfhe_bfv_tables *t, int which, uint32_t *out) {
  const auto &v = which == 0 ? t->t.bp_gate0 : which == 1 ? t->t.bp_gate1 : t->t.bp_rlc;
  memcpy(out, v.data(), v.size() * 4);
  return ZKFHE_OK;
}

}  // extern "C"
