// C ABI of the host-side circuit layer (include/zkfhe.h "BFV circuit"): witness tables without a GPU.
#include <array>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>

#include "../../include/zkfhe.h"
#include "bfv_circuit.hpp"
#include "bfv_phase0_fast.hpp"
#include "srs_secret.hpp"
#include "transcript.hpp"

using namespace zkhost;

struct zkfhe_transcript {
  Transcript tr;
  explicit zkfhe_transcript(uint32_t kind) : tr(kind) {}
};

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
  cfg.transcript = c->transcript;
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
    const bool kg = keygen_mode != 0;
    Context ctx0(CTX_PHASE0, false, kg), ctx_gate(CTX_GATE1, false, kg), ctx_rlc(CTX_RLC1, true, kg);
    std::vector<Cell> make_public;
    // prover-mode tables take the prover's phase 0 (machine words where the input allows, ZKFHE_PHASE0=generic forces the
    // restatement): tests/test_host_witness.py compares the two table for table
    BfvState st;
    if (kg || !bfv_phase0_fast(ctx0, input_json, strlen(input_json), prm, make_public, st)) st = bfv_phase0(ctx0, CircuitInput::parse_json(input_json), prm, make_public);
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

// ---- MockProver (README.md:18-22: `mock` = halo2 MockProver::run(..).assert_satisfied()): every constraint of the
// constraint system is evaluated on every row of the assigned table -- the gate a + b*c = d under each gate selector, the
// RLC gate a*gamma + b = c under each RLC selector, membership of every lookup cell in the table column, and equality of
// the two cells of every copy constraint (break-point duplicates, constants, lookups, public inputs).
int zkfhe_bfv_tables_poke_advice(zkfhe_bfv_tables *t, uint32_t column, uint32_t row, const uint8_t value_le[32]) {
  if (!t || !value_le || column >= t->t.advice.size() || row >= t->cfg.n()) return ZKFHE_EINVAL;
  U256 v;
  memcpy(v.l, value_le, 32);
  if (!(v < fe::MOD)) return ZKFHE_EINVAL;
  t->t.advice[column][row] = v;
  return ZKFHE_OK;
}

static int mock_check_impl(const zkfhe_bfv_tables *t, const uint8_t gamma_le[32], uint64_t *n_failures, char *err, size_t err_len) {
  if (!t || !gamma_le || !n_failures) return ZKFHE_EINVAL;
  const CircuitConfig &cfg = t->cfg;
  if (t->t.fixed.size() != cfg.n_fixed()) {
    if (err && err_len) snprintf(err, err_len, "tables were not built in keygen mode (no fixed columns / copy constraints)");
    return ZKFHE_EINVAL;
  }
  U256 gamma;
  memcpy(gamma.l, gamma_le, 32);
  if (!(gamma < fe::MOD)) return ZKFHE_EINVAL;
  const size_t n = cfg.n(), usable = cfg.u();
  uint64_t fails = 0;
  std::string first;
  auto fail = [&](const std::string &what) {
    if (!fails) first = what;
    ++fails;
  };
  auto at = [](const char *kind, size_t col, size_t row) { return std::string(kind) + " column " + std::to_string(col) + ", row " + std::to_string(row); };
  // gates: q(X) * (a(X) + a(wX) a(w^2 X) - a(w^3 X)) on every usable row (rotations wrap like the polynomial identity does)
  for (unsigned j = 0; j < cfg.n_gate(); ++j) {
    const U256 *a = t->t.advice[j];
    const std::vector<U256> &q = t->t.fixed[j];
    for (size_t r = 0; r < usable; ++r) {
      if (q[r].is_zero()) continue;
      const U256 lhs = fe::add(a[r], fe::mul(a[(r + 1) % n], a[(r + 2) % n]));
      if (!(fe::mul(q[r], fe::sub(lhs, a[(r + 3) % n])).is_zero())) fail("gate a + b*c = d not satisfied at " + at("gate", j, r));
    }
  }
  for (unsigned j = 0; j < cfg.n_rlc; ++j) {
    const U256 *a = t->t.advice[cfg.adv_rlc0() + j];
    const std::vector<U256> &q = t->t.fixed[cfg.fix_qrlc0() + j];
    for (size_t r = 0; r < usable; ++r) {
      if (q[r].is_zero()) continue;
      const U256 lhs = fe::add(fe::mul(a[r], gamma), a[(r + 1) % n]);
      if (!(fe::mul(q[r], fe::sub(lhs, a[(r + 2) % n])).is_zero())) fail("RLC gate a*gamma + b = c not satisfied at " + at("rlc", j, r));
    }
  }
  // lookups: every usable row of a lookup column holds a value of the table column
  const uint64_t table_size = (uint64_t)1 << cfg.lookup_bits;
  for (unsigned i = 0; i < cfg.n_lookup; ++i) {
    const U256 *a = t->t.advice[cfg.adv_lookup0() + i];
    for (size_t r = 0; r < usable; ++r)
      if (a[r].l[1] | a[r].l[2] | a[r].l[3] || a[r].l[0] >= table_size) fail("lookup input not in the table at " + at("lookup", i, r));
  }
  // copy constraints (permutation argument)
  auto value = [&](uint64_t id) -> U256 {
    const size_t col = id / n, row = id % n;
    if (col < cfg.n_advice()) return t->t.advice[col][row];
    if (col == cfg.perm_const()) return t->t.fixed[cfg.fix_const()][row];
    return row < t->t.instance.size() ? t->t.instance[row] : fe::zero();
  };
  for (const auto &cp : t->t.copies)
    if (!(value(cp.first) == value(cp.second)))
      fail("copy constraint violated between " + at("permutation", cp.first / n, cp.first % n) + " and " + at("permutation", cp.second / n, cp.second % n));
  *n_failures = fails;
  if (err && err_len) snprintf(err, err_len, "%s", first.c_str());
  return ZKFHE_OK;
}

// ---- Fiat-Shamir transcript and the Poseidon instance behind it (transcript.hpp, poseidon.hpp)
namespace {
bool load_fr(const uint8_t b[32], U256 &v) {
  memcpy(v.l, b, 32);
  return v < fe::MOD;
}
bool load_point(const uint8_t b[64], AffinePoint &p) {
  static const U256 QMOD = {{0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}};
  memcpy(p.x.l, b, 32);
  memcpy(p.y.l, b + 32, 32);
  return p.x < QMOD && p.y < QMOD;
}
}  // namespace

int zkfhe_bfv_mock_check(const zkfhe_bfv_tables *t, const uint8_t gamma_le[32], uint64_t *n_failures, char *err, size_t err_len) {
  try {   // strings and vectors are built on the way: nothing may propagate through the C ABI
    return mock_check_impl(t, gamma_le, n_failures, err, err_len);
  } catch (const std::bad_alloc &) {
    return ZKFHE_ENOMEM;
  } catch (const std::exception &e) {
    if (err && err_len) snprintf(err, err_len, "%s", e.what());
    return ZKFHE_EINVAL;
  }
}

int zkfhe_transcript_create(uint32_t kind, zkfhe_transcript **out) {
  if (!out || (kind != ZKFHE_TRANSCRIPT_POSEIDON && kind != ZKFHE_TRANSCRIPT_BLAKE2B)) return ZKFHE_EINVAL;
  try {
    *out = new zkfhe_transcript(kind);
    return ZKFHE_OK;
  } catch (const std::exception &) {
    return ZKFHE_ENOMEM;
  }
}
void zkfhe_transcript_destroy(zkfhe_transcript *t) { delete t; }

#define ZK_TR_GUARD(body)              \
  try {                                \
    body;                              \
    return ZKFHE_OK;                   \
  } catch (const std::exception &) {   \
    return ZKFHE_EINVAL;               \
  }

int zkfhe_transcript_common_scalar(zkfhe_transcript *t, const uint8_t s_le[32]) {
  U256 v;
  if (!t || !s_le || !load_fr(s_le, v)) return ZKFHE_EINVAL;
  ZK_TR_GUARD(t->tr.common_scalar(v));
}
int zkfhe_transcript_write_scalar(zkfhe_transcript *t, const uint8_t s_le[32]) {
  U256 v;
  if (!t || !s_le || !load_fr(s_le, v)) return ZKFHE_EINVAL;
  ZK_TR_GUARD(t->tr.write_scalar(v));
}
int zkfhe_transcript_common_point(zkfhe_transcript *t, const uint8_t xy_le[64]) {
  AffinePoint p;
  if (!t || !xy_le || !load_point(xy_le, p)) return ZKFHE_EINVAL;
  ZK_TR_GUARD(t->tr.common_point(p));   // Poseidon refuses the identity (snark-verifier does)
}
int zkfhe_transcript_write_point(zkfhe_transcript *t, const uint8_t xy_le[64]) {
  AffinePoint p;
  if (!t || !xy_le || !load_point(xy_le, p)) return ZKFHE_EINVAL;
  ZK_TR_GUARD(t->tr.write_point(p));
}
int zkfhe_transcript_squeeze(zkfhe_transcript *t, uint8_t challenge_le[32]) {
  if (!t || !challenge_le) return ZKFHE_EINVAL;
  ZK_TR_GUARD({
    const U256 c = t->tr.squeeze();
    memcpy(challenge_le, c.l, 32);
  });
}
int zkfhe_transcript_bytes(const zkfhe_transcript *t, uint8_t *out, size_t cap, size_t *len) {
  if (!t || !len) return ZKFHE_EINVAL;
  *len = t->tr.out.size();
  if (!out) return ZKFHE_OK;
  if (cap < t->tr.out.size()) return ZKFHE_EINVAL;
  memcpy(out, t->tr.out.data(), t->tr.out.size());
  return ZKFHE_OK;
}

int zkfhe_host_poly_mul_u32(const uint64_t *a, const uint64_t *b, size_t n, uint64_t *lo, uint64_t *hi) {
  if (!a || !b || !lo || !hi) return ZKFHE_EINVAL;
  if (n < 2 || n > 2048 || (n & (n - 1))) return ZKFHE_EINVAL;   // before a single element is read
  try {
    const std::vector<uint64_t> va(a, a + n), vb(b, b + n);
    if (!gl::fits(va, vb)) return ZKFHE_EINVAL;
    std::vector<uint64_t> l, h;
    gl::poly_mul_u32(va, vb, l, h);
    memcpy(lo, l.data(), l.size() * 8);
    memcpy(hi, h.data(), h.size() * 8);
  } catch (const std::bad_alloc &) {
    return ZKFHE_ENOMEM;
  } catch (const std::exception &) {
    return ZKFHE_EINVAL;
  }
  return ZKFHE_OK;
}

int zkfhe_poseidon_permute(uint8_t state_le[96]) {
  if (!state_le) return ZKFHE_EINVAL;
  pos::F s[3];
  for (int i = 0; i < 3; ++i) {
    U256 v;
    if (!load_fr(state_le + 32 * i, v)) return ZKFHE_EINVAL;
    s[i] = pos::from_canon(v);
  }
  pos::permute(s);
  for (int i = 0; i < 3; ++i) {
    const U256 v = pos::to_canon(s[i]);
    memcpy(state_le + 32 * i, v.l, 32);
  }
  return ZKFHE_OK;
}
int zkfhe_snark_encode(const uint8_t *instances_le, size_t n_instances, const uint8_t *proof, size_t proof_len, uint8_t *out, size_t cap, size_t *len) try {
  if (!len || (!instances_le && n_instances) || (!proof && proof_len)) return ZKFHE_EINVAL;
  const size_t need = 16 + 8 + 8 + 32 * n_instances + 8 + proof_len;
  *len = need;
  if (!out || cap < need) return (out && cap) ? ZKFHE_EINVAL : ZKFHE_OK;
  uint8_t *p = out;
  auto put64 = [&](uint64_t v) {
    memcpy(p, &v, 8);
    p += 8;
  };
  memcpy(p, "ZKFHESN2", 8);
  p += 8;
  put64(0);             // protocol: absent
  put64(1);             // instances: one column ...
  put64(n_instances);   // ... of n values, each the raw Montgomery limbs
  for (size_t i = 0; i < n_instances; ++i) {
    U256 v;
    if (!load_fr(instances_le + 32 * i, v)) return ZKFHE_EINVAL;
    const pos::F m = pos::from_canon(v);
    memcpy(p, m.l, 32);
    p += 32;
  }
  put64(proof_len);
  if (proof_len) memcpy(p, proof, proof_len);
  return ZKFHE_OK;
} catch (const std::exception &) {
  return ZKFHE_EINVAL;
}
int zkfhe_snark_decode(const uint8_t *snark, size_t snark_len, uint8_t *instances_le, size_t *n_instances, uint8_t *proof, size_t *proof_len) try {
  if (!snark || snark_len < 16) return ZKFHE_EINVAL;
  auto get64 = [&](size_t off) {
    uint64_t v;
    memcpy(&v, snark + off, 8);
    return v;
  };
  size_t n = 0, ioff = 0, poff = 0, plen = 0;
  bool mont_form = false;
  if (!memcmp(snark, "ZKFHESN1", 8)) {
    n = get64(8);
    if (n > (snark_len - 16) / 32) return ZKFHE_EINVAL;
    ioff = 16, poff = 16 + 32 * n, plen = snark_len - poff;
  } else if (!memcmp(snark, "ZKFHESN2", 8)) {
    const uint64_t prot = get64(8);
    if (prot > snark_len - 16 || snark_len - 16 - prot < 24) return ZKFHE_EINVAL;
    size_t o = 16 + prot;
    if (get64(o) != 1) return ZKFHE_EINVAL;   // one instance column
    n = get64(o + 8);
    o += 16;
    if (n > (snark_len - o) / 32 || snark_len - o - 32 * n < 8) return ZKFHE_EINVAL;
    ioff = o, o += 32 * n;
    plen = get64(o);
    poff = o + 8;
    if (plen != snark_len - poff) return ZKFHE_EINVAL;
    mont_form = true;
  } else {
    return ZKFHE_EINVAL;
  }
  if (n_instances) *n_instances = n;
  if (proof_len) *proof_len = plen;
  if (instances_le)
    for (size_t i = 0; i < n; ++i) {
      if (mont_form) {
        pos::F m;
        memcpy(m.l, snark + ioff + 32 * i, 32);
        U256 lim;
        memcpy(lim.l, m.l, 32);
        if (!(lim < fe::MOD)) return ZKFHE_EINVAL;
        const U256 c = pos::to_canon(m);
        memcpy(instances_le + 32 * i, c.l, 32);
      } else {
        memcpy(instances_le + 32 * i, snark + ioff + 32 * i, 32);
      }
    }
  if (proof && plen) memcpy(proof, snark + poff, plen);
  return ZKFHE_OK;
} catch (const std::exception &) {
  return ZKFHE_EINVAL;
}
int zkfhe_chacha20_block(const uint8_t key[32], const uint32_t counter_nonce[4], uint8_t out[64]) {
  if (!key || !counter_nonce || !out) return ZKFHE_EINVAL;
  chacha20_block(key, counter_nonce, out);
  return ZKFHE_OK;
}
int zkfhe_host_hash_mode(int mode) {
  if (mode == 0 || mode == 1) pos::hash_mode().store(mode);
  else if (mode != -1) return ZKFHE_EINVAL;
  return pos::hash_mode().load();
}
int zkfhe_poseidon_hash_many(const uint8_t *values_le, const size_t *counts, size_t n_jobs, int mode, uint8_t *digests_le) try {
  if (!counts || !digests_le || mode < 0 || mode > 2) return ZKFHE_EINVAL;
  if (mode && !pos::x8_available()) return ZKFHE_ENODEV;
  size_t total = 0;
  for (size_t j = 0; j < n_jobs; ++j) total += counts[j];
  if (total && !values_le) return ZKFHE_EINVAL;
  std::vector<U256> vals(total);
  for (size_t i = 0; i < total; ++i)
    if (!load_fr(values_le + 32 * i, vals[i])) return ZKFHE_EINVAL;
  std::vector<pos::Sponge> sp(n_jobs);
  std::vector<pos::AbsorbJob> jobs(n_jobs);
  std::vector<pos::AbsorbJob *> ptr(n_jobs);
  size_t off = 0;
  for (size_t j = 0; j < n_jobs; off += counts[j], ++j) {
    ptr[j] = &jobs[j];
    if (mode == 0) {
      for (size_t i = 0; i < counts[j]; ++i) sp[j].update(vals[off + i]);
    } else if (mode == 1) {
      sp[j].begin_bulk_prepare(vals.data() + off, counts[j], jobs[j]);
    } else {
      sp[j].begin_bulk(vals.data() + off, counts[j], jobs[j]);
    }
  }
  if (mode == 1 && !pos::x8_absorb_now(ptr.data(), n_jobs)) return ZKFHE_ENODEV;
  for (size_t j = 0; j < n_jobs; ++j) {
    if (mode) sp[j].end_bulk(jobs[j]);
    const U256 d = sp[j].squeeze();
    memcpy(digests_le + 32 * j, d.l, 32);
  }
  return ZKFHE_OK;
} catch (const std::bad_alloc &) {
  return ZKFHE_ENOMEM;
} catch (const std::exception &) {
  return ZKFHE_EINVAL;
}
int zkfhe_poseidon_constants(uint8_t *rc_le, uint8_t *mds_le) try {
  const pos::Constants &c = pos::constants();
  for (int r = 0; r < pos::ROUNDS && rc_le; ++r)
    for (int i = 0; i < pos::T; ++i) {
      const U256 v = pos::to_canon(c.rc[r][i]);
      memcpy(rc_le + 32 * (r * pos::T + i), v.l, 32);
    }
  for (int i = 0; i < pos::T && mds_le; ++i)
    for (int j = 0; j < pos::T; ++j) {
      const U256 v = pos::to_canon(c.mds[i][j]);
      memcpy(mds_le + 32 * (i * pos::T + j), v.l, 32);
    }
  return ZKFHE_OK;
} catch (const std::bad_alloc &) {
  return ZKFHE_ENOMEM;
} catch (const std::exception &) {
  return ZKFHE_EINVAL;
}

}  // extern "C"
