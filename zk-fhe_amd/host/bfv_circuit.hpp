// The BFV correct-encryption circuit: reference examples/bfv.rs:27-30 (parameters), :50-61 (CircuitInput),
// :63-165 (phase 0 + out-of-circuit precomputation), :171-301 (the phase-1 callback), restated on the
// host.  The operation order below fixes the cell stream, which the reference's configs/bfv.json pins.
#pragma once
#include <chrono>
#include <exception>
#include <cstdio>
#include <thread>
#include <array>
#include <cctype>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <functional>
#include <vector>

#include "poly.hpp"

namespace zkhost {

struct BfvParams {  // examples/bfv.rs:27-30 are compile-time consts upstream; a runtime parameter set here
  size_t N = 1024;
  uint64_t Q = 536870909;
  uint64_t T = 7;
  uint64_t B = 19;
};

// examples/bfv.rs:50-61 -- nine arrays of decimal strings
struct CircuitInput {
  std::vector<std::string> pk0, pk1, m, u, e0, e1, c0, c1, cyclo;

  static CircuitInput parse_json(const std::string &text) {
    // minimal parser for {"name": ["123", ...], ...}.  Pass 1 walks the object and records where each array's text lies (the
    // values are decimal strings: no escapes, no nesting); pass 2 turns every array into its strings -- one thread per array
    // once the text is long (N = 16384: 2.8 MB, 8 ms on one core).
    struct Span {
      std::string key;
      size_t lo, hi;   // text between '[' and ']'
    };
    std::vector<Span> spans;
    size_t i = 0;
    auto skip = [&]() { while (i < text.size() && isspace((unsigned char)text[i])) ++i; };
    auto expect = [&](char c) {
      skip();
      if (i >= text.size() || text[i] != c) throw std::runtime_error(std::string("input JSON: expected '") + c + "'");
      ++i;
    };
    expect('{');
    skip();
    while (i < text.size() && text[i] != '}') {
      skip();
      if (i >= text.size() || text[i] != '"') throw std::runtime_error("input JSON: expected string");
      const size_t j = text.find('"', i + 1);
      if (j == std::string::npos) throw std::runtime_error("input JSON: unterminated string");
      Span sp;
      sp.key = text.substr(i + 1, j - i - 1);
      i = j + 1;
      expect(':');
      expect('[');
      sp.lo = i;
      const size_t close = text.find(']', i);
      if (close == std::string::npos) throw std::runtime_error("input JSON: expected ']'");
      sp.hi = close;
      i = close + 1;
      spans.push_back(std::move(sp));
      skip();
      if (i < text.size() && text[i] == ',') ++i;
      skip();
    }
    auto parse_array = [&text](size_t lo, size_t hi, std::vector<std::string> &arr) {
      size_t commas = 0;
      for (size_t q = lo; q < hi; ++q) commas += text[q] == ',';
      arr.reserve(commas + 1);
      size_t q = lo;
      auto skipw = [&]() { while (q < hi && isspace((unsigned char)text[q])) ++q; };
      skipw();
      while (q < hi) {
        if (text[q] != '"') throw std::runtime_error("input JSON: expected string");
        const size_t e = text.find('"', q + 1);
        if (e == std::string::npos || e >= hi) throw std::runtime_error("input JSON: unterminated string");
        arr.emplace_back(text, q + 1, e - q - 1);
        q = e + 1;
        skipw();
        if (q < hi && text[q] == ',') ++q;
        skipw();
      }
    };
    CircuitInput in;
    const char *names[9] = {"pk0", "pk1", "m", "u", "e0", "e1", "c0", "c1", "cyclo"};
    std::vector<std::string> *dst[9] = {&in.pk0, &in.pk1, &in.m, &in.u, &in.e0, &in.e1, &in.c0, &in.c1, &in.cyclo};
    const Span *src[9];
    for (int k = 0; k < 9; ++k) {
      src[k] = nullptr;
      for (const Span &sp : spans)
        if (sp.key == names[k]) src[k] = &sp;   // a repeated key: the last one counts, as with a map
      if (!src[k]) throw std::runtime_error(std::string("input JSON: missing field ") + names[k]);
    }
    for (const Span &sp : spans) {   // fields the circuit does not read are still checked for form
      bool used = false;
      for (int k = 0; k < 9; ++k) used = used || src[k] == &sp;
      if (!used) {
        std::vector<std::string> ignored;
        parse_array(sp.lo, sp.hi, ignored);
      }
    }
    if (text.size() < ((size_t)1 << 18)) {
      for (int k = 0; k < 9; ++k) parse_array(src[k]->lo, src[k]->hi, *dst[k]);
    } else {
      std::exception_ptr err[9];
      std::thread th[9];
      for (int k = 0; k < 9; ++k)
        th[k] = std::thread([&, k] {
          try {
            parse_array(src[k]->lo, src[k]->hi, *dst[k]);
          } catch (...) {
            err[k] = std::current_exception();
          }
        });
      for (auto &t : th) t.join();
      for (int k = 0; k < 9; ++k)
        if (err[k]) std::rethrow_exception(err[k]);
    }
    return in;
  }
  static CircuitInput load(const std::string &path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open input file " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return parse_json(ss.str());
  }
};

// State carried from phase 0 to the phase-1 callback (the `move` closure of examples/bfv.rs:171)
struct BfvState {
  PolyChip pk0, pk1, m, u, e0, e1, expected_c0, expected_c1, cyclo;
  PolyChip pk0_u, pk1_u, quotient_0, quotient_1, quotient_0_times_cyclo, quotient_1_times_cyclo, remainder_0, remainder_1;
  Cell delta;
};

// examples/bfv.rs:63-165
// on_public (optional) is called as soon as the public inputs are complete (examples/bfv.rs:118-122), before the
// precomputation: the prover starts hashing them into the transcript while the products and divisions below run.
inline BfvState bfv_phase0(Context &ctx, const CircuitInput &input, const BfvParams &prm, std::vector<Cell> &make_public,
                           const std::function<void(const std::vector<Cell> &)> &on_public = nullptr) {
  const size_t N = prm.N;
  const uint64_t Q = prm.Q;
  const bool tr_on = getenv("ZKFHE_TRACE0") != nullptr;
  auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tl = tnow();
  auto mark = [&](const char *w) {
    if (!tr_on) return;
    const double t = tnow();
    fprintf(stderr, "[phase0] +%.3f ms %s\n", t - tl, w);
    tl = t;
  };
  Poly pk0_un, pk1_un, m_un, u_un, e0_un, e1_un, c0_un, c1_un, cyclo_un;
  {
    // examples/bfv.rs:71-79: nine independent decimal-to-integer conversions; on their own threads once they are long
    const std::vector<std::string> *src[9] = {&input.pk0, &input.pk1, &input.m, &input.u, &input.e0, &input.e1, &input.c0, &input.c1, &input.cyclo};
    Poly *dst[9] = {&pk0_un, &pk1_un, &m_un, &u_un, &e0_un, &e1_un, &c0_un, &c1_un, &cyclo_un};
    if (N < 4096) {
      for (int k = 0; k < 9; ++k) *dst[k] = Poly::from_string(*src[k], Q);
    } else {
      std::exception_ptr err[9];
      std::thread th[9];
      for (int k = 0; k < 9; ++k)
        th[k] = std::thread([&, k] {
          try {
            *dst[k] = Poly::from_string(*src[k], Q);
          } catch (...) {
            err[k] = std::current_exception();
          }
        });
      for (auto &t : th) t.join();
      for (int k = 0; k < 9; ++k)
        if (err[k]) std::rethrow_exception(err[k]);
    }
  }
  mark("from_string x9");
  ZK_ASSERT(pk0_un.deg() == N - 1 && pk1_un.deg() == N - 1 && m_un.deg() == N - 1 && u_un.deg() == N - 1 && e0_un.deg() == N - 1 &&
                e1_un.deg() == N - 1 && c0_un.deg() == N - 1 && c1_un.deg() == N - 1,
            "input polynomials must have degree N - 1 (examples/bfv.rs:82-89)");
  ZK_ASSERT(cyclo_un.deg() == N, "cyclo must have degree N (examples/bfv.rs:90)");
  BfvState st;
  st.pk0 = PolyChip::from_poly(pk0_un, ctx);
  st.pk1 = PolyChip::from_poly(pk1_un, ctx);
  st.m = PolyChip::from_poly(m_un, ctx);
  st.u = PolyChip::from_poly(u_un, ctx);
  st.e0 = PolyChip::from_poly(e0_un, ctx);
  st.e1 = PolyChip::from_poly(e1_un, ctx);
  st.expected_c0 = PolyChip::from_poly(c0_un, ctx);
  st.expected_c1 = PolyChip::from_poly(c1_un, ctx);
  st.cyclo = PolyChip::from_poly(cyclo_un, ctx);
  const uint64_t DELTA = Q / prm.T;
  st.delta = ctx.load_constant(fe::from_u64(DELTA));
  st.pk0.to_public(make_public);
  st.pk1.to_public(make_public);
  st.expected_c0.to_public(make_public);
  st.expected_c1.to_public(make_public);
  st.cyclo.to_public(make_public);
  if (on_public) on_public(make_public);
  mark("from_poly x9 + to_public");
  // PRECOMPUTATION (examples/bfv.rs:124-150)
  Poly pk0_u_un = pk0_un.mul(u_un);
  Poly pk1_u_un = pk1_un.mul(u_un);
  mark("mul x2 (GPU)");
  st.pk0_u = PolyChip::from_poly(pk0_u_un, ctx);
  st.pk1_u = PolyChip::from_poly(pk1_u_un, ctx);
  mark("from_poly pk_u x2");
  Poly pk0_u_red = pk0_u_un.reduce_by_modulus(Q);
  Poly pk1_u_red = pk1_u_un.reduce_by_modulus(Q);
  mark("reduce_by_modulus x2");
  auto qr0 = pk0_u_red.divide_by_cyclo(cyclo_un, Q);
  auto qr1 = pk1_u_red.divide_by_cyclo(cyclo_un, Q);
  mark("divide_by_cyclo x2");
  Poly q0c = qr0.first.mul(cyclo_un);
  Poly q1c = qr1.first.mul(cyclo_un);
  mark("mul cyclo x2");
  st.quotient_0 = PolyChip::from_poly(qr0.first, ctx);
  st.quotient_1 = PolyChip::from_poly(qr1.first, ctx);
  st.quotient_0_times_cyclo = PolyChip::from_poly(q0c, ctx);
  st.quotient_1_times_cyclo = PolyChip::from_poly(q1c, ctx);
  st.remainder_0 = PolyChip::from_poly(qr0.second, ctx);
  st.remainder_1 = PolyChip::from_poly(qr1.second, ctx);
  mark("from_poly x6");
  return st;
}

// examples/bfv.rs:171-301
inline void bfv_phase1(const BfvState &st, const BfvParams &prm, Context &ctx_gate, Context &ctx_rlc, const U256 &gamma) {
  const uint64_t Q = prm.Q, T = prm.T, B = prm.B;
  const RangeChip range(8);
  const RlcChip rlc(gamma);
  const GateChip &gate = range.gate;
  st.e0.constrain_coefficients_in_range(ctx_gate, range, B, Q);
  st.e1.constrain_coefficients_in_range(ctx_gate, range, B, Q);
  st.u.constrain_from_distribution_chi_key(ctx_gate, gate, Q - 1);
  st.m.constrain_coefficients_in_range(ctx_gate, range, T / 2, Q);
  // 1. c0
  st.pk0.constrain_mul(st.u, st.pk0_u, ctx_gate, ctx_rlc, rlc);
  PolyChip pk0_u = st.pk0_u.reduce_by_modulo(ctx_gate, range, Q);
  st.quotient_0.constrain_coefficients_in_modulus_field(ctx_gate, range, Q);
  st.remainder_0.constrain_coefficients_in_modulus_field(ctx_gate, range, Q);
  pk0_u = pk0_u.reduce_by_cyclo(st.cyclo, st.quotient_0, st.quotient_0_times_cyclo, st.remainder_0, range, ctx_gate, ctx_rlc, rlc, Q);
  const PolyChip m_delta = st.m.scalar_mul(ctx_gate, st.delta, gate);
  const PolyChip pk0_u_plus_m_delta = pk0_u.add(ctx_gate, m_delta, gate);
  PolyChip c0 = pk0_u_plus_m_delta.add(ctx_gate, st.e0, gate);
  c0 = c0.reduce_by_modulo(ctx_gate, range, Q);
  c0.constrain_equality(ctx_gate, st.expected_c0, gate);
  // 2. c1
  st.pk1.constrain_mul(st.u, st.pk1_u, ctx_gate, ctx_rlc, rlc);
  PolyChip pk1_u = st.pk1_u.reduce_by_modulo(ctx_gate, range, Q);
  st.quotient_1.constrain_coefficients_in_modulus_field(ctx_gate, range, Q);
  st.remainder_1.constrain_coefficients_in_modulus_field(ctx_gate, range, Q);
  pk1_u = pk1_u.reduce_by_cyclo(st.cyclo, st.quotient_1, st.quotient_1_times_cyclo, st.remainder_1, range, ctx_gate, ctx_rlc, rlc, Q);
  PolyChip c1 = pk1_u.add(ctx_gate, st.e1, gate);
  c1 = c1.reduce_by_modulo(ctx_gate, range, Q);
  c1.constrain_equality(ctx_gate, st.expected_c1, gate);
  ctx_gate.resolve_fractions();
}

// The RLC context alone (the 12 `compute_rlc_fixed_len` evaluations behind the four constrain_mul calls, in circuit
// order) -- used when the gate context is generated on the GPU.  evals[3*i .. 3*i+2] = a(gamma), b(gamma), c(gamma).
inline void bfv_phase1_rlc(const BfvState &st, Context &ctx_rlc, const U256 &gamma, U256 evals[12]) {
  const PolyChip *trip[4][3] = {{&st.pk0, &st.u, &st.pk0_u},
                                {&st.quotient_0, &st.cyclo, &st.quotient_0_times_cyclo},
                                {&st.pk1, &st.u, &st.pk1_u},
                                {&st.quotient_1, &st.cyclo, &st.quotient_1_times_cyclo}};
  if (ctx_rlc.record_structure) {  // keygen / mock: through the chip, which records copies and selectors
    const RlcChip rlc(gamma);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 3; ++j) evals[3 * i + j] = rlc.compute_rlc_fixed_len(ctx_rlc, trip[i][j]->assigned_coefficients).value;
    return;
  }
  // prover: values only.  The 12 Horner chains are independent: their cell blocks ([x0, x1, acc1, x2, acc2, ...], 2L - 1
  // cells each, RlcChip::compute_rlc_fixed_len) are laid out first and filled by four threads, one per constrain_mul.
  // acc stays canonical: acc * gamma is one Montgomery product with gamma * R.
  size_t base[4][3], total = ctx_rlc.advice.size();
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 3; ++j) {
      const size_t L = trip[i][j]->assigned_coefficients.size();
      base[i][j] = total;
      for (size_t s = 1; s < L; ++s) ctx_rlc.selector.push_back((uint32_t)(total + 2 * s - 2));
      total += 2 * L - 1;
    }
  ctx_rlc.advice.resize(total);
  const zk::Fr g = fe::to_mont(gamma);
  auto chain = [&](int i) {
    for (int j = 0; j < 3; ++j) {
      const std::vector<Cell> &in = trip[i][j]->assigned_coefficients;
      U256 *out = ctx_rlc.advice.data() + base[i][j];
      zk::Fr acc = fe::to_fr_raw(in[0].value);
      out[0] = in[0].value;
      for (size_t s = 1; s < in.size(); ++s) {
        acc = zk::fp_add<zk::FrP>(zk::fp_mul<zk::FrP>(acc, g), fe::to_fr_raw(in[s].value));
        out[2 * s - 1] = in[s].value;
        out[2 * s] = fe::from_fr_raw(acc);
      }
      evals[3 * i + j] = fe::from_fr_raw(acc);
    }
  };
  std::thread th[3];
  for (int i = 1; i < 4; ++i) th[i - 1] = std::thread(chain, i);
  chain(0);
  for (auto &t : th) t.join();
}

// --------------------------------------------------------------------------------------------- layout
struct CircuitConfig {  // configs/<name>.json "params"
  unsigned k = 13;
  unsigned n_gate0 = 3, n_gate1 = 153, n_lookup = 36, n_rlc = 5;
  unsigned unusable_rows = 109, lookup_bits = 8;
  unsigned transcript = 0;  // TranscriptKind (transcript.hpp): 0 = Poseidon (the reference's), 1 = Blake2b
  std::vector<uint32_t> bp_gate0, bp_gate1, bp_rlc;  // break points (replayed by the prover)

  size_t n() const { return (size_t)1 << k; }
  // row counts saturate at zero instead of wrapping: a configuration that has not been range-checked (3 < unusable_rows < 2^k)
  // then yields empty ranges, not 2^64-sized ones
  unsigned bf() const { return unusable_rows > 3 ? unusable_rows - 3 : 0; }
  size_t u() const { return n() > (size_t)bf() + 1 ? n() - bf() - 1 : 0; }
  size_t max_rows() const { return n() > unusable_rows ? n() - unusable_rows : 0; }
  unsigned n_gate() const { return n_gate0 + n_gate1; }
  unsigned n_advice() const { return n_gate() + n_lookup + n_rlc; }
  unsigned adv_lookup0() const { return n_gate(); }
  unsigned adv_rlc0() const { return n_gate() + n_lookup; }
  unsigned n_fixed() const { return n_gate() + n_rlc + 2; }
  unsigned fix_qrlc0() const { return n_gate(); }
  unsigned fix_const() const { return n_gate() + n_rlc; }
  unsigned fix_table() const { return n_gate() + n_rlc + 1; }
  unsigned n_perm() const { return n_advice() + 2; }
  unsigned perm_const() const { return n_advice(); }
  unsigned perm_inst() const { return n_advice() + 1; }
  unsigned chunk() const { return 2; }
  unsigned n_chunks() const { return (n_perm() + chunk() - 1) / chunk(); }
};

struct Placement {
  std::vector<uint32_t> col, row;                      // per stream cell
  std::vector<uint32_t> dup_i, dup_col, dup_row;       // duplicates at break points
  std::vector<uint32_t> break_points;
  uint32_t n_columns = 0;
};

// halo2-base `assign_all` walk; `replay` != nullptr replays pinned break points
inline Placement place_stream(size_t n_cells, const std::vector<uint32_t> &selector, size_t max_rows, bool rlc,
                              const std::vector<uint32_t> *replay) {
  Placement p;
  p.col.resize(n_cells);
  p.row.resize(n_cells);
  const uint32_t look = rlc ? 3 : 4;
  size_t si = 0, bi = 0;
  uint32_t col = 0, row = 0;
  for (size_t i = 0; i < n_cells; ++i) {
    while (si < selector.size() && selector[si] < i) ++si;
    const bool sel = si < selector.size() && selector[si] == i;
    bool brk;
    if (!replay) brk = (sel && row + look > max_rows) || row + 1 >= max_rows;
    else brk = bi < replay->size() && row == (*replay)[bi];
    if (brk) {
      p.dup_i.push_back((uint32_t)i);
      p.dup_col.push_back(col);
      p.dup_row.push_back(row);
      p.break_points.push_back(row);
      ++bi;
      ++col;
      row = 0;
    }
    p.col[i] = col;
    p.row[i] = row;
    ++row;
  }
  p.n_columns = col + 1;
  return p;
}

// The witness table of one proof: advice columns (canonical values), and -- in keygen mode -- the fixed
// columns and the copy constraints in permutation-column coordinates.
// [n_advice][n] advice values, either owned or placed in caller-provided (e.g. pinned, reused) memory
struct AdviceTable {
  U256 *data = nullptr;
  size_t n_cols = 0, n = 0;
  std::vector<U256> own;
  void init(size_t cols, size_t rows, U256 *external) {
    n_cols = cols;
    n = rows;
    if (external) {
      data = external;  // reused across proofs: every cell of the fixed layout is rewritten, the rest stays zero
    } else {
      own.assign(cols * rows, fe::zero());
      data = own.data();
    }
  }
  U256 *operator[](size_t c) { return data + c * n; }
  const U256 *operator[](size_t c) const { return data + c * n; }
  size_t size() const { return n_cols; }
};

struct Tables {
  AdviceTable advice;  // [n_advice][n]
  std::vector<std::vector<U256>> fixed;   // [n_fixed][n]   (keygen only)
  std::vector<U256> instance;
  std::vector<std::pair<uint64_t, uint64_t>> copies;  // cell id = perm_col * n + row   (keygen only)
  std::vector<uint32_t> bp_gate0, bp_gate1, bp_rlc;
};

class Assigner {
 public:
  const CircuitConfig &cfg;
  bool keygen;
  Tables t;
  Placement pl[3];
  uint32_t col0[3];
  std::vector<std::pair<U256, uint32_t>> const_rows;  // first-appearance order

  Assigner(const CircuitConfig &c, bool keygen_mode, U256 *advice_storage = nullptr) : cfg(c), keygen(keygen_mode) {
    t.advice.init(cfg.n_advice(), cfg.n(), advice_storage);
    if (keygen) t.fixed.assign(cfg.n_fixed(), std::vector<U256>(cfg.n(), fe::zero()));
    col0[CTX_PHASE0] = 0;
    col0[CTX_GATE1] = cfg.n_gate0;
    col0[CTX_RLC1] = cfg.adv_rlc0();
  }

  uint64_t cell_id(const CellRef &r) const {
    return (uint64_t)(col0[r.ctx] + pl[r.ctx].col[r.off]) * cfg.n() + pl[r.ctx].row[r.off];
  }

  // place one context's stream into its columns
  void place(const Context &ctx, bool use_pinned) {
    const uint32_t id = ctx.cid;
    const std::vector<uint32_t> *replay = nullptr;
    if (use_pinned) replay = id == CTX_PHASE0 ? &cfg.bp_gate0 : id == CTX_GATE1 ? &cfg.bp_gate1 : &cfg.bp_rlc;
    const unsigned ncols = id == CTX_PHASE0 ? cfg.n_gate0 : id == CTX_GATE1 ? cfg.n_gate1 : cfg.n_rlc;
    pl[id] = place_stream(ctx.advice.size(), ctx.selector, cfg.max_rows(), ctx.rlc, replay);
    ZK_ASSERT(pl[id].n_columns <= ncols || ctx.advice.empty(), "circuit does not fit the configured columns");
    const Placement &p = pl[id];
    for (size_t i = 0; i < ctx.advice.size(); ++i) t.advice[col0[id] + p.col[i]][p.row[i]] = ctx.advice[i];
    for (size_t d = 0; d < p.dup_i.size(); ++d) {
      t.advice[col0[id] + p.dup_col[d]][p.dup_row[d]] = ctx.advice[p.dup_i[d]];
      if (keygen)
        t.copies.push_back({(uint64_t)(col0[id] + p.dup_col[d]) * cfg.n() + p.dup_row[d],
                            (uint64_t)(col0[id] + p.col[p.dup_i[d]]) * cfg.n() + p.row[p.dup_i[d]]});
    }
    (id == CTX_PHASE0 ? t.bp_gate0 : id == CTX_GATE1 ? t.bp_gate1 : t.bp_rlc) = p.break_points;
    if (keygen) {
      const uint32_t fsel0 = id == CTX_PHASE0 ? 0 : id == CTX_GATE1 ? cfg.n_gate0 : cfg.fix_qrlc0();
      for (uint32_t o : ctx.selector) t.fixed[fsel0 + p.col[o]][p.row[o]] = fe::one();
    }
  }

  // lookup cells of the phase-1 gate context -> lookup advice columns (column-major fill)
  void place_lookups(const Context &ctx_gate) {
    uint32_t lc = 0, lr = 0;
    for (const CellRef &ref : ctx_gate.lookup) {
      if (lr >= cfg.max_rows()) {
        lr = 0;
        ++lc;
      }
      ZK_ASSERT(lc < cfg.n_lookup, "lookup cells do not fit the configured lookup columns");
      const uint64_t src = cell_id(ref);
      t.advice[cfg.adv_lookup0() + lc][lr] = t.advice[src / cfg.n()][src % cfg.n()];
      if (keygen) t.copies.push_back({(uint64_t)(cfg.adv_lookup0() + lc) * cfg.n() + lr, src});
      ++lr;
    }
  }

  // keygen only: constants column, copy constraints of the contexts, table column, instance wiring
  void finish_structure(const Context &ctx0, const Context &ctx_gate, const Context &ctx_rlc, const std::vector<Cell> &make_public) {
    const Context *ctxs[3] = {&ctx0, &ctx_gate, &ctx_rlc};
    std::map<std::array<uint64_t, 4>, uint32_t> row_of;
    auto key = [](const U256 &v) { return std::array<uint64_t, 4>{v.l[3], v.l[2], v.l[1], v.l[0]}; };
    for (const Context *c : ctxs)
      for (const auto &cv : c->consts)
        if (!row_of.count(key(cv.second))) {
          const uint32_t r = (uint32_t)row_of.size();
          row_of[key(cv.second)] = r;
          ZK_ASSERT(r < cfg.max_rows(), "too many distinct constants");
          t.fixed[cfg.fix_const()][r] = cv.second;
        }
    for (const Context *c : ctxs) {
      for (const auto &cp : c->copies) t.copies.push_back({cell_id(cp.first), cell_id(cp.second)});
      for (const auto &cv : c->consts) t.copies.push_back({cell_id(cv.first), (uint64_t)cfg.perm_const() * cfg.n() + row_of[key(cv.second)]});
    }
    for (uint32_t i = 0; i < (1u << cfg.lookup_bits); ++i) t.fixed[cfg.fix_table()][i] = fe::from_u64(i);
    ZK_ASSERT(make_public.size() <= cfg.max_rows(), "too many instances");
    for (size_t i = 0; i < make_public.size(); ++i) t.copies.push_back({cell_id(make_public[i].ref), (uint64_t)cfg.perm_inst() * cfg.n() + i});
  }
};

}  // namespace zkhost
