// The slice of halo2-base v0.3.0-ce / axiom-eth that reference src/poly_chip.rs drives
// (`Context`, `GateChip`, `RangeChip`, `RlcChip`; reference src/poly_chip.rs:4-11), restated.
// Cell layouts follow SURVEY.md Appendix A and are pinned by the reference's configs/bfv.json:
// the 158 break points come out exactly (tests/test_host_witness.py, against oracle/circuit_ref.py).
#pragma once
#include <cstdint>
#include <initializer_list>
#include <stdexcept>
#include <utility>
#include <vector>

#include "fe.hpp"

namespace zkhost {

struct CellRef {
  uint32_t ctx;
  uint32_t off;
};

// AssignedValue<F>
struct Cell {
  CellRef ref;
  U256 value;
};

struct Q {  // QuantumCell
  enum Kind { WITNESS, CONSTANT, EXISTING, FRACTION } kind;
  U256 v;
  CellRef ref;
};
inline Q Witness(const U256 &v) { return Q{Q::WITNESS, v, {0, 0}}; }
inline Q Constant(const U256 &v) { return Q{Q::CONSTANT, v, {0, 0}}; }
inline Q Constant(uint64_t v) { return Q{Q::CONSTANT, fe::from_u64(v), {0, 0}}; }
inline Q Existing(const Cell &c) { return Q{Q::EXISTING, c.value, c.ref}; }
inline Q WitnessFraction(const U256 &denominator) { return Q{Q::FRACTION, denominator, {0, 0}}; }  // 1/denominator, resolved later

enum : uint32_t { CTX_PHASE0 = 0, CTX_GATE1 = 1, CTX_RLC1 = 2 };

class Context {
 public:
  uint32_t cid;
  bool rlc;
  bool record_structure;  // keygen/mock: keep copy constraints and constants; prover: values only
  std::vector<U256> advice;
  std::vector<uint32_t> selector;                  // offsets with the gate enabled (ascending)
  std::vector<std::pair<CellRef, CellRef>> copies;
  std::vector<std::pair<CellRef, U256>> consts;    // cell == constant
  std::vector<CellRef> lookup;                     // cells to look up
  std::vector<uint32_t> fractions;                 // offsets holding a denominator to invert
  std::vector<uint32_t> inv_slots;                 // every is_zero's 1/x cell, whatever the value (structure; keygen mode)

  Context(uint32_t id, bool is_rlc, bool record) : cid(id), rlc(is_rlc), record_structure(record) {}

  Cell get(int64_t i) const {
    if (i < 0) i += (int64_t)advice.size();
    return Cell{{cid, (uint32_t)i}, advice[(size_t)i]};
  }
  Cell last() const { return get(-1); }

  void push(const Q &q) {
    const uint32_t off = (uint32_t)advice.size();
    advice.push_back(q.v);
    if (q.kind == Q::FRACTION) fractions.push_back(off);
    if (!record_structure) return;
    if (q.kind == Q::EXISTING) copies.push_back({q.ref, CellRef{cid, off}});
    else if (q.kind == Q::CONSTANT) consts.push_back({CellRef{cid, off}, q.v});
  }
  void assign_region(std::initializer_list<Q> cells, std::initializer_list<int> gate_offsets,
                     std::initializer_list<std::pair<int, int>> equalities = {}) {
    const uint32_t base = (uint32_t)advice.size();
    for (const Q &q : cells) push(q);
    for (int g : gate_offsets) selector.push_back(base + (uint32_t)g);
    if (record_structure)
      for (auto &e : equalities) copies.push_back({CellRef{cid, base + (uint32_t)e.first}, CellRef{cid, base + (uint32_t)e.second}});
  }
  Cell load_witness(const U256 &v) {
    push(Witness(v));
    return last();
  }
  Cell load_constant(const U256 &c) {
    push(Constant(c));
    return last();
  }
  void constrain_equal(const Cell &a, const Cell &b) {
    if (record_structure) copies.push_back({a.ref, b.ref});
  }
  void constrain_const(const Cell &a, const U256 &c) {
    if (record_structure) consts.push_back({a.ref, c});
  }
  // resolve deferred 1/x cells (halo2 `batch_invert_assigned`): Montgomery trick on the host
  void resolve_fractions();
};

class GateChip {
 public:
  Cell add(Context &ctx, const Q &a, const Q &b) const {
    ctx.assign_region({a, b, Constant(1), Witness(fe::add(a.v, b.v))}, {0});
    return ctx.last();
  }
  Cell sub(Context &ctx, const Q &a, const Q &b) const {
    ctx.assign_region({Witness(fe::sub(a.v, b.v)), b, Constant(1), a}, {0});
    return ctx.get(-4);
  }
  Cell mul(Context &ctx, const Q &a, const Q &b) const {
    ctx.assign_region({Constant(0), a, b, Witness(fe::mul(a.v, b.v))}, {0});
    return ctx.last();
  }
  Cell not_(Context &ctx, const Q &a) const { return sub(ctx, Constant(1), a); }
  Cell or_(Context &ctx, const Q &a, const Q &b) const {
    const U256 not_b = fe::sub(fe::one(), b.v);
    const U256 out = fe::sub(fe::add(a.v, b.v), fe::mul(a.v, b.v));
    ctx.assign_region({Witness(not_b), Constant(1), b, Constant(1), b, a, Witness(not_b), Witness(out)}, {0, 4}, {{0, 6}, {2, 4}});
    return ctx.last();
  }
  Cell is_zero(Context &ctx, const Cell &a) const {
    const bool z = a.value.is_zero();
    const U256 zv = fe::from_u64(z ? 1 : 0);
    if (ctx.record_structure) ctx.inv_slots.push_back((uint32_t)ctx.advice.size() + 2);
    ctx.assign_region({Witness(zv), Existing(a), z ? Witness(fe::one()) : WitnessFraction(a.value), Constant(1), Constant(0), Existing(a),
                       Witness(zv), Constant(0)},
                      {0, 4}, {{0, 6}});
    return ctx.get(-2);
  }
  Cell is_equal(Context &ctx, const Q &a, const Q &b) const {
    const Cell diff = sub(ctx, a, b);
    return is_zero(ctx, diff);
  }
  void assert_is_const(Context &ctx, const Cell &a, const U256 &c) const { ctx.constrain_const(a, c); }
};

class RangeChip {
 public:
  unsigned lookup_bits;
  GateChip gate;
  explicit RangeChip(unsigned lb = 8) : lookup_bits(lb) {
    if (lb != 8) throw std::invalid_argument("only lookup_bits = 8 (configs/bfv.json:18) is restated");
  }

  void range_check(Context &ctx, const Cell &a, unsigned range_bits) const {
    const unsigned lb = lookup_bits;
    const unsigned k = (range_bits + lb - 1) / lb;
    if (range_bits % lb) throw std::logic_error("range_check: only multiples of lookup_bits occur in this circuit");
    if (k == 1) {
      ctx.lookup.push_back(a.ref);
      return;
    }
    const uint32_t row = (uint32_t)ctx.advice.size();
    // inner_product(limbs, [1, 2^8, 2^16, ...]): [l0, l1, C(2^8), s1, l2, C(2^16), s2, ...], gates at 0,3,6,...
    U256 acc = fe::from_u64(fe::byte_at(a.value, 0));
    ctx.push(Witness(acc));
    for (unsigned i = 1; i < k; ++i) {
      const uint32_t limb = fe::byte_at(a.value, lb * i);
      // acc += limb * 2^(8 i): limb < 256 and the partial sums stay < 2^(8k) << r: plain shifted add
      U256 term = fe::zero();
      const unsigned sh = lb * i;
      term.l[sh / 64] |= (uint64_t)limb << (sh % 64);
      if (sh % 64 > 56 && sh / 64 + 1 < 4) term.l[sh / 64 + 1] |= (uint64_t)limb >> (64 - sh % 64);
      acc = fe::add(acc, term);
      ctx.selector.push_back((uint32_t)ctx.advice.size() - 1);
      ctx.push(Witness(fe::from_u64(limb)));
      ctx.push(Constant(fe::pow2(sh)));
      ctx.push(Witness(acc));
    }
    ctx.constrain_equal(a, ctx.last());
    ctx.lookup.push_back(CellRef{ctx.cid, row});
    for (unsigned i = 0; i + 1 < k; ++i) ctx.lookup.push_back(CellRef{ctx.cid, row + 1 + 3 * i});
  }

  void check_less_than(Context &ctx, const Q &a, const Q &b, unsigned num_bits) const {
    const U256 p2 = fe::pow2(num_bits);
    const U256 shift_a = fe::add(p2, a.v);
    ctx.assign_region({Witness(fe::sub(shift_a, b.v)), b, Constant(1), Witness(shift_a), Constant(fe::neg(p2)), Constant(1), a}, {0, 3});
    const Cell check = ctx.get(-7);
    range_check(ctx, check, num_bits);
  }
  // check_less_than_safe(a, b: u64) and check_big_less_than_safe(a, b: BigUint)
  void check_less_than_safe(Context &ctx, const Cell &a, const U256 &b) const {
    const unsigned range_bits = (b.bits() + lookup_bits - 1) / lookup_bits * lookup_bits;
    range_check(ctx, a, range_bits);
    check_less_than(ctx, Existing(a), Constant(b), range_bits);
  }
  Cell is_less_than(Context &ctx, const Q &a, const Q &b, unsigned num_bits) const {
    const unsigned k = (num_bits + lookup_bits - 1) / lookup_bits;
    const unsigned padded = k * lookup_bits;
    const U256 pp = fe::pow2(padded);
    const U256 shift_a = fe::add(pp, a.v);
    ctx.assign_region({Witness(fe::sub(shift_a, b.v)), b, Constant(1), Witness(shift_a), Constant(fe::neg(pp)), Constant(1), a}, {0, 3});
    const Cell cell = ctx.get(-7);
    range_check(ctx, cell, padded + lookup_bits);
    const CellRef lastl = ctx.lookup.back();
    return gate.is_zero(ctx, ctx.get(lastl.off));
  }
  // (div, rem) of the canonical value by b (< 2^64)
  std::pair<Cell, Cell> div_mod(Context &ctx, const Cell &a, uint64_t b, unsigned a_num_bits) const {
    BigInt q;
    uint64_t r;
    fe::to_bigint(a.value).div_mod_floor_u64(b, q, r);
    const U256 qv = fe::from_bigint(q);
    ctx.assign_region({Witness(fe::from_u64(r)), Constant(b), Witness(qv), Existing(a)}, {0});
    const Cell rem = ctx.get(-4), div = ctx.get(-2);
    // div < 2^a_num_bits / b + 1
    BigInt bound;
    bound.mag.assign(a_num_bits / 32 + 1, 0);
    bound.mag[a_num_bits / 32] = 1u << (a_num_bits % 32);
    bound = bound.div_trunc_u64(b) + BigInt(1);
    check_less_than_safe(ctx, div, fe::from_bigint(bound));
    check_less_than_safe(ctx, rem, fe::from_u64(b));
    return {div, rem};
  }
};

class RlcChip {
 public:
  U256 gamma;
  explicit RlcChip(const U256 &g) : gamma(g) {}
  // returns the cell holding sum_i x_i gamma^(L-1-i)
  Cell compute_rlc_fixed_len(Context &ctx_rlc, const std::vector<Cell> &inputs) const {
    if (!ctx_rlc.rlc) throw std::logic_error("compute_rlc_fixed_len needs the RLC context");
    ctx_rlc.push(Existing(inputs[0]));
    const zk::Fr g = fe::to_mont(gamma);
    zk::Fr acc = fe::to_mont(inputs[0].value);
    for (size_t i = 1; i < inputs.size(); ++i) {
      acc = zk::fp_add<zk::FrP>(zk::fp_mul<zk::FrP>(acc, g), fe::to_mont(inputs[i].value));
      ctx_rlc.selector.push_back((uint32_t)ctx_rlc.advice.size() - 1);
      ctx_rlc.push(Existing(inputs[i]));
      ctx_rlc.push(Witness(fe::from_mont(acc)));
    }
    return ctx_rlc.last();
  }
};

inline void Context::resolve_fractions() {
  if (fractions.empty()) return;
  std::vector<zk::Fr> pre(fractions.size());
  zk::Fr acc = zk::Fr::one();
  for (size_t i = 0; i < fractions.size(); ++i) {
    pre[i] = acc;
    acc = zk::fp_mul<zk::FrP>(acc, fe::to_mont(advice[fractions[i]]));
  }
  acc = zk::fp_inv<zk::FrP>(acc);
  for (size_t i = fractions.size(); i-- > 0;) {
    const zk::Fr x = fe::to_mont(advice[fractions[i]]);
    advice[fractions[i]] = fe::from_mont(zk::fp_mul<zk::FrP>(acc, pre[i]));
    acc = zk::fp_mul<zk::FrP>(acc, x);
  }
  fractions.clear();
}

}  // namespace zkhost
