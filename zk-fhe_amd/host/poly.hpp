// `Poly` and `PolyChip<F>` with the reference's names, argument order and bit-growth assertions
// (reference src/poly.rs:9-191, src/poly_chip.rs:19-399).  Rust `assert!` panics become C++ exceptions
// (the C ABI turns them into status codes).
#pragma once
#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

#include "halo2_base.hpp"
#include "poly_ntt64.hpp"

namespace zkhost {

inline unsigned log2_ceil(uint64_t x) {  // halo2_base::utils::log2_ceil (src/poly.rs:1,101)
  unsigned r = 0;
  while (((uint64_t)1 << r) < x) ++r;
  return r;
}
inline uint64_t bits_u64(uint64_t v) { return v ? 64 - __builtin_clzll(v) : 0; }

#define ZK_ASSERT(cond, msg)                                   \
  do {                                                         \
    if (!(cond)) throw std::runtime_error(std::string("assertion failed: ") + msg); \
  } while (0)

// Plain integer product of two equal-length u64 coefficient vectors (2N-1 exact coefficients).
// The prover installs the GPU implementation (zkfhe_witness_poly_mul_u64); without one, Poly::mul runs
// the reference's own O(N^2) host loop (src/poly.rs:86-90) -- used by `mock` and the host unit tests.
struct PolyMulBackend {
  virtual ~PolyMulBackend() {}
  virtual std::vector<BigInt> mul_u64(const std::vector<uint64_t> &a, const std::vector<uint64_t> &b) = 0;
  // the same product as 2n - 1 canonical 256-bit words owned by the backend (valid until its next call); nullptr = not offered
  virtual const U256 *mul_u64_raw(const std::vector<uint64_t> &, const std::vector<uint64_t> &) { return nullptr; }
};
inline PolyMulBackend *&poly_mul_backend() {
  static thread_local PolyMulBackend *p = nullptr;
  return p;
}

class Poly {
 public:
  std::vector<BigInt> coefficients;  // [a_deg, ..., a_1, a_0]
  size_t degree = 0;
  uint64_t max_bits = 0;

  // src/poly.rs:21-40
  static Poly from_string(const std::vector<std::string> &coeffs, uint64_t modulus) {
    Poly p;
    ZK_ASSERT(!coeffs.empty(), "polynomial needs at least one coefficient");
    const BigInt m(modulus);
    p.coefficients.reserve(coeffs.size());
    for (const auto &s : coeffs) {
      BigInt c = BigInt::parse_dec(s);
      ZK_ASSERT(c <= m, "coeff <= modulus (src/poly.rs:28)");
      p.coefficients.push_back(std::move(c));
    }
    p.degree = p.coefficients.size() - 1;
    p.max_bits = bits_u64(modulus);
    return p;
  }
  // src/poly.rs:46-59
  static Poly from_big_int(std::vector<BigInt> coeffs, uint64_t max_bits) {
    Poly p;
    for (const auto &c : coeffs) ZK_ASSERT(c.bits() <= max_bits, "coeff.bits() <= max_bits (src/poly.rs:51)");
    p.degree = coeffs.size() - 1;
    p.coefficients = std::move(coeffs);
    p.max_bits = max_bits;
    return p;
  }
  size_t deg() const { return degree; }

  // src/poly.rs:75-103
  Poly mul(const Poly &other) const {
    ZK_ASSERT(deg() == other.deg(), "deg_a == deg_b (src/poly.rs:78)");
    const size_t da = deg(), db = other.deg();
    std::vector<BigInt> c;
    bool small = true;
    for (const auto &x : coefficients) small = small && x.fits_u64();
    for (const auto &x : other.coefficients) small = small && x.fits_u64();
    if (small && is_cyclo_shape(other)) {
      // q * (x^N + 1) = q * x^N + q : exact shift-and-add
      c.assign(da + db + 1, BigInt());
      for (size_t i = 0; i <= da; ++i) {
        c[i] += coefficients[i];
        c[i + db] += coefficients[i];
      }
    } else if (small && ((da + 1) & da) == 0 && (poly_mul_backend() || (da + 1 <= 2048 && max_bits <= 32 && other.max_bits <= 32))) {
      std::vector<uint64_t> a(da + 1), b(db + 1);
      for (size_t i = 0; i <= da; ++i) a[i] = coefficients[i].to_u64();
      for (size_t i = 0; i <= db; ++i) b[i] = other.coefficients[i].to_u64();
      if (gl::fits(a, b)) {
        // short and narrow (N <= 2048, 32-bit coefficients): exact NTT convolution on this core (poly_ntt64.hpp)
        std::vector<uint64_t> lo, hi;
        gl::poly_mul_u32(a, b, lo, hi);
        c.resize(lo.size());
        for (size_t i = 0; i < c.size(); ++i) {
          U256 v = fe::zero();
          v.l[0] = lo[i];
          v.l[1] = hi[i];
          c[i] = fe::to_bigint(v);
        }
      } else if (poly_mul_backend()) {
        c = poly_mul_backend()->mul_u64(a, b);
      } else {
        c.assign(da + db + 1, BigInt());
        for (size_t i = 0; i <= da; ++i) {
          if (coefficients[i].is_zero()) continue;
          for (size_t j = 0; j <= db; ++j)
            if (!other.coefficients[j].is_zero()) c[i + j] += coefficients[i] * other.coefficients[j];
        }
      }
    } else {
      c.assign(da + db + 1, BigInt());
      for (size_t i = 0; i <= da; ++i) {
        if (coefficients[i].is_zero()) continue;
        for (size_t j = 0; j <= db; ++j)
          if (!other.coefficients[j].is_zero()) c[i + j] += coefficients[i] * other.coefficients[j];
      }
    }
    ZK_ASSERT(c.size() == da + db + 1, "c.len() == deg_c + 1");
    const uint64_t mb = max_bits + other.max_bits + log2_ceil(da + 1);
    return from_big_int(std::move(c), mb);
  }

  // src/poly.rs:180-191
  Poly reduce_by_modulus(uint64_t modulus) const {
    std::vector<BigInt> c;
    c.reserve(coefficients.size());
    for (const auto &x : coefficients) c.push_back(BigInt(x.mod_floor_u64(modulus)));
    return from_big_int(std::move(c), bits_u64(modulus));
  }

  // src/poly.rs:113-177.  For the divisor x^N + 1 (the function's stated assumption) the long division
  // has the closed form  quot = d[0 .. len-N),  rem[j] = d[len-N+j] - d[j-... ] (the loop below is the
  // reference's, with a moving head instead of Vec::remove(0)).
  std::pair<Poly, Poly> divide_by_cyclo(const Poly &cyclo, uint64_t modulus) const {
    const uint64_t modulus_bits = bits_u64(modulus);
    bool all_zero = true;
    for (const auto &x : coefficients) all_zero = all_zero && x.is_zero();
    if (coefficients.empty() || all_zero) {
      return {from_big_int(std::vector<BigInt>(cyclo.deg() + 1), modulus_bits),
              from_big_int(std::vector<BigInt>(2 * cyclo.deg() + 1), modulus_bits)};
    }
    std::vector<BigInt> dividend = coefficients;
    const std::vector<BigInt> &divisor = cyclo.coefficients;
    ZK_ASSERT(divisor[0].fits_u64() && !divisor[0].is_zero(), "leading coefficient of cyclo");
    const uint64_t lead = divisor[0].to_u64();
    std::vector<BigInt> quotient;
    std::vector<size_t> nz;  // the divisor's non-zero terms (two for x^N + 1): the update loop only visits those
    for (size_t i = 0; i < divisor.size(); ++i)
      if (!divisor[i].is_zero()) nz.push_back(i);
    size_t pos = 0;
    while (dividend.size() - pos > divisor.size() - 1) {
      BigInt ratio = dividend[pos].div_trunc_u64(lead);
      if (!ratio.is_zero())
        for (size_t i : nz) dividend[pos + i] -= ratio * divisor[i];
      quotient.push_back(std::move(ratio));
      ++pos;
    }
    std::vector<BigInt> remainder(dividend.begin() + pos, dividend.end());
    size_t qz = 0, rz = 0;
    while (qz < quotient.size() && quotient[qz].is_zero()) ++qz;
    while (rz < remainder.size() && remainder[rz].is_zero()) ++rz;
    quotient.erase(quotient.begin(), quotient.begin() + qz);
    remainder.erase(remainder.begin(), remainder.begin() + rz);
    ZK_ASSERT(!quotient.empty(), "quotient.len() - 1 underflows in the reference (src/poly.rs:158) for an empty quotient");
    if (quotient.size() < cyclo.deg() + 1) quotient.insert(quotient.begin(), cyclo.deg() + 1 - quotient.size(), BigInt());
    if (remainder.size() < 2 * cyclo.deg() + 1) remainder.insert(remainder.begin(), 2 * cyclo.deg() + 1 - remainder.size(), BigInt());
    for (auto &x : remainder) x = BigInt(x.mod_floor_u64(modulus));
    return {from_big_int(std::move(quotient), modulus_bits), from_big_int(std::move(remainder), modulus_bits)};
  }

 private:
  static bool is_cyclo_shape(const Poly &p) {
    const auto &c = p.coefficients;
    if (c.size() < 2 || !(c.front() == BigInt(1)) || !(c.back() == BigInt(1))) return false;
    for (size_t i = 1; i + 1 < c.size(); ++i)
      if (!c[i].is_zero()) return false;
    return true;
  }
};

// ---------------------------------------------------------------------------------------------
class PolyChip {
 public:
  std::vector<Cell> assigned_coefficients;
  uint64_t max_num_bits = 0;
  size_t degree = 0;

  PolyChip() {}
  PolyChip(std::vector<Cell> cells, uint64_t bits) : assigned_coefficients(std::move(cells)), max_num_bits(bits) {
    degree = assigned_coefficients.size() - 1;
  }

  // :27-42
  static PolyChip from_poly(const Poly &poly, Context &ctx) {
    std::vector<Cell> cells;
    cells.reserve(poly.deg() + 1);
    for (size_t i = 0; i <= poly.deg(); ++i) cells.push_back(ctx.load_witness(fe::from_bigint(poly.coefficients[i])));
    return PolyChip(std::move(cells), poly.max_bits);
  }
  // :58-62
  void to_public(std::vector<Cell> &make_public) const {
    for (const auto &c : assigned_coefficients) make_public.push_back(c);
  }
  // :81-116
  void constrain_mul(const PolyChip &b, const PolyChip &c, Context &ctx_gate, Context &ctx_rlc, const RlcChip &rlc) const {
    ZK_ASSERT(c.max_num_bits < fe::MOD_BITS, "c_max_bits < p_bits (src/poly_chip.rs:94)");
    const Cell a_eval = rlc.compute_rlc_fixed_len(ctx_rlc, assigned_coefficients);
    const Cell b_eval = rlc.compute_rlc_fixed_len(ctx_rlc, b.assigned_coefficients);
    const Cell c_eval = rlc.compute_rlc_fixed_len(ctx_rlc, c.assigned_coefficients);
    ctx_gate.assign_region({Constant(0), Existing(a_eval), Existing(b_eval), Existing(c_eval)}, {0});
  }
  // :122-144
  PolyChip add(Context &ctx, const PolyChip &other, const GateChip &gate) const {
    std::vector<Cell> out;
    out.reserve(degree + 1);
    for (size_t i = 0; i <= degree; ++i)
      out.push_back(gate.add(ctx, Existing(assigned_coefficients[i]), Existing(other.assigned_coefficients[i])));
    const uint64_t mb = std::max(max_num_bits, other.max_num_bits) + 1;
    ZK_ASSERT(mb < fe::MOD_BITS, "Risk of overflow detected in add");
    return PolyChip(std::move(out), mb);
  }
  // :150-174
  PolyChip scalar_mul(Context &ctx, const Cell &scalar, const GateChip &gate) const {
    const uint64_t mb = max_num_bits + scalar.value.bits();
    ZK_ASSERT(mb < fe::MOD_BITS, "Risk of overflow detected in scalar_mul");
    std::vector<Cell> out;
    out.reserve(degree + 1);
    for (size_t i = 0; i <= degree; ++i) out.push_back(gate.mul(ctx, Existing(assigned_coefficients[i]), Existing(scalar)));
    return PolyChip(std::move(out), mb);
  }
  // :183-223
  PolyChip reduce_by_cyclo(const PolyChip &cyclo, const PolyChip &quotient, const PolyChip &quotient_times_cyclo,
                           const PolyChip &remainder, const RangeChip &range, Context &ctx_gate, Context &ctx_rlc,
                           const RlcChip &rlc, uint64_t modulus) const {
    const uint64_t modulus_bits = bits_u64(modulus);
    ZK_ASSERT(quotient.max_num_bits <= modulus_bits, "quotient.max_num_bits <= modulus_bits");
    ZK_ASSERT(remainder.max_num_bits <= modulus_bits, "remainder.max_num_bits <= modulus_bits");
    ZK_ASSERT(std::max(quotient_times_cyclo.max_num_bits, remainder.max_num_bits) + 1 < fe::MOD_BITS, "sum fits the field");
    const size_t cyclo_deg = cyclo.degree;
    quotient.constrain_mul(cyclo, quotient_times_cyclo, ctx_gate, ctx_rlc, rlc);
    const PolyChip sum = quotient_times_cyclo.add(ctx_gate, remainder, range.gate);
    const PolyChip sum_mod = sum.reduce_by_modulo(ctx_gate, range, modulus);
    const PolyChip sum_trimmed = sum_mod.safe_trim_leading_zeroes(ctx_gate, range, degree);
    sum_trimmed.constrain_equality(ctx_gate, *this, range.gate);
    return remainder.safe_trim_leading_zeroes(ctx_gate, range, cyclo_deg - 1);
  }
  // :226-252
  PolyChip reduce_by_modulo(Context &ctx, const RangeChip &range, uint64_t modulus) const {
    std::vector<Cell> out;
    out.reserve(degree + 1);
    const unsigned num_bits = (unsigned)max_num_bits;
    for (size_t i = 0; i <= degree; ++i) out.push_back(range.div_mod(ctx, assigned_coefficients[i], modulus, num_bits).second);
    return PolyChip(std::move(out), bits_u64(modulus));
  }
  // :255-264
  void constrain_equality(Context &ctx, const PolyChip &other, const GateChip &gate) const {
    for (size_t i = 0; i <= degree; ++i) {
      const Cell b = gate.is_equal(ctx, Existing(assigned_coefficients[i]), Existing(other.assigned_coefficients[i]));
      gate.assert_is_const(ctx, b, fe::one());
    }
  }
  // :270-317
  void constrain_coefficients_in_range(Context &ctx, const RangeChip &range, uint64_t z, uint64_t y) const {
    ZK_ASSERT(z < y, "z < y");
    const unsigned y_bits = (unsigned)bits_u64(y);
    for (const Cell &coeff : assigned_coefficients) {
      range.check_less_than_safe(ctx, coeff, fe::from_u64(y));
      const Cell in1 = range.is_less_than(ctx, Existing(coeff), Constant(z + 1), y_bits);
      const Cell not_in2 = range.is_less_than(ctx, Existing(coeff), Constant(y - z), y_bits);
      const Cell in2 = range.gate.not_(ctx, Existing(not_in2));
      const Cell in_range = range.gate.or_(ctx, Existing(in1), Existing(in2));
      range.gate.assert_is_const(ctx, in_range, fe::one());
    }
  }
  // :320-354
  void constrain_from_distribution_chi_key(Context &ctx, const GateChip &gate, uint64_t z) const {
    for (const Cell &coeff : assigned_coefficients) {
      const Cell f1 = gate.sub(ctx, Existing(coeff), Constant(0));
      const Cell f2 = gate.sub(ctx, Existing(coeff), Constant(1));
      const Cell f3 = gate.sub(ctx, Existing(coeff), Constant(z));
      const Cell f12 = gate.mul(ctx, Existing(f1), Existing(f2));
      const Cell f123 = gate.mul(ctx, Existing(f12), Existing(f3));
      gate.assert_is_const(ctx, f123, fe::zero());
    }
  }
  // :357-366
  void constrain_coefficients_in_modulus_field(Context &ctx, const RangeChip &range, uint64_t modulus) const {
    for (const Cell &coeff : assigned_coefficients) range.check_less_than_safe(ctx, coeff, fe::from_u64(modulus));
  }

 private:
  // :374-399
  PolyChip safe_trim_leading_zeroes(Context &ctx, const RangeChip &range, size_t deg) const {
    ZK_ASSERT(deg <= degree, "degree <= self.degree");
    for (size_t i = 0; i < degree - deg; ++i) range.gate.assert_is_const(ctx, assigned_coefficients[i], fe::zero());
    return PolyChip(std::vector<Cell>(assigned_coefficients.begin() + (degree - deg), assigned_coefficients.end()), max_num_bits);
  }
};

}  // namespace zkhost
