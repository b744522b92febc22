// Minimal signed arbitrary-precision integer: the subset of num-bigint that reference src/poly.rs uses
// (parse_bytes base 10, + - *, `/` truncating, mod_floor, bits(), comparisons, is_zero).
#pragma once
#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace zkhost {

// limb storage with a small inline buffer: the circuit's integers stay below 2^192 (six limbs), so the ~30 k BigInt
// temporaries of one witness never touch the heap; larger values (pairing exponents) spill to a heap block
class LimbVec {
 public:
  LimbVec() {}
  LimbVec(size_t n, uint32_t v) { assign(n, v); }
  LimbVec(const LimbVec &o) { copy_from(o); }
  LimbVec(LimbVec &&o) noexcept { steal(o); }
  ~LimbVec() { release(); }
  LimbVec &operator=(const LimbVec &o) {
    if (this != &o) {
      n_ = 0;
      copy_from(o);
    }
    return *this;
  }
  LimbVec &operator=(LimbVec &&o) noexcept {
    if (this != &o) {
      release();
      steal(o);
    }
    return *this;
  }
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  uint32_t &operator[](size_t i) { return data()[i]; }
  const uint32_t &operator[](size_t i) const { return data()[i]; }
  uint32_t &back() { return data()[n_ - 1]; }
  const uint32_t &back() const { return data()[n_ - 1]; }
  uint32_t *begin() { return data(); }
  uint32_t *end() { return data() + n_; }
  const uint32_t *begin() const { return data(); }
  const uint32_t *end() const { return data() + n_; }
  void clear() { n_ = 0; }
  void pop_back() { --n_; }
  void push_back(uint32_t v) {
    reserve(n_ + 1);
    data()[n_++] = v;
  }
  void assign(size_t n, uint32_t v) {
    n_ = 0;
    reserve(n);
    std::fill(data(), data() + n, v);
    n_ = (uint32_t)n;
  }

 private:
  static constexpr uint32_t INLINE = 6;
  uint32_t n_ = 0, cap_ = INLINE;
  uint32_t *heap_ = nullptr;
  uint32_t buf_[INLINE];
  uint32_t *data() { return heap_ ? heap_ : buf_; }
  const uint32_t *data() const { return heap_ ? heap_ : buf_; }
  void reserve(size_t want) {
    if (want <= cap_) return;
    size_t nc = std::max<size_t>(want, (size_t)cap_ * 2);
    uint32_t *nh = new uint32_t[nc];
    std::copy(data(), data() + n_, nh);
    delete[] heap_;
    heap_ = nh;
    cap_ = (uint32_t)nc;
  }
  void release() {
    delete[] heap_;
    heap_ = nullptr;
    cap_ = INLINE;
    n_ = 0;
  }
  void copy_from(const LimbVec &o) {
    reserve(o.n_);
    std::copy(o.data(), o.data() + o.n_, data());
    n_ = o.n_;
  }
  void steal(LimbVec &o) {
    n_ = o.n_;
    if (o.heap_) {
      heap_ = o.heap_;
      cap_ = o.cap_;
      o.heap_ = nullptr;
      o.cap_ = INLINE;
    } else {
      heap_ = nullptr;
      cap_ = INLINE;
      std::copy(o.buf_, o.buf_ + o.n_, buf_);
    }
    o.n_ = 0;
  }
};

class BigInt {
 public:
  bool neg = false;
  LimbVec mag;  // little-endian limbs, no leading zeros; empty == 0

  BigInt() {}
  BigInt(uint64_t v) { set_u64(v); }
  static BigInt from_i64(int64_t v) {
    BigInt r(v < 0 ? (uint64_t)(-(v + 1)) + 1 : (uint64_t)v);
    r.neg = v < 0;
    return r;
  }
  void set_u64(uint64_t v) {
    mag.clear();
    neg = false;
    if (v) mag.push_back((uint32_t)v);
    if (v >> 32) mag.push_back((uint32_t)(v >> 32));
  }
  bool is_zero() const { return mag.empty(); }
  // number of significant bits of |x| (BigInt::bits in num-bigint)
  uint64_t bits() const {
    if (mag.empty()) return 0;
    return (uint64_t)(mag.size() - 1) * 32 + (32 - __builtin_clz(mag.back()));
  }
  bool fits_u64() const { return !neg && mag.size() <= 2; }
  uint64_t to_u64() const {
    uint64_t v = 0;
    if (mag.size() > 0) v |= mag[0];
    if (mag.size() > 1) v |= (uint64_t)mag[1] << 32;
    return v;
  }

  static BigInt parse_dec(const std::string &s) {
    BigInt r;
    size_t i = 0;
    bool ng = false;
    if (i < s.size() && (s[i] == '-' || s[i] == '+')) ng = s[i++] == '-';
    if (i >= s.size()) throw std::invalid_argument("empty integer literal");
    // digits are consumed nine at a time (10^9 < 2^32): one multiply-add of the limb vector per group
    while (i < s.size()) {
      uint32_t group = 0, scale = 1;
      for (int k = 0; k < 9 && i < s.size(); ++k, ++i) {
        if (s[i] < '0' || s[i] > '9') throw std::invalid_argument("bad decimal digit in '" + s + "'");
        group = group * 10 + (uint32_t)(s[i] - '0');
        scale *= 10;
      }
      r.mul_small(scale);
      r.add_small(group);
    }
    r.neg = ng && !r.is_zero();
    return r;
  }

  std::string to_dec() const {
    if (is_zero()) return "0";
    BigInt t = *this;
    t.neg = false;
    std::string out;
    while (!t.is_zero()) {
      uint32_t rem = t.divmod_small(1000000000u);
      for (int k = 0; k < 9; ++k) {
        out.push_back((char)('0' + rem % 10));
        rem /= 10;
        if (t.is_zero() && rem == 0) break;
      }
    }
    while (out.size() > 1 && out.back() == '0') out.pop_back();
    if (neg) out.push_back('-');
    std::reverse(out.begin(), out.end());
    return out;
  }

  static int cmp_mag(const BigInt &a, const BigInt &b) {
    if (a.mag.size() != b.mag.size()) return a.mag.size() < b.mag.size() ? -1 : 1;
    for (size_t i = a.mag.size(); i-- > 0;)
      if (a.mag[i] != b.mag[i]) return a.mag[i] < b.mag[i] ? -1 : 1;
    return 0;
  }
  static int cmp(const BigInt &a, const BigInt &b) {
    if (a.neg != b.neg) return a.neg ? -1 : 1;
    int c = cmp_mag(a, b);
    return a.neg ? -c : c;
  }
  bool operator==(const BigInt &o) const { return cmp(*this, o) == 0; }
  bool operator!=(const BigInt &o) const { return cmp(*this, o) != 0; }
  bool operator<(const BigInt &o) const { return cmp(*this, o) < 0; }
  bool operator<=(const BigInt &o) const { return cmp(*this, o) <= 0; }

  BigInt operator-() const {
    BigInt r = *this;
    if (!r.is_zero()) r.neg = !r.neg;
    return r;
  }
  BigInt operator+(const BigInt &o) const {
    BigInt r;
    if (neg == o.neg) {
      r.mag = add_mag(mag, o.mag);
      r.neg = neg;
    } else {
      int c = cmp_mag(*this, o);
      if (c == 0) return r;
      if (c > 0) {
        r.mag = sub_mag(mag, o.mag);
        r.neg = neg;
      } else {
        r.mag = sub_mag(o.mag, mag);
        r.neg = o.neg;
      }
    }
    return r;
  }
  BigInt operator-(const BigInt &o) const { return *this + (-o); }
  BigInt &operator+=(const BigInt &o) { return *this = *this + o; }
  BigInt &operator-=(const BigInt &o) { return *this = *this - o; }
  BigInt operator*(const BigInt &o) const {
    BigInt r;
    if (is_zero() || o.is_zero()) return r;
    r.mag.assign(mag.size() + o.mag.size(), 0);
    for (size_t i = 0; i < mag.size(); ++i) {
      uint64_t carry = 0;
      for (size_t j = 0; j < o.mag.size(); ++j) {
        uint64_t t = (uint64_t)mag[i] * o.mag[j] + r.mag[i + j] + carry;
        r.mag[i + j] = (uint32_t)t;
        carry = t >> 32;
      }
      r.mag[i + o.mag.size()] += (uint32_t)carry;
    }
    r.trim();
    r.neg = neg != o.neg;
    return r;
  }

  // truncating division by a u64 (num-bigint `/`), remainder has the sign of the dividend
  static void divrem_u64(const BigInt &a, uint64_t d, BigInt &q, uint64_t &rem_mag) {
    if (d == 0) throw std::domain_error("division by zero");
    q.mag.assign(a.mag.size(), 0);
    unsigned __int128 rem = 0;
    for (size_t i = a.mag.size(); i-- > 0;) {
      rem = (rem << 32) | a.mag[i];
      q.mag[i] = (uint32_t)(rem / d);
      rem %= d;
    }
    q.trim();
    q.neg = a.neg && !q.is_zero();
    rem_mag = (uint64_t)rem;
  }
  BigInt div_trunc_u64(uint64_t d) const {
    BigInt q;
    uint64_t r;
    divrem_u64(*this, d, q, r);
    return q;
  }
  // Integer::mod_floor with a positive u64 modulus: result in [0, m)
  uint64_t mod_floor_u64(uint64_t m) const {
    BigInt q;
    uint64_t r;
    divrem_u64(*this, m, q, r);
    if (neg && r) r = m - r;
    return r;
  }
  // floor division by positive u64: (q, r) with r in [0, d)
  void div_mod_floor_u64(uint64_t d, BigInt &q, uint64_t &r) const {
    divrem_u64(*this, d, q, r);
    if (neg && r) {
      q = q - BigInt(1);
      r = d - r;
    }
  }

 private:
  void trim() {
    while (!mag.empty() && mag.back() == 0) mag.pop_back();
    if (mag.empty()) neg = false;
  }
  void mul_small(uint32_t m) {
    uint64_t carry = 0;
    for (auto &l : mag) {
      uint64_t t = (uint64_t)l * m + carry;
      l = (uint32_t)t;
      carry = t >> 32;
    }
    if (carry) mag.push_back((uint32_t)carry);
  }
  void add_small(uint32_t a) {
    uint64_t carry = a;
    for (size_t i = 0; i < mag.size() && carry; ++i) {
      uint64_t t = (uint64_t)mag[i] + carry;
      mag[i] = (uint32_t)t;
      carry = t >> 32;
    }
    if (carry) mag.push_back((uint32_t)carry);
  }
  uint32_t divmod_small(uint32_t d) {
    uint64_t rem = 0;
    for (size_t i = mag.size(); i-- > 0;) {
      uint64_t cur = (rem << 32) | mag[i];
      mag[i] = (uint32_t)(cur / d);
      rem = cur % d;
    }
    trim();
    return (uint32_t)rem;
  }
  static LimbVec add_mag(const LimbVec &a, const LimbVec &b) {
    const auto &x = a.size() >= b.size() ? a : b;
    const auto &y = a.size() >= b.size() ? b : a;
    LimbVec r(x.size() + 1, 0);
    uint64_t carry = 0;
    for (size_t i = 0; i < x.size(); ++i) {
      uint64_t t = (uint64_t)x[i] + (i < y.size() ? y[i] : 0) + carry;
      r[i] = (uint32_t)t;
      carry = t >> 32;
    }
    r[x.size()] = (uint32_t)carry;
    while (!r.empty() && r.back() == 0) r.pop_back();
    return r;
  }
  static LimbVec sub_mag(const LimbVec &a, const LimbVec &b) {  // |a| > |b|
    LimbVec r(a.size(), 0);
    int64_t borrow = 0;
    for (size_t i = 0; i < a.size(); ++i) {
      int64_t t = (int64_t)a[i] - (i < b.size() ? b[i] : 0) - borrow;
      borrow = t < 0;
      r[i] = (uint32_t)(t + (borrow ? ((int64_t)1 << 32) : 0));
    }
    while (!r.empty() && r.back() == 0) r.pop_back();
    return r;
  }
};

}  // namespace zkhost
