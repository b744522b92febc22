// bfv_phase0_fast (machine words) against bfv_phase0 (the BigInt restatement of src/poly.rs / examples/bfv.rs:63-165) with a
// schoolbook product as the PolyMulBackend: the wide configuration (N = 4096, 60-bit Q: three-word products) that the host-only
// C ABI cannot reach without a GPU.  usage: phase0_check <input.json> <N> <Q>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <sstream>

#include "bfv_phase0_fast.hpp"
using namespace zkhost;

struct SchoolbookMul : PolyMulBackend {
  std::vector<U256> out;
  std::vector<BigInt> mul_u64(const std::vector<uint64_t> &a, const std::vector<uint64_t> &b) override {
    const U256 *r = mul_u64_raw(a, b);
    std::vector<BigInt> o(2 * a.size() - 1);
    for (size_t i = 0; i < o.size(); ++i) o[i] = fe::to_bigint(r[i]);
    return o;
  }
  const U256 *mul_u64_raw(const std::vector<uint64_t> &a, const std::vector<uint64_t> &b) override {
    typedef unsigned __int128 u128;
    const size_t n = a.size();
    out.assign(2 * n - 1, U256{{0, 0, 0, 0}});
    for (size_t i = 0; i < n; ++i)
      for (size_t j = 0; j < n; ++j) {
        const u128 p = (u128)a[i] * b[j];
        U256 &o = out[i + j];
        u128 s = (u128)o.l[0] + (uint64_t)p;
        o.l[0] = (uint64_t)s;
        s = (u128)o.l[1] + (uint64_t)(p >> 64) + (uint64_t)(s >> 64);
        o.l[1] = (uint64_t)s;
        o.l[2] += (uint64_t)(s >> 64);
      }
    return out.data();
  }
};

int main(int argc, char **argv) {
  if (argc < 4) return 2;
  std::ifstream f(argv[1]);
  std::stringstream ss;
  ss << f.rdbuf();
  const std::string text = ss.str();
  BfvParams prm;
  prm.N = strtoull(argv[2], nullptr, 10);
  prm.Q = strtoull(argv[3], nullptr, 10);
  prm.T = 7;
  prm.B = 19;
  SchoolbookMul m;
  poly_mul_backend() = &m;
  Context c0(CTX_PHASE0, false, false), c1(CTX_PHASE0, false, false);
  std::vector<Cell> pub0, pub1;
  BfvState s0;
  const bool took = bfv_phase0_fast(c0, text.data(), text.size(), prm, pub0, s0);
  const BfvState s1 = bfv_phase0(c1, CircuitInput::parse_json(text), prm, pub1);
  bool same = took && c0.advice.size() == c1.advice.size() && !memcmp(c0.advice.data(), c1.advice.data(), c0.advice.size() * 32) && pub0.size() == pub1.size();
  for (size_t i = 0; same && i < pub0.size(); ++i) same = pub0[i].value == pub1[i].value && pub0[i].ref.off == pub1[i].ref.off;
  const PolyChip *a[17] = {&s0.pk0, &s0.pk1, &s0.m, &s0.u, &s0.e0, &s0.e1, &s0.expected_c0, &s0.expected_c1, &s0.cyclo, &s0.pk0_u, &s0.pk1_u, &s0.quotient_0, &s0.quotient_1,
                           &s0.quotient_0_times_cyclo, &s0.quotient_1_times_cyclo, &s0.remainder_0, &s0.remainder_1};
  const PolyChip *b[17] = {&s1.pk0, &s1.pk1, &s1.m, &s1.u, &s1.e0, &s1.e1, &s1.expected_c0, &s1.expected_c1, &s1.cyclo, &s1.pk0_u, &s1.pk1_u, &s1.quotient_0, &s1.quotient_1,
                           &s1.quotient_0_times_cyclo, &s1.quotient_1_times_cyclo, &s1.remainder_0, &s1.remainder_1};
  for (int k = 0; same && k < 17; ++k) {
    same = a[k]->max_num_bits == b[k]->max_num_bits && a[k]->degree == b[k]->degree && a[k]->assigned_coefficients.size() == b[k]->assigned_coefficients.size();
    for (size_t i = 0; same && i < a[k]->assigned_coefficients.size(); ++i)
      same = a[k]->assigned_coefficients[i].value == b[k]->assigned_coefficients[i].value && a[k]->assigned_coefficients[i].ref.off == b[k]->assigned_coefficients[i].ref.off;
  }
  same = same && s0.delta.value == s1.delta.value && s0.delta.ref.off == s1.delta.ref.off;
  printf("phase0: fast path taken %d, cells %zu, identical %d\n", (int)took, c0.advice.size(), (int)same);
  return same ? 0 : 1;
}
