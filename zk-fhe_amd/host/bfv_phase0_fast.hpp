// Phase 0 of the BFV circuit (examples/bfv.rs:63-165) for the PROVER, in machine words.
//
// bfv_phase0 (bfv_circuit.hpp) restates the reference line by line: decimal strings -> BigInt (src/poly.rs:21-40), the
// products pk_i * u (src/poly.rs:75-103), `reduce_by_modulus` (:180-191), the long division by the cyclotomic polynomial
// (:113-177), quotient * cyclo, and a `load_witness` per coefficient (src/poly_chip.rs:27-42).  That is what keygen, `mock`
// and the unit tests run.  A proof needs only the VALUES of that computation, and for the inputs the circuit is meant for --
// Q below 2^63, decimal coefficients of at most 19 digits, cyclo = x^N + 1 -- every one of them is a 64- or 128-bit integer
// and every step has a closed form:
//   * pk_i * u          exact product (gl::poly_mul_u32 on this core for N <= 2048 and 32-bit coefficients, else the GPU
//                       convolution of the PolyMulBackend): coefficients below N Q^2 < 2^141, three words;
//   * reduce_by_modulus coefficient mod Q: three 128-by-64 divisions, word by word;
//   * divide_by_cyclo   with the dividend d[0 .. 2N-1) (big-endian, as everywhere in the reference) the loop of
//                       src/poly.rs:133-142 gives  quotient = d[0 .. N-1),  remainder[0] = d[N-1],
//                       remainder[j] = d[N-1+j] - d[j-1]; stripping leading zeros and padding back to N+1 / 2N+1 coefficients
//                       (:148-166) is a left-pad with zeros; the remainder is then taken mod_floor Q (:169-172);
//   * quotient * cyclo  = quotient * x^N + quotient (shift and add).
// The function below computes exactly these values, builds the same cell stream in the same order (so the phase-0 columns,
// the public inputs and the proof bytes are the ones bfv_phase0 gives -- tests/test_host_witness.py compares the two paths
// table for table, the GPU suite proof for proof) and carries the same max_bits bookkeeping.  Anything outside its domain
// (odd syntax, a coefficient above Q, a cyclo of another shape, a quotient the reference would panic on, ...) makes it return
// false WITHOUT touching its outputs: the caller then runs bfv_phase0, which reports the error the reference's way.
// k = 13: 1.2 ms -> 0.3 ms per proof; k = 19 (N = 16384, 2.8 MB of JSON): 25 ms -> 8 ms before the GPU has anything to do.
#pragma once
#include <thread>

#include "bfv_circuit.hpp"

namespace zkhost {

struct FastInput {
  std::vector<uint64_t> a[9];   // pk0, pk1, m, u, e0, e1, c0, c1, cyclo -- file order (big-endian coefficients)
};

// {"name": ["123", ...], ...}: the same grammar as CircuitInput::parse_json, values straight to u64.  false = not for the
// fast path (the generic parser decides whether it is an error).
inline bool fast_parse_input(const char *text, size_t len, uint64_t Q, FastInput &in) {
  struct Span {
    const char *key;
    size_t key_len, lo, hi;
  };
  Span spans[16];
  int n_spans = 0;
  size_t i = 0;
  auto skip = [&]() { while (i < len && isspace((unsigned char)text[i])) ++i; };
  skip();
  if (i >= len || text[i] != '{') return false;
  ++i;
  skip();
  while (i < len && text[i] != '}') {
    if (text[i] != '"' || n_spans == 16) return false;
    const char *q = (const char *)memchr(text + i + 1, '"', len - i - 1);
    if (!q) return false;
    Span sp;
    sp.key = text + i + 1;
    sp.key_len = (size_t)(q - sp.key);
    i = (size_t)(q - text) + 1;
    skip();
    if (i >= len || text[i] != ':') return false;
    ++i;
    skip();
    if (i >= len || text[i] != '[') return false;
    ++i;
    sp.lo = i;
    const char *c = (const char *)memchr(text + i, ']', len - i);
    if (!c) return false;
    sp.hi = (size_t)(c - text);
    i = sp.hi + 1;
    spans[n_spans++] = sp;
    skip();
    if (i < len && text[i] == ',') ++i;
    skip();
  }
  if (i >= len) return false;
  static const char *names[9] = {"pk0", "pk1", "m", "u", "e0", "e1", "c0", "c1", "cyclo"};
  const Span *src[9];
  for (int k = 0; k < 9; ++k) {
    src[k] = nullptr;
    for (int s = 0; s < n_spans; ++s)
      if (spans[s].key_len == strlen(names[k]) && !memcmp(spans[s].key, names[k], spans[s].key_len)) src[k] = &spans[s];   // the last one counts
    if (!src[k]) return false;
  }
  for (int s = 0; s < n_spans; ++s) {   // a field the circuit does not read: let the generic parser check its form
    bool used = false;
    for (int k = 0; k < 9; ++k) used = used || src[k] == &spans[s];
    if (!used) return false;
  }
  bool ok[9];
  auto parse_array = [&](int k) {
    std::vector<uint64_t> &out = in.a[k];
    const char *p = text + src[k]->lo, *end = text + src[k]->hi;
    size_t commas = 0;
    for (const char *c = p; c < end; ++c) commas += *c == ',';
    out.clear();
    out.reserve(commas + 1);
    ok[k] = false;
    auto skipw = [&]() { while (p < end && isspace((unsigned char)*p)) ++p; };
    skipw();
    while (p < end) {
      if (*p != '"') return;
      ++p;
      uint64_t v = 0;
      int digits = 0;
      while (p < end && *p >= '0' && *p <= '9') {
        if (++digits > 19) return;   // 10^19 - 1 < 2^64: no overflow up to here
        v = v * 10 + (uint64_t)(*p - '0');
        ++p;
      }
      if (!digits || p >= end || *p != '"' || v > Q) return;   // `coeff <= modulus` (src/poly.rs:28): note <=
      ++p;
      out.push_back(v);
      skipw();
      if (p < end && *p == ',') {
        ++p;
        skipw();
        if (p >= end) return;   // trailing comma
      }
    }
    ok[k] = !out.empty();
  };
  if (len < ((size_t)1 << 18)) {
    for (int k = 0; k < 9; ++k) parse_array(k);
  } else {
    std::thread th[9];
    for (int k = 0; k < 9; ++k) th[k] = std::thread(parse_array, k);
    for (auto &t : th) t.join();
  }
  for (int k = 0; k < 9; ++k)
    if (!ok[k]) return false;
  return true;
}

inline bool phase0_force_generic() {
  const char *e = getenv("ZKFHE_PHASE0");   // read per call: the tests switch paths inside one process
  return e && strcmp(e, "generic") == 0;
}

// See the header comment.  ctx must be empty and in prover mode (values only).
inline bool bfv_phase0_fast(Context &ctx, const char *text, size_t text_len, const BfvParams &prm, std::vector<Cell> &make_public, BfvState &st,
                            const std::function<void(const std::vector<Cell> &)> &on_public = nullptr) {
  typedef unsigned __int128 u128;
  const size_t N = prm.N;
  const uint64_t Q = prm.Q;
  if (phase0_force_generic() || ctx.record_structure || !ctx.advice.empty() || N < 2 || (N & (N - 1)) || Q < 2 || (Q >> 63) || prm.T == 0) return false;
  FastInput in;
  if (!fast_parse_input(text, text_len, Q, in)) return false;
  for (int k = 0; k < 8; ++k)
    if (in.a[k].size() != N) return false;
  const std::vector<uint64_t> &cy = in.a[8];
  if (cy.size() != N + 1 || cy[0] != 1 || cy[N] != 1) return false;
  for (size_t i = 1; i < N; ++i)
    if (cy[i]) return false;
  // products pk_i * u first (they may fail for want of a backend, and nothing is to be written before success is certain)
  const std::vector<uint64_t> &pk0 = in.a[0], &pk1 = in.a[1], &u = in.a[3];
  const size_t L = 2 * N - 1;
  std::vector<U256> prod[2];   // canonical integers below 2^192
  {
    bool narrow = N <= 2048;
    for (size_t i = 0; i < N && narrow; ++i) narrow = !((pk0[i] | pk1[i] | u[i]) >> 32);
    if (narrow) {
      std::vector<uint64_t> lo, hi;
      for (int s = 0; s < 2; ++s) {
        gl::poly_mul_u32(s ? pk1 : pk0, u, lo, hi);
        if (lo.size() != L) return false;
        prod[s].resize(L);
        for (size_t i = 0; i < L; ++i) prod[s][i] = U256{{lo[i], hi[i], 0, 0}};
      }
    } else {
      PolyMulBackend *be = poly_mul_backend();
      if (!be) return false;
      for (int s = 0; s < 2; ++s) {
        const U256 *raw = be->mul_u64_raw(s ? pk1 : pk0, u);
        if (!raw) return false;
        prod[s].assign(raw, raw + L);
        for (size_t i = 0; i < L; ++i)
          if (prod[s][i].l[3]) return false;
      }
    }
  }
  // per side: d = (pk_i u) mod Q, quotient, remainder, quotient * cyclo
  std::vector<uint64_t> quo[2], rem[2], qc[2];
  bool side_ok[2] = {false, false};
  auto side = [&](int s) {
    std::vector<uint64_t> d(L);
    bool any = false;
    for (size_t i = 0; i < L; ++i) {
      const U256 &v = prod[s][i];
      uint64_t r = v.l[2] % Q;
      r = (uint64_t)((((u128)r << 64) | v.l[1]) % Q);
      d[i] = (uint64_t)((((u128)r << 64) | v.l[0]) % Q);
      any = any || d[i];
    }
    quo[s].assign(N + 1, 0), rem[s].assign(2 * N + 1, 0), qc[s].assign(2 * N + 1, 0);
    if (any) {   // the all-zero dividend has its own branch in the reference (src/poly.rs:118-123): all-zero quotient and remainder
      bool qnz = false;
      for (size_t i = 0; i + 1 < N; ++i) {
        quo[s][2 + i] = d[i];
        qnz = qnz || d[i];
      }
      if (!qnz) return;   // empty quotient after the strip: the reference underflows (src/poly.rs:158) -- the generic path reports it
      for (size_t j = 0; j < N; ++j) {
        const uint64_t hi = d[N - 1 + j], lo = j ? d[j - 1] : 0;
        rem[s][N + 1 + j] = hi >= lo ? hi - lo : hi + Q - lo;   // mod_floor Q of a value in (-Q, Q)
      }
    }
    for (size_t i = 0; i <= N; ++i) {   // q x^N + q, big-endian: coefficient i of q lands at i and at i + N
      qc[s][i] += quo[s][i];
      qc[s][i + N] += quo[s][i];
    }
    side_ok[s] = true;
  };
  if (N >= 4096) {
    std::thread t1(side, 1);
    side(0);
    t1.join();
  } else {
    side(0), side(1);
  }
  if (!side_ok[0] || !side_ok[1]) return false;

  // ---- from here on nothing fails: the cell stream, in the order of examples/bfv.rs:101-165 ------------------------------------
  const uint64_t qbits = bits_u64(Q), lgN = log2_ceil(N), lgN1 = log2_ceil(N + 1);
  const uint64_t mul_bits = 2 * qbits + lgN, qc_bits = 2 * qbits + lgN1;   // Poly::mul: a.max_bits + b.max_bits + log2_ceil(deg + 1)
  struct Seg {
    PolyChip *chip;
    const uint64_t *lo;   // 64-bit values ...
    const U256 *wide;     // ... or whole words (the products)
    size_t len, off;
    uint64_t bits;
  };
  Seg seg[17] = {
      {&st.pk0, in.a[0].data(), nullptr, N, 0, qbits},          {&st.pk1, in.a[1].data(), nullptr, N, 0, qbits},
      {&st.m, in.a[2].data(), nullptr, N, 0, qbits},            {&st.u, in.a[3].data(), nullptr, N, 0, qbits},
      {&st.e0, in.a[4].data(), nullptr, N, 0, qbits},           {&st.e1, in.a[5].data(), nullptr, N, 0, qbits},
      {&st.expected_c0, in.a[6].data(), nullptr, N, 0, qbits},  {&st.expected_c1, in.a[7].data(), nullptr, N, 0, qbits},
      {&st.cyclo, in.a[8].data(), nullptr, N + 1, 0, qbits},
      // (the constant delta sits here in the stream)
      {&st.pk0_u, nullptr, prod[0].data(), L, 0, mul_bits},    {&st.pk1_u, nullptr, prod[1].data(), L, 0, mul_bits},
      {&st.quotient_0, quo[0].data(), nullptr, N + 1, 0, qbits}, {&st.quotient_1, quo[1].data(), nullptr, N + 1, 0, qbits},
      {&st.quotient_0_times_cyclo, qc[0].data(), nullptr, 2 * N + 1, 0, qc_bits}, {&st.quotient_1_times_cyclo, qc[1].data(), nullptr, 2 * N + 1, 0, qc_bits},
      {&st.remainder_0, rem[0].data(), nullptr, 2 * N + 1, 0, qbits}, {&st.remainder_1, rem[1].data(), nullptr, 2 * N + 1, 0, qbits}};
  size_t total = 0;
  for (int k = 0; k < 17; ++k) {
    if (k == 9) ++total;   // delta
    seg[k].off = total;
    total += seg[k].len;
  }
  ctx.advice.resize(total);
  auto fill = [&](int k) {
    const Seg &sg = seg[k];
    std::vector<Cell> &cells = sg.chip->assigned_coefficients;
    cells.resize(sg.len);
    U256 *adv = ctx.advice.data() + sg.off;
    for (size_t i = 0; i < sg.len; ++i) {
      const U256 v = sg.wide ? sg.wide[i] : U256{{sg.lo[i], 0, 0, 0}};
      adv[i] = v;
      cells[i] = Cell{{ctx.cid, (uint32_t)(sg.off + i)}, v};
    }
    sg.chip->max_num_bits = sg.bits;
    sg.chip->degree = sg.len - 1;
  };
  const bool threaded = N >= 4096;
  for (int k = 0; k < 9; ++k) fill(k);
  const size_t delta_off = seg[9].off - 1;
  ctx.advice[delta_off] = fe::from_u64(Q / prm.T);
  st.delta = Cell{{ctx.cid, (uint32_t)delta_off}, ctx.advice[delta_off]};
  make_public.clear();
  make_public.reserve(4 * N + N + 1);
  st.pk0.to_public(make_public);
  st.pk1.to_public(make_public);
  st.expected_c0.to_public(make_public);
  st.expected_c1.to_public(make_public);
  st.cyclo.to_public(make_public);
  if (on_public) on_public(make_public);
  if (threaded) {
    std::thread th[7];
    for (int k = 10; k < 17; ++k) th[k - 10] = std::thread(fill, k);
    fill(9);
    for (auto &t : th) t.join();
  } else {
    for (int k = 9; k < 17; ++k) fill(k);
  }
  return true;
}

}  // namespace zkhost
