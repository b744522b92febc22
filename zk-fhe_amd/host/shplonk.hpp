// Opening bookkeeping shared by the prover (prove.hip) and the verifier (verifier.cpp): which polynomial is opened at
// which rotation, in which order the evaluations are written, and how halo2's SHPLONK groups them.
//
// Restates (third-party halo2_proofs, reached from reference examples/bfv.rs:311; mirrored by oracle/halo2_ref.py
// `open_queries` / `intermediate_sets`):
//   * plonk/prover.rs create_proof: evaluations are written as advice, fixed, random polynomial, sigma, permutation
//     products, lookups; the multi-open QUERY list is advice, permutation products, lookups, fixed, sigma, h(X), random;
//   * poly/kzg/multiopen/shplonk.rs construct_intermediate_sets: commitments in order of first appearance, each with the
//     set of its points; distinct point sets in order of first appearance, each with its commitments in that order.
// Points are identified by a rotation id; two ids never name the same point here (w^u != w^-1 because blinding rows exist).
#pragma once
#include <cstddef>
#include <utility>
#include <vector>

#include "bfv_circuit.hpp"

namespace zkhost {

enum RotId { ROT_0 = 0, ROT_1 = 1, ROT_2 = 2, ROT_3 = 3, ROT_LAST = 4, ROT_PREV = 5, N_ROT_IDS = 6 };  // w^0..w^3, w^u, w^-1

// Items in evaluation-write order: advice columns, fixed columns, H (not written), random polynomial, sigma columns,
// permutation products, then per lookup its product, permuted input, permuted table.
struct OpenLayout {
  size_t adv0, fixed0, H, rand, sigma0, pz0, lk0, count;
  std::vector<std::vector<int>> rots;  // per item, in write order
  explicit OpenLayout(const CircuitConfig &cfg) {
    adv0 = 0;
    fixed0 = adv0 + cfg.n_advice();
    H = fixed0 + cfg.n_fixed();
    rand = H + 1;
    sigma0 = rand + 1;
    pz0 = sigma0 + cfg.n_perm();
    lk0 = pz0 + cfg.n_chunks();
    count = lk0 + 3 * (size_t)cfg.n_lookup;
    rots.resize(count);
    for (unsigned c = 0; c < cfg.n_advice(); ++c)
      rots[adv0 + c] = c < cfg.n_gate() ? std::vector<int>{0, 1, 2, 3} : (c < cfg.adv_rlc0() ? std::vector<int>{0} : std::vector<int>{0, 1, 2});
    for (unsigned c = 0; c < cfg.n_fixed(); ++c) rots[fixed0 + c] = {0};
    rots[H] = {0};
    rots[rand] = {0};
    for (unsigned c = 0; c < cfg.n_perm(); ++c) rots[sigma0 + c] = {0};
    for (unsigned j = 0; j < cfg.n_chunks(); ++j)
      rots[pz0 + j] = j + 1 != cfg.n_chunks() ? std::vector<int>{ROT_0, ROT_1, ROT_LAST} : std::vector<int>{ROT_0, ROT_1};
    for (unsigned i = 0; i < cfg.n_lookup; ++i) {
      rots[lk0 + 3 * i] = {ROT_0, ROT_1};         // product z
      rots[lk0 + 3 * i + 1] = {ROT_0, ROT_PREV};  // permuted input
      rots[lk0 + 3 * i + 2] = {ROT_0};            // permuted table
    }
  }
  int eval_slot(size_t item, int rot) const {  // position of `rot` in the item's written evaluations
    for (size_t t = 0; t < rots[item].size(); ++t)
      if (rots[item][t] == rot) return (int)t;
    return -1;
  }
};

struct OpenSet {
  std::vector<int> rots;        // the point set (rotation ids, ascending id)
  std::vector<size_t> members;  // item indices, in commitment order
};

inline std::vector<std::pair<size_t, int>> open_queries(const CircuitConfig &cfg, const OpenLayout &L) {
  std::vector<std::pair<size_t, int>> q;
  for (unsigned c = 0; c < cfg.n_advice(); ++c)
    for (int r : L.rots[L.adv0 + c]) q.push_back({L.adv0 + c, r});
  // permutation::prover::Evaluated::open: every product at x and w x, then at w^last x for all but the last set, in reverse
  for (unsigned j = 0; j < cfg.n_chunks(); ++j) {
    q.push_back({L.pz0 + j, ROT_0});
    q.push_back({L.pz0 + j, ROT_1});
  }
  for (unsigned j = cfg.n_chunks() - 1; j-- > 0;) q.push_back({L.pz0 + j, ROT_LAST});
  // lookup::prover::Evaluated::open: product(x), input(x), table(x), input(w^-1 x), product(w x)
  for (unsigned i = 0; i < cfg.n_lookup; ++i) {
    const size_t z = L.lk0 + 3 * i, a = z + 1, s = z + 2;
    q.push_back({z, ROT_0});
    q.push_back({a, ROT_0});
    q.push_back({s, ROT_0});
    q.push_back({a, ROT_PREV});
    q.push_back({z, ROT_1});
  }
  for (unsigned c = 0; c < cfg.n_fixed(); ++c) q.push_back({L.fixed0 + c, ROT_0});
  for (unsigned c = 0; c < cfg.n_perm(); ++c) q.push_back({L.sigma0 + c, ROT_0});
  q.push_back({L.H, ROT_0});
  q.push_back({L.rand, ROT_0});
  return q;
}

inline std::vector<OpenSet> intermediate_sets(const OpenLayout &L, const std::vector<std::pair<size_t, int>> &queries, std::vector<int> &super_set) {
  std::vector<size_t> order;                   // commitments by first appearance
  std::vector<unsigned> mask(L.count, 0);      // their point sets
  unsigned all = 0;
  for (const auto &qr : queries) {
    if (!mask[qr.first]) order.push_back(qr.first);
    mask[qr.first] |= 1u << qr.second;
    all |= 1u << qr.second;
  }
  std::vector<OpenSet> sets;
  std::vector<unsigned> keys;
  for (size_t item : order) {
    size_t s = 0;
    for (; s < keys.size(); ++s)
      if (keys[s] == mask[item]) break;
    if (s == keys.size()) {
      keys.push_back(mask[item]);
      OpenSet os;
      for (int r = 0; r < N_ROT_IDS; ++r)
        if (mask[item] >> r & 1) os.rots.push_back(r);
      sets.push_back(os);
    }
    sets[s].members.push_back(item);
  }
  super_set.clear();
  for (int r = 0; r < N_ROT_IDS; ++r)
    if (all >> r & 1) super_set.push_back(r);
  return sets;
}

}  // namespace zkhost
