// `verify` (reference README.md:48-52: halo2_proofs verify_proof + VerifierSHPLONK + one pairing check), on the
// host CPU like the reference's.  Mirrors oracle/halo2_ref.py `verify` (same transcript, same expression order,
// same SHPLONK combination) and is tested against oracle-made and GPU-made proofs.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/zkfhe.h"
#include "bfv_circuit.hpp"
#include "pairing.hpp"
#include "shplonk.hpp"
#include "srs_secret.hpp"
#include "transcript.hpp"
#include "vk.hpp"

using namespace zkhost;
using zk::Fr;

zk::Fr zk_fr_root_of_unity(int log_n);  // csrc/core.hip
// the SRS side (srs.hip) borrows the host G2 arithmetic of this file; declared in prover_internal.hpp as well
void zk_srs_g2_from_secret(const U256 &s, uint8_t g2_raw[128], uint8_t sg2_raw[128]);
bool zk_g2_raw_to_canon(const uint8_t raw[128], uint8_t canon[128]);
bool zk_g2_canon_to_raw(const uint8_t canon[128], uint8_t raw[128]);

namespace {

static const uint64_t DELTA_CANON_V[4] = {0x870e56bbe533e9a2ULL, 0x5b5f898e5e963f25ULL, 0x64ec26aad4c86e71ULL, 0x09226b6e22c6f0caULL};

Fr M(const U256 &v) { return fe::to_mont(v); }
Fr Mu(uint64_t v) { return fe::to_mont(fe::from_u64(v)); }
Fr fpow(Fr b, uint64_t e) {
  Fr r = Fr::one();
  while (e) {
    if (e & 1) r = r * b;
    b = b * b;
    e >>= 1;
  }
  return r;
}
Fr finv(const Fr &a) { return zk::fp_inv<zk::FrP>(a); }

struct Reader {
  const uint8_t *p;
  size_t len, pos = 0;
  Transcript tr;
  Reader(uint32_t kind, const uint8_t *d, size_t l) : p(d), len(l), tr(kind) {}
  void common_scalar(const U256 &s) { tr.common_scalar(s); }
  AffinePoint read_point() {
    if (pos + 32 > len) throw std::runtime_error("proof truncated");
    uint8_t b[32];
    memcpy(b, p + pos, 32);
    pos += 32;
    AffinePoint a;
    const ptenc::Layout &L = ptenc::layout();
    if (ptenc::is_identity_encoding(b)) {
      // the identity has exactly one encoding (halo2curves rejects anything else)
      for (int i = 0; i < 31; ++i)
        if (b[i]) throw std::runtime_error("non-canonical encoding of the identity");
      if (b[31] != L.identity_bit) throw std::runtime_error("non-canonical encoding of the identity");
      a.x = fe::zero();
      a.y = fe::zero();
    } else {
      const unsigned sign = (b[31] & L.sign_bit) ? 1u : 0u;
      b[31] &= L.x_mask;
      memcpy(a.x.l, b, 32);
      static const U256 QMOD = {{0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}};
      if (!(a.x < QMOD)) throw std::runtime_error("point x not reduced");
      const zk::Fq x = pairing::fq_from_canon(a.x);
      const zk::Fq y2 = x * x * x + pairing::fq_from_u64(3);
      // sqrt: y = y2^((q+1)/4)
      uint32_t e[8];
      U256 ee;
      fe::add_raw(ee, QMOD, fe::one());
      for (int i = 0; i < 4; ++i) ee.l[i] = (ee.l[i] >> 2) | (i < 3 ? ee.l[i + 1] << 62 : 0);
      memcpy(e, ee.l, 32);
      zk::Fq y = zk::fp_pow<zk::FqP>(y2, e);
      if (!(y * y == y2)) throw std::runtime_error("point not on curve");
      zk::Fq yc = zk::fp_from_mont<zk::FqP>(y);
      if ((yc.l[0] & 1u) != sign) y = zk::fp_neg<zk::FqP>(y);
      yc = zk::fp_from_mont<zk::FqP>(y);
      memcpy(a.y.l, yc.l, 32);
    }
    tr.common_point(a);   // Poseidon: refuses the identity, as snark-verifier does
    return a;
  }
  U256 read_scalar() {
    if (pos + 32 > len) throw std::runtime_error("proof truncated");
    U256 s;
    memcpy(s.l, p + pos, 32);
    pos += 32;
    if (!(s < fe::MOD)) throw std::runtime_error("scalar not reduced");
    tr.common_scalar(s);
    return s;
  }
  U256 squeeze() { return tr.squeeze(); }
};

// host MSM over G1 (affine canonical in, XYZZ accumulate): small Pippenger, c = 8
zk::G1Affine to_dev_affine(const AffinePoint &p) {
  zk::G1Affine a;
  a.x = pairing::fq_from_canon(p.x);
  a.y = pairing::fq_from_canon(p.y);
  return a;
}
AffinePoint msm_host(const std::vector<Fr> &scalars, const std::vector<AffinePoint> &pts) {
  const int c = 8, W = 32;
  std::vector<U256> k(scalars.size());
  for (size_t i = 0; i < scalars.size(); ++i) k[i] = fe::from_mont(scalars[i]);
  std::vector<zk::G1Affine> base(pts.size());
  for (size_t i = 0; i < pts.size(); ++i) base[i] = to_dev_affine(pts[i]);
  zk::G1X total = zk::G1X::identity();
  for (int w = W - 1; w >= 0; --w) {
    for (int d = 0; d < c; ++d) total = zk::g1x_dbl(total);
    std::vector<zk::G1X> buckets(255, zk::G1X::identity());
    for (size_t i = 0; i < k.size(); ++i) {
      const unsigned dgt = (unsigned)((k[i].l[(w * c) >> 6] >> ((w * c) & 63)) & 0xff);
      if (dgt) zk::g1x_add_affine(buckets[dgt - 1], base[i], false);
    }
    zk::G1X run = zk::G1X::identity(), sum = zk::G1X::identity();
    for (int b = 254; b >= 0; --b) {
      zk::g1x_add(run, buckets[b]);
      zk::g1x_add(sum, run);
    }
    zk::g1x_add(total, sum);
  }
  const zk::G1Affine r = zk::g1x_to_affine(total);
  AffinePoint out;
  const zk::Fq x = zk::fp_from_mont<zk::FqP>(r.x), y = zk::fp_from_mont<zk::FqP>(r.y);
  memcpy(out.x.l, x.l, 32);
  memcpy(out.y.l, y.l, 32);
  return out;
}

struct Vk {
  CircuitConfig cfg;
  std::vector<AffinePoint> fixed_commit, sigma_commit;
  U256 digest;
};

const char VK_MAGIC[8] = {'Z', 'K', 'F', 'H', 'E', 'V', 'K', '2'};

Vk parse_vk(const uint8_t *d, size_t len) {
  if (len < 8 + 10 * 4 + 32 || memcmp(d, VK_MAGIC, 8)) throw std::runtime_error("not a zkfhe vk file (ZKFHEVK2)");
  uint32_t h[10];
  memcpy(h, d + 8, 40);
  Vk vk;
  vk.cfg.k = h[0];
  vk.cfg.n_gate0 = h[1];
  vk.cfg.n_gate1 = h[2];
  vk.cfg.n_lookup = h[3];
  vk.cfg.n_rlc = h[4];
  vk.cfg.unusable_rows = h[5];
  vk.cfg.lookup_bits = h[6];
  vk.cfg.transcript = h[7];
  // bounds first: n(), u() and the root of unity are only defined for these (2-adicity of Fr is 28)
  if (vk.cfg.k < 3 || vk.cfg.k > 28) throw std::runtime_error("vk: k out of range");
  if (vk.cfg.unusable_rows < 4 || vk.cfg.unusable_rows >= vk.cfg.n()) throw std::runtime_error("vk: unusable_rows out of range");
  if (vk.cfg.n_gate0 > 4096 || vk.cfg.n_gate1 > 4096 || vk.cfg.n_lookup > 4096 || vk.cfg.n_rlc > 4096 || vk.cfg.n_gate() == 0)
    throw std::runtime_error("vk: column counts out of range");
  if (vk.cfg.lookup_bits == 0 || vk.cfg.lookup_bits > 20 || ((size_t)1 << vk.cfg.lookup_bits) > vk.cfg.max_rows()) throw std::runtime_error("vk: lookup_bits out of range");
  if (vk.cfg.transcript > TR_BLAKE2B) throw std::runtime_error("vk: unknown transcript kind");
  const size_t nf = h[8], ns = h[9];
  if (nf != vk.cfg.n_fixed() || ns != vk.cfg.n_perm()) throw std::runtime_error("vk commitment counts do not match its configuration");
  if (len != 8 + 40 + 32 + 64 * (nf + ns)) throw std::runtime_error("vk file has the wrong size");
  memcpy(vk.digest.l, d + 48, 32);
  const uint8_t *p = d + 80;
  static const U256 QMOD = {{0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}};
  auto rd = [&](std::vector<AffinePoint> &v, size_t cnt) {
    v.resize(cnt);
    for (size_t i = 0; i < cnt; ++i, p += 64) {
      memcpy(v[i].x.l, p, 32);
      memcpy(v[i].y.l, p + 32, 32);
      if (!(v[i].x < QMOD) || !(v[i].y < QMOD)) throw std::runtime_error("vk: commitment coordinate not reduced");
    }
  };
  rd(vk.fixed_commit, nf);
  rd(vk.sigma_commit, ns);
  // the digest is what the transcript starts from: it must be the digest OF this configuration and these commitments
  if (!(vk_digest(vk.cfg, vk.fixed_commit, vk.sigma_commit) == vk.digest)) throw std::runtime_error("vk digest does not match its contents");
  return vk;
}

struct Item {  // one opened polynomial
  int kind;    // 0 = commitment point, 1 = H (combination of the quotient pieces), 2 = generator
  AffinePoint commit;
  std::vector<int> rots;  // 0,1,2,3 ; 4 = last ; 5 = -1
  std::vector<Fr> evals;
};

// the verifier's half of an SRS: G2 and s*G2 (the library's own setups derive s from a seed; an external ceremony supplies them)
struct SrsG2 {
  pairing::Pt<pairing::Fq2> g2, sg2;
};
SrsG2 srs_g2_from_seed(const uint8_t *srs_seed, size_t seed_len) {
  const U256 s = srs_secret(srs_seed, seed_len);
  SrsG2 r;
  r.g2 = pairing::g2_generator();
  r.sg2 = pairing::ec_mul(r.g2, s);
  return r;
}

bool verify_impl(const Vk &vk, const std::vector<U256> &inst, const uint8_t *proof, size_t proof_len, const SrsG2 &srs) {
  const CircuitConfig &cfg = vk.cfg;
  const size_t n = cfg.n(), u = cfg.u();
  const Fr w = zk_fr_root_of_unity((int)cfg.k);
  // halo2 verify_proof: Error::InstanceTooLarge.  Without it L_{i+n} = L_i would let value move between instance i and i+n.
  if (inst.size() > u) throw std::runtime_error("more instances than usable rows");
  Reader tr(cfg.transcript, proof, proof_len);
  tr.common_scalar(vk.digest);
  for (const U256 &v : inst) tr.common_scalar(v);
  std::vector<AffinePoint> adv_commit;
  for (unsigned c = 0; c < cfg.n_gate0; ++c) adv_commit.push_back(tr.read_point());
  const Fr gamma_rlc = M(tr.squeeze());
  for (unsigned c = cfg.n_gate0; c < cfg.n_advice(); ++c) adv_commit.push_back(tr.read_point());
  tr.squeeze();  // theta
  std::vector<AffinePoint> la_commit, ls_commit;
  for (unsigned i = 0; i < cfg.n_lookup; ++i) {
    la_commit.push_back(tr.read_point());
    ls_commit.push_back(tr.read_point());
  }
  const Fr beta = M(tr.squeeze()), gamma = M(tr.squeeze());
  std::vector<AffinePoint> pz_commit, lz_commit;
  for (unsigned j = 0; j < cfg.n_chunks(); ++j) pz_commit.push_back(tr.read_point());
  for (unsigned i = 0; i < cfg.n_lookup; ++i) lz_commit.push_back(tr.read_point());
  const AffinePoint rand_commit = tr.read_point();
  const Fr y = M(tr.squeeze());
  AffinePoint h_commit[3];
  for (auto &p : h_commit) p = tr.read_point();
  const Fr x = M(tr.squeeze());
  // ---- evaluations, in the prover's write order
  std::vector<Item> items;
  std::map<std::pair<int, int>, Fr> ev_adv, ev_pz, ev_lz, ev_la, ev_ls;
  std::vector<Fr> ev_fixed(cfg.n_fixed()), ev_sigma(cfg.n_perm());
  auto read_item = [&](const AffinePoint &cm, std::vector<int> rots) {
    Item it;
    it.kind = 0;
    it.commit = cm;
    it.rots = rots;
    for (size_t r = 0; r < rots.size(); ++r) it.evals.push_back(M(tr.read_scalar()));
    items.push_back(it);
    return items.back().evals;
  };
  for (unsigned c = 0; c < cfg.n_advice(); ++c) {
    std::vector<int> rots = c < cfg.n_gate() ? std::vector<int>{0, 1, 2, 3} : (c < cfg.adv_rlc0() ? std::vector<int>{0} : std::vector<int>{0, 1, 2});
    const auto e = read_item(adv_commit[c], rots);
    for (size_t r = 0; r < rots.size(); ++r) ev_adv[{(int)c, rots[r]}] = e[r];
  }
  for (unsigned c = 0; c < cfg.n_fixed(); ++c) ev_fixed[c] = read_item(vk.fixed_commit[c], {0})[0];
  const size_t h_slot = items.size();
  {
    Item it;
    it.kind = 1;
    it.rots = {0};
    it.evals = {Fr::zero()};
    items.push_back(it);
  }
  read_item(rand_commit, {0});
  for (unsigned c = 0; c < cfg.n_perm(); ++c) ev_sigma[c] = read_item(vk.sigma_commit[c], {0})[0];
  for (unsigned j = 0; j < cfg.n_chunks(); ++j) {
    std::vector<int> rots = j + 1 != cfg.n_chunks() ? std::vector<int>{0, 1, 4} : std::vector<int>{0, 1};
    const auto e = read_item(pz_commit[j], rots);
    for (size_t r = 0; r < rots.size(); ++r) ev_pz[{(int)j, rots[r]}] = e[r];
  }
  for (unsigned i = 0; i < cfg.n_lookup; ++i) {
    auto e = read_item(lz_commit[i], {0, 1});
    ev_lz[{(int)i, 0}] = e[0];
    ev_lz[{(int)i, 1}] = e[1];
    e = read_item(la_commit[i], {0, 5});
    ev_la[{(int)i, 0}] = e[0];
    ev_la[{(int)i, 5}] = e[1];
    ev_ls[{(int)i, 0}] = read_item(ls_commit[i], {0})[0];
  }
  // ---- Lagrange values and the instance evaluation at x
  const Fr xn = fpow(x, n), zh = xn - Fr::one(), ninv = finv(Mu(n));
  auto lagr_many = [&](size_t first, size_t count) {  // L_i(x) = w^i zh / (n (x - w^i)), batch inverted
    std::vector<Fr> den(count), wi(count);
    Fr cur = fpow(w, first);
    for (size_t i = 0; i < count; ++i) {
      wi[i] = cur;
      den[i] = x - cur;
      cur = cur * w;
    }
    std::vector<Fr> pre(count);
    Fr acc = Fr::one();
    for (size_t i = 0; i < count; ++i) {
      pre[i] = acc;
      acc = acc * den[i];
    }
    acc = finv(acc);
    std::vector<Fr> out(count);
    for (size_t i = count; i-- > 0;) {
      out[i] = wi[i] * zh * ninv * (acc * pre[i]);
      acc = acc * den[i];
    }
    return out;
  };
  Fr inst_x = Fr::zero();
  if (!inst.empty()) {
    const auto L = lagr_many(0, inst.size());
    for (size_t i = 0; i < inst.size(); ++i) inst_x = inst_x + M(inst[i]) * L[i];
  }
  const Fr l0 = lagr_many(0, 1)[0], llast = lagr_many(u, 1)[0];
  Fr lblind = Fr::zero();
  for (const Fr &v : lagr_many(u + 1, n - u - 1)) lblind = lblind + v;
  const Fr lactive = Fr::one() - llast - lblind;
  // ---- fold every constraint expression with y (order of oracle/halo2_ref.py expressions_at)
  Fr acc = Fr::zero();
  auto fold = [&](const Fr &e) { acc = acc * y + e; };
  for (unsigned j = 0; j < cfg.n_gate(); ++j)
    fold(ev_fixed[j] * (ev_adv[{(int)j, 0}] + ev_adv[{(int)j, 1}] * ev_adv[{(int)j, 2}] - ev_adv[{(int)j, 3}]));
  for (unsigned j = 0; j < cfg.n_rlc; ++j) {
    const int col = (int)(cfg.adv_rlc0() + j);
    fold(ev_fixed[cfg.fix_qrlc0() + j] * (ev_adv[{col, 0}] * gamma_rlc + ev_adv[{col, 1}] - ev_adv[{col, 2}]));
  }
  const int m = (int)cfg.n_chunks() - 1;
  const Fr one = Fr::one();
  fold(l0 * (one - ev_pz[{0, 0}]));
  fold(llast * (ev_pz[{m, 0}] * ev_pz[{m, 0}] - ev_pz[{m, 0}]));
  for (int j = 1; j <= m; ++j) fold(l0 * (ev_pz[{j, 0}] - ev_pz[{j - 1, 4}]));
  U256 dc;
  memcpy(dc.l, DELTA_CANON_V, 32);
  const Fr delta = M(dc);
  std::vector<Fr> dpow(cfg.n_perm());
  {
    Fr cur = Fr::one();
    for (auto &d : dpow) {
      d = cur;
      cur = cur * delta;
    }
  }
  auto permcol = [&](unsigned c) { return c < cfg.n_advice() ? ev_adv[{(int)c, 0}] : (c == cfg.perm_const() ? ev_fixed[cfg.fix_const()] : inst_x); };
  for (int j = 0; j <= m; ++j) {
    Fr left = ev_pz[{j, 1}], right = ev_pz[{j, 0}];
    for (unsigned c = j * cfg.chunk(); c < (j + 1) * cfg.chunk() && c < cfg.n_perm(); ++c) {
      const Fr v = permcol(c);
      left = left * (v + beta * ev_sigma[c] + gamma);
      right = right * (v + beta * dpow[c] * x + gamma);
    }
    fold(lactive * (left - right));
  }
  for (unsigned i = 0; i < cfg.n_lookup; ++i) {
    const Fr z0 = ev_lz[{(int)i, 0}], z1 = ev_lz[{(int)i, 1}], a = ev_adv[{(int)(cfg.adv_lookup0() + i), 0}], s = ev_fixed[cfg.fix_table()];
    const Fr ap = ev_la[{(int)i, 0}], apm = ev_la[{(int)i, 5}], sp = ev_ls[{(int)i, 0}];
    fold(l0 * (one - z0));
    fold(llast * (z0 * z0 - z0));
    fold(lactive * (z1 * ((ap + beta) * (sp + gamma)) - z0 * ((a + beta) * (s + gamma))));
    fold(l0 * (ap - sp));
    fold(lactive * ((ap - sp) * (ap - apm)));
  }
  items[h_slot].evals[0] = acc * finv(zh);
  // ---- SHPLONK (halo2 VerifierSHPLONK::verify_proof)
  const Fr yq = M(tr.squeeze()), v = M(tr.squeeze());
  const AffinePoint h1 = tr.read_point();
  const Fr uu = M(tr.squeeze());
  const AffinePoint h2 = tr.read_point();
  if (tr.pos != proof_len) throw std::runtime_error("trailing bytes in proof");
  Fr pts_rot[N_ROT_IDS];
  pts_rot[ROT_0] = x;
  pts_rot[ROT_1] = x * w;
  pts_rot[ROT_2] = pts_rot[1] * w;
  pts_rot[ROT_3] = pts_rot[2] * w;
  pts_rot[ROT_LAST] = x * fpow(w, u);
  pts_rot[ROT_PREV] = x * finv(w);
  const OpenLayout layout(cfg);
  if (items.size() != layout.count || h_slot != layout.H) throw std::logic_error("opening layout mismatch");
  std::vector<int> all_rots;
  const std::vector<OpenSet> sets = intermediate_sets(layout, open_queries(cfg, layout), all_rots);
  std::vector<Fr> scal;
  std::vector<AffinePoint> pts;
  Fr r_outer = Fr::zero(), vj = Fr::one(), z_0 = Fr::one(), z_0_diff_inv = Fr::one();
  for (size_t j = 0; j < sets.size(); ++j) {
    const auto &rots = sets[j].rots;
    Fr zdiff = Fr::one();
    for (int r : all_rots)
      if (std::find(rots.begin(), rots.end(), r) == rots.end()) zdiff = zdiff * (uu - pts_rot[r]);
    if (j == 0) {
      // normalised by the first set: its own vanishing polynomial multiplies h1, everything else is divided by z_diff_0
      for (int r : rots) z_0 = z_0 * (uu - pts_rot[r]);
      z_0_diff_inv = finv(zdiff);
      zdiff = Fr::one();
    } else {
      zdiff = zdiff * z_0_diff_inv;
    }
    const Fr coef = vj * zdiff;
    std::vector<Fr> comb(rots.size(), Fr::zero());
    Fr pw = Fr::one();
    for (size_t mi : sets[j].members) {
      const Item &it = items[mi];
      if (it.kind == 1) {
        Fr xp = Fr::one();
        for (int t = 0; t < 3; ++t) {
          pts.push_back(h_commit[t]);
          scal.push_back(coef * pw * xp);
          xp = xp * xn;
        }
      } else {
        pts.push_back(it.commit);
        scal.push_back(coef * pw);
      }
      for (size_t t = 0; t < rots.size(); ++t) comb[t] = comb[t] + pw * it.evals[layout.eval_slot(mi, rots[t])];
      pw = pw * yq;
    }
    // r_j(u) by Lagrange interpolation through (points of the set, comb)
    Fr ru = Fr::zero();
    for (size_t a = 0; a < rots.size(); ++a) {
      Fr num = Fr::one(), den = Fr::one();
      for (size_t b = 0; b < rots.size(); ++b)
        if (a != b) {
          num = num * (uu - pts_rot[rots[b]]);
          den = den * (pts_rot[rots[a]] - pts_rot[rots[b]]);
        }
      ru = ru + comb[a] * num * finv(den);
    }
    r_outer = r_outer + coef * ru;
    vj = vj * v;
  }
  AffinePoint gen;
  gen.x = fe::from_u64(1);
  gen.y = fe::from_u64(2);
  pts.push_back(gen);
  scal.push_back(Fr::zero() - r_outer);
  pts.push_back(h1);
  scal.push_back(Fr::zero() - z_0);
  pts.push_back(h2);
  scal.push_back(uu);
  const AffinePoint &w_commit = h2;
  const AffinePoint F = msm_host(scal, pts);
  // e(h2, s G2) = e(F, G2)
  const pairing::Pt<pairing::Fq2> &g2 = srs.g2, &sg2 = srs.sg2;
  pairing::G1 Fp{F.x, F.y}, nW;
  nW.x = w_commit.x;
  nW.y = w_commit.is_identity() ? fe::zero() : [&] {
    static const U256 QMOD = {{0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}};
    U256 r;
    fe::sub_raw(r, QMOD, w_commit.y);
    return r;
  }();
  return pairing::pairing_product_is_one({{Fp, g2}, {nW, sg2}});
}

const U256 QMOD_C = {{0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}};

void g2_to_raw(const pairing::Pt<pairing::Fq2> &p, uint8_t raw[128]) {
  const zk::Fq *c[4] = {&p.x.c[0], &p.x.c[1], &p.y.c[0], &p.y.c[1]};
  for (int i = 0; i < 4; ++i) memcpy(raw + 32 * i, c[i]->l, 32);
}
bool g2_from_canon(const uint8_t canon[128], pairing::Pt<pairing::Fq2> &p) {
  U256 c[4];
  memcpy(c, canon, 128);
  for (const auto &x : c)
    if (!(x < QMOD_C)) return false;
  p.x.c = {pairing::fq_from_canon(c[0]), pairing::fq_from_canon(c[1])};
  p.y.c = {pairing::fq_from_canon(c[2]), pairing::fq_from_canon(c[3])};
  p.inf = false;
  return pairing::g2_on_curve(p);
}

}  // namespace

void zk_srs_g2_from_secret(const U256 &s, uint8_t g2_raw[128], uint8_t sg2_raw[128]) {
  const pairing::Pt<pairing::Fq2> g = pairing::g2_generator();
  g2_to_raw(g, g2_raw);
  g2_to_raw(pairing::ec_mul(g, s), sg2_raw);
}
bool zk_g2_raw_to_canon(const uint8_t raw[128], uint8_t canon[128]) {
  for (int i = 0; i < 4; ++i) {
    zk::Fq m;
    memcpy(m.l, raw + 32 * i, 32);
    U256 lim;
    memcpy(lim.l, m.l, 32);
    if (!(lim < QMOD_C)) return false;   // a Montgomery residue is reduced too
    const zk::Fq c = zk::fp_from_mont<zk::FqP>(m);
    memcpy(canon + 32 * i, c.l, 32);
  }
  pairing::Pt<pairing::Fq2> p;
  return g2_from_canon(canon, p);
}
bool zk_g2_canon_to_raw(const uint8_t canon[128], uint8_t raw[128]) {
  pairing::Pt<pairing::Fq2> p;
  if (!g2_from_canon(canon, p)) return false;
  g2_to_raw(p, raw);
  return true;
}

extern "C" {

int zkfhe_srs_file_g2(const char *path, uint32_t *k_out, uint8_t g2_le[128], uint8_t s_g2_le[128]) {
  if (!path || !g2_le || !s_g2_le) return ZKFHE_EINVAL;
  FILE *f = fopen(path, "rb");
  if (!f) return ZKFHE_EINVAL;
  uint32_t k = 0;
  uint8_t raw[256];
  bool ok = fread(&k, 4, 1, f) == 1 && k >= 1 && k <= 28;
  const long want = 4 + (long)2 * 64 * ((long)1 << (ok ? k : 1)) + 256;
  ok = ok && fseek(f, 0, SEEK_END) == 0 && ftell(f) == want && fseek(f, want - 256, SEEK_SET) == 0 && fread(raw, 256, 1, f) == 1;
  fclose(f);
  if (!ok) return ZKFHE_EINVAL;
  if (!zk_g2_raw_to_canon(raw, g2_le) || !zk_g2_raw_to_canon(raw + 128, s_g2_le)) return ZKFHE_EINVAL;
  if (k_out) *k_out = k;
  return ZKFHE_OK;
}

int zkfhe_bfv_verify(const uint8_t *vk_bytes, size_t vk_len, const uint8_t *instances, size_t n_instances, const uint8_t *proof, size_t proof_len,
                     const uint8_t *srs_seed, size_t seed_len, int *accepted, char *err, size_t err_len) {
  if (!vk_bytes || !proof || !accepted || (!instances && n_instances) || (!srs_seed && seed_len)) return ZKFHE_EINVAL;
  *accepted = 0;
  try {
    const Vk vk = parse_vk(vk_bytes, vk_len);
    std::vector<U256> inst(n_instances);
    if (n_instances) memcpy(inst.data(), instances, n_instances * 32);
    for (const auto &v : inst)
      if (!(v < fe::MOD)) throw std::runtime_error("instance not reduced");
    *accepted = verify_impl(vk, inst, proof, proof_len, srs_g2_from_seed(srs_seed, seed_len)) ? 1 : 0;
    return ZKFHE_OK;
  } catch (const std::exception &e) {
    if (err && err_len) snprintf(err, err_len, "%s", e.what());
    return ZKFHE_OK;  // a malformed proof is a rejected proof, not an API error
  }
}

int zkfhe_bfv_verify_g2(const uint8_t *vk_bytes, size_t vk_len, const uint8_t *instances, size_t n_instances, const uint8_t *proof, size_t proof_len,
                        const uint8_t g2[128], const uint8_t s_g2[128], int *accepted, char *err, size_t err_len) {
  if (!vk_bytes || !proof || !accepted || !g2 || !s_g2 || (!instances && n_instances)) return ZKFHE_EINVAL;
  *accepted = 0;
  try {
    const Vk vk = parse_vk(vk_bytes, vk_len);
    std::vector<U256> inst(n_instances);
    if (n_instances) memcpy(inst.data(), instances, n_instances * 32);
    for (const auto &v : inst)
      if (!(v < fe::MOD)) throw std::runtime_error("instance not reduced");
    auto load = [](const uint8_t *b) {
      pairing::Pt<pairing::Fq2> p;
      // reduced coordinates, on the twist y^2 = x^3 + 3/(9+i)  (subgroup membership is the caller's responsibility, as in halo2's ParamsKZG::read)
      if (!g2_from_canon(b, p)) throw std::runtime_error("G2 point: a coordinate is not reduced or the point is not on the curve");
      return p;
    };
    SrsG2 srs;
    srs.g2 = load(g2);
    srs.sg2 = load(s_g2);
    *accepted = verify_impl(vk, inst, proof, proof_len, srs) ? 1 : 0;
    return ZKFHE_OK;
  } catch (const std::exception &e) {
    if (err && err_len) snprintf(err, err_len, "%s", e.what());
    return ZKFHE_OK;
  }
}

}  // extern "C"
