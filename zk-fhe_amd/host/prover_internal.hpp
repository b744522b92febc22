// Shared declarations of the circuit-level translation units (srs.hip, keygen.hip, prove.hip): handles, the per-context
// workspace, small host helpers.  Not part of the public ABI.
#pragma once
#include <chrono>
#include <ctime>
#include <cstdio>
#include <cstring>
#include <map>
#include <atomic>
#include <mutex>
#include <thread>
#include <stdexcept>

#include "../../include/zkfhe.h"
#include "bfv_circuit.hpp"
#include "bfv_phase0_fast.hpp"
#include "gpu_witness.hip.hpp"
#include "prover_kernels.hip.hpp"
#include "shplonk.hpp"
#include "transcript.hpp"
#include "prefix_cache.hpp"
#include "vk.hpp"

using namespace zkhost;
using zk::Fr;
using zk::G1Affine;

namespace zkhost {
CircuitConfig config_from_c(const zkfhe_bfv_config *c);
BfvParams params_from_c(const zkfhe_bfv_params *p);
}  // namespace zkhost

static const uint64_t DELTA_CANON[4] = {0x870e56bbe533e9a2ULL, 0x5b5f898e5e963f25ULL, 0x64ec26aad4c86e71ULL, 0x09226b6e22c6f0caULL};
static const uint64_t COSET_G = 7;

#define CK(x)                 \
  do {                        \
    int rc__ = (x);           \
    if (rc__) return rc__;    \
  } while (0)

struct zkfhe_srs {
  uint32_t k = 0;
  zkfhe_basis *g = nullptr, *g_lagrange = nullptr;
  // the same Lagrange points with narrower windows, for columns of small values (advice, permuted lookups): their cost is
  // the per-bucket work (merge, marginals), not the additions, so 8x fewer buckets beats 30 % more windows
  zkfhe_basis *g_lagrange_small = nullptr;
  // point-range shard (zkfhe_srs_create_sharded): the bases above hold points [lo, hi) only and every commitment goes through
  // zkfhe_msm_batch_sharded over `comm`
  zkfhe_comm *comm = nullptr;
  size_t lo = 0, hi = 0;
  // what zkfhe_srs_save writes besides k: the points as they came (unsharded SRS only) and the verifier's half, raw
  // Montgomery coordinates x.c0 | x.c1 | y.c0 | y.c1
  std::vector<G1Affine> g_host, gl_host;
  uint8_t g2_raw[128], sg2_raw[128];
  bool have_g2 = false;
  // more than one rank, or a one-rank communicator with a real transport (the single-GPU test of the whole collective path)
  bool sharded() const { return comm != nullptr && zkfhe_comm_active(comm) != 0; }
};

// commitment MSM of `n_cols` full columns (stride n): the whole basis, or this rank's rows + all-gather + sum
static inline int srs_msm(zkfhe_ctx *ctx, const zkfhe_srs *srs, const zkfhe_basis *basis, const Fr *cols, size_t n_cols, G1Affine *dev_out) {
  if (!srs->sharded()) return zkfhe_msm_batch(ctx, basis, (const zkfhe_fr *)cols, n_cols, (zkfhe_g1_affine *)dev_out);
  // the collective runs on the communicator's stream: srs_join / srs_record before anything reads dev_out
  return zkfhe_msm_batch_sharded_async(ctx, srs->comm, basis, (const zkfhe_fr *)(cols + srs->lo), (size_t)1 << srs->k, n_cols, (zkfhe_g1_affine *)dev_out);
}

// Commitments the HOST reads (pinned result blocks of the workspace).  A slot is PT_SLOT = 128 bytes.  One GPU: the MSM stores
// the accumulator-form sum (zkfhe_g1_xyzz) and the host normalises all points of a Fiat-Shamir round with one field inversion
// (points_canon) -- halo2's create_proof batch-normalises a round's commitments the same way, and
// the 40 us inversion in one GPU lane at the end of every MSM call is gone.  Sharded SRS: the all-gathered partials are affine
// and so is their sum: those results are packed at a 64-byte stride in the same blocks (pt_stride).
static constexpr size_t PT_SLOT = sizeof(zk::G1X);
static inline size_t pt_stride(const zkfhe_srs *srs) { return srs->sharded() ? sizeof(G1Affine) : sizeof(zk::G1X); }
static inline int srs_msm_pts(zkfhe_ctx *ctx, const zkfhe_srs *srs, const zkfhe_basis *basis, const Fr *cols, size_t n_cols, void *pinned_out) {
  if (!srs->sharded()) return zkfhe_msm_batch_xyzz(ctx, basis, (const zkfhe_fr *)cols, n_cols, (zkfhe_g1_xyzz *)pinned_out);
  return zkfhe_msm_batch_sharded_async(ctx, srs->comm, basis, (const zkfhe_fr *)(cols + srs->lo), (size_t)1 << srs->k, n_cols, (zkfhe_g1_affine *)pinned_out);
}

struct DevBuf {
  int device = 0;  // not the context: a workspace outlives the zkfhe_ctx it was made for if the caller destroys that first
  void *p = nullptr;
  size_t bytes = 0;
  int alloc(zkfhe_ctx *c, size_t b) {
    device = c->device;
    bytes = b;
    return zkfhe_dev_alloc(c, b, &p);
  }
  void release() {
    if (p) {
      (void)hipSetDevice(device);
      (void)hipFree(p);
    }
    p = nullptr;
  }
  Fr *fr() const { return (Fr *)p; }
};


struct View {  // a slice of a larger device allocation
  void *p = nullptr;
  Fr *fr() const { return (Fr *)p; }
};

struct Workspace {
  // every polynomial of one proof, Lagrange form, contiguous: [advice | la | ls | pz | lz | instance] (all_l) and the
  // same order on the extended coset (all_ext) -- one iNTT launch and one coset-NTT launch cover all of them
  DevBuf all_l, all_ext;
  View adv_l, la_l, ls_l, pz_l, lz_l, inst_l, adv_ext, la_ext, ls_ext, pz_ext, lz_ext, inst_ext;
  size_t inst_count = 0;   // rows of inst_l that hold public inputs of the last proof; the rest of the column is zero since allocation
  size_t n_all = 0;
  U256 *host_adv = nullptr;    // pinned [n_advice][n] witness table, reused by every proof on this context
  U256 *host_blind = nullptr;  // pinned staging for blinding rows / permuted lookup columns
  U256 *host_pool = nullptr;   // pinned staging for the coefficient arrays of the GPU witness generator
  U256 *host_poly = nullptr;   // pinned: the product of a phase-0 polynomial multiplication comes back here
  size_t host_poly_len = 0;
  uint8_t *ring = nullptr;      // pinned bump arena for the small tables of one proof: uploads from it need no host wait
  size_t ring_off = 0;
  static constexpr size_t RING_BYTES = (size_t)4 << 20;
  // device mirror of the ring: stage() places a table in the ring and returns its address in the mirror, flush_staged() moves
  // everything staged since the last flush with ONE copy (a Fiat-Shamir round has a dozen small tables: expression groups,
  // powers, pointer / scalar lists, rotation sets ...)
  DevBuf dev_ring;
  size_t ring_flushed = 0;
  // results that the host reads (commitments, evaluations, check flags) are WRITTEN by the kernels into this pinned,
  // device-visible block -- no device-to-host copy commands: [points | evaluations | flags]
  uint8_t *host_out = nullptr;
  size_t out_pts_cap = 0, out_ev_off = 0, out_ev_cap = 0, out_flag_off = 0;
  uint8_t *out_pts() const { return host_out; }   // out_pts_cap slots of PT_SLOT bytes (prover_internal.hpp "Commitments the HOST reads")
  U256 *out_ev() const { return (U256 *)(host_out + out_ev_off); }
  int *out_flags() const { return (int *)(host_out + out_flag_off); }   // [0] permutation closes, [1] lookups close, [2] lookup input in table
  uint8_t *host_pts = nullptr; // pinned: the phase-0 commitments (PT_SLOT bytes each), read after ev_pts
  hipEvent_t ev_pts = nullptr;
  // the random polynomial of the vanishing argument depends on no challenge: it is uploaded and committed at the start of
  // the proof on an auxiliary context (own stream, scratch and tickets) beside the phase-0 / witness work
  zkfhe_ctx *aux = nullptr;
  uint8_t *host_rand_pt = nullptr;  // pinned: the commitment (one PT_SLOT)
  hipEvent_t ev_rand = nullptr;
  // early phase-1 commitment (everything that does not depend on the phase-1 challenge): points + lookup error flag, pinned
  uint8_t *host_early = nullptr;   // PT_SLOT bytes per point
  int *host_early_err = nullptr;
  hipEvent_t ev_early = nullptr;
  DevBuf stream, pool, invtmp, wblind;  // device: phase-1 gate stream, coefficient arrays, deferred inverses, blinding rows + flag
  DevBuf tmp_c, partials, h_ext, h_c, misc, points, num, den, small, jobs, evout, polyio;
  DevBuf qgather;   // one proof over W GPUs: the W shares of the quotient, [coset row][rank][n] (allocated by the first sharded proof)
  std::vector<DevBuf *> all() {
    return {&all_l, &all_ext, &tmp_c, &partials, &h_ext, &h_c, &misc, &points, &num, &den, &small, &jobs, &evout, &polyio, &stream, &pool, &invtmp, &wblind, &dev_ring, &qgather};
  }
};

struct zkfhe_bfv_pk;
// keygen.hip
int up(zkfhe_ctx *ctx, Workspace *ws, void *dst, const void *src, size_t bytes);
int get_workspace(zkfhe_ctx *ctx, zkfhe_bfv_pk *pk, Workspace **out);
void free_workspace(Workspace *ws);
int extend_cols(zkfhe_ctx *ctx, zkfhe_bfv_pk *pk, Workspace *ws, const Fr *lagr, size_t count, Fr *ext);
// prove.hip
int alloc_witness_buffers(zkfhe_ctx *ctx, const zkfhe_bfv_pk *pk, Workspace *ws);
// verifier.cpp (host pairing arithmetic): G2 and s G2 as raw Montgomery coordinates; raw <-> canonical; on-curve check of raw coordinates
void zk_srs_g2_from_secret(const U256 &s, uint8_t g2_raw[128], uint8_t sg2_raw[128]);
bool zk_g2_raw_to_canon(const uint8_t raw[128], uint8_t canon[128]);   // false: a coordinate is not reduced or the point is off the twist
bool zk_g2_canon_to_raw(const uint8_t canon[128], uint8_t raw[128]);

static inline double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static inline Fr mont(const U256 &v) { return fe::to_mont(v); }
static inline Fr mont_u64(uint64_t v) { return fe::to_mont(fe::from_u64(v)); }
static inline U256 canon(const Fr &f) { return fe::from_mont(f); }
static inline Fr fr_pow(Fr b, uint64_t e) {
  Fr r = Fr::one();
  while (e) {
    if (e & 1) r = r * b;
    b = b * b;
    e >>= 1;
  }
  return r;
}
static inline Fr fr_inv(const Fr &a) { return zk::fp_inv<zk::FrP>(a); }

static inline AffinePoint point_canon(const G1Affine &p) {
  AffinePoint a;
  zk::Fq x = zk::fp_from_mont<zk::FqP>(p.x), y = zk::fp_from_mont<zk::FqP>(p.y);
  memcpy(a.x.l, x.l, 32);
  memcpy(a.y.l, y.l, 32);
  return a;
}

static inline unsigned grid_for(zkfhe_ctx *ctx, size_t work) {
  size_t b = (work + 255) / 256;
  size_t cap = (size_t)ctx->num_cu * 16;
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

// upload canonical values and convert to Montgomery on the device
static inline int upload_canon(zkfhe_ctx *ctx, Fr *dst, const U256 *src, size_t count) {
  ZK_HIP(ctx, hipMemcpyAsync(dst, src, count * 32, hipMemcpyHostToDevice, ctx->stream));
  return zkfhe_fr_to_mont(ctx, (const zkfhe_fr *)dst, (zkfhe_fr *)dst, count);
}

// Small host -> device tables of a proof (expression groups, powers, pointer lists, ...): staged in the workspace's pinned
// arena and copied without waiting -- the arena is only recycled at the start of the next proof, after a stream sync.


// A small host table of the proof in flight -> device: copied into the pinned ring now, visible on the device (at the returned
// address) after the next flush_staged().  nullptr when the ring is full.
static inline void *stage_table(Workspace *ws, const void *src, size_t bytes) {
  const size_t need = (bytes + 63) & ~(size_t)63;
  if (!ws->ring || !ws->dev_ring.p || ws->ring_off + need > Workspace::RING_BYTES) return nullptr;
  uint8_t *slot = ws->ring + ws->ring_off;
  memcpy(slot, src, bytes);
  void *dev = (char *)ws->dev_ring.p + ws->ring_off;
  ws->ring_off += need;
  return dev;
}
// The same for a block the caller fills in place before the next flush_staged(): *host = where to write, returns the device address
static inline void *stage_reserve(Workspace *ws, size_t bytes, void **host) {
  const size_t need = (bytes + 63) & ~(size_t)63;
  if (!ws->ring || !ws->dev_ring.p || ws->ring_off + need > Workspace::RING_BYTES) return nullptr;
  *host = ws->ring + ws->ring_off;
  void *dev = (char *)ws->dev_ring.p + ws->ring_off;
  ws->ring_off += need;
  return dev;
}
static inline int flush_staged(zkfhe_ctx *ctx, Workspace *ws) {
  if (ws->ring_off > ws->ring_flushed) {
    ZK_HIP(ctx, hipMemcpyAsync((char *)ws->dev_ring.p + ws->ring_flushed, ws->ring + ws->ring_flushed, ws->ring_off - ws->ring_flushed, hipMemcpyHostToDevice, ctx->stream));
    ws->ring_flushed = ws->ring_off;
  }
  return ZKFHE_OK;
}
#define STAGE(var, type, ws, src, bytes)                                                             \
  type *var = (type *)::stage_table((ws), (src), (bytes));                                                   \
  if (!var) return zk_fail_msg(ctx, ZKFHE_ENOMEM, "the proof's small tables do not fit the staging ring")

// commit `n_cols` columns straight into the pinned result block (the last MSM kernel stores its affine points there) and wait
static inline int commit_cols_out(zkfhe_ctx *ctx, const zkfhe_srs *srs, const zkfhe_basis *basis, const Fr *cols, size_t n_cols, Workspace *ws, std::vector<AffinePoint> &out);

// after srs_msm: host = false -> the context's stream waits for the commitments queued so far (a device-side reader follows);
// host = true -> the calling thread waits (for them and for everything queued on the context's stream before them)
static inline int srs_join(zkfhe_ctx *ctx, const zkfhe_srs *srs, bool host) {
  if (srs->sharded()) return zkfhe_comm_join(ctx, srs->comm, host ? 1 : 0);
  if (host) ZK_HIP(ctx, zk_wait(ctx));
  return ZKFHE_OK;
}
// an event behind the commitments queued so far (not behind the kernels the caller queues next)
static inline int srs_record(zkfhe_ctx *ctx, const zkfhe_srs *srs, hipEvent_t ev) {
  if (srs->sharded()) return zkfhe_comm_record_event(ctx, srs->comm, (void *)ev);
  ZK_HIP(ctx, hipEventRecord(ev, ctx->stream));
  return ZKFHE_OK;
}

// commit `n_cols` columns (device, Montgomery) and return canonical affine points
static inline int commit_cols(zkfhe_ctx *ctx, const zkfhe_srs *srs, const zkfhe_basis *basis, const Fr *cols, size_t n_cols, G1Affine *dev_out, std::vector<AffinePoint> &out) {
  CK(srs_msm(ctx, srs, basis, cols, n_cols, dev_out));
  CK(srs_join(ctx, srs, false));
  std::vector<G1Affine> h(n_cols);
  CK(zkfhe_download(ctx, h.data(), dev_out, n_cols * sizeof(G1Affine)));
  out.resize(n_cols);
  for (size_t i = 0; i < n_cols; ++i) out[i] = point_canon(h[i]);
  return ZKFHE_OK;
}

// n points of a pinned result block (written by srs_msm_pts / zkfhe_msm_sparse_xyzz, complete) -> canonical affine coordinates
static inline void points_canon(const zkfhe_srs *srs, const void *block, size_t n, AffinePoint *out) {
  if (srs->sharded()) {
    const G1Affine *a = (const G1Affine *)block;
    for (size_t i = 0; i < n; ++i) out[i] = point_canon(a[i]);
    return;
  }
  std::vector<G1Affine> a(n);
  zk::g1x_normalize_batch((const zk::G1X *)block, n, a.data());
  for (size_t i = 0; i < n; ++i) out[i] = point_canon(a[i]);
}

static inline int commit_cols_out(zkfhe_ctx *ctx, const zkfhe_srs *srs, const zkfhe_basis *basis, const Fr *cols, size_t n_cols, Workspace *ws, std::vector<AffinePoint> &out) {
  if (!ws->host_out || n_cols > ws->out_pts_cap) return commit_cols(ctx, srs, basis, cols, n_cols, (G1Affine *)ws->points.p, out);
  CK(srs_msm_pts(ctx, srs, basis, cols, n_cols, ws->out_pts()));
  CK(srs_join(ctx, srs, true));
  out.resize(n_cols);
  points_canon(srs, ws->out_pts(), n_cols, out.data());
  return ZKFHE_OK;
}

struct GpuPolyMul : PolyMulBackend {
  zkfhe_ctx *ctx;
  Workspace *ws;
  GpuPolyMul(zkfhe_ctx *c, Workspace *w) : ctx(c), ws(w) {}
  std::vector<BigInt> mul_u64(const std::vector<uint64_t> &a, const std::vector<uint64_t> &b) override;
  const U256 *mul_u64_raw(const std::vector<uint64_t> &a, const std::vector<uint64_t> &b) override;

 private:
  std::vector<U256> pageable;   // the product when it does not fit the pinned block
};


struct zkfhe_bfv_pk {
  CircuitConfig cfg;
  BfvParams prm;
  DevBuf fixed_l, sigma_l, fixed_ext, sigma_ext, l_ext, xs_ext, dpow, ext3_pw;
  std::vector<AffinePoint> fixed_commit, sigma_commit;
  U256 vk_digest;
  // structure of the phase-1 gate stream, recorded at keygen for the GPU witness generator
  size_t gate1_cells = 0, n_lookup_cells = 0, n_inv_slots = 0;
  // cosets of the extended domain the quotient is evaluated on: 3 (degree < 3n), or all 4 with ZKFHE_CHECK_QUOTIENT set when
  // the key is built (the fourth gives the degree check).  Every extended array has this many rows per column.
  int ext_rows = 3;
  DevBuf lookup_src, inv_slots, place_start, place_len;   // device: u32 lists
  // per-context prover workspaces: one proof at a time per zkfhe_ctx, any number of contexts (streams)
  // may prove concurrently against the same key (everything above is read-only after keygen)
  std::map<uint64_t, Workspace *> workspaces;   // keyed by zkfhe_ctx::uid (never reused), not by address
  std::mutex mu;
  // transcript state after `vk digest | pk0 | pk1`, per public key seen (prefix_cache.hpp); shared by the proofs in flight
  PrefixCache prefix;
  // transcript states of proofs announced ahead of time (zkfhe_bfv_pk_prehash): one-shot, consumed by the proof of the same inputs
  PreHash prehash;
};

