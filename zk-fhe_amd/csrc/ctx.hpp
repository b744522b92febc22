// Internal context shared by the C-ABI translation units (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <array>
#include <map>
#include <string>
#include <vector>

#include "../../include/zkfhe.h"
#include "bn254.hip.hpp"

struct NttDomain {
  int log_n = 0;
  zk::Fr *fwd = nullptr;  // omega^j, j < n
  zk::Fr *inv = nullptr;  // omega^-j, j < n
  zk::Fr n_inv;           // (2^log_n)^-1
  zk::Fr omega, omega_inv;
  // the same tables as constant operands of the nine-limb multiply (fr29.hip.hpp: value * 2^261, i.e. 32 times the standard
  // Montgomery form): what the NTT kernels read
  zk::Fr *fwd29 = nullptr, *inv29 = nullptr;
  zk::Fr n_inv29;
  zk::Fr *n_inv29_dev = nullptr;   // the same constant resident on the device (the inverse transforms read it: no upload per call)
};

struct zkfhe_ctx {
  uint64_t uid = 0;   // never reused (an address can be): what per-context caches such as the prover workspaces are keyed by
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::string err;
  int num_cu = 0;
  std::map<int, NttDomain> domains;
  // tables of the 2^13 tile kernel (ntt13.hip): unpacked twiddles per direction (keyed by the domain table they come from) and
  // coset pre-multiplier tables (keyed by shift, extension and rows); device memory, freed with the context
  std::map<const void *, void *> tw13;
  std::map<std::array<uint64_t, 6>, void *> pre13;
  // tables of the four-step passes of the longer rows (ntt_dif8.hip): butterfly constants and, per (size, direction, coset shift),
  // [M constants | n entries]; device memory, bounded, freed with the context
  std::map<std::array<uint64_t, 8>, void *> dif8;
  // grow-only scratch arenas (bytes)
  // profiling (zkfhe_prof_*): [0] = the summing kernel of a wide MSM call (k_msm_table / k_msm_accumulate), [1] = the NTT tile kernel (k_ntt13 / k_ntt_tile)
  bool prof_on = false;
  hipEvent_t pe0 = nullptr, pe1 = nullptr;
  hipEvent_t wait_ev = nullptr;  // hipEventBlockingSync: host waits sleep instead of spinning (zk_wait)
  double prof_ms[3] = {0, 0, 0}, prof_bytes[3] = {0, 0, 0}, prof_ops[3] = {0, 0, 0};   // [2] = k_msm_table of a call of a few columns
  uint64_t prof_launches[3] = {0, 0, 0};
  // pinned bounce buffer for small host<->device transfers (pageable copies go through the runtime's shared staging path)
  void *bounce = nullptr;
  static constexpr size_t BOUNCE_BYTES = (size_t)1 << 20;
  void *scratch[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t scratch_sz[4] = {0, 0, 0, 0};
  // host-side marks of the last proof made on this context, ms from its start (zkfhe_ctx_last_proof_marks): [0] the phase-0
  // commitment is back from the GPU, [1] the first challenge is squeezed (behind the public inputs' sponge), [2] the proof is done
  float proof_marks[3] = {0, 0, 0};
  unsigned *tickets = nullptr;   // zeroed counters: "last workgroup done" tickets of the table-path MSM, one per column (self-resetting)
};

struct zkfhe_basis {
  size_t n = 0;
  int c = 0;        // window bits
  int windows = 0;  // number of signed windows
  zk::G1Affine *table = nullptr;  // [windows][n] : 2^(c*w) * P_i
  // digit-multiple table (msm.hip k_msm_table): mult[(i*mw + w) * 2^(mc-1) + j-1] = j * 2^(mc*w) * P_i, j = 1..2^(mc-1),
  // mw = ceil(255/mc) signed windows; mc is the widest width whose table fits the per-basis budget (23.6 GB at n = 2^13,
  // mc = 12).  nullptr: calls against this basis take the bucket pipeline over `table`.
  zk::G1Affine *mult = nullptr;
  int mc = 0, mw = 0;
  size_t mult_bytes = 0;   // resident bytes of `mult`
  bool narrowed = false;   // the device did not have the room for the width the budget allowed when the basis was made
};

int zk_fail(zkfhe_ctx *ctx, int code, const char *what, hipError_t e, const char *file, int line);
int zk_fail_msg(zkfhe_ctx *ctx, int code, const std::string &msg);

#define ZK_HIP(ctx, expr)                                                             \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) return zk_fail((ctx), ZKFHE_EHIP, #expr, _e, __FILE__, __LINE__); \
  } while (0)

#define ZK_LAUNCH_CHECK(ctx) ZK_HIP(ctx, hipGetLastError())

// HIP's current device is per host thread: every public entry point selects the context's device first, so contexts
// of different GPUs (one process per GPU under torchrun, or several worker threads) never launch on the wrong one.
#define ZK_ENTER(ctx)                                   \
  do {                                                  \
    if (ctx) (void)hipSetDevice((ctx)->device);         \
  } while (0)

#define ZK_ARG(ctx, cond)                                                        \
  do {                                                                           \
    if (!(cond)) return zk_fail_msg((ctx), ZKFHE_EINVAL, std::string("bad argument: ") + #cond); \
  } while (0)

// profiling brackets: call zk_prof_begin before and zk_prof_end after the kernel launch
inline void zk_prof_begin(zkfhe_ctx *ctx) {
  if (ctx->prof_on) (void)hipEventRecord(ctx->pe0, ctx->stream);
}
inline void zk_prof_end(zkfhe_ctx *ctx, int which, double bytes) {
  if (!ctx->prof_on) return;
  (void)hipEventRecord(ctx->pe1, ctx->stream);
  (void)hipEventSynchronize(ctx->pe1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, ctx->pe0, ctx->pe1);
  ctx->prof_ms[which] += ms;
  ctx->prof_bytes[which] += bytes;
  ctx->prof_launches[which] += 1;
}

// Wait for everything queued on the context's stream.  hipStreamSynchronize spins; with a dozen proving threads per GPU
// that is a dozen cores per GPU doing nothing, so the wait goes through an event created with hipEventBlockingSync.
inline hipError_t zk_wait(zkfhe_ctx *ctx) {
  static const bool spin = getenv("ZKFHE_SPIN_WAIT") != nullptr;
  if (spin || !ctx->wait_ev) return hipStreamSynchronize(ctx->stream);
  hipError_t e = hipEventRecord(ctx->wait_ev, ctx->stream);
  if (e != hipSuccess) return e;
  return hipEventSynchronize(ctx->wait_ev);
}

// the top three / six DIF stages of long rows in four-step form (ntt_dif8.hip)
int zk_dif8_pass(zkfhe_ctx *ctx, const zk::Fr *src, zk::Fr *dst, size_t n_cols, int log_n, int S, int inverse, const zk::Fr *shifts_host, unsigned rows);
// zkfhe_msm_batch with a column stride (msm.hip)
int zk_msm_batch_strided(zkfhe_ctx *ctx, const zkfhe_basis *basis, const zkfhe_fr *scalars_dev, size_t col_stride, size_t n_cols, zkfhe_g1_affine *out_dev);
int zk_msm_batch_strided_form(zkfhe_ctx *ctx, const zkfhe_basis *basis, const zkfhe_fr *scalars_dev, size_t col_stride, size_t n_cols, void *out_dev, bool xyzz);
#define ZK_CK(x)             \
  do {                       \
    const int rc__ = (x);    \
    if (rc__) return rc__;   \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: set once per (device, kernel), under a lock --
// several contexts (devices, threads) may reach a launch site at the same time (core.hip)
int zk_func_max_lds(zkfhe_ctx *ctx, const void *kernel, int bytes);
// returns a device scratch arena of at least `bytes` (slot 0..3), grow-only, stream-ordered reuse
int zk_scratch(zkfhe_ctx *ctx, int slot, size_t bytes, void **out);
// zkfhe_basis_create with a share of the digit-multiple table budget (msm.hip)
int zk_basis_create_scaled(zkfhe_ctx *ctx, const zkfhe_g1_affine *bases_host, size_t n, int window_bits, double table_budget_scale, zkfhe_basis **out);
// stream-ordered device-to-device copy (own kernel for large blocks)
int zk_copy_d2d(zkfhe_ctx *ctx, void *dst, const void *src, size_t bytes);
int zk_domain(zkfhe_ctx *ctx, int log_n, const NttDomain **out);
int zk_pre13(zkfhe_ctx *ctx, const zk::Fr &g, int lef, int rows, bool scaled, const void **out);
// Lagrange columns -> the first `rows` cosets of the extended domain (lagrange_to_coeff + coeff_to_extended in one call): tmp_dev
// receives the coefficient form (at n = 2^13: times n, the n^-1 lives in the coset tables), column c at c * 2^log_n; out as
// zk_coset_ntt_rows
int zk_extend_lagrange(zkfhe_ctx *ctx, const zk::Fr *lagr_dev, zk::Fr *tmp_dev, zk::Fr *out_dev, size_t n_cols, int log_n, int lef, const zk::Fr &g, int rows);
// forward coset extension of the first `rows` cosets (ntt.hip)
int zk_coset_ntt_rows(zkfhe_ctx *ctx, const zk::Fr *in_dev, zk::Fr *out_dev, size_t n_cols, int log_n, int lef, const zk::Fr &g, int rows);

// host-side Fr helpers (same code as the device, compiled for the host)
zk::Fr zk_fr_from_u64(uint64_t v);
zk::Fr zk_fr_root_of_unity(int log_n);

static inline unsigned zk_blocks(size_t n, unsigned threads) { return (unsigned)((n + threads - 1) / threads); }
