// Intra-proof multi-GPU: one process per GPU, the big commitments of ONE proof sharded by point range over the ranks
// (BASELINE config 5 "windows split across 8 GPUs + ncclAllGather").
//
// Rank r owns the bases [lo_r, hi_r) of an SRS half (its own per-window table, 1/W of the memory) and reads the same rows of
// every column: its partial MSM is a group element, sum_r partial_r is the commitment.  EC addition is not an RCCL reduction
// op, so the 64-byte affine partials are all-gathered as raw bytes (n_cols * 64 B per rank -- latency-bound on xGMI, one
// collective per commitment batch) and every rank adds the W partials of each column itself: all ranks end with the same
// points, hence the same transcript and the same proof bytes as a single GPU.
//
// Transport: RCCL (librccl.so, resolved with dlopen at zkfhe_comm_create so that single-GPU users do not need it) --
// ncclAllGather of ncclUint8 on the context's stream, device buffers, no host round trip.  For tests and for hosts that
// bring their own transport (MPI, gloo, a Rust channel) zkfhe_comm_create_with_transport takes a host all-gather callback
// instead; the partials then go through pinned host memory.
#include <dlfcn.h>

#include <cstring>
#include <string>
#include <vector>

#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>   // the installed header: compile-time check of every signature used below and of ncclUint8
#else
// A build host without the RCCL development files: the slice of the API that is used, as RCCL 2.x declares it (the library is
// only ever resolved with dlopen at run time).
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct {
  char internal[128];
} ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId *uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream);
const char *ncclGetErrorString(ncclResult_t result);
}
#endif

#include "ctx.hpp"

using namespace zk;

namespace {

// The slice of the RCCL API that is used.  Types and prototypes come from the installed <rccl/rccl.h> (compile-time check of
// every signature and of ncclUint8); the library itself is resolved with dlopen at zkfhe_comm_create, nothing is linked.
struct Rccl {
  void *lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string err;
  bool load() {
    if (lib) return true;
    for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) {
      err = std::string("librccl.so not found: ") + dlerror();
      return false;
    }
    GetUniqueId = (decltype(GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    CommInitRank = (decltype(CommInitRank))dlsym(lib, "ncclCommInitRank");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
    GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllGather) {
      err = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
      return false;
    }
    return true;
  }
};
static_assert(sizeof(ncclUniqueId) == 128, "zkfhe_comm_unique_id hands out 128 bytes");
Rccl &rccl() {
  static Rccl r;
  return r;
}

// out[c] = sum_r parts[r * n_cols + c]   (affine in, affine out; one thread per column, W - 1 additions and one inversion)
__global__ void __launch_bounds__(64) k_sum_partials(const G1Affine *__restrict__ parts, unsigned world, size_t n_cols, G1Affine *__restrict__ out) {
  const size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (c >= n_cols) return;
  G1X acc = g1x_from_affine(parts[c]);
  for (unsigned r = 1; r < world; ++r) g1x_add_affine(acc, parts[(size_t)r * n_cols + c], false);
  out[c] = g1x_to_affine(acc);
}

}  // namespace

struct zkfhe_comm {
  int rank = 0, world = 1;
  ncclComm_t nccl = nullptr;
  zkfhe_allgather_fn transport = nullptr;   // host callback instead of RCCL
  void *transport_user = nullptr;
  uint8_t *host_send = nullptr, *host_recv = nullptr;   // pinned staging of the callback transport
  size_t host_cap = 0;
  // zkfhe_msm_batch_sharded_async (RCCL / one-rank communicators): the all-gather of the partial commitments and their sum run
  // on the communicator's own stream, so that whatever the caller queues on the context's stream next -- the partial MSM of the
  // following batch, the witness kernels -- overlaps the collective.  Two gather buffers alternate: batch i + 1 may fill its
  // partials while batch i is still on the wire.
  hipStream_t aux = nullptr;
  hipEvent_t ev_msm = nullptr, ev_done[3] = {nullptr, nullptr, nullptr};   // [2]: behind the last zkfhe_comm_all_gather_async
  void *buf[2] = {nullptr, nullptr};
  size_t buf_sz[2] = {0, 0};
  bool pending[2] = {false, false};
  int next = 0, last = -1;
  bool has_aux() const { return nccl != nullptr || world == 1; }   // the callback transport blocks the host: it stays on the context's stream
};

extern "C" {

int zkfhe_comm_unique_id(uint8_t id_out[128]) {
  if (!id_out) return ZKFHE_EINVAL;
  if (!rccl().load()) return ZKFHE_ENODEV;
  ncclUniqueId id;
  if (rccl().GetUniqueId(&id) != ncclSuccess) return ZKFHE_EHIP;
  memcpy(id_out, id.internal, 128);
  return ZKFHE_OK;
}

int zkfhe_comm_create(zkfhe_ctx *ctx, int rank, int world, const uint8_t unique_id[128], zkfhe_comm **out) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, out != nullptr && world >= 1 && rank >= 0 && rank < world);
  *out = nullptr;
  zkfhe_comm *c = new zkfhe_comm();
  c->rank = rank;
  c->world = world;
  // world == 1 with an id: a real one-rank RCCL communicator (every collective then goes through librccl: the smoke test of
  // the dlopen'ed entry points on a single GPU); world == 1 without: no transport at all
  if (world > 1 || unique_id != nullptr) {
    if (!unique_id) {
      delete c;
      return zk_fail_msg(ctx, ZKFHE_EINVAL, "zkfhe_comm_create: world > 1 needs the unique id of rank 0");
    }
    if (!rccl().load()) {
      delete c;
      return zk_fail_msg(ctx, ZKFHE_ENODEV, rccl().err);
    }
    ncclUniqueId id;
    memcpy(id.internal, unique_id, 128);
    const ncclResult_t rc = rccl().CommInitRank(&c->nccl, world, id, rank);
    if (rc != ncclSuccess) {
      delete c;
      return zk_fail_msg(ctx, ZKFHE_EHIP, std::string("ncclCommInitRank failed: ") + (rccl().GetErrorString ? rccl().GetErrorString(rc) : "?"));
    }
  }
  *out = c;
  return ZKFHE_OK;
}

int zkfhe_comm_create_with_transport(zkfhe_ctx *ctx, int rank, int world, zkfhe_allgather_fn allgather, void *user, zkfhe_comm **out) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, out != nullptr && world >= 1 && rank >= 0 && rank < world && (world == 1 || allgather != nullptr));
  zkfhe_comm *c = new zkfhe_comm();
  c->rank = rank;
  c->world = world;
  c->transport = allgather;
  c->transport_user = user;
  *out = c;
  return ZKFHE_OK;
}

int zkfhe_comm_destroy(zkfhe_ctx *ctx, zkfhe_comm *comm) {
  ZK_ENTER(ctx);
  if (!comm) return ZKFHE_OK;
  if (ctx) (void)hipStreamSynchronize(ctx->stream);
  if (comm->aux) (void)hipStreamSynchronize(comm->aux);
  if (comm->nccl) rccl().CommDestroy(comm->nccl);
  for (int b = 0; b < 2; ++b)
    if (comm->buf[b]) (void)hipFree(comm->buf[b]);
  for (int b = 0; b < 3; ++b)
    if (comm->ev_done[b]) (void)hipEventDestroy(comm->ev_done[b]);
  if (comm->ev_msm) (void)hipEventDestroy(comm->ev_msm);
  if (comm->aux) (void)hipStreamDestroy(comm->aux);
  if (comm->host_send) (void)hipHostFree(comm->host_send);
  if (comm->host_recv) (void)hipHostFree(comm->host_recv);
  delete comm;
  return ZKFHE_OK;
}

int zkfhe_comm_rank(const zkfhe_comm *comm) { return comm ? comm->rank : 0; }
int zkfhe_comm_world(const zkfhe_comm *comm) { return comm ? comm->world : 1; }
int zkfhe_comm_active(const zkfhe_comm *comm) { return comm && (comm->world > 1 || comm->nccl != nullptr) ? 1 : 0; }

void zkfhe_comm_point_range(const zkfhe_comm *comm, size_t n, size_t *lo, size_t *hi) {
  const size_t w = comm ? (size_t)comm->world : 1, r = comm ? (size_t)comm->rank : 0;
  if (lo) *lo = n * r / w;
  if (hi) *hi = n * (r + 1) / w;
}

// every rank: send_dev (bytes) -> recv_dev (world * bytes, rank-major), on the context's stream
int zkfhe_comm_all_gather(zkfhe_ctx *ctx, zkfhe_comm *comm, const void *send_dev, void *recv_dev, size_t bytes) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, comm != nullptr && send_dev != nullptr && recv_dev != nullptr);
  if (comm->nccl) {
    const ncclResult_t rc = rccl().AllGather(send_dev, recv_dev, bytes, ncclUint8, comm->nccl, ctx->stream);
    if (rc != ncclSuccess) return zk_fail_msg(ctx, ZKFHE_EHIP, std::string("ncclAllGather failed: ") + (rccl().GetErrorString ? rccl().GetErrorString(rc) : "?"));
    return ZKFHE_OK;
  }
  if (comm->world == 1) return zk_copy_d2d(ctx, recv_dev, send_dev, bytes);
  // callback transport: through pinned host memory
  if (comm->host_cap < bytes * (size_t)comm->world) {
    // the old buffers go first and the capacity is only recorded once both new ones exist: a failed allocation leaves the
    // communicator with no staging (cap 0), never with dangling pointers
    if (comm->host_send) (void)hipHostFree(comm->host_send);
    if (comm->host_recv) (void)hipHostFree(comm->host_recv);
    comm->host_send = comm->host_recv = nullptr;
    comm->host_cap = 0;
    const size_t cap = bytes * (size_t)comm->world;
    uint8_t *a = nullptr, *b = nullptr;
    ZK_HIP(ctx, hipHostMalloc((void **)&a, cap, hipHostMallocDefault));
    const hipError_t e2 = hipHostMalloc((void **)&b, cap, hipHostMallocDefault);
    if (e2 != hipSuccess) {
      (void)hipHostFree(a);
      ZK_HIP(ctx, e2);
    }
    comm->host_send = a;
    comm->host_recv = b;
    comm->host_cap = cap;
  }
  ZK_HIP(ctx, hipMemcpyAsync(comm->host_send, send_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
  ZK_HIP(ctx, zk_wait(ctx));
  if (comm->transport(comm->transport_user, comm->host_send, bytes, comm->host_recv) != 0) return zk_fail_msg(ctx, ZKFHE_EINVAL, "all-gather transport callback failed");
  ZK_HIP(ctx, hipMemcpyAsync(recv_dev, comm->host_recv, bytes * (size_t)comm->world, hipMemcpyHostToDevice, ctx->stream));
  return ZKFHE_OK;
}

// all-gather on an explicit stream (the RCCL / one-rank halves of zkfhe_comm_all_gather)
static int gather_on(zkfhe_ctx *ctx, zkfhe_comm *comm, const void *send_dev, void *recv_dev, size_t bytes, hipStream_t stream) {
  if (comm->nccl) {
    const ncclResult_t rc = rccl().AllGather(send_dev, recv_dev, bytes, ncclUint8, comm->nccl, stream);
    if (rc != ncclSuccess) return zk_fail_msg(ctx, ZKFHE_EHIP, std::string("ncclAllGather failed: ") + (rccl().GetErrorString ? rccl().GetErrorString(rc) : "?"));
    return ZKFHE_OK;
  }
  ZK_HIP(ctx, hipMemcpyAsync(recv_dev, send_dev, bytes, hipMemcpyDeviceToDevice, stream));
  return ZKFHE_OK;
}

static int comm_aux_init(zkfhe_ctx *ctx, zkfhe_comm *comm) {
  if (comm->aux) return ZKFHE_OK;
  ZK_HIP(ctx, hipStreamCreateWithFlags(&comm->aux, hipStreamNonBlocking));
  ZK_HIP(ctx, hipEventCreateWithFlags(&comm->ev_msm, hipEventDisableTiming));
  for (int b = 0; b < 3; ++b) ZK_HIP(ctx, hipEventCreateWithFlags(&comm->ev_done[b], hipEventDisableTiming | hipEventBlockingSync));
  return ZKFHE_OK;
}

// zkfhe_comm_all_gather on the communicator's own stream: the collective starts when everything queued on the context's stream so
// far is done and runs beside whatever the caller queues there next (the prover: the quotient of the next coset while the previous
// coset's share is on the wire); zkfhe_comm_join orders the context's stream (or the caller) behind it.  send_dev and recv_dev must
// stay untouched until then.  The callback transport blocks the host anyway: it completes on the context's stream as before.
int zkfhe_comm_all_gather_async(zkfhe_ctx *ctx, zkfhe_comm *comm, const void *send_dev, void *recv_dev, size_t bytes) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, comm != nullptr && send_dev != nullptr && recv_dev != nullptr);
  if (!comm->has_aux()) return zkfhe_comm_all_gather(ctx, comm, send_dev, recv_dev, bytes);
  ZK_CK(comm_aux_init(ctx, comm));
  ZK_HIP(ctx, hipEventRecord(comm->ev_msm, ctx->stream));
  ZK_HIP(ctx, hipStreamWaitEvent(comm->aux, comm->ev_msm, 0));
  ZK_CK(gather_on(ctx, comm, send_dev, recv_dev, bytes, comm->aux));
  ZK_HIP(ctx, hipEventRecord(comm->ev_done[2], comm->aux));
  comm->last = 2;   // the stream is in order: this event is behind every collective queued before it
  return ZKFHE_OK;
}

int zkfhe_msm_batch_sharded_async(zkfhe_ctx *ctx, zkfhe_comm *comm, const zkfhe_basis *basis_slice, const zkfhe_fr *scalars_dev, size_t col_stride,
                                  size_t n_cols, zkfhe_g1_affine *out_dev) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, comm != nullptr && basis_slice != nullptr);
  if (!n_cols) return ZKFHE_OK;
  if (!comm->has_aux()) return zkfhe_msm_batch_sharded(ctx, comm, basis_slice, scalars_dev, col_stride, n_cols, out_dev);   // complete on the context's stream
  ZK_CK(comm_aux_init(ctx, comm));
  const int b = comm->next;
  comm->next ^= 1;
  const size_t need = (size_t)(comm->world + 1) * n_cols * sizeof(G1Affine) + 64;
  if (comm->buf_sz[b] < need) {
    ZK_HIP(ctx, hipStreamSynchronize(comm->aux));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (comm->buf[b]) ZK_HIP(ctx, hipFree(comm->buf[b]));
    comm->buf[b] = nullptr, comm->buf_sz[b] = 0;
    ZK_HIP(ctx, hipMalloc(&comm->buf[b], need + need / 4));
    comm->buf_sz[b] = need + need / 4;
    comm->pending[b] = false;
  }
  // this buffer carried the batch before the previous one: that collective must be over before the partials below overwrite it
  if (comm->pending[b]) ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, comm->ev_done[b], 0));
  G1Affine *mine = (G1Affine *)((char *)comm->buf[b] + 64), *all = mine + n_cols;
  int rc = zk_msm_batch_strided(ctx, basis_slice, scalars_dev, col_stride, n_cols, (zkfhe_g1_affine *)mine);
  if (rc) return rc;
  ZK_HIP(ctx, hipEventRecord(comm->ev_msm, ctx->stream));
  ZK_HIP(ctx, hipStreamWaitEvent(comm->aux, comm->ev_msm, 0));
  rc = gather_on(ctx, comm, mine, all, n_cols * sizeof(G1Affine), comm->aux);
  if (rc) return rc;
  k_sum_partials<<<zk_blocks(n_cols, 64), 64, 0, comm->aux>>>(all, (unsigned)comm->world, n_cols, (G1Affine *)out_dev);
  ZK_LAUNCH_CHECK(ctx);
  ZK_HIP(ctx, hipEventRecord(comm->ev_done[b], comm->aux));
  comm->pending[b] = true;
  comm->last = b;
  return ZKFHE_OK;
}

// block_host = 0: the context's stream waits for every collective queued by the _async calls so far; 1: the calling thread does
// (and with it for everything that was on the context's stream when the last of them was queued)
int zkfhe_comm_join(zkfhe_ctx *ctx, zkfhe_comm *comm, int block_host) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, comm != nullptr);
  if (!comm->aux || comm->last < 0) {
    if (block_host) ZK_HIP(ctx, zk_wait(ctx));
    return ZKFHE_OK;
  }
  if (block_host) ZK_HIP(ctx, hipEventSynchronize(comm->ev_done[comm->last]));
  else ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, comm->ev_done[comm->last], 0));
  return ZKFHE_OK;
}

// records a HIP event behind the collectives queued so far (on the communicator's stream, or on the context's when the
// communicator has none): what a caller waits on instead of the context's stream to leave later kernels running
int zkfhe_comm_record_event(zkfhe_ctx *ctx, zkfhe_comm *comm, void *hip_event) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, comm != nullptr && hip_event != nullptr);
  ZK_HIP(ctx, hipEventRecord((hipEvent_t)hip_event, comm->aux && comm->last >= 0 ? comm->aux : ctx->stream));
  return ZKFHE_OK;
}

int zkfhe_msm_batch_sharded(zkfhe_ctx *ctx, zkfhe_comm *comm, const zkfhe_basis *basis_slice, const zkfhe_fr *scalars_dev, size_t col_stride,
                            size_t n_cols, zkfhe_g1_affine *out_dev) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, comm != nullptr && basis_slice != nullptr);
  if (!n_cols) return ZKFHE_OK;
  if (comm->world == 1 && !comm->nccl) return zk_msm_batch_strided(ctx, basis_slice, scalars_dev, col_stride, n_cols, out_dev);
  if (comm->has_aux()) {   // the asynchronous form, joined: the result is ordered on the context's stream as before
    int rc_a = zkfhe_msm_batch_sharded_async(ctx, comm, basis_slice, scalars_dev, col_stride, n_cols, out_dev);
    if (rc_a) return rc_a;
    return zkfhe_comm_join(ctx, comm, 0);
  }
  // scratch slot 3: [my partials | everyone's partials]  (slots 0..2 belong to the MSM itself)
  void *p;
  int rc = zk_scratch(ctx, 3, (size_t)(comm->world + 1) * n_cols * sizeof(G1Affine) + 64, &p);
  if (rc) return rc;
  G1Affine *mine = (G1Affine *)((char *)p + 64), *all = mine + n_cols;
  rc = zk_msm_batch_strided(ctx, basis_slice, scalars_dev, col_stride, n_cols, (zkfhe_g1_affine *)mine);
  if (rc) return rc;
  rc = zkfhe_comm_all_gather(ctx, comm, mine, all, n_cols * sizeof(G1Affine));
  if (rc) return rc;
  k_sum_partials<<<zk_blocks(n_cols, 64), 64, 0, ctx->stream>>>(all, (unsigned)comm->world, n_cols, (G1Affine *)out_dev);
  ZK_LAUNCH_CHECK(ctx);
  return ZKFHE_OK;
}

}  // extern "C"
