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
  if (comm->nccl) rccl().CommDestroy(comm->nccl);
  if (comm->host_send) (void)hipHostFree(comm->host_send);
  if (comm->host_recv) (void)hipHostFree(comm->host_recv);
  delete comm;
  return ZKFHE_OK;
}

int zkfhe_comm_rank(const zkfhe_comm *comm) { return comm ? comm->rank : 0; }
int zkfhe_comm_world(const zkfhe_comm *comm) { return comm ? comm->world : 1; }

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

int zkfhe_msm_batch_sharded(zkfhe_ctx *ctx, zkfhe_comm *comm, const zkfhe_basis *basis_slice, const zkfhe_fr *scalars_dev, size_t col_stride,
                            size_t n_cols, zkfhe_g1_affine *out_dev) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, comm != nullptr && basis_slice != nullptr);
  if (!n_cols) return ZKFHE_OK;
  if (comm->world == 1 && !comm->nccl) return zk_msm_batch_strided(ctx, basis_slice, scalars_dev, col_stride, n_cols, out_dev);
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
