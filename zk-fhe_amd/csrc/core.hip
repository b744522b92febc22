// Context, device memory, timing and coefficient-wise Fr kernels of the C ABI (include/zkfhe.h).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <map>
#include <mutex>
#include <utility>

#include "ctx.hpp"
#include "fq29.hip.hpp"
#include "fr29.hip.hpp"

using namespace zk;

static thread_local std::string g_create_err;

int zk_fail(zkfhe_ctx *ctx, int code, const char *what, hipError_t e, const char *file, int line) {
  char buf[512];
  snprintf(buf, sizeof(buf), "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
  if (ctx) ctx->err = buf; else g_create_err = buf;
  return code;
}
int zk_fail_msg(zkfhe_ctx *ctx, int code, const std::string &msg) {
  if (ctx) ctx->err = msg; else g_create_err = msg;
  return code;
}

int zk_func_max_lds(zkfhe_ctx *ctx, const void *kernel, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void *>, int> done;   // (device, kernel) -> bytes granted
  std::lock_guard<std::mutex> l(mu);
  int &have = done[std::make_pair(ctx->device, kernel)];
  if (have >= bytes) return ZKFHE_OK;
  ZK_HIP(ctx, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  have = bytes;
  return ZKFHE_OK;
}

int zk_scratch(zkfhe_ctx *ctx, int slot, size_t bytes, void **out) {
  if (ctx->scratch_sz[slot] < bytes) {
    if (ctx->scratch[slot]) {
      ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
      ZK_HIP(ctx, hipFree(ctx->scratch[slot]));
      ctx->scratch[slot] = nullptr;
      ctx->scratch_sz[slot] = 0;
    }
    size_t want = bytes + bytes / 4;
    ZK_HIP(ctx, hipMalloc(&ctx->scratch[slot], want));
    ctx->scratch_sz[slot] = want;
  }
  *out = ctx->scratch[slot];
  return ZKFHE_OK;
}

Fr zk_fr_from_u64(uint64_t v) {
  Fr t = Fr::zero();
  t.l[0] = (u32)v;
  t.l[1] = (u32)(v >> 32);
  return fp_to_mont<FrP>(t);
}

Fr zk_fr_root_of_unity(int log_n) {
  // 7^((r-1)/2^28), canonical (SURVEY.md section 4 KAT 4)
  Fr c;
  const u32 w[8] = {0x60c37c9cu, 0xd34f1ed9u, 0xd39329c8u, 0x3215cf6du, 0x3dd31f74u, 0x98865ea9u, 0x166d18b7u, 0x03ddb9f5u};
  for (int i = 0; i < 8; ++i) c.l[i] = w[i];
  Fr r = fp_to_mont<FrP>(c);
  for (int i = 0; i < 28 - log_n; ++i) r = fp_sqr<FrP>(r);
  return r;
}

// Device-to-device copy.  The runtime's blit kernel runs a few workgroups (measured 0.3-1 TB/s on multi-GB column blocks
// at k = 19); large 16-byte-aligned copies go through a grid-stride kernel that fills the chip instead.
__global__ void __launch_bounds__(256) k_copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int zk_copy_d2d(zkfhe_ctx *ctx, void *dst, const void *src, size_t bytes) {
  if (!bytes) return ZKFHE_OK;
  if (bytes >= ((size_t)1 << 20) && (bytes & 15) == 0 && ((uintptr_t)dst & 15) == 0 && ((uintptr_t)src & 15) == 0) {
    const size_t n16 = bytes / 16;
    size_t blocks = (n16 + 255) / 256;
    const size_t cap = (size_t)ctx->num_cu * 32;
    k_copy16<<<(unsigned)(blocks > cap ? cap : blocks), 256, 0, ctx->stream>>>((const uint4 *)src, (uint4 *)dst, n16);
    ZK_LAUNCH_CHECK(ctx);
    return ZKFHE_OK;
  }
  ZK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return ZKFHE_OK;
}

extern "C" {

const char *zkfhe_version(void) { return "zkfhe-mi355x 0.1 (gfx950)"; }

int zkfhe_ctx_create(int device_id, void *hip_stream, zkfhe_ctx **out) {
  // one hardware queue per concurrent context (ROCm default: 4 per process, streams sharing a queue serialise); a no-op
  // when the host application already initialised the HIP runtime or set the variable itself
  setenv("GPU_MAX_HW_QUEUES", "16", 0);
  // host threads that wait for the GPU sleep instead of spinning (a dozen proving threads per GPU would otherwise keep a
  // dozen cores busy doing nothing); ZKFHE_SPIN_WAIT=1 keeps the runtime's default
  if (!getenv("ZKFHE_SPIN_WAIT")) {
    (void)hipSetDevice(device_id);
    (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
    (void)hipGetLastError();
  }
  if (!out) return zk_fail_msg(nullptr, ZKFHE_EINVAL, "out is NULL");
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0)
    return zk_fail_msg(nullptr, ZKFHE_ENODEV, std::string("no HIP device: ") + hipGetErrorString(e));
  if (device_id < 0 || device_id >= count) return zk_fail_msg(nullptr, ZKFHE_EINVAL, "device_id out of range");
  hipDeviceProp_t prop;
  ZK_HIP(nullptr, hipGetDeviceProperties(&prop, device_id));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return zk_fail_msg(nullptr, ZKFHE_ENODEV, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
  ZK_HIP(nullptr, hipSetDevice(device_id));
  zkfhe_ctx *ctx = new zkfhe_ctx();
  static std::atomic<uint64_t> next_uid{1};
  ctx->uid = next_uid.fetch_add(1);
  ctx->device = device_id;
  ctx->num_cu = prop.multiProcessorCount;
  if (hip_stream) {
    ctx->stream = (hipStream_t)hip_stream;
  } else {
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return zk_fail(nullptr, ZKFHE_EHIP, "hipStreamCreate", e, __FILE__, __LINE__); }
    ctx->own_stream = true;
  }
  if (hipHostMalloc(&ctx->bounce, zkfhe_ctx::BOUNCE_BYTES, hipHostMallocDefault) != hipSuccess) ctx->bounce = nullptr;
  if (hipEventCreateWithFlags(&ctx->wait_ev, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) ctx->wait_ev = nullptr;
  hipEventCreate(&ctx->ev0);
  hipEventCreate(&ctx->ev1);
  hipEventCreate(&ctx->pe0);
  hipEventCreate(&ctx->pe1);
  *out = ctx;
  return ZKFHE_OK;
}

int zkfhe_ctx_destroy(zkfhe_ctx *ctx) {
  ZK_ENTER(ctx);
  if (!ctx) return ZKFHE_OK;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  if (ctx->bounce) (void)hipHostFree(ctx->bounce);
  for (auto &kv : ctx->domains) {
    hipFree(kv.second.fwd);
    hipFree(kv.second.inv);
    hipFree(kv.second.fwd29);
    hipFree(kv.second.inv29);
    hipFree(kv.second.n_inv29_dev);
  }
  for (auto &kv : ctx->tw13) hipFree(kv.second);
  for (auto &kv : ctx->pre13) hipFree(kv.second);
  for (auto &kv : ctx->dif8) hipFree(kv.second);
  for (int i = 0; i < 4; ++i)
    if (ctx->scratch[i]) hipFree(ctx->scratch[i]);
  if (ctx->tickets) hipFree(ctx->tickets);
  if (ctx->wait_ev) hipEventDestroy(ctx->wait_ev);
  hipEventDestroy(ctx->ev0);
  hipEventDestroy(ctx->ev1);
  if (ctx->own_stream) hipStreamDestroy(ctx->stream);
  delete ctx;
  return ZKFHE_OK;
}

const char *zkfhe_last_error(const zkfhe_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int zkfhe_sync(zkfhe_ctx *ctx) {
  ZK_ENTER(ctx);
  ZK_HIP(ctx, zk_wait(ctx));
  return ZKFHE_OK;
}

void *zkfhe_stream(zkfhe_ctx *ctx) { return (void *)ctx->stream; }

int zkfhe_device_info(zkfhe_ctx *ctx, char *arch_name, size_t arch_len, int *num_cu, size_t *hbm_bytes) {
  ZK_ENTER(ctx);
  hipDeviceProp_t prop;
  ZK_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
  if (arch_name && arch_len) { strncpy(arch_name, prop.gcnArchName, arch_len - 1); arch_name[arch_len - 1] = 0; }
  if (num_cu) *num_cu = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
  return ZKFHE_OK;
}

int zkfhe_dev_alloc(zkfhe_ctx *ctx, size_t bytes, void **dptr) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, dptr != nullptr);
  ZK_HIP(ctx, hipSetDevice(ctx->device));
  hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
  if (e == hipErrorOutOfMemory) return zk_fail_msg(ctx, ZKFHE_ENOMEM, "hipMalloc: out of device memory");
  ZK_HIP(ctx, e);
  return ZKFHE_OK;
}
int zkfhe_dev_free(zkfhe_ctx *ctx, void *dptr) {
  ZK_ENTER(ctx);
  if (!dptr) return ZKFHE_OK;
  if (!ctx) {  // the owning context is gone already (hipFree synchronises on its own)
    (void)hipFree(dptr);
    return ZKFHE_OK;
  }
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ZK_HIP(ctx, hipFree(dptr));
  return ZKFHE_OK;
}
int zkfhe_upload(zkfhe_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) {
  ZK_ENTER(ctx);
  if (!bytes) return ZKFHE_OK;
  if (ctx->bounce && bytes <= zkfhe_ctx::BOUNCE_BYTES) {
    // small transfers (tables of a few KB, dozens per proof) through the context's own pinned buffer: a pageable copy goes
    // through the runtime's process-wide staging path and serialises the proving threads
    memcpy(ctx->bounce, src_host, bytes);
    ZK_HIP(ctx, hipMemcpyAsync(dst_dev, ctx->bounce, bytes, hipMemcpyHostToDevice, ctx->stream));
  } else {
    ZK_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
  }
  ZK_HIP(ctx, zk_wait(ctx));
  return ZKFHE_OK;
}
int zkfhe_download(zkfhe_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes) {
  ZK_ENTER(ctx);
  if (!bytes) return ZKFHE_OK;
  if (ctx->bounce && bytes <= zkfhe_ctx::BOUNCE_BYTES) {
    ZK_HIP(ctx, hipMemcpyAsync(ctx->bounce, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, zk_wait(ctx));
    memcpy(dst_host, ctx->bounce, bytes);
    return ZKFHE_OK;
  }
  ZK_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
  ZK_HIP(ctx, zk_wait(ctx));
  return ZKFHE_OK;
}
int zkfhe_copy_dev(zkfhe_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes) {
  ZK_ENTER(ctx);
  return zk_copy_d2d(ctx, dst_dev, src_dev, bytes);
}
int zkfhe_memset_dev(zkfhe_ctx *ctx, void *dst_dev, int byte, size_t bytes) {
  ZK_ENTER(ctx);
  ZK_HIP(ctx, hipMemsetAsync(dst_dev, byte, bytes, ctx->stream));
  return ZKFHE_OK;
}

int zkfhe_prof_enable(zkfhe_ctx *ctx, int on) {
  ZK_ENTER(ctx);
  ctx->prof_on = on != 0;
  return ZKFHE_OK;
}
int zkfhe_prof_reset(zkfhe_ctx *ctx) {
  ZK_ENTER(ctx);
  for (int i = 0; i < 3; ++i) {
    ctx->prof_ms[i] = ctx->prof_bytes[i] = ctx->prof_ops[i] = 0;
    ctx->prof_launches[i] = 0;
  }
  return ZKFHE_OK;
}
int zkfhe_prof_read(zkfhe_ctx *ctx, int which, double *total_ms, uint64_t *launches, double *algorithmic_bytes) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, which >= 0 && which < 3);
  if (total_ms) *total_ms = ctx->prof_ms[which];
  if (launches) *launches = ctx->prof_launches[which];
  if (algorithmic_bytes) *algorithmic_bytes = ctx->prof_bytes[which];
  return ZKFHE_OK;
}

int zkfhe_prof_read_ops(zkfhe_ctx *ctx, int which, double *ops) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, which >= 0 && which < 3 && ops != nullptr);
  *ops = ctx->prof_ops[which];
  return ZKFHE_OK;
}

int zkfhe_ctx_last_proof_marks(zkfhe_ctx *ctx, float marks_ms[3]) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, marks_ms != nullptr);
  for (int i = 0; i < 3; ++i) marks_ms[i] = ctx->proof_marks[i];
  return ZKFHE_OK;
}

int zkfhe_timer_start(zkfhe_ctx *ctx) {
  ZK_ENTER(ctx);
  ZK_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  return ZKFHE_OK;
}
int zkfhe_timer_stop_ms(zkfhe_ctx *ctx, float *ms) {
  ZK_ENTER(ctx);
  ZK_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  ZK_HIP(ctx, hipEventSynchronize(ctx->ev1));
  ZK_HIP(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
  return ZKFHE_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// coefficient-wise kernels: one Fr (32 B = 2 x 16 B vector loads) per thread, grid-stride
// ---------------------------------------------------------------------------------------------
enum { OP_ADD = 0, OP_SUB = 1, OP_MUL = 2 };

template <int OP>
__global__ void __launch_bounds__(256) k_fr_binop(const Fr *__restrict__ a, const Fr *__restrict__ b, Fr *__restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    Fr x = a[i], y = b[i];
    Fr r = OP == OP_ADD ? x + y : OP == OP_SUB ? x - y : x * y;
    out[i] = r;
  }
}

// MODE 0: * s ; 1: to_mont ; 2: from_mont
template <int MODE>
__global__ void __launch_bounds__(256) k_fr_unop(const Fr *__restrict__ a, Fr s, Fr *__restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    Fr x = a[i];
    out[i] = MODE == 0 ? x * s : MODE == 1 ? fr29_to_mont(x) : fp_from_mont<FrP>(x);
  }
}

__global__ void __launch_bounds__(256) k_fr_sqr_chain(const Fr *__restrict__ a, Fr *__restrict__ out, size_t n, int iters) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr x = a[i];
  for (int k = 0; k < iters; ++k) x = fp_sqr<FrP>(x);
  out[i] = x;
}

// the same probe for the radix-2^29 product of the MSM kernels (fq29.hip.hpp): packed 256-bit words in, nine limbs in registers
__global__ void __launch_bounds__(256) k_fq29_sqr_chain(const Fq *__restrict__ a, Fq *__restrict__ out, size_t n, int iters) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  F29 x = f29_unpack(a[i]);
  for (int k = 0; k < iters; ++k) x = f29_sqr(x);
  out[i] = f29_pack(f29_canonical(x));
}

// Batch inversion, Montgomery trick per thread over a strided chunk of CHUNK elements:
// thread t owns elements t, t+T, t+2T, ... (T = total threads) so every load/store is coalesced.
// prefix products go to `tmp` (n elements).  Zero elements are skipped and stay zero.
#define BI_CHUNK 8
// from 2^20 elements on (the grand-product denominators of a k = 13 proof: 1.6 M): at least 16 per inversion -- a lone proof is
// 0.25 ms slower, 96 proofs through 16 streams 1.4 % faster (242.5 / 239.7 against 239.0 / 236.7 proofs/s)
#define BI_LONG ((size_t)1 << 20)
__global__ void __launch_bounds__(256) k_fr_batch_invert(Fr *__restrict__ a, Fr *__restrict__ tmp, size_t n, size_t T) {
  size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t >= T) return;
  Fr acc = Fr::one();
  int cnt = 0;
  for (size_t i = t; i < n; i += T, ++cnt) {
    tmp[i] = acc;
    Fr x = a[i];
    if (!x.is_zero()) acc = acc * x;
  }
  acc = fp_inv<FrP>(acc);
  for (int k = cnt - 1; k >= 0; --k) {
    size_t i = t + (size_t)k * T;
    Fr x = a[i];
    if (x.is_zero()) continue;
    Fr inv = acc * tmp[i];
    acc = acc * x;
    a[i] = inv;
  }
}

static unsigned ew_grid(zkfhe_ctx *ctx, size_t n) {
  size_t b = (n + 255) / 256;
  size_t cap = (size_t)ctx->num_cu * 8;
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

extern "C" {

int zkfhe_fr_add(zkfhe_ctx *ctx, const zkfhe_fr *a, const zkfhe_fr *b, zkfhe_fr *out, size_t n) {
  ZK_ENTER(ctx);
  if (!n) return ZKFHE_OK;
  k_fr_binop<OP_ADD><<<ew_grid(ctx, n), 256, 0, ctx->stream>>>((const Fr *)a, (const Fr *)b, (Fr *)out, n);
  ZK_LAUNCH_CHECK(ctx);
  return ZKFHE_OK;
}
int zkfhe_fr_sub(zkfhe_ctx *ctx, const zkfhe_fr *a, const zkfhe_fr *b, zkfhe_fr *out, size_t n) {
  ZK_ENTER(ctx);
  if (!n) return ZKFHE_OK;
  k_fr_binop<OP_SUB><<<ew_grid(ctx, n), 256, 0, ctx->stream>>>((const Fr *)a, (const Fr *)b, (Fr *)out, n);
  ZK_LAUNCH_CHECK(ctx);
  return ZKFHE_OK;
}
int zkfhe_fr_mul(zkfhe_ctx *ctx, const zkfhe_fr *a, const zkfhe_fr *b, zkfhe_fr *out, size_t n) {
  ZK_ENTER(ctx);
  if (!n) return ZKFHE_OK;
  k_fr_binop<OP_MUL><<<ew_grid(ctx, n), 256, 0, ctx->stream>>>((const Fr *)a, (const Fr *)b, (Fr *)out, n);
  ZK_LAUNCH_CHECK(ctx);
  return ZKFHE_OK;
}
int zkfhe_fr_scale(zkfhe_ctx *ctx, const zkfhe_fr *a, const zkfhe_fr *s_host, zkfhe_fr *out, size_t n) {
  ZK_ENTER(ctx);
  if (!n) return ZKFHE_OK;
  ZK_ARG(ctx, s_host != nullptr);
  Fr s;
  memcpy(&s, s_host, 32);
  k_fr_unop<0><<<ew_grid(ctx, n), 256, 0, ctx->stream>>>((const Fr *)a, s, (Fr *)out, n);
  ZK_LAUNCH_CHECK(ctx);
  return ZKFHE_OK;
}
int zkfhe_fr_to_mont(zkfhe_ctx *ctx, const zkfhe_fr *a, zkfhe_fr *out, size_t n) {
  ZK_ENTER(ctx);
  if (!n) return ZKFHE_OK;
  k_fr_unop<1><<<ew_grid(ctx, n), 256, 0, ctx->stream>>>((const Fr *)a, Fr::zero(), (Fr *)out, n);
  ZK_LAUNCH_CHECK(ctx);
  return ZKFHE_OK;
}
int zkfhe_fr_from_mont(zkfhe_ctx *ctx, const zkfhe_fr *a, zkfhe_fr *out, size_t n) {
  ZK_ENTER(ctx);
  if (!n) return ZKFHE_OK;
  k_fr_unop<2><<<ew_grid(ctx, n), 256, 0, ctx->stream>>>((const Fr *)a, Fr::zero(), (Fr *)out, n);
  ZK_LAUNCH_CHECK(ctx);
  return ZKFHE_OK;
}
int zkfhe_fr_batch_invert(zkfhe_ctx *ctx, zkfhe_fr *a, size_t n) {
  ZK_ENTER(ctx);
  if (!n) return ZKFHE_OK;
  void *tmp;
  int rc = zk_scratch(ctx, 0, n * sizeof(Fr), &tmp);
  if (rc) return rc;
  // elements per thread (= per inversion, ~110 product-times each against the 3 products per element of the prefix trick): 8 while
  // that is what it takes to put four waves on every SIMD, up to 32 on the long arrays of k >= 15 (k = 19: 3.5 -> 1.9 ms per proof)
  static const int forced = getenv("ZKFHE_BI_CHUNK") ? atoi(getenv("ZKFHE_BI_CHUNK")) : 0;
  size_t chunk = forced > 0 ? (size_t)forced : n / ((size_t)ctx->num_cu * 1024);
  if (forced <= 0) chunk = chunk < BI_CHUNK ? BI_CHUNK : (chunk > 32 ? 32 : chunk);
  if (forced <= 0 && n >= BI_LONG && chunk < 16) chunk = 16;
  size_t T = (n + chunk - 1) / chunk;
  k_fr_batch_invert<<<zk_blocks(T, 256), 256, 0, ctx->stream>>>((Fr *)a, (Fr *)tmp, n, T);
  ZK_LAUNCH_CHECK(ctx);
  return ZKFHE_OK;
}
int zkfhe_fr_sqr_chain(zkfhe_ctx *ctx, const zkfhe_fr *a, zkfhe_fr *out, size_t n, int iters) {
  ZK_ENTER(ctx);
  if (!n) return ZKFHE_OK;
  k_fr_sqr_chain<<<zk_blocks(n, 256), 256, 0, ctx->stream>>>((const Fr *)a, (Fr *)out, n, iters);
  ZK_LAUNCH_CHECK(ctx);
  return ZKFHE_OK;
}

int zkfhe_fq29_sqr_chain(zkfhe_ctx *ctx, const zkfhe_fq *a, zkfhe_fq *out, size_t n, int iters) {
  ZK_ENTER(ctx);
  if (!n) return ZKFHE_OK;
  k_fq29_sqr_chain<<<zk_blocks(n, 256), 256, 0, ctx->stream>>>((const Fq *)a, (Fq *)out, n, iters);
  ZK_LAUNCH_CHECK(ctx);
  return ZKFHE_OK;
}

}  // extern "C"
