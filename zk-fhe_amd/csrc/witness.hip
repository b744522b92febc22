// BFV witness kernels: the out-of-circuit polynomial arithmetic of
// reference src/poly.rs and the per-coefficient div_mod witnesses of src/poly_chip.rs:226-252.
//
//   zkfhe_witness_poly_mul_u64  <- Poly::mul (src/poly.rs:75-103): plain integer product of two
//       degree-(N-1) polynomials.  The reference does the O(N^2) BigInt schoolbook; here the product is
//       an NTT convolution over Fr: coefficients are < 2^64, the product coefficients are
//       < N * 2^128 <= 2^148 << r, so the result mod r IS the integer result (exact, bit-identical).
//   zkfhe_witness_div_mod       <- RangeChip::div_mod as called by PolyChip::reduce_by_modulo
//       (src/poly_chip.rs:236-246): floor division of the canonical value by the ciphertext modulus Q.
#include <cstring>

#include "ctx.hpp"

using namespace zk;

namespace {

__global__ void __launch_bounds__(256) k_u64_to_fr_padded(const uint64_t *__restrict__ a, size_t n, Fr *__restrict__ out, size_t n_out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_out; i += (size_t)gridDim.x * blockDim.x) {
    Fr t = Fr::zero();
    if (i < n) {
      const uint64_t v = a[i];
      t.l[0] = (u32)v;
      t.l[1] = (u32)(v >> 32);
      t = fp_to_mont<FrP>(t);
    }
    out[i] = t;
  }
}

// a: canonical value < 2^128 in Montgomery form.  q < 2^63.
__global__ void __launch_bounds__(256) k_div_mod(const Fr *__restrict__ a, uint64_t q, Fr *__restrict__ div, Fr *__restrict__ rem, size_t n,
                                                 int *__restrict__ err) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const Fr c = fp_from_mont<FrP>(a[i]);
    if (c.l[4] | c.l[5] | c.l[6] | c.l[7]) {
      atomicExch(err, 1);  // value does not fit 128 bits: caller violated the contract
    }
    const uint64_t lo = (uint64_t)c.l[0] | ((uint64_t)c.l[1] << 32);
    const uint64_t hi = (uint64_t)c.l[2] | ((uint64_t)c.l[3] << 32);
    const uint64_t qh = hi / q;
    uint64_t r = hi % q;
    uint64_t ql = 0;
    for (int b = 63; b >= 0; --b) {
      r = (r << 1) | ((lo >> b) & 1);  // r < q < 2^63 before the shift: no overflow
      ql <<= 1;
      if (r >= q) {
        r -= q;
        ql |= 1;
      }
    }
    Fr d = Fr::zero(), m = Fr::zero();
    d.l[0] = (u32)ql;
    d.l[1] = (u32)(ql >> 32);
    d.l[2] = (u32)qh;
    d.l[3] = (u32)(qh >> 32);
    m.l[0] = (u32)r;
    m.l[1] = (u32)(r >> 32);
    div[i] = fp_to_mont<FrP>(d);
    rem[i] = fp_to_mont<FrP>(m);
  }
}

}  // namespace

extern "C" {

int zkfhe_witness_poly_mul_u64(zkfhe_ctx *ctx, const uint64_t *a_dev, const uint64_t *b_dev, size_t n, zkfhe_fr *out_dev) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, a_dev && b_dev && out_dev);
  ZK_ARG(ctx, n >= 1 && (n & (n - 1)) == 0 && n <= ((size_t)1 << 20));
  int log_m = 1;
  while (((size_t)1 << log_m) < 2 * n) ++log_m;
  const size_t m = (size_t)1 << log_m;
  // scratch slot 2: zkfhe_ntt_batch / zkfhe_fr_* (called below) use slots 0, 1 and 3 only
  void *sp;
  int src = zk_scratch(ctx, 2, 2 * m * sizeof(Fr), &sp);
  if (src) return src;
  Fr *buf = (Fr *)sp;
  unsigned grid = zk_blocks(m, 256);
  k_u64_to_fr_padded<<<grid, 256, 0, ctx->stream>>>(a_dev, n, buf, m);
  k_u64_to_fr_padded<<<grid, 256, 0, ctx->stream>>>(b_dev, n, buf + m, m);
  hipError_t e = hipGetLastError();
  int rc = e == hipSuccess ? ZKFHE_OK : zk_fail(ctx, ZKFHE_EHIP, "k_u64_to_fr_padded", e, __FILE__, __LINE__);
  if (!rc) rc = zkfhe_ntt_batch(ctx, (zkfhe_fr *)buf, 2, log_m, 0);
  if (!rc) rc = zkfhe_fr_mul(ctx, (const zkfhe_fr *)buf, (const zkfhe_fr *)(buf + m), (zkfhe_fr *)buf, m);
  if (!rc) rc = zkfhe_ntt_batch(ctx, (zkfhe_fr *)buf, 1, log_m, 1);
  if (!rc) {
    e = hipMemcpyAsync(out_dev, buf, (2 * n - 1) * sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream);
    if (e != hipSuccess) rc = zk_fail(ctx, ZKFHE_EHIP, "hipMemcpyAsync", e, __FILE__, __LINE__);
  }
  return rc;
}

int zkfhe_witness_div_mod(zkfhe_ctx *ctx, const zkfhe_fr *a_dev, uint64_t q, zkfhe_fr *div_dev, zkfhe_fr *rem_dev, size_t n) {
  ZK_ENTER(ctx);
  ZK_ARG(ctx, q >= 1 && q < ((uint64_t)1 << 63));
  if (!n) return ZKFHE_OK;
  ZK_ARG(ctx, a_dev && div_dev && rem_dev);
  void *p;
  int rc = zk_scratch(ctx, 3, 64, &p);
  if (rc) return rc;
  int *err = (int *)p + 8;
  ZK_HIP(ctx, hipMemsetAsync(err, 0, sizeof(int), ctx->stream));
  unsigned grid = zk_blocks(n, 256);
  const unsigned cap = (unsigned)ctx->num_cu * 8;
  if (grid > cap) grid = cap;
  k_div_mod<<<grid, 256, 0, ctx->stream>>>((const Fr *)a_dev, q, (Fr *)div_dev, (Fr *)rem_dev, n, err);
  ZK_LAUNCH_CHECK(ctx);
  int h = 0;
  ZK_HIP(ctx, hipMemcpyAsync(&h, err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (h) return zk_fail_msg(ctx, ZKFHE_EINVAL, "zkfhe_witness_div_mod: an input value does not fit 128 bits");
  return ZKFHE_OK;
}

}  // extern "C"
