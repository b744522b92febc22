/* zkfhe.h -- C ABI of the MI355X (gfx950) backend for zk-fhe's BFV-proof hot path.
 *
 * The reference (enricobottazzi/zk-fhe) is a Rust crate with no FFI of its own; its prover hot loops
 * live in third-party crates reached through the single call `run_eth(bfv_encryption_circuit, args)`
 * (reference examples/bfv.rs:311).  Each entry point below names the Rust seam it replaces -- the
 * function a maintainer would re-route through `extern "C"` (binding stubs: INTEGRATION.md):
 *
 *   zkfhe_ntt_batch        <- halo2_proofs::arithmetic::best_fft(&mut [Fr], omega, log_n) and
 *                             poly::EvaluationDomain::{lagrange_to_coeff, coeff_to_lagrange}
 *   zkfhe_coset_ntt_batch  <- EvaluationDomain::{coeff_to_extended, extended_to_coeff}
 *   zkfhe_basis_create     <- poly::kzg::commitment::ParamsKZG {g, g_lagrange} (the SRS halves)
 *   zkfhe_msm_batch        <- arithmetic::best_multiexp(&[Fr], &[G1Affine]) as used by
 *                             ParamsKZG::{commit, commit_lagrange}
 *   zkfhe_fr_*             <- the coefficient-wise Fr mul/add/sub loops of `parallelize(...)` bodies
 *                             (and src/poly_chip.rs:122-174 add / scalar_mul witness values)
 *   zkfhe_fr_batch_invert  <- ff::BatchInvert / halo2 `batch_invert_assigned`
 *   zkfhe_witness_*        <- the per-coefficient witness loops of src/poly_chip.rs:226-252
 *                             (reduce_by_modulo -> RangeChip::div_mod) and src/poly.rs:75-191
 *
 * Data layouts (identical to halo2curves' in-memory representation):
 *   Fr, Fq      4 x uint64_t little-endian limbs, Montgomery form with R = 2^256, value < p.
 *   G1 affine   {Fq x, Fq y} = 64 bytes; the identity is (0, 0).
 * Conventions: every call returns 0 on success or a negative ZKFHE_E* code and never throws or aborts
 * across the ABI; zkfhe_last_error() gives the message.  `*_dev` pointers are device (HBM) addresses
 * obtained from zkfhe_dev_alloc (or any hipMalloc'd buffer of the same process); everything else is
 * host memory.  One context per GPU; calls on one context are ordered on its stream and are
 * asynchronous with respect to the host unless stated (zkfhe_sync / zkfhe_download wait).
 * There is no CPU fallback: without the HIP runtime and a gfx950 device zkfhe_ctx_create fails.
 */
#ifndef ZKFHE_H
#define ZKFHE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZKFHE_OK 0
#define ZKFHE_EINVAL (-1)   /* bad argument */
#define ZKFHE_EHIP (-2)     /* HIP runtime error (message in zkfhe_last_error) */
#define ZKFHE_ENOMEM (-3)
#define ZKFHE_ENODEV (-4)   /* no usable gfx950 device */

typedef struct zkfhe_ctx zkfhe_ctx;
typedef struct zkfhe_basis zkfhe_basis;

typedef struct { uint64_t l[4]; } zkfhe_fr;
typedef struct { uint64_t l[4]; } zkfhe_fq;
typedef struct { zkfhe_fq x, y; } zkfhe_g1_affine;
/* accumulator form of a G1 point (EFD "XYZZ"): x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2; the identity has ZZ = ZZZ = 0.  128 bytes,
 * raw Montgomery coordinates.  What an MSM holds before its one field inversion. */
typedef struct { zkfhe_fq x, y, zz, zzz; } zkfhe_g1_xyzz;

/* ---- context / memory -------------------------------------------------------------------- */
/* hip_stream: an existing hipStream_t to run on, or NULL to let the context create its own. */
int zkfhe_ctx_create(int device_id, void *hip_stream, zkfhe_ctx **out);
int zkfhe_ctx_destroy(zkfhe_ctx *ctx);
const char *zkfhe_last_error(const zkfhe_ctx *ctx);   /* ctx may be NULL: last creation error */
int zkfhe_sync(zkfhe_ctx *ctx);
void *zkfhe_stream(zkfhe_ctx *ctx);                   /* the hipStream_t the context launches on */
int zkfhe_device_info(zkfhe_ctx *ctx, char *arch_name, size_t arch_len, int *num_cu, size_t *hbm_bytes);

int zkfhe_dev_alloc(zkfhe_ctx *ctx, size_t bytes, void **dptr);
int zkfhe_dev_free(zkfhe_ctx *ctx, void *dptr);
int zkfhe_upload(zkfhe_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);     /* waits */
int zkfhe_download(zkfhe_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);   /* waits */
int zkfhe_copy_dev(zkfhe_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes);
int zkfhe_memset_dev(zkfhe_ctx *ctx, void *dst_dev, int byte, size_t bytes);

/* HIP-event timing on the context's stream (what bench.py uses for per-kernel durations). */
int zkfhe_timer_start(zkfhe_ctx *ctx);
int zkfhe_timer_stop_ms(zkfhe_ctx *ctx, float *ms);   /* waits for the stop event */

/* Per-kernel profiling with HIP events on the context's stream.  While enabled, zkfhe_msm_batch and
 * zkfhe_ntt_batch bracket their dominant kernel (which 0: the summing kernel of a wide MSM call -- k_msm_table or k_msm_accumulate --, 1: the NTT tile kernel (k_ntt13 at n >= 2^13),
 * 2: k_msm_table of a call of a few columns) with an event pair
 * and wait for it, accumulating duration, launch count and ALGORITHMIC bytes (MSM: 96 B per term, NTT: 64 B per
 * point -- BASELINE.md).  Meant for a separate, untimed pass (it serialises the stream). */
int zkfhe_prof_enable(zkfhe_ctx *ctx, int on);
int zkfhe_prof_reset(zkfhe_ctx *ctx);
int zkfhe_prof_read(zkfhe_ctx *ctx, int which, double *total_ms, uint64_t *launches, double *algorithmic_bytes);
/* arithmetic units of the profiled launches: which = 0 -> mixed point additions (k_msm_accumulate), 1 -> butterflies (NTT tile kernel) */
int zkfhe_prof_read_ops(zkfhe_ctx *ctx, int which, double *ops);

/* Host-side marks of the LAST zkfhe_bfv_prove on this context, in ms from its start: [0] the phase-0 commitment is back from the
 * GPU (what the transcript absorbs next), [1] the first challenge is squeezed -- it stands behind the sponge over the public
 * inputs (examples/bfv.rs:118-122: 5 N + 1 values, one sequential Poseidon chain with the reference's transcript), so [1] - [0]
 * is the time a lone proof waits for the HOST --, [2] the proof is complete. */
int zkfhe_ctx_last_proof_marks(zkfhe_ctx *ctx, float marks_ms[3]);

/* ---- coefficient-wise Fr arithmetic (device buffers, out may alias a or b) ----------------- */
int zkfhe_fr_add(zkfhe_ctx *ctx, const zkfhe_fr *a_dev, const zkfhe_fr *b_dev, zkfhe_fr *out_dev, size_t n);
int zkfhe_fr_sub(zkfhe_ctx *ctx, const zkfhe_fr *a_dev, const zkfhe_fr *b_dev, zkfhe_fr *out_dev, size_t n);
int zkfhe_fr_mul(zkfhe_ctx *ctx, const zkfhe_fr *a_dev, const zkfhe_fr *b_dev, zkfhe_fr *out_dev, size_t n);
/* out[i] = a[i] * s   (s: one host-side Fr) */
int zkfhe_fr_scale(zkfhe_ctx *ctx, const zkfhe_fr *a_dev, const zkfhe_fr *s_host, zkfhe_fr *out_dev, size_t n);
/* canonical integer <-> Montgomery form */
int zkfhe_fr_to_mont(zkfhe_ctx *ctx, const zkfhe_fr *a_dev, zkfhe_fr *out_dev, size_t n);
int zkfhe_fr_from_mont(zkfhe_ctx *ctx, const zkfhe_fr *a_dev, zkfhe_fr *out_dev, size_t n);
/* in place a[i] <- a[i]^-1, zero stays zero (halo2 batch_invert convention) */
int zkfhe_fr_batch_invert(zkfhe_ctx *ctx, zkfhe_fr *a_dev, size_t n);
/* modmul micro-benchmark: out[i] = a[i]^(2^iters) by repeated squaring (ALU-roofline probe) */
int zkfhe_fr_sqr_chain(zkfhe_ctx *ctx, const zkfhe_fr *a_dev, zkfhe_fr *out_dev, size_t n, int iters);
/* the same probe for the radix-2^29 product the MSM kernels use (nine 29-bit limbs, Montgomery constant 2^261): a[i] < q as a
 * packed 256-bit integer, out[i] = a[i]^(2^iters) * (2^-261)^(2^iters - 1) mod q, canonical */
int zkfhe_fq29_sqr_chain(zkfhe_ctx *ctx, const zkfhe_fq *a_dev, zkfhe_fq *out_dev, size_t n, int iters);

/* ---- NTT ---------------------------------------------------------------------------------- */
/* n_cols independent transforms of length 2^log_n, column c at cols_dev + c * 2^log_n, in place,
 * natural order in and out.  inverse = 0: out[i] = sum_j a[j] w^(ij) with w = halo2's omega for
 * this log_n (= ROOT_OF_UNITY^(2^(28-log_n)));  inverse = 1: w^-1 and a final multiplication by
 * n^-1 (lagrange_to_coeff).  1 <= log_n <= 26. */
int zkfhe_ntt_batch(zkfhe_ctx *ctx, zkfhe_fr *cols_dev, size_t n_cols, int log_n, int inverse);
/* The same transform out of place (in_dev and out_dev distinct, not overlapping): EvaluationDomain::lagrange_to_coeff /
 * coeff_to_lagrange consume one Polynomial and return another.  This is the form the kernels run natively at n = 2^13 (two
 * workgroups per column, each reading all of it): the in-place call above goes through a scratch copy there. */
int zkfhe_ntt_batch_to(zkfhe_ctx *ctx, const zkfhe_fr *in_dev, zkfhe_fr *out_dev, size_t n_cols, int log_n, int inverse);

/* coeff_to_extended (inverse = 0): column c holds 2^log_n coefficients at in_dev + c*2^log_n; writes
 * the 2^(log_n+log_ext_factor) evaluations over the coset g*<w_ext> to out_dev + c*2^(log_n+lef) in
 * COSET-MAJOR order: out[k1*2^log_n + k2] = f(g * w_ext^(k1 + 2^lef * k2)), i.e. row k1 is the
 * evaluation over (g*w_ext^k1)*<w>; a rotation by w stays inside a row.
 * extended_to_coeff (inverse = 1): the same layout in -> 2^(log_n+lef) coefficients out (natural order).
 * g is `zeta`-coset generator of halo2's extended domain, passed by the caller (host Fr). */
int zkfhe_coset_ntt_batch(zkfhe_ctx *ctx, const zkfhe_fr *in_dev, zkfhe_fr *out_dev, size_t n_cols,
                          int log_n, int log_ext_factor, const zkfhe_fr *g_host, int inverse);

/* ---- MSM (KZG commit) ---------------------------------------------------------------------- */
/* Uploads n affine bases (host memory) and builds the per-window tables 2^(c*w) * P_i used by the
 * single-bucket-set Pippenger (DESIGN.md "MSM").  window_bits = 0 picks the default for n. */
int zkfhe_basis_create(zkfhe_ctx *ctx, const zkfhe_g1_affine *bases_host, size_t n, int window_bits,
                       zkfhe_basis **out);
int zkfhe_basis_destroy(zkfhe_ctx *ctx, zkfhe_basis *basis);
size_t zkfhe_basis_len(const zkfhe_basis *basis);
/* out_dev[c] = sum_i scalars_dev[c*n + i] * bases[i], c < n_cols; n = zkfhe_basis_len; result affine */
int zkfhe_msm_batch(zkfhe_ctx *ctx, const zkfhe_basis *basis, const zkfhe_fr *scalars_dev, size_t n_cols,
                    zkfhe_g1_affine *out_dev);

/* A few non-zero scalars against a basis (one with a digit-multiple table: zkfhe_basis_has_multiples; n <= 2^16, default
 * window bits): out_dev[slot] = sum of scalar * P_row over the terms of that slot, slot < n_slots.  scalar: Montgomery Fr.
 * One wave per slot -- the right tool when a column is all zero except a handful of cells. */
typedef struct { zkfhe_fr scalar; uint32_t row; uint32_t slot; } zkfhe_sparse_term;
int zkfhe_msm_sparse(zkfhe_ctx *ctx, const zkfhe_basis *basis, const zkfhe_sparse_term *terms_dev, size_t n_terms, size_t n_slots,
                     zkfhe_g1_affine *out_dev);
/* The same two calls with the sums left in the accumulator form: the call's last kernel skips its field inversion (a 40 us
 * dependent chain in one lane at the end of every call) and the caller normalises many points with ONE inversion on the host --
 * what halo2's create_proof does with a round's commitments (`commit_lagrange` returns projective points, the round is
 * batch-normalised: SURVEY.md Appendix B step 2).  out_dev may be pinned host memory, like every *_dev output of this header.
 * zkfhe_g1_xyzz_to_affine runs on the calling thread (no device work): Montgomery's trick over the ZZZ, one inversion for the
 * array; in / out are host pointers and must not overlap. */
int zkfhe_msm_batch_xyzz(zkfhe_ctx *ctx, const zkfhe_basis *basis, const zkfhe_fr *scalars_dev, size_t n_cols,
                         zkfhe_g1_xyzz *out_dev);
int zkfhe_msm_sparse_xyzz(zkfhe_ctx *ctx, const zkfhe_basis *basis, const zkfhe_sparse_term *terms_dev, size_t n_terms, size_t n_slots,
                          zkfhe_g1_xyzz *out_dev);
int zkfhe_g1_xyzz_to_affine(const zkfhe_g1_xyzz *in, size_t n, zkfhe_g1_affine *out);
int zkfhe_basis_has_multiples(const zkfhe_basis *basis);
/* Digit width of the basis' digit-multiple table (0: none).  *wide_calls (optional) = 1 when calls of many columns take the
 * table path too (k_msm_table), 0 when they take the bucket pipeline (k_msm_accumulate ...) and only calls of <= 8 columns
 * go through the table. */
int zkfhe_basis_table_bits(const zkfhe_basis *basis, int *wide_calls);
/* Resident bytes of that table; *narrowed (optional) = 1 when it is narrower than its budget allowed because the device did not
 * have the room when the basis was made (ZKFHE_TABLE_GB: default 48 and at most a quarter of the free memory; a reserve of an
 * eighth of the device, at least 24 GB, always stays free for keys and workspaces -- ZKFHE_TABLE_RESERVE_GB). */
size_t zkfhe_basis_table_bytes(const zkfhe_basis *basis, int *narrowed);

/* ---- intra-proof multi-GPU: commitments sharded by point range (replaces nothing in the reference,
 * whose prover is single-process -- the seam is again best_multiexp inside ParamsKZG::{commit, commit_lagrange}) ----------
 * One process per GPU.  Rank r of W owns the bases [n r / W, n (r+1) / W) of each SRS half and the same rows of every
 * column; zkfhe_msm_batch_sharded computes its partial sums, all-gathers the 64-byte affine partials as raw bytes
 * (ncclAllGather of ncclUint8 over RCCL / xGMI, on the context's stream) and adds them on every rank, so all ranks hold the
 * same commitments, the same transcript and the same proof bytes as a single GPU.  An SRS made with
 * zkfhe_srs_create_sharded carries the communicator: zkfhe_bfv_keygen / zkfhe_bfv_prove then shard every commitment with
 * no further change (every rank calls them with the same inputs and seed).
 * zkfhe_comm_create needs librccl.so at run time (dlopen; nothing is linked).  zkfhe_comm_create_with_transport takes a host
 * all-gather callback instead (MPI, gloo, a test shim): recv = the `world` send buffers of `bytes` bytes each, rank-major;
 * return 0 on success. */
typedef struct zkfhe_comm zkfhe_comm;
typedef int (*zkfhe_allgather_fn)(void *user, const void *send, size_t bytes, void *recv);
int zkfhe_comm_unique_id(uint8_t id_out[128]);   /* rank 0: ncclGetUniqueId, to be passed to the other ranks out of band */
int zkfhe_comm_create(zkfhe_ctx *ctx, int rank, int world, const uint8_t unique_id[128], zkfhe_comm **out);
int zkfhe_comm_create_with_transport(zkfhe_ctx *ctx, int rank, int world, zkfhe_allgather_fn allgather, void *user, zkfhe_comm **out);
int zkfhe_comm_destroy(zkfhe_ctx *ctx, zkfhe_comm *comm);
int zkfhe_comm_rank(const zkfhe_comm *comm);
int zkfhe_comm_world(const zkfhe_comm *comm);
/* non-zero when commitments made with this communicator go through a collective: world > 1, or a one-rank RCCL communicator */
int zkfhe_comm_active(const zkfhe_comm *comm);
void zkfhe_comm_point_range(const zkfhe_comm *comm, size_t n, size_t *lo, size_t *hi);
int zkfhe_comm_all_gather(zkfhe_ctx *ctx, zkfhe_comm *comm, const void *send_dev, void *recv_dev, size_t bytes);
/* The same collective on the communicator's own stream (RCCL and one-rank communicators; the callback transport completes on the
 * context's stream as zkfhe_comm_all_gather does): it starts behind everything queued on the context's stream so far and overlaps
 * what is queued there next; zkfhe_comm_join orders the context's stream or the caller behind it.  Both buffers stay untouched
 * until then. */
int zkfhe_comm_all_gather_async(zkfhe_ctx *ctx, zkfhe_comm *comm, const void *send_dev, void *recv_dev, size_t bytes);
/* basis_slice: a basis made of this rank's point range; column c's scalars for that range at scalars_dev + c * col_stride
 * (col_stride = the full column length when scalars_dev points at row lo of column 0).  out_dev[c]: the full MSM, on every rank. */
int zkfhe_msm_batch_sharded(zkfhe_ctx *ctx, zkfhe_comm *comm, const zkfhe_basis *basis_slice, const zkfhe_fr *scalars_dev,
                            size_t col_stride, size_t n_cols, zkfhe_g1_affine *out_dev);
/* The same with the collective off the context's stream: the partial MSM runs on the context's stream, the all-gather and the sum
 * on the communicator's own, so kernels queued on the context's stream after the call (the next batch's partial MSM, witness
 * kernels) overlap the exchange over xGMI.  out_dev is complete after zkfhe_comm_join: block_host = 0 makes the context's stream
 * wait for every collective queued so far, 1 the calling thread.  zkfhe_comm_record_event records a HIP event (hipEvent_t)
 * behind them instead.  With a callback transport (host all-gather) the call completes on the context's stream like
 * zkfhe_msm_batch_sharded and the join is a plain wait.  Every rank must issue these calls in the same order. */
int zkfhe_msm_batch_sharded_async(zkfhe_ctx *ctx, zkfhe_comm *comm, const zkfhe_basis *basis_slice, const zkfhe_fr *scalars_dev, size_t col_stride,
                                  size_t n_cols, zkfhe_g1_affine *out_dev);
int zkfhe_comm_join(zkfhe_ctx *ctx, zkfhe_comm *comm, int block_host);
int zkfhe_comm_record_event(zkfhe_ctx *ctx, zkfhe_comm *comm, void *hip_event);

/* ---- G1 helpers (device, used by tests and by the SRS builder) ------------------------------ */
/* out[i] = a[i] + b[i] (affine in, affine out; handles doubling / inverse / identity) */
int zkfhe_g1_add(zkfhe_ctx *ctx, const zkfhe_g1_affine *a_dev, const zkfhe_g1_affine *b_dev,
                 zkfhe_g1_affine *out_dev, size_t n);
/* out[i] = k[i] * p[i] */
int zkfhe_g1_mul(zkfhe_ctx *ctx, const zkfhe_g1_affine *p_dev, const zkfhe_fr *k_dev,
                 zkfhe_g1_affine *out_dev, size_t n);

/* ---- BFV witness kernels -------------------------------- */
/* Negacyclic product in R_q = Z_q[x]/(x^N+1) is NOT what the reference computes: Poly::mul
 * (src/poly.rs:75-103) is the plain integer product of two degree-(N-1) polynomials, 2N-1 coefficients.
 * a, b: N canonical integers < 2^64 each (uint64), big-endian coefficient order as in bfv.in;
 * out: 2N-1 Montgomery Fr values (exact integers, < 2^132 << r).  N a power of two <= 2^20. */
int zkfhe_witness_poly_mul_u64(zkfhe_ctx *ctx, const uint64_t *a_dev, const uint64_t *b_dev, size_t n,
                               zkfhe_fr *out_dev);
/* The same product on the host for short, narrow polynomials (n a power of two <= 2048, every coefficient below 2^32: the
 * k = 13 circuit's pk_i * u): an exact NTT convolution over p = 2^64 - 2^32 + 1 on one core (host/poly_ntt64.hpp), which is
 * what Poly::mul uses for them inside zkfhe_bfv_prove.  lo / hi: 2n - 1 coefficients as 128-bit integers.  ZKFHE_EINVAL when
 * the operands do not fit. */
int zkfhe_host_poly_mul_u32(const uint64_t *a, const uint64_t *b, size_t n, uint64_t *lo, uint64_t *hi);
/* RangeChip::div_mod witness (src/poly_chip.rs:236-246): for canonical values a[i] < 2^128 held as
 * Montgomery Fr, q a u64 modulus: div[i] = floor(a/q), rem[i] = a mod q (both returned as Montgomery Fr). */
int zkfhe_witness_div_mod(zkfhe_ctx *ctx, const zkfhe_fr *a_dev, uint64_t q, zkfhe_fr *div_dev,
                          zkfhe_fr *rem_dev, size_t n);

/* ---- Fiat-Shamir transcript (host only; replaces snark-verifier `PoseidonTranscript<NativeLoader>` /
 * halo2_proofs `Blake2bWrite`, reached from reference examples/bfv.rs:311 via gen_snark_shplonk) -------------- */
#define ZKFHE_TRANSCRIPT_POSEIDON 0   /* T = 3, RATE = 2, R_F = 8, R_P = 57 over BN254 Fr: what the reference proves / verifies with */
#define ZKFHE_TRANSCRIPT_BLAKE2B 1    /* halo2's "Halo2-Transcript" Blake2b with Challenge255 */
typedef struct zkfhe_transcript zkfhe_transcript;
int zkfhe_transcript_create(uint32_t kind, zkfhe_transcript **out);
void zkfhe_transcript_destroy(zkfhe_transcript *t);
/* scalars: canonical 32-byte little-endian Fr; points: canonical affine x || y (64 bytes, little-endian Fq each).
 * common_* only absorb; write_* also append the 32-byte encoding to the byte stream: a scalar as it is, a point as halo2curves'
 * bn256 G1Affine::to_bytes -- x little-endian, (y & 1) << 6 in byte 31, the identity as 0x80 in byte 31 and zeros. */
int zkfhe_transcript_common_scalar(zkfhe_transcript *t, const uint8_t s_le[32]);
int zkfhe_transcript_write_scalar(zkfhe_transcript *t, const uint8_t s_le[32]);
int zkfhe_transcript_common_point(zkfhe_transcript *t, const uint8_t xy_le[64]);
int zkfhe_transcript_write_point(zkfhe_transcript *t, const uint8_t xy_le[64]);
int zkfhe_transcript_squeeze(zkfhe_transcript *t, uint8_t challenge_le[32]);
int zkfhe_transcript_bytes(const zkfhe_transcript *t, uint8_t *out, size_t cap, size_t *len);
/* The Poseidon instance itself: one permutation of three canonical 32-byte LE words in place, and the generated
 * constants (65 x 3 round constants, 3 x 3 MDS, canonical 32-byte LE each) for cross-checks. */
int zkfhe_poseidon_permute(uint8_t state_le[96]);
int zkfhe_poseidon_constants(uint8_t round_constants_le[65 * 3 * 32], uint8_t mds_le[9 * 32]);
/* n_jobs independent sponge hashes (fresh sponge, absorb counts[j] canonical scalars taken in order from values_le, squeeze) --
 * the parity and timing hook of the eight-lane sponge engine the prover's transcripts share when several proofs are in flight
 * (host/poseidon_x8.cpp: one sponge per AVX-512 IFMA lane, ragged lengths, lanes refilled as they run dry).
 * mode 0: the scalar / single-sponge path (what a lone transcript runs); 1: eight lanes on the calling thread; 2: through the
 * hash service's worker threads, as the prover does.  Modes 1 and 2 return ZKFHE_ENODEV on a CPU without AVX-512 IFMA.
 * No GPU involved; the same digests in every mode. */
/* How the Poseidon transcripts of the proofs in flight in this process hash: 0 = every transcript on its own (lowest latency:
 * the default), 1 = long runs of all of them through the shared eight-lane service (about half the host CPU per proof, about
 * twice the hashing latency: for hosts with few CPUs per GPU), -1 = query.  Returns the mode in force (ZKFHE_EINVAL for any
 * other argument).  Process-wide; takes effect for transcripts' next runs.  Initial value from ZKFHE_HASH_MODE=latency|shared. */
int zkfhe_host_hash_mode(int mode);
int zkfhe_poseidon_hash_many(const uint8_t *values_le, const size_t *counts, size_t n_jobs, int mode, uint8_t *digests_le);

/* ---- BFV circuit: witness tables, keygen, prove (reference examples/bfv.rs + halo2-scaffold run_eth) ---- */
/* Runtime form of the compile-time constants at examples/bfv.rs:27-30. */
typedef struct { uint64_t n; uint64_t q; uint64_t t; uint64_t b; } zkfhe_bfv_params;
/* configs/<name>.json "params" + "break_points" (README.md:38).  With n_break_* == 0 and replay == 0 the
 * break points are computed (keygen stage); with replay != 0 they are replayed (prover stage). */
typedef struct {
  uint32_t k, n_gate0, n_gate1, n_lookup, n_rlc, unusable_rows, lookup_bits;
  const uint32_t *bp_gate0; uint32_t n_bp_gate0;
  const uint32_t *bp_gate1; uint32_t n_bp_gate1;
  const uint32_t *bp_rlc;   uint32_t n_bp_rlc;
  int replay;
  uint32_t transcript;   /* ZKFHE_TRANSCRIPT_*: part of the verifying key (bound into its digest) */
} zkfhe_bfv_config;

typedef struct zkfhe_bfv_tables zkfhe_bfv_tables;
/* Host-only (no GPU): runs the circuit (examples/bfv.rs:63-304) on the JSON input text (CircuitInput,
 * examples/bfv.rs:50-61) with RLC challenge `gamma` (32-byte LE canonical Fr) and places the cell streams
 * into columns.  keygen_mode != 0 also produces the fixed columns and the copy constraints. */
int zkfhe_bfv_build_tables(const char *input_json, const zkfhe_bfv_params *params, const zkfhe_bfv_config *config,
                           const uint8_t gamma[32], int keygen_mode, zkfhe_bfv_tables **out, char *err, size_t err_len);
void zkfhe_bfv_tables_free(zkfhe_bfv_tables *t);
/* Column counts the circuit needs at 2^k rows (halo2-base auto-configuration, the first half of the reference's keygen):
 * counts_out = { n_gate0, n_gate1, n_lookup, n_rlc }.  Host only. */
int zkfhe_bfv_auto_config(const char *input_json, const zkfhe_bfv_params *params, uint32_t k, uint32_t unusable_rows, uint32_t lookup_bits,
                          uint32_t counts_out[4], char *err, size_t err_len);
/* what: 0 n_advice, 1 n_fixed, 2 n rows, 3 n_instance, 4 n_copies, 5/6/7 number of break points gate0/gate1/rlc,
 *       8/9/10 cells in the phase-0 / phase-1 gate / RLC stream, 11 lookup cells */
size_t zkfhe_bfv_tables_count(const zkfhe_bfv_tables *t, int what);
/* canonical (non-Montgomery) 4 x u64 values */
int zkfhe_bfv_tables_copy_advice(const zkfhe_bfv_tables *t, uint64_t *out);     /* n_advice * n * 4 */
int zkfhe_bfv_tables_copy_fixed(const zkfhe_bfv_tables *t, uint64_t *out);      /* n_fixed * n * 4  */
int zkfhe_bfv_tables_copy_instance(const zkfhe_bfv_tables *t, uint64_t *out);   /* n_instance * 4   */
int zkfhe_bfv_tables_copy_copies(const zkfhe_bfv_tables *t, uint64_t *out);     /* n_copies * 2 (cell = perm_col * n + row) */
int zkfhe_bfv_tables_copy_break_points(const zkfhe_bfv_tables *t, int which, uint32_t *out);

/* Process-wide admission gate of the proofs in flight: at most n of them inside the GPU-heavy middle of a proof (grand products,
 * their commitment, coset extension, quotient) at a time, first come first served; 0 = no gate (default, or ZKFHE_GATE), n < 0 =
 * query.  Returns the previous setting.  Spreads proofs that would otherwise move through the Fiat-Shamir rounds in lockstep; pays
 * when the streams are kept full, not for a batch that starts and ends together.  Ignored by sharded proofs. */
int zkfhe_prover_gate(int n);

/* `mock` (README.md:18-22, halo2 MockProver::run(..).assert_satisfied()): evaluates every constraint on every row of
 * tables built with keygen_mode != 0 and the same gamma -- gate and RLC-gate identities under their selectors, lookup
 * membership, and both cells of every copy constraint.  *n_failures = number of violated rows / constraints, err = the
 * first one.  Host only.  zkfhe_bfv_tables_poke_advice overwrites one advice cell (negative tests of the checker). */
int zkfhe_bfv_mock_check(const zkfhe_bfv_tables *t, const uint8_t gamma_le[32], uint64_t *n_failures, char *err, size_t err_len);
int zkfhe_bfv_tables_poke_advice(zkfhe_bfv_tables *t, uint32_t column, uint32_t row, const uint8_t value_le[32]);

/* Unsafe seeded test SRS (the reference's gen_srs is an unsafe seeded setup as well, README.md:34): s derived
 * from the seed, g[i] = s^i G, g_lagrange[i] = L_i(s) G, both computed on the GPU and kept as MSM bases. */
typedef struct zkfhe_srs zkfhe_srs;
/* Passing exactly this string as the seed derives s the way the REFERENCE's setup does: halo2-scaffold gen_srs ->
 * ParamsKZG::<Bn256>::setup(k, ChaCha20Rng::from_seed([0u8; 32])), s = Fr::from_u512 of the first 64 bytes of the ChaCha20
 * keystream of the all-zero key (the published RFC 7539 zero-key vector).  Any other seed: Blake2b-512("zkfhe-srs", seed) mod r. */
#define ZKFHE_SRS_HALO2_UNSAFE "halo2:ParamsKZG::setup(k, ChaCha20Rng::from_seed([0u8; 32]))"
int zkfhe_srs_create(zkfhe_ctx *ctx, uint32_t k, const uint8_t *seed, size_t seed_len, zkfhe_srs **out);
/* params/kzg_bn254_<k>.srs (README.md:34; .gitignore:17), the file halo2 `ParamsKZG::write` / `::read` exchange
 * (SerdeFormat::RawBytes): u32 k little-endian | g[2^k] | g_lagrange[2^k] | g2 | s_g2 -- G1 points as x | y, G2 points as
 * x.c0 | x.c1 | y.c0 | y.c1, every coordinate four little-endian u64 Montgomery limbs, i.e. the library's in-memory layout.
 * save: an unsharded SRS made by zkfhe_srs_create, zkfhe_srs_load, or zkfhe_srs_from_points + zkfhe_srs_set_g2.
 * load: checks the frame, that every coordinate is reduced and every point on its curve (as halo2's RawBytes read does). */
int zkfhe_srs_save(zkfhe_ctx *ctx, const zkfhe_srs *srs, const char *path);
/* Releases the host copies of the points an unsharded SRS keeps for zkfhe_srs_save (128 B x 2^k); a later save is refused. */
int zkfhe_srs_drop_host_copy(zkfhe_srs *srs);
int zkfhe_srs_load(zkfhe_ctx *ctx, const char *path, zkfhe_srs **out);
/* The verifier's half, as zkfhe_bfv_verify_g2 takes it: canonical little-endian x.c0 | x.c1 | y.c0 | y.c1 of G2 and s G2.
 * zkfhe_srs_g2: of an SRS in memory (ZKFHE_EINVAL when it has none: from_points without set_g2); zkfhe_srs_file_g2: read from
 * the tail of a params file on the host alone (no GPU: `verify` needs nothing else of the SRS); *k_out = the file's k. */
int zkfhe_srs_g2(const zkfhe_srs *srs, uint8_t g2_le[128], uint8_t s_g2_le[128]);
int zkfhe_srs_set_g2(zkfhe_srs *srs, const uint8_t g2_le[128], const uint8_t s_g2_le[128]);
int zkfhe_srs_file_g2(const char *path, uint32_t *k_out, uint8_t g2_le[128], uint8_t s_g2_le[128]);
/* One ChaCha20 block (RFC 7539 section 2.3; words 12..15 of the state as given): host-only hook that pins the keystream the
 * reference derivation above reads to the published vectors. */
int zkfhe_chacha20_block(const uint8_t key[32], const uint32_t counter_nonce[4], uint8_t out[64]);
/* An SRS from outside (a ceremony file such as the reference's params/kzg_bn254_<k>.srs, README.md:34-38, read by the
 * caller): 2^k points g[i] = s^i G and g_lagrange[i] = L_i(s) G, host memory, affine, Montgomery limbs (halo2curves'
 * in-memory G1Affine).  The library only builds its MSM tables from them; nothing is checked about the ceremony. */
int zkfhe_srs_from_points(zkfhe_ctx *ctx, uint32_t k, const zkfhe_g1_affine *g_host, const zkfhe_g1_affine *g_lagrange_host, zkfhe_srs **out);
/* The same seeded setup, but only this rank's point range of both halves (1 / world of the table memory); keygen and prove
 * called with it shard every commitment over `comm` (see "intra-proof multi-GPU").  comm must outlive the SRS. */
int zkfhe_srs_create_sharded(zkfhe_ctx *ctx, zkfhe_comm *comm, uint32_t k, const uint8_t *seed, size_t seed_len, zkfhe_srs **out);
int zkfhe_srs_destroy(zkfhe_ctx *ctx, zkfhe_srs *srs);
/* zkfhe_basis_table_bits of the Lagrange half of the SRS (the basis of the advice / permutation / lookup commitments). */
int zkfhe_srs_table_bits(const zkfhe_srs *srs, int *wide_calls);
/* bits[0] / bits[1]: digit width of the monomial / Lagrange half's table (0 = none); *bytes: resident bytes of both; *narrowed: a
 * half is narrower than its budget allowed (no room on the device at creation: slower calls, same results).  Outputs optional. */
int zkfhe_srs_table_info(const zkfhe_srs *srs, int bits[2], uint64_t *bytes, int *narrowed);

/* keygen (README.md:28-38): circuit structure from the (empty) input, fixed + sigma polynomials, their
 * commitments, the vk digest; everything the prover needs stays resident in HBM. */
typedef struct zkfhe_bfv_pk zkfhe_bfv_pk;
int zkfhe_bfv_keygen(zkfhe_ctx *ctx, const zkfhe_srs *srs, const char *input_json, const zkfhe_bfv_params *params,
                     const zkfhe_bfv_config *config, zkfhe_bfv_pk **out);
int zkfhe_bfv_pk_destroy(zkfhe_ctx *ctx, zkfhe_bfv_pk *pk);
/* A key keeps one prover workspace per context that proved against it (0.3 GB at k = 13, several GB at k = 19) until the
 * key is destroyed.  Call this before zkfhe_ctx_destroy when the key outlives the context. */
int zkfhe_bfv_pk_release_ctx(zkfhe_ctx *ctx, zkfhe_bfv_pk *pk);
/* 32-byte LE vk digest; counts of fixed / sigma commitments; the commitments as canonical affine (x||y, 64 B each) */
int zkfhe_bfv_pk_info(const zkfhe_bfv_pk *pk, uint8_t vk_digest[32], uint32_t *n_fixed, uint32_t *n_sigma);
int zkfhe_bfv_pk_commitments(const zkfhe_bfv_pk *pk, uint8_t *fixed_out, uint8_t *sigma_out);
/* Per-public-key transcript cache.  The instance column starts with pk0 | pk1 (reference examples/bfv.rs:118-119), the same
 * 2 N values for every encryption under one BFV public key; the prover keeps the Fiat-Shamir state after `vk digest | pk0 | pk1`
 * for the last few public keys it has seen with this proving key and starts a proof whose pk0 | pk1 match from there (k = 13:
 * 1 024 of the 2 561 sequential Poseidon permutations before the first challenge).  Same state, same bytes.  Nothing beyond the
 * public key is cached.  capacity >= 0 sets how many keys are remembered (default 8, ZKFHE_PREFIX_CACHE; 0 = off; at most 64),
 * negative only queries; hits / misses / entries are optional outputs. */
int zkfhe_bfv_pk_prefix_cache(const zkfhe_bfv_pk *pk, int capacity, uint64_t *hits, uint64_t *misses, uint64_t *entries);
/* Announce a proof ahead of time.  The first challenge of a proof stands behind one SEQUENTIAL sponge over the 5 N + 1 public inputs
 * (reference examples/bfv.rs:118-122; with the reference's Poseidon transcript 10 241 permutations at N = 4096, 40 961 at N = 16384:
 * 30 / 120 ms on a core, as long as the proof's GPU work) that depends on the input alone.  A caller that knows the input of a LATER
 * proof calls this while the current proof is on the GPU: the public inputs are parsed and absorbed on a helper thread, and the
 * zkfhe_bfv_prove of the same input_json (the same bytes) starts from the parked state (it waits for it if it is not complete yet).
 * The call returns after copying the text.  One-shot: an announcement serves one proof, the oldest announcement of a text first; at
 * most 16 may be pending (ZKFHE_EINVAL beyond).  An input that does not parse serves nobody (zkfhe_bfv_prove words the error).  Same
 * state, same proof bytes.  Host only.  started / taken / pending: optional counters; input_json == NULL only reads them. */
int zkfhe_bfv_pk_prehash(const zkfhe_bfv_pk *pk, const char *input_json, uint64_t *started, uint64_t *taken, uint64_t *pending);
int zkfhe_bfv_pk_break_points(const zkfhe_bfv_pk *pk, int which, uint32_t *out, uint32_t *count);

/* Serialised verifying key: magic "ZKFHEVK2", 8 x u32 configuration (k, n_gate0, n_gate1, n_lookup, n_rlc, unusable_rows,
 * lookup_bits, transcript), u32 n_fixed, u32 n_sigma, 32-byte vk digest, then the fixed and sigma commitments as canonical
 * affine x||y (64 B each).  What `keygen` writes to data/<name>.vk. */
int zkfhe_bfv_pk_export_vk(const zkfhe_bfv_pk *pk, uint8_t *out, size_t cap, size_t *len);
/* Parity hook for the GPU witness generator: the phase-1 gate-context cell stream of examples/bfv.rs:171-301 (1 231 992 cells at
 * the reference's parameters) as produced on the device, canonical 32-byte little-endian values, for a caller-chosen challenge
 * gamma.  cells_out == NULL only returns the cell count. */
int zkfhe_bfv_witness_stream(zkfhe_ctx *ctx, const zkfhe_bfv_pk *pk, const char *input_json, const uint8_t gamma_le[32], uint8_t *cells_out,
                             size_t cap_cells, size_t *n_cells);
/* Lookup argument, step 1 (halo2 lookup::prover `permute_expression_pair`) for the 8-bit table
 * {0..255} over `usable_rows` rows: column c (Montgomery Fr, n rows) -> A' = the inputs sorted, S' = the table permuted so
 * that S'[i] = A'[i] wherever A'[i] differs from A'[i-1], the unused table values filling the other rows in ascending
 * order.  Rows >= usable_rows of the outputs are left untouched (the prover blinds them).  *not_in_table = 1 if an input
 * exceeds 255.  One workgroup per column. */
int zkfhe_lookup_permute(zkfhe_ctx *ctx, const zkfhe_fr *cols_dev, size_t n_cols, size_t n, uint32_t usable_rows, zkfhe_fr *a_dev,
                         zkfhe_fr *s_dev, int *not_in_table);
/* Proving key on disk (the reference's keygen writes data/<name>.pk, README.md:38): configuration, break points, commitments
 * and the fixed / permutation columns; the extended-domain tables are rebuilt on load.  A key is bound to the SRS it was
 * generated with. */
int zkfhe_bfv_pk_save(zkfhe_ctx *ctx, const zkfhe_bfv_pk *pk, const char *path);
int zkfhe_bfv_pk_load(zkfhe_ctx *ctx, const zkfhe_srs *srs, const char *path, zkfhe_bfv_pk **out);

/* data/<name>.snark (README.md:42-52).  snark-verifier-sdk's `Snark { protocol, instances: Vec<Vec<Fr>>, proof: Vec<u8> }` is
 * written with bincode (fixed-width little-endian integers, u64 lengths); its `protocol` field is the compiled PlonkProtocol of
 * the reference's constraint system, which this library does not have.  This container keeps the two
 * fields that ARE common, byte for byte as bincode writes them, behind a 16-byte header:
 *   "ZKFHESN2" | u64 protocol_len = 0 (absent) | u64 1 | u64 n | n x Fr | u64 proof_len | proof bytes
 * with Fr as halo2curves' serde derives it: four little-endian u64 limbs of the MONTGOMERY form (x * 2^256 mod r) -- so a Rust
 * reader skips 16 bytes and `bincode::deserialize::<(Vec<Vec<Fr>>, Vec<u8>)>`s the rest.  Host only.
 * encode: instances as canonical 32-byte LE scalars; out = NULL / cap too small: *len = bytes needed (ZKFHE_EINVAL if cap > 0).
 * decode: also accepts the round-1..3 container ("ZKFHESN1" | u64 n | n canonical scalars | proof); NULL outputs = sizes only. */
int zkfhe_snark_encode(const uint8_t *instances_le, size_t n_instances, const uint8_t *proof, size_t proof_len, uint8_t *out, size_t cap, size_t *len);
int zkfhe_snark_decode(const uint8_t *snark, size_t snark_len, uint8_t *instances_le, size_t *n_instances, uint8_t *proof, size_t *proof_len);

/* verify (README.md:48-52), host CPU only (no GPU, like the reference's verifier): replays the transcript, checks the
 * quotient identity at x, and ends in one BN254 pairing-product check.  instances: n_instances canonical 32-byte LE
 * scalars.  srs_seed: the seed the (unsafe, test) SRS was derived from.  *accepted = 1 iff the proof verifies; a
 * malformed proof is reported as accepted = 0 with the reason in err. */
int zkfhe_bfv_verify(const uint8_t *vk_bytes, size_t vk_len, const uint8_t *instances, size_t n_instances, const uint8_t *proof,
                     size_t proof_len, const uint8_t *srs_seed, size_t seed_len, int *accepted, char *err, size_t err_len);
/* The same check against the verifier's half of an external SRS: G2 and s*G2 as four 32-byte little-endian canonical Fq values
 * each (x.c0, x.c1, y.c0, y.c1), the layout of halo2curves' G2Affine coordinates. */
int zkfhe_bfv_verify_g2(const uint8_t *vk_bytes, size_t vk_len, const uint8_t *instances, size_t n_instances, const uint8_t *proof, size_t proof_len,
                        const uint8_t g2[128], const uint8_t s_g2[128], int *accepted, char *err, size_t err_len);

/* prove (README.md:42-44): witness generation + create_proof.
 * seed: 32 bytes; every blinding scalar of the proof is derived from it (Blake2b(seed || counter)).  ZERO KNOWLEDGE RESTS
 * ON THE SEED: it must be fresh, secret randomness for every proof (the reference draws StdRng::from_entropy()); a known
 * or reused seed makes the blinding rows predictable and the openings then leak the witness (u, e0, e1, m).  A fixed
 * seed is only for reproducible tests.
 * proof_out must hold proof_cap bytes; *proof_len receives the length.  instances_out (may be NULL): canonical
 * 32-byte LE scalars; *n_instances is its capacity on entry and the instance count on return -- if the capacity is too
 * small the call fails with ZKFHE_EINVAL and *n_instances holds the required count.
 * timings_ms (may be NULL): [witness, commit, quotient, open, total]. */
int zkfhe_bfv_prove(zkfhe_ctx *ctx, const zkfhe_srs *srs, const zkfhe_bfv_pk *pk, const char *input_json,
                    const uint8_t seed[32], uint8_t *proof_out, size_t proof_cap, size_t *proof_len,
                    uint8_t *instances_out, size_t *n_instances, float *timings_ms);

const char *zkfhe_version(void);

#ifdef __cplusplus
}
#endif
#endif
