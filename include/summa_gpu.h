/*
 * summa_gpu.h -- C ABI of the MI355X (gfx950) back-end for the Halo2/KZG hot path of
 * summa-dev/circuits-halo2's zk_prover: BN254 G1 multi-scalar multiplication and BN254 Fr
 * radix-2 NTT/iNTT.
 *
 * What it replaces.  The reference reaches this arithmetic only through
 *   zk_prover/src/circuits/utils.rs:75-76   keygen_vk / keygen_pk
 *   zk_prover/src/circuits/utils.rs:94-101  create_proof (full_prover)
 *   zk_prover/src/circuits/utils.rs:171-178 create_proof (create_proof_checked)
 *   zk_prover/src/circuits/utils.rs:55,64,70 ParamsKZG::{read, downsize, setup}
 * and the code itself lives in the un-vendored crates halo2_proofs 0.2.0
 * (summa-dev/halo2#8386d6e) and halo2curves 0.1.0 (zk_prover/Cargo.lock:2223-2276).  The seam
 * is two free functions of `halo2_proofs::arithmetic` plus the `EvaluationDomain` /
 * `ParamsKZG` methods built on them; each entry point below names the one it stands in for.
 * INTEGRATION.md shows the Rust-side binding (extern "C" block + [patch] of halo2_proofs).
 *
 * Data conventions (identical to halo2curves' in-memory layout, so `&[Fr]` / `&[G1Affine]`
 * are passed as raw pointers with zero copies or conversions):
 *   Fr        32 bytes: 4 x u64 little-endian limbs, Montgomery form (x * 2^256 mod r)
 *   G1Affine  64 bytes: x || y, each a Montgomery Fq as above; identity = 64 zero bytes
 * All results are fully reduced and, for points, affine-normalised, hence canonical and
 * bit-comparable with the CPU prover's.
 *
 * Conventions: every function returns SG_OK (0) or a negative sg_status; nothing throws or
 * aborts; pointers are borrowed for the duration of the call; the library is thread-safe and
 * re-entrant: it keeps a small number of independent contexts ("lanes": own streams, MSM engines,
 * NTT plans, staging buffers), a call owns one lane for its whole duration, and calls from different
 * host threads run side by side on the device.  There are FOUR lanes by default
 * (sg_set_param("lanes", 1..8)); a caller that arrives while all of them are taken WAITS for one,
 * for as long as that lane's current call lasts -- a thread pool wider than the lane count (rayon's
 * default is one thread per core) gains nothing beyond it.  "_dev" entry points are
 * asynchronous unless they return a value to the host: the work is ordered on the given stream and the
 * library's work space is kept per stream, so independent ops may be issued on several streams.  "host" entry points
 * take host pointers and move data themselves; "_dev" entry points take HIP device pointers
 * (e.g. torch tensors' data_ptr()) and a hipStream_t passed as void* (NULL = default stream).
 */
#ifndef SUMMA_GPU_H
#define SUMMA_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  SG_OK = 0,
  SG_ERR_INVALID = -1,   /* bad argument (null pointer, size/length mismatch, k out of range) */
  SG_ERR_NO_DEVICE = -2, /* no usable HIP device / sg_init not called */
  SG_ERR_HIP = -3,       /* a HIP runtime call failed; see sg_last_error() */
  SG_ERR_NOMEM = -4,     /* device or host allocation failed */
  SG_ERR_UNSUPPORTED = -5, /* the inputs are outside what this entry point handles; a general path exists */
  SG_ERR_WITNESS = -6      /* the data, not the call, is at fault: e.g. a lookup input that is not in its table
                              (upstream: permute_expression_pair -> Error::ConstraintSystemFailure) */
} sg_status;

/* ---- context ------------------------------------------------------------------------- */
/* Binds the calling process to one GPU (one process per GPU is the intended deployment;
 * under torch.distributed pass LOCAL_RANK).  Idempotent for the same device. */
int sg_init(int device);
void sg_shutdown(void);
/* Device memory the library has outgrown while running (work spaces reallocated larger, window tables replaced by
 * sg_srs_precompute) is retired, not freed -- hipFree waits for the whole device, which would stall every other lane.
 * It is bounded by the final sizes (work spaces grow geometrically) and is returned here and by sg_shutdown.  The call
 * also gives back the lanes' own work spaces (MSM / NTT work space, plans, staging: rebuilt by the next call that needs
 * them), so that afterwards the library holds what it held after sg_init plus the live SRS handles and proving keys.
 * It waits for the device to go idle: use it between workloads (e.g. after switching k), with no other call in flight. */
int sg_collect_retired(void);
/* Host memory for the host-pointer entry points (sg_msm_g1, sg_commit, sg_ntt_fr and the other calls without `_dev`): what a
 * [patch] of best_multiexp / best_fft hands over are ordinary vectors in pageable memory, which the runtime has to stage
 * on every call.  A caller that keeps its vectors in place (an SRS, a column pool, an arena) page-locks them ONCE with
 * sg_host_register(base, bytes) -- hipHostRegister underneath -- and every later call whose argument lies inside a
 * registered range moves it by direct DMA at link speed.  Ranges may not overlap; sg_host_unregister takes the base
 * address again (before the memory is freed); sg_shutdown drops what is left.  Not needed for memory that is already
 * page-locked (hipHostMalloc). */
int sg_host_register(void* host, size_t bytes);
int sg_host_unregister(void* host);
/* Wait until everything enqueued on `stream` (NULL: the default stream) has finished -- hipStreamSynchronize, in the way the
 * library itself waits: parameter "host.wait_sleep_us" (sg_set_param; 0 = the runtime's polling wait, the default; N > 0 =
 * poll every N microseconds and sleep in between, for processes that keep many proofs in flight on few CPU cores). */
int sg_stream_wait(void* stream);
/* Message of the last failure on the calling thread (static storage, never NULL). */
const char* sg_last_error(void);
/* Number of HIP devices visible (0 when there is none; never fails). */
int sg_device_count(void);
/* The device this process is bound to (sg_init, or 0 after the first lazily initialised call); -1 before either. */
int sg_device(void);
/* HIP's current device is a property of the host thread and starts at 0 in every new thread.  The library selects its
 * device itself on every call; a caller's OWN HIP calls on a worker thread -- the allocations and streams whose pointers
 * it then hands over -- do not.  sg_bind_thread() makes the bound device current on the calling thread (no-op before
 * the library is bound); sp_key_create / sp_create_proof / sp_verify_proof call it on entry.  Matters on a multi-GPU
 * host only: one process per GPU, rank r on device r, worker threads that would otherwise allocate on device 0. */
int sg_bind_thread(void);
/* "summa_gpu <version> gfx950" */
const char* sg_version(void);

/* ---- M1: halo2_proofs::arithmetic::best_multiexp(coeffs: &[Fr], bases: &[G1Affine]) -> G1
 * (reached via ParamsKZG::commit / commit_lagrange).  upstream asserts
 * coeffs.len() == bases.len(); here a single n covers both.  n = 0 yields the identity.
 * sg_msm_g1 with n <= 64 (parameter "msm.tiny_max") is one kernel launch and one host wait -- what the verifier's 37-point
 * left-hand side needs once per proof served.
 * sg_msm_g1_dev returns the point, i.e. it is complete on return: its kernels run on the calling lane's own stream, ordered
 * after whatever `stream` holds at the time of the call (the lanes' streams are on distinct hardware queues, so calls from
 * several threads overlap whatever streams the callers use: DESIGN.md section 4.4). */
int sg_msm_g1(const uint8_t* scalars, const uint8_t* bases, size_t n, uint8_t out_affine[64]);
int sg_msm_g1_dev(const void* d_scalars, const void* d_bases, size_t n, void* stream, uint8_t out_affine[64]);

/* A batch of independent MSMs -- e.g. the advice / permuted-lookup / quotient-piece commitments
 * that halo2's create_proof computes in a loop within one Fiat-Shamir phase (SURVEY.md §3.1
 * steps 3-10; proposed in §8b).  out_affine receives count x 64 bytes.  The MSMs are pipelined
 * over two streams (one MSM's bucket reduction overlaps the next one's accumulation). */
int sg_msm_g1_batch(const uint8_t* const* scalars, const uint8_t* const* bases, const size_t* n, size_t count,
                    uint8_t* out_affine);
int sg_msm_g1_batch_dev(const void* const* d_scalars, const void* const* d_bases, const size_t* n, size_t count,
                        void* stream, uint8_t* out_affine);

/* Sum of a few affine points, on the host (the local "reduce" after the all_gather of the
 * per-GPU partial results of a point-sharded MSM; EC addition is not an RCCL reduction op). */
int sg_g1_sum_affine(const uint8_t* points, size_t n, uint8_t out_affine[64]);

/* SRS cache: keeps ParamsKZG's g[] / g_lagrange[] resident in HBM across proofs
 * (the reference re-reads the file per Snapshot: backend/src/apis/round.rs:136-145). */
int sg_srs_upload(uint32_t k, const uint8_t* g, const uint8_t* g_lagrange, uint64_t* handle_out);
/* The same from device buffers (2^k x 64 B each, e.g. the receive side of an RCCL broadcast of the setup artifacts: the
 * bases never visit the host), and the reverse: copies of the resident bases into caller-owned device buffers (the send
 * side; either output may be NULL).  Copies are ordered on `stream`; sg_srs_upload_dev returns after they completed. */
int sg_srs_upload_dev(uint32_t k, const void* d_g, const void* d_g_lagrange, void* stream, uint64_t* handle_out);
int sg_srs_copy_dev(uint64_t handle, void* d_g_out, void* d_g_lagrange_out, void* stream);
/* The validation `ParamsKZG::read` performs with SerdeFormat::RawBytes (points off the curve are rejected; the
 * RawBytesUnchecked format skips it): *bad_out = number of points of the resident SRS (g and g_lagrange) that are neither on
 * y^2 = x^3 + 3 nor the identity. */
int sg_srs_check(uint64_t handle, uint64_t* bad_out);
int sg_srs_free(uint64_t handle);
/* ParamsKZG::commit (basis = 0, monomial g[]) / commit_lagrange (basis = 1): n <= 2^k scalars.
 * basis = 2 is commit_lagrange as well -- the same commitment -- with a hint: the column is (mostly) piecewise constant, as
 * the permutation / lookup grand products and the sorted lookup columns of a halo2 proof are over the unused rows.  With
 *   sum_i s_i L_i = sum_i (s_i - s_{i+1}) Q_i,   Q_i = L_0 + ... + L_i,  s_n = 0
 * the library then runs the MSM of the DIFFERENCES against the prefix-summed basis Q (sg_srs_precompute(handle, 2, ..)), in
 * which every constant run costs nothing (zero digits are skipped).  Without that table, or for n < 2^k, basis 2 is basis 1. */
int sg_commit(uint64_t srs_handle, int basis, const uint8_t* scalars, size_t n, uint8_t out_affine[64]);
int sg_commit_dev(uint64_t srs_handle, int basis, const void* d_scalars, size_t n, void* stream,
                  uint8_t out_affine[64]);
/* Fixed-base acceleration for a resident SRS (every MSM of create_proof is a ParamsKZG::commit /
 * commit_lagrange against one of the two fixed bases): builds, once, the table
 *   row w = 2^(bit offset of window w) * basis[i],   W x 2^k affine points (W = ceil(255 / window_bits)),
 * after which sg_commit / sg_commit_dev / sg_commit_batch_dev on that basis run all W digits of a scalar
 * into ONE bucket set (one bucket reduction instead of W; wider windows at small k).  Same result
 * bits.  window_bits = 0 chooses min(16, k).  Memory: W * 64 * 2^k bytes per basis.
 * basis = 2: the prefix sums Q of g_lagrange (computed on first use, 64 * 2^k bytes) and their table, for difference-form
 * commitments (see sg_commit); use the window_bits of bases 0 / 1 when the columns are to share fused jobs. */
int sg_srs_precompute(uint64_t handle, int basis, uint32_t window_bits);
/* `count` commitments of n scalars each against one basis, issued as fused jobs; out_affine: count x 64 B */
int sg_commit_batch_dev(uint64_t srs_handle, int basis, const void* const* d_scalars, size_t count, size_t n,
                        void* stream, uint8_t* out_affine);
/* the same with one basis per polynomial (0 = g, 1 = g_lagrange, 2 = g_lagrange in difference form): the commitments of one prover phase that mix
 * Lagrange- and coefficient-form polynomials as ONE fused job (fixed-base when both tables were precomputed).
 * A basis value may carry the hint SG_BASIS_SPARSE (basis | 16): "this column is witness-like -- mostly zeros and small values".
 * When every column of the call carries it the job is scheduled as the short latency chain it is (shorter accumulation tasks);
 * the commitments are the same with and without the hint. */
#define SG_BASIS_SPARSE 16
int sg_commit_batch_mixed_dev(uint64_t srs_handle, const int* basis, const void* const* d_scalars, size_t count, size_t n,
                              void* stream, uint8_t* out_affine);
/* Commit combiner, for provers that keep several proofs in flight from several host threads (one proof per thread).  Between
 * sg_commit_combine_begin() and sg_commit_combine_end() the calling thread's sg_commit_batch_mixed_dev calls may be FUSED with
 * those of the other threads that declared themselves: whichever caller finds no job running waits a bounded time (parameter
 * "commit.combine_wait_us", default 300) for the others to arrive, then everything pending with the same SRS and length
 * runs as one job -- one sort front-end, one bucket reduction, one host tail for all of them.  Same commitments, same
 * return values; a caller never waits for a thread that may not come, only for that deadline or for the running job.
 * A fused job that fails as a whole (device memory, a bad pointer of ONE member) does not fail its members: each of them is
 * then run as a job of its own and gets its own return value.
 * sg_commit_combine_stats: fused jobs run / requests served so far (requests / jobs = average fusion). */
int sg_commit_combine_begin(void);
int sg_commit_combine_end(void);
int sg_commit_combining(void);   /* 1 between _begin and _end on the calling thread, else 0 */
int sg_commit_combine_stats(uint64_t* jobs, uint64_t* requests);
/* Device pointers of a cached SRS (for callers that drive the *_dev entry points). */
int sg_srs_device_ptrs(uint64_t handle, const void** d_g, const void** d_g_lagrange, uint32_t* k);

/* ---- N1: halo2_proofs::arithmetic::best_fft(a: &mut [Fr], omega: Fr, log_n: u32)
 * in place, natural order in and out: A[j] = sum_i a[i] * omega^(i*j). */
int sg_ntt_fr(uint8_t* a, const uint8_t omega[32], uint32_t log_n);
int sg_ntt_fr_dev(void* d_a, const uint8_t omega[32], uint32_t log_n, void* stream);

/* A batch of independent in-place transforms of one size with one omega (the per-column loops
 * of create_proof: 9 x lagrange_to_coeff, ...; SURVEY.md §8b `sg_ntt_fr_batch`).
 * divisor == NULL: best_fft; otherwise EvaluationDomain::ifft semantics. */
int sg_ntt_fr_batch_dev(void* const* d_a, size_t count, const uint8_t omega[32], const uint8_t* divisor,
                        uint32_t log_n, void* stream);

/* The same out of place: d_out[i] = transform of d_in[i], the inputs untouched (a column that is still needed in Lagrange
 * form -- every advice column, every grand product -- is transformed without a device-to-device copy in front). */
int sg_ntt_fr_batch_oop_dev(const void* const* d_in, void* const* d_out, size_t count, const uint8_t omega[32], const uint8_t* divisor,
                            uint32_t log_n, void* stream);

/* ---- N2: EvaluationDomain::ifft(a, omega_inv, log_n, divisor): best_fft with omega_inv,
 * then every element times `divisor` (n^-1 for lagrange_to_coeff). */
int sg_intt_fr(uint8_t* a, const uint8_t omega_inv[32], const uint8_t divisor[32], uint32_t log_n);
int sg_intt_fr_dev(void* d_a, const uint8_t omega_inv[32], const uint8_t divisor[32], uint32_t log_n, void* stream);
/* EvaluationDomain::lagrange_to_coeff with the domain's own constants for 2^k. */
int sg_lagrange_to_coeff(uint8_t* a, uint32_t k);
int sg_lagrange_to_coeff_dev(void* d_a, uint32_t k, void* stream);

/* ---- N3: EvaluationDomain::coeff_to_extended: coeffs[2^k] -> out[2^ext_k]
 * (a[i] *= zeta^(i mod 3), zero-pad, best_fft with omega_ext); fused into one transform. */
int sg_coeff_to_extended(const uint8_t* coeffs, uint32_t k, uint32_t ext_k, uint8_t* out);
int sg_coeff_to_extended_dev(const void* d_coeffs, uint32_t k, uint32_t ext_k, void* d_out, void* stream);
/* the same for `count` columns (the 9 coeff_to_extended calls of a proof): one launch per pass while 2^ext_k is small */
int sg_coeff_to_extended_batch_dev(const void* const* d_coeffs, void* const* d_out, size_t count, uint32_t k, uint32_t ext_k,
                                   void* stream);

/* ---- N4: EvaluationDomain::extended_to_coeff: in place over 2^ext_k elements (iNTT with
 * omega_ext^-1 and 2^-ext_k, then a[i] *= zeta^-(i mod 3)); the caller truncates to
 * n * quotient_poly_degree as halo2 does.  divide_by_vanishing_poly multiplies element i by
 * t_evaluations[i mod 2^(ext_k-k)] = 1/((zeta*omega_ext^i)^n - 1). */
int sg_extended_to_coeff(uint8_t* ext, uint32_t k, uint32_t ext_k);
int sg_extended_to_coeff_dev(void* d_ext, uint32_t k, uint32_t ext_k, void* stream);
int sg_divide_by_vanishing_poly(uint8_t* ext, uint32_t k, uint32_t ext_k);
int sg_divide_by_vanishing_poly_dev(void* d_ext, uint32_t k, uint32_t ext_k, void* stream);

/* ---- the quotient on quotient-degree many cosets instead of the whole extended domain (used by the library's own prover;
 * the halo2-layout calls above stay for callers that keep halo2's evaluate_h).  deg h < d * 2^k (d = cs.degree() - 1), so h is
 * determined by its values on d cosets of the 2^k domain H; halo2's extended domain of 2^ext_k points is 2^(ext_k - k) such
 * cosets, c_b H with c_b = zeta * omega_ext^b, and the first d of them are used.  Layout "coset-major":
 *   out[b * 2^k + a] = f(c_b * omega^a)  =  element a * 2^(ext_k - k) + b of sg_coeff_to_extended's output.
 * A rotation by omega^r is an index shift of r inside a block.  On a coset X^n is the constant c_b^n: the vanishing polynomial
 * is a constant there, and the values are those of P_b = h mod (X^n - c_b^n) = sum_t c_b^(n t) h_t, so the d pieces h_t (n
 * coefficients each; what the prover commits to) come from d inverse transforms of size 2^k and a d x d Vandermonde solve per
 * coefficient.  Same h as extended_to_coeff(divide_by_vanishing_poly(.)) -- it is unique -- from d / 2^(ext_k - k) of the rows. */
int sg_coeff_to_cosets_batch_dev(const void* const* d_coeffs, void* const* d_out, size_t count, uint32_t k, uint32_t ext_k,
                                 uint32_t n_cosets, void* stream);
/* d_values: n_cosets * 2^k coset-major values of the quotient's NUMERATOR (destroyed); d_pieces: n_cosets vectors of 2^k */
int sg_cosets_to_pieces_dev(void* d_values, void* const* d_pieces, uint32_t k, uint32_t ext_k, uint32_t n_cosets, void* stream);

/* ---- domain constants (EvaluationDomain::new): omega = ROOT_OF_UNITY^(2^(28-k)) etc.
 * which: 0 omega, 1 omega^-1, 2 (2^k)^-1, 3 zeta (Fr::ZETA, the extended-coset shift). */
int sg_domain_constant(uint32_t k, int which, uint8_t out[32]);

/* ---- row S: ParamsKZG::setup's fixed-base products out[i] = scalars[i] * G1::generator()
 * (affine); also used to synthesise bases for benchmarks. */
int sg_g1_fixed_base_mul(const uint8_t* scalars, size_t n, uint8_t* out_affine);
int sg_g1_fixed_base_mul_dev(const void* d_scalars, size_t n, void* d_out_affine, void* stream);

/* The two G2 points of ParamsKZG (verifier side): out = scalar * G2 generator in halo2curves' G2Affine
 * layout (x.c0 || x.c1 || y.c0 || y.c1, 32-B Montgomery Fq each); g2 = 1 * G2, s_g2 = tau * G2.  Host code. */
int sg_g2_generator_mul(const uint8_t scalar[32], uint8_t out[128]);
/* The pairing check that ends `verify_proof` (halo2 `SingleStrategy::process` -> `multi_miller_loop` +
 * `final_exponentiation`, reached from `full_verifier` / `create_proof_checked`, zk_prover/src/circuits/utils.rs:
 * 123-130, 185-191; precompile 0x08 in contracts/src/InclusionVerifier.sol:185-202):
 * *ok = 1 iff prod_i e(g1[i], g2[i]) == 1.  g1: n x 64 B G1Affine, g2: n x 128 B G2Affine (halo2curves layout,
 * Montgomery; identity = zeros).  Host code.  Points off the curve are SG_ERR_INVALID; G2 points are taken from
 * the (trusted) SRS and are not subgroup-checked. */
int sg_pairing_check(const uint8_t* g1_points, const uint8_t* g2_points, size_t n, int* ok);
/* the same, with the final exponentiation's addition chain cross-checked against the plain exponentiation (tests) */
int sg_pairing_check_slow(const uint8_t* g1_points, const uint8_t* g2_points, size_t n, int* ok);
/* Keccak-256 as Ethereum uses it (`ethers::utils::keccak256`: usernames, zk_prover/src/merkle_sum_tree/entry.rs:21; the
 * hash of the EVM transcript).  Host code, for host-language bindings without a fast Keccak of their own. */
int sg_keccak256(const uint8_t* data, size_t len, uint8_t out[32]);
/* ParamsKZG::<Bn256>::setup(k, rng) (zk_prover/src/circuits/utils.rs:70) with tau = the field
 * element the caller drew from its RNG (32 B Montgomery Fr): g[i] = tau^i * G,
 * g_lagrange[i] = L_i(tau) * G, 2^k points of 64 B each.  The G2 elements of the SRS are only
 * used by the verifier's pairing check and are not produced here. */
int sg_kzg_setup(uint32_t k, const uint8_t tau[32], uint8_t* g, uint8_t* g_lagrange);
int sg_kzg_setup_dev(uint32_t k, const uint8_t tau[32], void* d_g, void* d_g_lagrange, void* stream);

/* ---- N5: best_fft over G1 (halo2's FftGroup for curve points), used by ParamsKZG::downsize
 * (zk_prover/src/circuits/utils.rs:64) to recompute g_lagrange for the truncated g[]:
 * out[j] = sum_i omega^(ij) * in[i], optionally times `scale` (NULL: none).  Set-up time only. */
int sg_g1_fft_dev(const void* d_in, void* d_out, const uint8_t omega[32], const uint8_t* scale, uint32_t log_n,
                  void* stream);
/* g_lagrange[0..2^k) from g[0..2^k): inverse G1 FFT with the domain's omega^-1 and n^-1 */
int sg_g1_to_lagrange(const uint8_t* g, uint32_t k, uint8_t* g_lagrange);

/* ---- helpers: Fr canonical <-> Montgomery (PrimeField::from_repr / to_repr in bulk) */
int sg_fr_to_montgomery_dev(const void* d_in, void* d_out, size_t n, void* stream);
int sg_fr_from_montgomery_dev(const void* d_in, void* d_out, size_t n, void* stream);
/* halo2 lookup::prover::permute_expression_pair (§3.1 step 4) for range tables, on the device: rows = the usable
 * rows; A' (d_permuted_input) = the input rows in increasing order, S' (d_permuted_table) = the table rows
 * rearranged so that every row has A'[i] == S'[i] or A'[i] == A'[i-1].  Handles tables whose values are all below
 * 2^16 (SG_ERR_UNSUPPORTED otherwise: sort on the host as upstream does); SG_ERR_WITNESS if an input value is not
 * in the table.  Synchronises the stream (the status is known on return). */
int sg_lookup_permute_small_dev(const void* d_input, const void* d_table, size_t rows, void* d_permuted_input,
                                void* d_permuted_table, void* stream);
/* The same without the wait: *d_status (u32 in device-visible memory, e.g. mapped page-locked host memory) receives 0 (done), 1
 * (an input value is not in the table: what the call above reports as SG_ERR_WITNESS) or 2 (not a range table: SG_ERR_UNSUPPORTED)
 * when the kernels have run; the outputs are Montgomery words and valid only under status 0.  No memset, conversion or copy
 * launches (the write pass cleans the work space of the next call on the stream).  A prover that knows its table is a range
 * table (a property of the proving key) issues this, goes on, and looks at the status where it waits anyway.  Asynchronous. */
int sg_lookup_permute_small_async_dev(const void* d_input, const void* d_table, size_t rows, void* d_permuted_input,
                                      void* d_permuted_table, void* d_status, void* stream);
/* n uniform field elements written to d_out (blinding rows, the random polynomial of create_proof -- upstream draws
 * them from `OsRng`; here the OS supplies a 32-byte key per proof and ChaCha20, RFC 8439's block function, expands it
 * on the device): element i = the first 32 bytes of block(key, counter = i, nonce = (attempt, stream_id)) with the top
 * two bits cleared, redrawn with attempt + 1 while >= r.  Deterministic in (key, stream_id, i). */
int sg_fr_random_dev(const uint8_t key[32], uint64_t stream_id, void* d_out, size_t n, void* stream);
/* m <= 8 draws in one launch: draw d fills d_out[d] with n[d] elements of stream `first_stream_id + d` (exactly what m calls
 * of sg_fr_random_dev with consecutive stream ids produce: the blinding rows of a phase's columns cost one launch) */
int sg_fr_random_batch_dev(const uint8_t key[32], uint64_t first_stream_id, void* const* d_out, const size_t* n, uint32_t m, void* stream);

/* ---- first "next" row: device-resident helpers between NTTs and MSMs.
 * halo2_proofs::arithmetic::eval_polynomial(poly, point) = sum_i poly[i] * point^i
 * (the 35 evaluations of create_proof) */
int sg_fr_eval_poly(const uint8_t* coeffs, size_t n, const uint8_t x[32], uint8_t out[32]);
int sg_fr_eval_poly_dev(const void* d_coeffs, size_t n, const uint8_t x[32], void* stream, uint8_t out[32]);
/* m evaluations in one go: out[j] = polys[j](points[j]); all polynomials have n coefficients (n <= 2^26) */
int sg_fr_eval_poly_batch_dev(const void* const* d_polys, size_t n, const uint8_t* points, uint32_t m, void* stream,
                              uint8_t* out);
/* ff::BatchInvert::batch_invert: in place, zeros stay zero */
int sg_fr_batch_invert_dev(void* d_a, size_t n, void* stream);
/* exclusive prefix product, the core of the permutation / lookup grand products (steps 5-6):
 * out[0] = 1, out[i] = a[0] * ... * a[i-1] for i <= n  (n + 1 outputs, n < 2^21) */
int sg_fr_prefix_product_dev(const void* d_a, size_t n, void* d_out, void* stream);
/* One chunk of halo2's permutation grand product (permutation::prover::commit; SURVEY.md §3.1
 * step 5): for the `ncols` columns of the chunk (values / permuted sigma columns in Lagrange
 * basis, 2^k rows, k <= 21)
 *   z[0] = z0 (NULL: 1),  z[i+1] = z[i] * prod_c (v_c[i] + delta_start delta^c omega^i beta + gamma)
 *                                        / prod_c (v_c[i] + beta sigma_c[i] + gamma)
 * n values are written; the caller chains chunks through z0 and overwrites the blinding rows. */
int sg_permutation_product_dev(const void* const* d_values, const void* const* d_sigma, uint32_t ncols,
                               const uint8_t beta[32], const uint8_t gamma[32], const uint8_t delta_start[32],
                               uint32_t k, const uint8_t* z0, void* d_z, void* stream);
/* halo2's lookup grand product (lookup::prover::commit_product; step 6), inputs already
 * theta-compressed / permuted by the caller:
 *   z[0] = 1,  z[i+1] = z[i] (a[i] + beta)(s[i] + gamma) / ((a'[i] + beta)(s'[i] + gamma)) */
int sg_lookup_product_dev(const void* d_input, const void* d_table, const void* d_permuted_input,
                          const void* d_permuted_table, const uint8_t beta[32], const uint8_t gamma[32], size_t n,
                          void* d_z, void* stream);
/* ALL grand products of one proof in batched launches (permutation::prover::commit over every chunk, then
 * lookup::prover::commit_product for every lookup -- halo2 plonk/permutation/prover.rs, plonk/lookup/prover.rs; reached
 * from the reference's create_proof call, zk_prover/src/circuits/utils.rs:94-101): the products share one denominator
 * pass, ONE batch inversion, one numerator pass and one three-launch running product (grid.y = product), and a chunk's z
 * continues from the previous chunk's value at row `usable_rows` through a device-side scalar -- no host round trip, nothing
 * to synchronise.  Same values as the calls above chained by hand.
 *   d_values / d_sigma: the permutation's columns in order, chunk after chunk (chunk j has chunk_cols[j] <= 8 of them);
 *   d_lookup_cols: 4 per lookup -- input, table, permuted input, permuted table (theta-compressed by the caller);
 *   d_z: n_chunks + n_lookups outputs of 2^k values each (the caller overwrites the blinding rows); k <= 20,
 *   n_chunks + n_lookups <= 8. */
int sg_grand_products_dev(const void* const* d_values, const void* const* d_sigma, const uint32_t* chunk_cols, uint32_t n_chunks,
                          const void* const* d_lookup_cols, uint32_t n_lookups, const uint8_t beta[32], const uint8_t gamma[32],
                          uint32_t k, size_t usable_rows, void* const* d_z, void* stream);
/* The same; additionally d_closing[p] (32 B each, device-visible memory, may be NULL) <- z_p[usable_rows] for every product p,
 * written by the kernels that produce that row: the value a satisfied argument ends on is 1, and a prover that checks it reads
 * mapped host memory after its next wait instead of issuing copies. */
int sg_grand_products_closing_dev(const void* const* d_values, const void* const* d_sigma, const uint32_t* chunk_cols, uint32_t n_chunks,
                                  const void* const* d_lookup_cols, uint32_t n_lookups, const uint8_t beta[32], const uint8_t gamma[32],
                                  uint32_t k, size_t usable_rows, void* const* d_z, void* d_closing, void* stream);
/* out[i] = a[i] * b[i] */
int sg_fr_mul_dev(const void* d_a, const void* d_b, size_t n, void* d_out, void* stream);

/* halo2 arithmetic::kate_division(a, b): quotient of a(X) by (X - b) -- applied once per opening point by the
 * SHPLONK multi-open (§8f-3).  d_q receives n elements (q_0 .. q_{n-2}, then 0) and must not alias d_a;
 * remainder_out (optional, host) receives a(b).  n <= 2^21. */
int sg_fr_kate_division_dev(const void* d_a, size_t n, const uint8_t b[32], void* d_q, uint8_t* remainder_out,
                            void* stream);
/* m <= 16 exact divisions q_j = a_j / (X - points_j) of n coefficients each (n written per quotient, the last one 0), one
 * launch per scan step for all of them.  The multi-open's use: by partial fractions 1 / prod_j (X - p_j) = sum_j c_j / (X - p_j),
 * so the divisions of a rotation set are independent divisions of the same polynomial (which vanishes on the whole set)
 * instead of a chain, and all sets go in one batch.  Remainders are not returned.  Asynchronous on `stream` (`points` and the
 * pointer arrays are read before the call returns; revisions 1-2 waited for the stream before returning). */
int sg_fr_kate_division_batch_dev(const void* const* d_a, size_t n, const uint8_t* points, uint32_t m, void* const* d_q,
                                  void* stream);
/* sg_fr_kate_division_dev with the remainder a(b) (32 B, Montgomery) written to device-visible memory d_remainder by the kernel
 * instead of being returned: asynchronous on `stream` (a prover checks the remainder of its final division after the next
 * commitment has been issued, not before). */
int sg_fr_kate_division_rem_dev(const void* d_a, size_t n, const uint8_t b[32], void* d_q, void* d_remainder, void* stream);
/* Range check of caller-supplied columns: *d_count (u32 in device memory) = number of elements of the m <= 16 columns (n each)
 * whose 32-byte word value is >= r (halo2curves never produces such words; `Fr::from_repr` refuses them [UPSTREAM]).
 * Asynchronous on `stream`. */
int sg_fr_count_noncanonical_dev(const void* const* d_cols, uint32_t m, size_t n, void* d_count, void* stream);
/* The same check without a counter: *d_flag (u32) is set to 1 when any such element exists and left alone otherwise -- the caller
 * clears it beforehand.  d_flag may be page-locked host memory mapped into the device (hipHostMalloc(.., hipHostMallocMapped)):
 * then the check costs one kernel, no memset and no copy back, and the host reads the word after any later wait on `stream`. */
int sg_fr_flag_noncanonical_dev(const void* const* d_cols, uint32_t m, size_t n, void* d_flag, void* stream);
/* out[i] = sum_j coeffs[j] * polys[j][i], 1 <= m <= 32 (the random linear combinations of the multi-open) */
int sg_fr_lincomb_dev(const void* const* d_polys, const uint8_t* coeffs, uint32_t m, size_t n, void* d_out, void* stream);
/* the same plus a polynomial of n_low <= 8 coefficients given by value (32 B Montgomery each): out[i] += low[i] for i < n_low.
 * What SHPLONK's q_i(X) - r_i(X) needs -- r_i interpolates a rotation set's evaluations, at most four coefficients -- without
 * a zeroed column, an upload and a second pass for it.  m = 0 (d_polys, coeffs may be NULL): out = the low polynomial itself,
 * zeros from row n_low on -- an instance column from its few values in one launch. */
int sg_fr_lincomb_low_dev(const void* const* d_polys, const uint8_t* coeffs, uint32_t m, size_t n, const uint8_t* low, uint32_t n_low,
                          void* d_out, void* stream);
/* n_sets <= 8 such combinations of one length in ONE launch: combination s takes the next set_sizes[s] <= 32 entries of d_polys /
 * coeffs (at most 48 in all), adds the n_lows[s] <= 4 coefficients lows[s * 4 ..] (32 B Montgomery each; lows is n_sets x 4 x 32
 * bytes, n_lows / lows may be NULL) and writes d_outs[s].  The five rotation sets of SHPLONK's q_i(X) - r_i(X) are one launch
 * instead of five (halo2 `multiopen::shplonk::prover::create_proof`: `rotation_sets.map(|set| ...)`).  Asynchronous on `stream`. */
int sg_fr_lincomb_sets_dev(const void* const* d_polys, const uint8_t* coeffs, const uint32_t* set_sizes, uint32_t n_sets, size_t n,
                           const uint8_t* lows, const uint32_t* n_lows, void* const* d_outs, void* stream);

/* ---- the generic (circuit-independent) parts of halo2's `evaluate_h`
 * (halo2_proofs plonk/evaluation.rs, Evaluator::evaluate_h; the same terms, in the same
 * order, are folded by the generated verifier: contracts/src/InclusionVerifier.sol:903-997).
 * All arrays are evaluations over the extended coset (2^ext_k rows, row i = zeta * omega_ext^i),
 * `values` is the running numerator, updated in place as values = values * y + term for each
 * term.  l0 / l_last / l_active are the extended-coset evaluations of the Lagrange selector
 * polynomials the proving key holds.
 *
 * Permutation argument: nsets grand-product polynomials z_s over ncols columns in chunks of
 * chunk_len (= degree - 2), sigma = the permutation polynomials of the proving key;
 * last_rotation_abs = blinding_factors + 1 (z_{s-1} is read at omega^-last_rotation_abs):
 *   l0 (1 - z_0);  l_last (z_last^2 - z_last);  l0 (z_s - z_{s-1}(omega^-last X)), s >= 1;
 *   l_active (z_s(omega X) prod_j (v_j + beta sigma_j + gamma)
 *             - z_s(X) prod_j (v_j + beta zeta delta^j' omega_ext^i + gamma))    per set          */
int sg_quotient_permutation_dev(void* d_values, const void* const* d_z, uint32_t nsets, const void* const* d_cols,
                                const void* const* d_sigma, uint32_t ncols, uint32_t chunk_len, const void* d_l0,
                                const void* d_l_last, const void* d_l_active, const uint8_t beta[32],
                                const uint8_t gamma[32], const uint8_t y[32], uint32_t k, uint32_t ext_k,
                                uint32_t last_rotation_abs, void* stream);
/* the same over coset-major arrays (n_cosets blocks of 2^k rows; block b = the coset zeta * omega_ext^b * H): one launch */
int sg_quotient_permutation_cosets_dev(void* d_values, const void* const* d_z, uint32_t nsets, const void* const* d_cols,
                                       const void* const* d_sigma, uint32_t ncols, uint32_t chunk_len, const void* d_l0,
                                       const void* d_l_last, const void* d_l_active, const uint8_t beta[32],
                                       const uint8_t gamma[32], const uint8_t y[32], uint32_t k, uint32_t ext_k,
                                       uint32_t n_cosets, uint32_t last_rotation_abs, void* stream);
/* Lookup argument (one lookup; inputs already theta-compressed by the caller):
 *   l0 (1 - z);  l_last (z^2 - z);
 *   l_active (z(omega X)(a' + beta)(s' + gamma) - z(X)(a + beta)(s + gamma));
 *   l0 (a' - s');  l_active (a' - s')(a' - a'(omega^-1 X))                                         */
int sg_quotient_lookup_dev(void* d_values, const void* d_z, const void* d_permuted_input, const void* d_permuted_table,
                           const void* d_input, const void* d_table, const void* d_l0, const void* d_l_last,
                           const void* d_l_active, const uint8_t beta[32], const uint8_t gamma[32], const uint8_t y[32],
                           uint32_t k, uint32_t ext_k, void* stream);

int sg_quotient_lookup_cosets_dev(void* d_values, const void* d_z, const void* d_permuted_input, const void* d_permuted_table,
                                  const void* d_input, const void* d_table, const void* d_l0, const void* d_l_last,
                                  const void* d_l_active, const uint8_t beta[32], const uint8_t gamma[32], const uint8_t y[32],
                                  uint32_t k, uint32_t n_cosets, void* stream);

/* Custom-gate block: halo2's GraphEvaluator program (plonk/evaluation.rs: `calculations`, `constants`,
 * `rotations`; value sources and calculations keep upstream's names).  calculations[i] defines
 * intermediate i; for every extended row the value of the LAST calculation becomes values[row], the old
 * values[row] being available as SG_VS_PREVIOUS_VALUE (upstream ends its program with
 * Horner(PreviousValue, gate polynomials, Y)).  Column sources name (column index, index into
 * rotations[]); a rotation r reads row + r * 2^(ext_k - k).  The program is compiled per call into a
 * straight-line LDS-slot program (csrc/gates.hip); at most 64 simultaneously live values. */
enum { SG_VS_CONSTANT = 0, SG_VS_INTERMEDIATE = 1, SG_VS_FIXED = 2, SG_VS_ADVICE = 3, SG_VS_INSTANCE = 4,
       SG_VS_CHALLENGE = 5, SG_VS_BETA = 6, SG_VS_GAMMA = 7, SG_VS_THETA = 8, SG_VS_Y = 9, SG_VS_PREVIOUS_VALUE = 10 };
enum { SG_OP_ADD = 0, SG_OP_SUB = 1, SG_OP_MUL = 2, SG_OP_SQUARE = 3, SG_OP_DOUBLE = 4, SG_OP_NEGATE = 5,
       SG_OP_HORNER = 6, SG_OP_STORE = 7 };
typedef struct { uint32_t kind, index, rotation; } sg_value_source;
typedef struct {
  uint32_t op;
  sg_value_source a, b;              /* HORNER: a = start value, b = factor */
  uint32_t parts_offset, parts_len;  /* HORNER: horner_parts[parts_offset .. +parts_len) */
} sg_calculation;
typedef struct {
  const uint8_t* constants; uint32_t n_constants;      /* 32-B Montgomery Fr each */
  const int32_t* rotations; uint32_t n_rotations;
  const sg_calculation* calculations; uint32_t n_calculations;
  const sg_value_source* horner_parts; uint32_t n_horner_parts;
} sg_graph;
int sg_quotient_gates_dev(void* d_values, const sg_graph* graph, const void* const* d_fixed, uint32_t n_fixed,
                          const void* const* d_advice, uint32_t n_advice, const void* const* d_instance,
                          uint32_t n_instance, const uint8_t* challenges, uint32_t n_challenges,
                          const uint8_t beta[32], const uint8_t gamma[32], const uint8_t theta[32], const uint8_t y[32],
                          uint32_t k, uint32_t ext_k, void* stream);

/* the same over coset-major arrays of n_cosets * 2^k rows (a rotation r reads row + r inside its block of 2^k rows) */
int sg_quotient_gates_cosets_dev(void* d_values, const sg_graph* graph, const void* const* d_fixed, uint32_t n_fixed,
                                 const void* const* d_advice, uint32_t n_advice, const void* const* d_instance,
                                 uint32_t n_instance, const uint8_t* challenges, uint32_t n_challenges,
                                 const uint8_t beta[32], const uint8_t gamma[32], const uint8_t theta[32], const uint8_t y[32],
                                 uint32_t k, uint32_t n_cosets, void* stream);

/* halo2's `evaluate_h` for the quotient on cosets in ONE call: values <- the custom gates (program `gates`, previous value zero:
 * d_values needs no clearing), folded with the permutation argument's terms (as sg_quotient_permutation_cosets_dev) and the lookup
 * argument's (as sg_quotient_lookup_cosets_dev, its input column being program `lookup_input` over the same columns -- a lookup
 * with ONE input and ONE table expression, so that theta does not enter).  When both programs are ones the library has
 * straight-line code for (the reference circuit's, N_CURRENCIES 1 .. 4) a row's value stays in registers through all three
 * blocks: one launch, every column read once, no memset, no blob upload (csrc/numerator.hip); any other pair runs the separate
 * kernels one after the other (d_input_work: n_cosets * 2^k rows of work space for the input column, NULL = the library's own).
 * The words written are the same either way.  d_perm_cols / d_sigma: the ncols permutation columns and their sigma polynomials in
 * coset-major form; d_z: the nsets = ceil(ncols / chunk_len) grand products; parameter "quotient.fused_numerator" = 0 forces
 * the separate kernels.  Asynchronous on `stream`.
 * [UPSTREAM halo2_proofs::plonk::evaluation::Evaluator::evaluate_h: custom gates, permutations, lookups -- in this order] */
int sg_quotient_numerator_cosets_dev(void* d_values, const sg_graph* gates, const sg_graph* lookup_input, const void* const* d_fixed,
                                     uint32_t n_fixed, const void* const* d_advice, uint32_t n_advice, const void* const* d_instance,
                                     uint32_t n_instance, const uint8_t* challenges, uint32_t n_challenges, const void* const* d_z,
                                     uint32_t nsets, const void* const* d_perm_cols, const void* const* d_sigma, uint32_t ncols,
                                     uint32_t chunk_len, const void* d_l0, const void* d_l_last, const void* d_l_active,
                                     const void* d_lookup_z, const void* d_permuted_input, const void* d_permuted_table,
                                     const void* d_table, void* d_input_work, const uint8_t beta[32], const uint8_t gamma[32],
                                     const uint8_t theta[32], const uint8_t y[32], uint32_t k, uint32_t ext_k, uint32_t n_cosets,
                                     uint32_t last_rotation_abs, void* stream);
/* what the interpreter makes of a program: instructions per row and simultaneously live values (LDS slots per row; the
 * occupancy of the kernel is set by the latter).  Host only -- no device needed. */
int sg_gates_program_info(const sg_graph* graph, uint32_t n_fixed, uint32_t n_advice, uint32_t n_instance, uint32_t n_challenges,
                          uint32_t* n_ops_out, uint32_t* n_slots_out);
/* the lowered program itself (tooling, tests): words_out = [n_slots, result_kind, result_index, n_ops, then 4 words per
 * instruction]; *n_words_out = the size needed, nothing is written when cap_words is smaller.  Host only. */
int sg_gates_program_words(const sg_graph* graph, uint32_t n_fixed, uint32_t n_advice, uint32_t n_instance, uint32_t n_challenges,
                           uint32_t* words_out, uint32_t cap_words, uint32_t* n_words_out);

/* ---- keygen's circuit side: what keygen_vk / keygen_pk need of MstInclusionCircuit<LEVELS, N_CURRENCIES, N_BYTES>::synthesize over
 * 2^k rows in the reference's own floor plan (include/summa_circuit.hpp): fixed_out = 11 columns x 2^k x 32 B, sigma_out = 6
 * columns x 2^k x 32 B (Montgomery Fr, column-major).  Host only -- no device is touched. */
int sg_mst_inclusion_keygen_columns(uint32_t k, uint32_t levels, uint32_t n_currencies, uint32_t n_bytes, uint8_t* fixed_out,
                                    uint8_t* sigma_out, uint32_t* rows_used_out);

/* ---- witness side: the Merkle sum tree of
 * zk_prover/src/merkle_sum_tree (node.rs:16-84, utils/build_tree.rs:5-78) over Poseidon(t = 2,
 * rate 1, R_F = 8, R_P = 56, x^5; chips/poseidon/poseidon_spec.rs:14-37).  All values 32-B Fr
 * (Montgomery); usernames are keccak256(username) reduced mod r by the caller (entry.rs:21).
 *   leaf hash   = H(username, balance_0 .. balance_{NC-1})
 *   middle node = H(bal_l + bal_r (per currency) .., hash_l, hash_r), balances = the sums */
int sg_mst_leaves_dev(const void* d_usernames, const void* d_balances, size_t n, uint32_t n_currencies,
                      void* d_hashes, void* stream);
int sg_mst_level_dev(const void* d_child_hashes, const void* d_child_balances, size_t n_parents, uint32_t n_currencies,
                     void* d_hashes, void* d_balances, void* stream);
/* whole tree of 2^depth (already padded) entries; node arrays are level-major: 2^depth leaves,
 * 2^(depth-1) parents, ..., the root last (2^(depth+1) - 1 nodes; balances NC per node) */
int sg_mst_build_dev(const void* d_usernames, const void* d_leaf_balances, uint32_t depth, uint32_t n_currencies,
                     void* d_node_hashes, void* d_node_balances, void* stream);
/* `MstInclusionCircuit::synthesize` (zk_prover/src/circuits/merkle_sum_tree.rs:228-520) on the device, for `n_users`
 * users of a tree built by sg_mst_build_dev: the three advice columns in the reference's own floor plan.  The floor plan
 * is a function of <LEVELS, N_CURRENCIES, N_BYTES> only and arrives as `d_program` (device memory): n_items x 5 u32
 * {kind, column, row, symbol, extra} followed by n_absorbs x 3 u32 {add-input row, permute row, symbol}; kind 0 copies one
 * value, 1 writes the running sum of a byte-wise range check (extra = N_BYTES), 2 lays out a whole Poseidon sponge
 * (initial state, add-input and 37-row permute regions; extra = first absorb | count << 20).  A symbol names a tree
 * node relative to the user's leaf index: bits 0-3 kind (0 username, 1 node hash, 2 node balance, 3 path bit), 4-9
 * level, 10-12 mode (0 path node, 1 sibling, 2 child of the sibling, 3 ordered child of the path's parent), 13-19
 * lane (child / currency).  d_advice: n_users x 3 x rows elements (cleared here); d_user_indices: n_users u32. */
int sg_mst_inclusion_witness_dev(const void* d_program, uint32_t n_items, uint32_t n_absorbs, const void* d_usernames,
                                 const void* d_node_hashes, const void* d_node_balances, uint32_t depth, uint32_t n_currencies,
                                 const void* d_user_indices, uint32_t n_users, void* d_advice, uint64_t rows, void* stream);

/* ---- tuning / introspection (not part of the reference seam) */
typedef struct {
  float digits_ms, sort_ms, accumulate_ms, reduce_ms, total_ms; /* HIP-event times on the stream; accumulate_ms: the accumulation launch
                                                                  * (and merge rounds) alone, from behind its wait for other jobs' accumulations */
  uint32_t window_bits, windows, tasks, max_bucket;
  uint32_t accumulate_threads; /* threads of the msm_accumulate launch (sized by an upper bound of `tasks`) */
  float order_ms;              /* between sort and accumulation: the task-ordering kernels + the time queued behind other jobs' accumulations */
} sg_msm_timings;
/* as sg_msm_g1_dev, additionally fills per-phase HIP event timings */
int sg_msm_g1_dev_timed(const void* d_scalars, const void* d_bases, size_t n, void* stream, uint8_t out_affine[64],
                        sg_msm_timings* timings);
int sg_commit_dev_timed(uint64_t srs_handle, int basis, const void* d_scalars, size_t n, void* stream,
                        uint8_t out_affine[64], sg_msm_timings* timings);
/* name: "lanes" (1..8, default 4: concurrent calls that get a context of their own, see the conventions at the top),
 * "commit.combine_wait_us" (default 300), "commit.combine_target" (default 4: a runner stops waiting once this many requests
 *   are pending), "commit.combine_runners" (default 1: fused jobs that may run side by side, each on a lane of its own) -- see
 *   sg_commit_combine_begin; process-wide, they stay as set until set again,
 * "host.wait_sleep_us" (0 | 1..1000: see sg_stream_wait; process-wide, takes effect at once),
 * "msm.host_chunks" (0 = by size | 1..8: the host-pointer entry points sg_msm_g1 / sg_commit cut inputs of 2^18 pairs and more into
 *   that many chunks, which run as jobs on two engines while the next chunk is uploaded; default 2),
 * "msm.tiny_max" (0..64, default 64: sg_msm_g1 of at most this many points -- the verifier's 37 -- is ONE launch that reads its
 *   inputs from mapped host memory instead of eleven launches and three staging copies; 0: always the engine's pipeline),
 * "msm.log_fuse_entries" (16..30; 0 = the defaults: a fused job holds at most 2^x (window, scalar) entries -- 27 for fixed-base
 *   jobs (sg_commit_batch*, the commit combiner: 64 polynomials of 2^17 rows over a 16-window table), 25 for generic ones
 *   (sg_msm_g1_batch*); larger batches are cut into several jobs; a value set here applies to both kinds),
 * "msm.acc_trace" (0 | 1: debug -- every wave of msm_accumulate records when it starts and leaves; the job's host tail prints the
 *   percentiles to stderr: tools/acc_trace.sh, profiles/r04_sweeps/accumulate_tail.txt),
 * "ntt.radix4" (0 = by size, default | 1 always | 2 never: an NTT pass runs two butterfly stages per sweep over its LDS tile --
 *   four elements per thread, half the barriers; same words; by size = transforms of 2^20 points and more and batched launches of
 *   four vectors and more, where it is 3-6 % faster: profiles/r05_sweeps/ntt_radix4.txt),
 * "ntt.coset_scale_pass" (0 | 1: A-B aid -- sg_coeff_to_cosets_batch_dev multiplies by the coset shifts inside the first NTT pass
 *   (0, default) or in a pass of its own before the transforms (1, rounds 3-4); same results), "msm.acc_log" (see sg_msm_launch_log),
 * "debug.fail_next_fused_job" (test hook: the next FUSED job of the commit combiner reports SG_ERR_NOMEM without running, so that
 *   its members fall back to jobs of their own),
 * "msm.window_bits", "msm.log_seg", "msm.log_red_chunk", "msm.quad", "ntt.tile_log", "ntt.threads",
 * "ntt.max_single_log", "ntt.max_multi_log";
 * how calls in flight share the device (DESIGN.md sections 4.1 and 4.4, docs/history.md section 4.11; the defaults are what the measurements chose):
 *   "side_prio" (1 | 0): every kernel but the MSM's accumulation runs at wave priority 3, so that a kernel of another call that
 *     lands beside an accumulation is not starved of issue slots by it (device-wide, not per lane);
 *   "msm.acc_waves" / "msm.acc_waves_fixed" (0 = default, 2, 3, 8): waves per SIMD of the persistent accumulation launch of
 *     generic / fixed-base jobs -- default: 3 (a full register file) for a generic job that has the device to itself, 2 (a
 *     third of the file left to other kernels) when other jobs are in flight and for fixed-base jobs; 8: one ticket per wave;
 *   "msm.acc_chain" (1 | 0): accumulations of different calls run one after the other;
 *   "msm.red_lean" (0 | 1 | 2): the bucket reduction's 168-register twin never / when other jobs are in flight / always;
 *   "msm.fused_frontend" (0 | 1 | 2): the scans and the task-length histogram of a job inside its sort's own kernels (five launches
 *     fewer) never / when the job has the device to itself (default) / always. */
int sg_set_param(const char* name, int value);
/* Reads a parameter back (so that a caller that changes a process-wide one for the length of a job can restore what it found):
 * the process-wide ones ("lanes", "commit.*", "host.wait_sleep_us", "msm.host_chunks", "msm.tiny_max") return their live value, the per-lane
 * ones ("msm.*", "ntt.*", "side_prio") the value most recently set through sg_set_param, 0 when none was (built-in default). */
int sg_get_param(const char* name, int* value);
/* Profiling aid.  With parameter "msm.acc_log" = 1 (setting it clears the log) every msm_accumulate launch of the process is
 * recorded in the order in which the chained launches run on the device, so the i-th msm_accumulate of a kernel trace ordered by
 * start time is record i: exact attribution of a profile's launches to jobs (tools/proof_budget.py).  A record is 8 u32 words:
 * entries (lo, hi), n, M (polynomials of the job), threads of the launch, fixed-base (1) / generic (0), jobs in flight when it
 * was issued, entries per task.  *n_records = records held (may exceed cap_records; only cap_records are written). */
int sg_msm_launch_log(uint32_t* out_words, size_t cap_records, size_t* n_records);
/* ABI revision of this header: bumped whenever a struct that a caller allocates grows or an entry point's meaning changes.
 *   2 (round 4): sg_msm_timings gained `order_ms` (44 bytes; accumulate_ms excludes the wait in the accumulation chain)
 *   3 (round 5): added sg_get_param, sg_msm_launch_log, sg_abi_version, sg_fr_lincomb_sets_dev, sg_quotient_numerator_cosets_dev,
 *     sg_fr_flag_noncanonical_dev, sg_lookup_permute_small_async_dev, sg_grand_products_closing_dev, sg_fr_kate_division_rem_dev;
 *     sg_fr_kate_division_batch_dev no longer waits for the stream before it returns (it is asynchronous like its neighbours);
 *     nothing removed or resized
 * A binding built against revision r must refuse a library whose sg_abi_version() < r. */
#define SG_ABI_VERSION 3
int sg_abi_version(void);
/* Time `reps` back-to-back launches of the operation with HIP events on the library's
 * stream; returns average milliseconds per launch in *ms_out (used by bench.py for the
 * roofline block). op: 0 = ntt (d_a in place, forward with the domain omega of log_n). */
int sg_time_ntt_dev(void* d_a, uint32_t log_n, int reps, float* ms_out);

#ifdef __cplusplus
}
#endif
#endif /* SUMMA_GPU_H */
