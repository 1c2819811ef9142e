// Device-resident polynomial helpers around the MSM/NTT hot path (the
// first "next" row that needs no circuit knowledge): halo2's `eval_polynomial` (the 35
// evaluations of create_proof step 11), `BatchInvert::batch_invert` and the exclusive prefix
// product the permutation / lookup grand products are built from (steps 5-6).  With these the
// vectors that feed the commitments never leave HBM between an NTT and an MSM.
//
// All vectors: n x 32 B Fr, Montgomery-2^256 words (halo2curves layout); arithmetic on
// 9 x 29-bit limbs (bn254_f29.cuh) in the 2^261 domain.
#include "poly.h"
#include "side_prio.cuh"

namespace sg {
SG_DEFINE_SIDE_PRIO_SETTER(poly_set_side_prio)

typedef Fr29 P;
__device__ __forceinline__ f29 load_hat(const fp_words* p) {  // x~ words -> x^ (< 2p)
  uint32_t w[8];
  fp_words_load(p, w);
  return f29_words_to_r261<P>(w);
}
__device__ __forceinline__ void store_hat(fp_words* p, const f29& x_hat) {  // x^ -> canonical x~ words
  uint32_t w[8];
  f29_to_words(f29_reduce_with<P>(x_hat, P::r256), w);
  fp_words_store(p, w);
}

// ---- eval_polynomial: sum_i c[i] x^i ------------------------------------------------------
// thread: Horner over CH consecutive coefficients; workgroup: pairwise fold with x^(CH*2^l);
// one partial per workgroup, folded again by the same kernel until one value is left.
// Coefficients and the running value stay in the memory (2^256) domain, only x is in the 2^261 domain: acc~ * x^ * 2^-261 =
// (acc x)~, so no coefficient is converted.  16 coefficients per thread in the batched path (32 workgroups per 2^17-term
// polynomial: the 40 evaluations of a proof fill the chip); longer polynomials (> 2^24 terms, where two levels of 2^12 no longer
// reach) and the single-polynomial path take 32 per thread.
static constexpr uint32_t EV_THREADS = 256;
template <uint32_t EV_CH>
__device__ __forceinline__ void eval_poly_block(const fp_words* __restrict__ c, uint32_t n, const words8& xw,
                                                uint32_t log_stride, fp_words* __restrict__ out,
                                                uint32_t (*sh)[9]) {
  // element i of this level has weight x^(i << log_stride)
  const uint32_t tid = threadIdx.x;
  f29 x = f29_words_to_r261<P>(xw.l);
  for (uint32_t k = 0; k < log_stride; k++) x = f29_sqr<P>(x);   // x^(2^log_stride)
  // Round 5: a workgroup's EV_CH * 256 coefficients are read COALESCED -- thread t takes the elements t, t + 256, t + 512, ...
  // of the block and runs Horner in y = x^256 over them; the fold below then pairs neighbours with x, x^2, x^4, ... x^128.
  // (Sixteen CONSECUTIVE coefficients per thread made every load instruction touch 64 different 512-byte-strided sectors: 42 % VALU
  // busy at a fifth of the memory rate, profiles/r05a_proof_budget.json.)  The powers x^(2^l), l = 0 .. 8, come from nine lanes.
  __shared__ uint32_t s_xp[9][9];
  if (tid < 9) {
    f29 xp = x;
    for (uint32_t l = 0; l < tid; l++) xp = f29_sqr<P>(xp);
#pragma unroll
    for (int q = 0; q < 9; q++) s_xp[tid][q] = xp.l[q];
  }
  __syncthreads();
  f29 y;
#pragma unroll
  for (int q = 0; q < 9; q++) y.l[q] = s_xp[8][q];
  const uint32_t base = blockIdx.x * EV_THREADS * EV_CH + tid;
  f29 acc = f29_zero();
  if (base < n) {
    // the highest element of this thread's column that exists, then down in steps of 256
    uint32_t j = min(EV_CH - 1, (n - 1 - base) / EV_THREADS);
    acc = f29_load_r256<P>(c + base + j * EV_THREADS);               // any 256-bit word value: bound < 6
    while (j-- > 0) acc = f29_mul_add<P>(acc, y, f29_load_r256<P>(c + base + j * EV_THREADS));   // acc y + c_i in one chain: < 8 * 2 / 170 + 1 + 6 < 8
  }
#pragma unroll
  for (int q = 0; q < 9; q++) sh[tid][q] = acc.l[q];
  __syncthreads();
  // pairwise fold with x^(2^l) at level l; the sums stay lazy: the bound grows by 2 per level (< 8 + 16 after eight), well inside
  // what the next product takes, so only the last value is reduced.
  uint32_t level = 0;
  for (uint32_t s = 1; s < EV_THREADS; s <<= 1, level++) {
    if ((tid & (2 * s - 1)) == 0) {
      f29 lo, hi, xp;
#pragma unroll
      for (int q = 0; q < 9; q++) { lo.l[q] = sh[tid][q]; hi.l[q] = sh[tid + s][q]; xp.l[q] = s_xp[level][q]; }
      lo = f29_mul_add<P>(hi, xp, lo);                 // lo + hi x^(..): bound + 2 per level; hi's bound (< 24) * 2 stays below 170
#pragma unroll
      for (int q = 0; q < 9; q++) sh[tid][q] = lo.l[q];
    }
    __syncthreads();
  }
  if (tid == 0) {
    f29 r;
#pragma unroll
    for (int q = 0; q < 9; q++) r.l[q] = sh[0][q];
    f29_store_canonical<P>(out + blockIdx.x, f29_reduce_small<P>(r));   // the lazy sums back below 2p
  }
}
static constexpr uint32_t EV_CH = 32, EV_LOG = 13;   // the single-polynomial path (any length): log2(EV_CH * EV_THREADS)
__global__ void __launch_bounds__(256) eval_poly_kernel(const fp_words* __restrict__ c, uint32_t n, words8 xw,
                                                        uint32_t log_stride, fp_words* __restrict__ out) {
  side_kernel_prio();
  __shared__ uint32_t sh[EV_THREADS][9];
  eval_poly_block<EV_CH>(c, n, xw, log_stride, out, sh);
}
// m polynomials of one length, each at its own point: grid (blocks, m); level 0 reads the polynomials, level
// 1 folds the per-block partials (partials[j * stride ..]) into out[j]
struct EvalBatchArgs {
  const fp_words* polys[EVAL_BATCH_MAX];
  words8 x[EVAL_BATCH_MAX];
};
template <uint32_t CH>
__global__ void __launch_bounds__(256) eval_poly_batch_kernel(EvalBatchArgs a, uint32_t n, uint32_t level,
                                                              uint32_t stride, fp_words* __restrict__ partial,
                                                              fp_words* __restrict__ out) {
  side_kernel_prio();
  __shared__ uint32_t sh[EV_THREADS][9];
  const uint32_t j = blockIdx.y;
  constexpr uint32_t LOG = CH == 16 ? 12 : 13;   // log2(CH * EV_THREADS)
  if (level == 0) eval_poly_block<CH>(a.polys[j], n, a.x[j], 0, partial + (size_t)j * stride, sh);
  else eval_poly_block<CH>(partial + (size_t)j * stride, n, a.x[j], LOG, out + j, sh);
}

// ---- batch inversion (zeros stay zero, like ff::BatchInvert) --------------------------------
static constexpr uint32_t BI_CH = 8;
__global__ void __launch_bounds__(256) batch_invert_kernel(fp_words* __restrict__ a, uint32_t n) {
  side_kernel_prio();
  const uint32_t first = (blockIdx.x * blockDim.x + threadIdx.x) * BI_CH;
  if (first >= n) return;
  const uint32_t cnt = min(BI_CH, n - first);
  // straight-line (selects, no branches around the products) and small enough for the compiler to unroll both loops
  // in full, so pre[] stays in registers; a zero or out-of-range element leaves the running product as it is.  The
  // elements are read again on the way back (L2) rather than held: 8 more field elements per thread would halve the
  // occupancy.
  f29 pre[BI_CH];           // prefix products over the non-zero elements
  uint32_t live = 0;        // bit i: element i is in range and non-zero
  f29 acc = f29_one<P>();
#pragma unroll
  for (int i = 0; i < (int)BI_CH; i++) {
    pre[i] = acc;
    const f29 v = load_hat(a + min(first + i, n - 1));
    const bool on = (uint32_t)i < cnt && !f29_is_zero_mod_p<P>(v);
    live |= (uint32_t)on << i;
    const f29 next = f29_mul<P>(acc, v);
#pragma unroll
    for (int q = 0; q < 9; q++) acc.l[q] = on ? next.l[q] : acc.l[q];
  }
  f29 inv = f29_inv<P>(acc);
  // two loops of 4: one loop of 8 x (2 products + the store's domain change) is past the compiler's size limit for
  // "#pragma unroll" (-pragma-unroll-threshold) and would be unrolled by 4 only -- pre[] indexed at run time, in scratch
  auto back = [&](const int i) {
    const bool on = (live >> i) & 1;
    const f29 v = load_hat(a + min(first + i, n - 1));
    const f29 out = f29_mul<P>(inv, pre[i]);
    const f29 next = f29_mul<P>(inv, v);
    if (on) store_hat(a + first + i, out);
#pragma unroll
    for (int q = 0; q < 9; q++) inv.l[q] = on ? next.l[q] : inv.l[q];
  };
#pragma unroll
  for (int k = 0; k < 4; k++) back(7 - k);
#pragma unroll
  for (int k = 0; k < 4; k++) back(3 - k);
  static_assert(BI_CH == 8, "");
}

// ---- exclusive prefix product: out[0] = 1, out[i] = a[0] * ... * a[i-1] ------------------------
// three launches: per-block products, scan of the block products (one workgroup), final pass
static constexpr uint32_t PP_CH = 8, PP_THREADS = 256, PP_BLOCK = PP_CH * PP_THREADS;
__device__ __forceinline__ f29 block_exclusive_scan_mul(f29 mine, uint32_t (*sh)[9], uint32_t tid, uint32_t nthr,
                                                        f29* total) {
  // Hillis-Steele inclusive scan with multiplication, then shift
#pragma unroll
  for (int q = 0; q < 9; q++) sh[tid][q] = mine.l[q];
  __syncthreads();
  f29 v = mine;
  for (uint32_t d = 1; d < nthr; d <<= 1) {
    f29 o = f29_one<P>();
    if (tid >= d) {
#pragma unroll
      for (int q = 0; q < 9; q++) o.l[q] = sh[tid - d][q];
    }
    __syncthreads();
    v = f29_mul<P>(v, o);
#pragma unroll
    for (int q = 0; q < 9; q++) sh[tid][q] = v.l[q];
    __syncthreads();
  }
  if (total) {
#pragma unroll
    for (int q = 0; q < 9; q++) total->l[q] = sh[nthr - 1][q];
  }
  f29 ex = f29_one<P>();
  if (tid) {
#pragma unroll
    for (int q = 0; q < 9; q++) ex.l[q] = sh[tid - 1][q];
  }
  __syncthreads();
  return ex;
}
__global__ void __launch_bounds__(256) prefix_product_blocks(const fp_words* __restrict__ a, uint32_t n,
                                                             fp_words* __restrict__ bprod) {
  side_kernel_prio();
  __shared__ uint32_t sh[PP_THREADS][9];
  const uint32_t tid = threadIdx.x, first = (blockIdx.x * PP_THREADS + tid) * PP_CH;
  f29 acc = f29_one<P>();
  for (uint32_t i = 0; i < PP_CH; i++)
    if (first + i < n) acc = f29_mul<P>(acc, load_hat(a + first + i));
  f29 total;
  block_exclusive_scan_mul(acc, sh, tid, PP_THREADS, &total);
  if (tid == 0) store_hat(bprod + blockIdx.x, total);
}
__global__ void __launch_bounds__(1024) prefix_product_scan_blocks(fp_words* __restrict__ bprod, uint32_t nblk) {
  side_kernel_prio();
  __shared__ uint32_t sh[1024][9];
  const uint32_t tid = threadIdx.x;
  f29 mine = tid < nblk ? load_hat(bprod + tid) : f29_one<P>();
  f29 ex = block_exclusive_scan_mul(mine, sh, tid, 1024, nullptr);
  if (tid < nblk) store_hat(bprod + tid, ex);
}
__global__ void __launch_bounds__(256) prefix_product_write(const fp_words* __restrict__ a, uint32_t n,
                                                            const fp_words* __restrict__ bprod, words8 init,
                                                            uint32_t has_init, uint32_t count_out,
                                                            fp_words* __restrict__ out) {
  side_kernel_prio();
  __shared__ uint32_t sh[PP_THREADS][9];
  const uint32_t tid = threadIdx.x, first = (blockIdx.x * PP_THREADS + tid) * PP_CH;
  f29 v[PP_CH];
  f29 acc = f29_one<P>();
#pragma unroll
  for (uint32_t i = 0; i < PP_CH; i++) {
    v[i] = (first + i < n) ? load_hat(a + first + i) : f29_one<P>();
    acc = f29_mul<P>(acc, v[i]);
  }
  f29 run = f29_mul<P>(block_exclusive_scan_mul(acc, sh, tid, PP_THREADS, nullptr), load_hat(bprod + blockIdx.x));
  if (has_init) run = f29_mul<P>(run, f29_words_to_r261<P>(init.l));
#pragma unroll
  for (uint32_t i = 0; i < PP_CH; i++) {
    if (first + i < count_out) store_hat(out + first + i, run);
    run = f29_mul<P>(run, v[i]);
  }
}

// ---- element-wise: out = a * b ------------------------------------------------------------------
__global__ void mul_elementwise_kernel(const fp_words* __restrict__ a, const fp_words* __restrict__ b, uint32_t n,
                                       fp_words* __restrict__ out) {
  side_kernel_prio();
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // a~ * b^ * 2^-261 = (ab)~
  f29_store_canonical<P>(out + i, f29_mul<P>(f29_load_r256<P>(a + i), load_hat(b + i)));
}

// ---- Kate division: a(X) = q(X) (X - b) + a(b) -------------------------------------------------
// s_i = a_i + b s_{i+1} (s_n = 0), q_{i-1} = s_i, remainder s_0: a suffix scan whose combine step is
// "multiply by a power of b and add".  Three launches like the prefix product: block values, scan of
// the block values (one workgroup, <= 1024 blocks), final pass with the carries.
static constexpr uint32_t KD_CH = 8, KD_THREADS = 256, KD_BLOCK = KD_CH * KD_THREADS;
// S_t = sum_{u >= t} mine_u w^(u - t) over the nthr threads of a workgroup (w_pow[j] = w^(2^j), hat)
__device__ __forceinline__ f29 block_suffix_geometric(f29 mine, const f29* w_pow, uint32_t (*sh)[9], uint32_t tid,
                                                      uint32_t nthr) {
#pragma unroll
  for (int q = 0; q < 9; q++) sh[tid][q] = mine.l[q];
  __syncthreads();
  f29 v = mine;                                              // lazy between steps: `mine` may come with a bound up to 8
  uint32_t j = 0;
  for (uint32_t d = 1; d < nthr; d <<= 1, j++) {
    f29 o = f29_zero();
    if (tid + d < nthr) {
#pragma unroll
      for (int q = 0; q < 9; q++) o.l[q] = sh[tid + d][q];
    }
    __syncthreads();
    v = f29_mul_add<P>(o, w_pow[j], v);                      // v + w^d * o: the bound grows by < 2 per step (< 2 + 16 after eight)
#pragma unroll
    for (int q = 0; q < 9; q++) sh[tid][q] = v.l[q];
    __syncthreads();
  }
  return v;
}
__device__ __forceinline__ f29 kd_local(const fp_words* __restrict__ a, uint32_t n, uint32_t first, const f29& b_hat,
                                        f29* vals) {
  f29 acc = f29_zero();                                      // Horner from the top of the chunk down
#pragma unroll
  for (uint32_t k = KD_CH; k-- > 0;) {
    vals[k] = (first + k < n) ? load_hat(a + first + k) : f29_zero();
    acc = f29_mul_add<P>(acc, b_hat, vals[k]);               // acc * b + a_k: < 4 * 2 / 170 + 1 + 2 < 4
  }
  return acc;
}
// the scan weights b^(KD_CH * 2^j) (within a block) and b^(KD_BLOCK * 2^j) (across blocks), hat form: 21 dependent
// squarings that every thread of every launch used to repeat (they were a third of a division's latency); computed
// once on the host with the same limb code and passed as a kernel argument
struct KatePowers {
  f29 b_hat;
  f29 chunk[8];   // b^(8 * 2^j)
  f29 block[10];  // b^(2048 * 2^j)
};
static KatePowers kate_powers(const words8& b) {
  KatePowers pw;
  pw.b_hat = f29_mul<P>(f29_from_words<0>(b.l), f29_const<P>(P::r266));   // = f29_words_to_r261
  f29 t = pw.b_hat;
  for (int i = 0; i < 3; i++) t = f29_sqr<P>(t);
  for (int j = 0; j < 8; j++) {
    pw.chunk[j] = t;
    t = f29_sqr<P>(t);
  }
  for (int j = 0; j < 10; j++) {   // t = b^(8 * 2^8) = b^2048 here
    pw.block[j] = t;
    t = f29_sqr<P>(t);
  }
  return pw;
}
__global__ void __launch_bounds__(256) kate_blocks(const fp_words* __restrict__ a, uint32_t n, KatePowers pw,
                                                   fp_words* __restrict__ bval) {
  side_kernel_prio();
  __shared__ uint32_t sh[KD_THREADS][9];
  const uint32_t tid = threadIdx.x, first = (blockIdx.x * KD_THREADS + tid) * KD_CH;
  f29 vals[KD_CH];
  f29 local = kd_local(a, n, first, pw.b_hat, vals);
  f29 s = block_suffix_geometric(local, pw.chunk, sh, tid, KD_THREADS);
  if (tid == 0) store_hat(bval + blockIdx.x, s);
}
// carry[k] = sum_{u > k} bval[u] * (b^KD_BLOCK)^(u - k - 1): the value of s just above block k
__global__ void __launch_bounds__(1024) kate_scan_blocks(fp_words* __restrict__ bval, uint32_t nblk, KatePowers pw) {
  side_kernel_prio();
  __shared__ uint32_t sh[1024][9];
  const uint32_t tid = threadIdx.x, nthr = blockDim.x;        // nthr = power of two >= nblk: log2(nthr) scan steps
  f29 mine = tid < nblk ? load_hat(bval + tid) : f29_zero();
  block_suffix_geometric(mine, pw.block, sh, tid, nthr);      // sh[t] = inclusive suffix value
  f29 carry = f29_zero();
  if (tid + 1 < nthr) {
#pragma unroll
    for (int q = 0; q < 9; q++) carry.l[q] = sh[tid + 1][q];
  }
  __syncthreads();
  if (tid < nblk) store_hat(bval + tid, carry);
}
__global__ void __launch_bounds__(256) kate_write(const fp_words* __restrict__ a, uint32_t n, KatePowers pw,
                                                  const fp_words* __restrict__ carry, fp_words* __restrict__ q_out,
                                                  fp_words* __restrict__ rem_out) {
  side_kernel_prio();
  __shared__ uint32_t sh[KD_THREADS][9];
  const uint32_t tid = threadIdx.x, first = (blockIdx.x * KD_THREADS + tid) * KD_CH;
  f29 vals[KD_CH];
  const f29& b_hat = pw.b_hat;
  const f29* w = pw.chunk;
  f29 local = kd_local(a, n, first, b_hat, vals);
  // the top thread's chunk sees the block carry: s(lo) = local + b^KD_CH * carry
  // (a single-block division has no carries: carry == nullptr)
  if (carry && tid == KD_THREADS - 1) local = f29_mul_add<P>(load_hat(carry + blockIdx.x), w[0], local);
  block_suffix_geometric(local, w, sh, tid, KD_THREADS);      // sh[t] = s at the bottom of thread t's chunk
  f29 run = f29_zero();                                        // s just above this thread's chunk
  if (tid + 1 < KD_THREADS) {
#pragma unroll
    for (int q = 0; q < 9; q++) run.l[q] = sh[tid + 1][q];
  } else if (carry) {
    run = load_hat(carry + blockIdx.x);
  }
#pragma unroll
  for (uint32_t k = KD_CH; k-- > 0;) {
    const uint32_t i = first + k;
    run = f29_mul_add<P>(run, b_hat, vals[k]);                 // s_i = s_(i+1) b + a_i: < 20 * 2 / 170 + 1 + 2
    if (i < n) {
      if (i >= 1) store_hat(q_out + i - 1, run);
      else if (rem_out) store_hat(rem_out, run);
    }
  }
  if (first <= n - 1 && n - 1 < first + KD_CH) store_hat(q_out + n - 1, f29_zero());   // padding slot
}

// ---- several Kate divisions in one launch per step (grid.y = division): the quotients of SHPLONK's rotation sets.  By
// partial fractions 1 / prod_j (X - p_j) = sum_j c_j / (X - p_j), so the |S| divisions of a set are independent divisions
// of ONE polynomial (it vanishes on the whole set) instead of a chain; all sets' divisions go in one batch.  The scan
// weights live in device memory (one KatePowers per division, uploaded by the caller).
static constexpr uint32_t KATE_BATCH_MAX = 16;
// ---- canonical-range check of caller-supplied columns (halo2curves' Fr::from_repr refuses words >= r; the gate
// interpreter's lazy bounds assume them) -- a streaming pass, one 32-byte load per element
struct CanonCols {
  const fp_words* col[16];
};
__global__ void __launch_bounds__(256) count_noncanonical_kernel(CanonCols cols, uint32_t n, uint32_t* __restrict__ count, uint32_t flag_only) {
  side_kernel_prio();
  // r as 8 LE words
  const uint32_t R[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4* src = reinterpret_cast<const uint4*>(cols.col[blockIdx.y] + i);
  const uint4 lo = src[0], hi = src[1];
  const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  bool ge = true;   // w >= r, decided from the most significant word that differs
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (w[k] != R[k]) ge = w[k] > R[k];
  }
  if (ge) {
    if (flag_only) {   // a plain store of the same value from every finder: no atomic, so the word may live in mapped host memory
      *reinterpret_cast<volatile uint32_t*>(count) = 1u;
      __threadfence_system();
    } else {
      atomicAdd(count, 1u);
    }
  }
}
hipError_t poly_flag_noncanonical(const fp_words* const* d_cols, uint32_t m, size_t n, uint32_t* d_flag, hipStream_t stream) {
  if (!m || !n) return hipSuccess;
  CanonCols cols{};
  for (uint32_t j = 0; j < m; j++) cols.col[j] = d_cols[j];
  count_noncanonical_kernel<<<dim3((unsigned)((n + 255) / 256), m), 256, 0, stream>>>(cols, (uint32_t)n, d_flag, 1u);
  return hipGetLastError();
}
hipError_t poly_count_noncanonical(const fp_words* const* d_cols, uint32_t m, size_t n, uint32_t* d_count, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(d_count, 0, sizeof(uint32_t), stream);
  if (e != hipSuccess || !m || !n) return e;
  CanonCols cols{};
  for (uint32_t j = 0; j < m; j++) cols.col[j] = d_cols[j];
  count_noncanonical_kernel<<<dim3((unsigned)((n + 255) / 256), m), 256, 0, stream>>>(cols, (uint32_t)n, d_count, 0u);
  return hipGetLastError();
}

struct KateBatch {
  const fp_words* a[KATE_BATCH_MAX];
  fp_words* q[KATE_BATCH_MAX];
};
__global__ void __launch_bounds__(256) kate_blocks_batch(KateBatch bt, uint32_t n, const KatePowers* __restrict__ pws,
                                                         fp_words* __restrict__ bval, uint32_t nblk) {
  side_kernel_prio();
  __shared__ uint32_t sh[KD_THREADS][9];
  __shared__ KatePowers pw;
  for (uint32_t i = threadIdx.x; i < sizeof(KatePowers) / 4; i += blockDim.x)
    reinterpret_cast<uint32_t*>(&pw)[i] = reinterpret_cast<const uint32_t*>(pws + blockIdx.y)[i];
  __syncthreads();
  const uint32_t tid = threadIdx.x, first = (blockIdx.x * KD_THREADS + tid) * KD_CH;
  f29 vals[KD_CH];
  f29 local = kd_local(bt.a[blockIdx.y], n, first, pw.b_hat, vals);
  f29 s = block_suffix_geometric(local, pw.chunk, sh, tid, KD_THREADS);
  if (tid == 0) store_hat(bval + (size_t)blockIdx.y * nblk + blockIdx.x, s);
}
__global__ void __launch_bounds__(1024) kate_scan_blocks_batch(fp_words* __restrict__ bval, uint32_t nblk,
                                                               const KatePowers* __restrict__ pws) {
  side_kernel_prio();
  __shared__ uint32_t sh[1024][9];
  __shared__ KatePowers pw;
  for (uint32_t i = threadIdx.x; i < sizeof(KatePowers) / 4; i += blockDim.x)
    reinterpret_cast<uint32_t*>(&pw)[i] = reinterpret_cast<const uint32_t*>(pws + blockIdx.y)[i];
  __syncthreads();
  fp_words* mine_b = bval + (size_t)blockIdx.y * nblk;
  const uint32_t tid = threadIdx.x, nthr = blockDim.x;
  f29 mine = tid < nblk ? load_hat(mine_b + tid) : f29_zero();
  block_suffix_geometric(mine, pw.block, sh, tid, nthr);
  f29 carry = f29_zero();
  if (tid + 1 < nthr) {
#pragma unroll
    for (int q = 0; q < 9; q++) carry.l[q] = sh[tid + 1][q];
  }
  __syncthreads();
  if (tid < nblk) store_hat(mine_b + tid, carry);
}
__global__ void __launch_bounds__(256) kate_write_batch(KateBatch bt, uint32_t n, const KatePowers* __restrict__ pws,
                                                        const fp_words* __restrict__ carry_all, uint32_t nblk) {
  side_kernel_prio();
  __shared__ uint32_t sh[KD_THREADS][9];
  __shared__ KatePowers pw;
  for (uint32_t i = threadIdx.x; i < sizeof(KatePowers) / 4; i += blockDim.x)
    reinterpret_cast<uint32_t*>(&pw)[i] = reinterpret_cast<const uint32_t*>(pws + blockIdx.y)[i];
  __syncthreads();
  const fp_words* __restrict__ a = bt.a[blockIdx.y];
  fp_words* __restrict__ q_out = bt.q[blockIdx.y];
  const fp_words* carry = nblk > 1 ? carry_all + (size_t)blockIdx.y * nblk : nullptr;
  const uint32_t tid = threadIdx.x, first = (blockIdx.x * KD_THREADS + tid) * KD_CH;
  f29 vals[KD_CH];
  const f29 b_hat = pw.b_hat;
  f29 local = kd_local(a, n, first, b_hat, vals);
  if (carry && tid == KD_THREADS - 1) local = f29_mul_add<P>(load_hat(carry + blockIdx.x), pw.chunk[0], local);
  block_suffix_geometric(local, pw.chunk, sh, tid, KD_THREADS);
  f29 run = f29_zero();
  if (tid + 1 < KD_THREADS) {
#pragma unroll
    for (int q = 0; q < 9; q++) run.l[q] = sh[tid + 1][q];
  } else if (carry) {
    run = load_hat(carry + blockIdx.x);
  }
#pragma unroll
  for (uint32_t k = KD_CH; k-- > 0;) {
    const uint32_t i = first + k;
    run = f29_mul_add<P>(run, b_hat, vals[k]);                 // s_i = s_(i+1) b + a_i: < 20 * 2 / 170 + 1 + 2; s_0 = a(b) is the remainder, dropped (exact divisions)
    if (i < n && i >= 1) store_hat(q_out + i - 1, run);
  }
  if (first <= n - 1 && n - 1 < first + KD_CH) store_hat(q_out + n - 1, f29_zero());   // padding slot
}

// ---- out[i] = sum_j c_j * p_j[i] ------------------------------------------------------------------
struct LinCombArgs {
  const fp_words* polys[LINCOMB_MAX];
  words8 coeff[LINCOMB_MAX];
  words8 low[LINCOMB_LOW_MAX];   // a polynomial of n_low coefficients added to the combination (memory-domain words)
  uint32_t n_low;
};
__global__ void __launch_bounds__(256) lincomb_kernel(LinCombArgs a, uint32_t m, uint32_t n, fp_words* __restrict__ out) {
  side_kernel_prio();
  __shared__ uint32_t s_c[LINCOMB_MAX][9];
  if (threadIdx.x < m) {
    f29 c = f29_words_to_r261<P>(a.coeff[threadIdx.x].l);
#pragma unroll
    for (int q = 0; q < 9; q++) s_c[threadIdx.x][q] = c.l[q];
  }
  __syncthreads();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  f29 acc = f29_zero();
  for (uint32_t j = 0; j < m; j += 2) {                       // two terms per reduction
    f29 c0, c1 = f29_zero(), p1 = f29_zero();
#pragma unroll
    for (int q = 0; q < 9; q++) c0.l[q] = s_c[j][q];
    const f29 p0 = f29_load_r256<P>(a.polys[j] + i);           // p~ * c^ = (pc)~
    if (j + 1 < m) {
#pragma unroll
      for (int q = 0; q < 9; q++) c1.l[q] = s_c[j + 1][q];
      p1 = f29_load_r256<P>(a.polys[j + 1] + i);
    }
    acc = f29_add(acc, f29_mul2<P>(p0, c0, p1, c1));          // bound grows by 2 per pair
    if ((j & 31) == 30) acc = f29_mul<P>(acc, f29_one<P>());   // keep the lazy sum far below 170 p... (tilde stays tilde)
  }
  if (i < a.n_low) acc = f29_add(acc, f29_from_words<0>(a.low[i].l));   // the low-degree addend: a few rows only
  f29_store_canonical<P>(out + i, f29_mul<P>(acc, f29_one<P>()));
}

struct LinCombSetsArgs {
  const fp_words* polys[LINCOMB_SETS_POLYS];
  words8 coeff[LINCOMB_SETS_POLYS];
  words8 low[LINCOMB_SETS_MAX][LINCOMB_SETS_LOW];
  fp_words* out[LINCOMB_SETS_MAX];
  uint32_t first[LINCOMB_SETS_MAX + 1];
  uint32_t n_low[LINCOMB_SETS_MAX];
};
__global__ void __launch_bounds__(256) lincomb_sets_kernel(LinCombSetsArgs a, uint32_t n) {
  side_kernel_prio();
  __shared__ uint32_t s_c[LINCOMB_MAX][9];
  const uint32_t set = blockIdx.y, j0 = a.first[set], m = a.first[set + 1] - j0;
  if (threadIdx.x < m) {
    f29 c = f29_words_to_r261<P>(a.coeff[j0 + threadIdx.x].l);
#pragma unroll
    for (int q = 0; q < 9; q++) s_c[threadIdx.x][q] = c.l[q];
  }
  __syncthreads();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  f29 acc = f29_zero();
  for (uint32_t j = 0; j < m; j += 2) {                       // two terms per reduction (as lincomb_kernel)
    f29 c0, c1 = f29_zero(), p1 = f29_zero();
#pragma unroll
    for (int q = 0; q < 9; q++) c0.l[q] = s_c[j][q];
    const f29 p0 = f29_load_r256<P>(a.polys[j0 + j] + i);
    if (j + 1 < m) {
#pragma unroll
      for (int q = 0; q < 9; q++) c1.l[q] = s_c[j + 1][q];
      p1 = f29_load_r256<P>(a.polys[j0 + j + 1] + i);
    }
    acc = f29_add(acc, f29_mul2<P>(p0, c0, p1, c1));
    if ((j & 31) == 30) acc = f29_mul<P>(acc, f29_one<P>());
  }
  if (i < a.n_low[set]) acc = f29_add(acc, f29_from_words<0>(a.low[set][i].l));
  f29_store_canonical<P>(a.out[set] + i, f29_mul<P>(acc, f29_one<P>()));
}

// ------------------------------------------------------------------ host side
hipError_t poly_eval(const fp_words* d_coeffs, size_t n, const words8& x, fp_words* d_tmp_a, fp_words* d_tmp_b,
                     fp_words* d_out, hipStream_t stream) {
  // level sizes shrink by EV_CH * EV_THREADS per launch; weights: element i of level l is x^(i * stride_l)
  const fp_words* cur = d_coeffs;
  size_t m = n;
  uint32_t log_stride = 0;
  fp_words* bufs[2] = {d_tmp_a, d_tmp_b};
  int which = 0;
  while (true) {
    const uint32_t blocks = (uint32_t)((m + (size_t)EV_CH * EV_THREADS - 1) / ((size_t)EV_CH * EV_THREADS));
    fp_words* dst = blocks == 1 ? d_out : bufs[which];
    eval_poly_kernel<<<blocks, EV_THREADS, 0, stream>>>(cur, (uint32_t)m, x, log_stride, dst);
    if (blocks == 1) break;
    cur = dst;
    m = blocks;
    log_stride += EV_LOG;
    which ^= 1;
  }
  return hipGetLastError();
}
size_t poly_eval_tmp_elems(size_t n) { return (n + (size_t)EV_CH * EV_THREADS - 1) / ((size_t)EV_CH * EV_THREADS) + 1; }

hipError_t poly_batch_invert(fp_words* d_a, size_t n, hipStream_t stream) {
  if (!n) return hipSuccess;
  const size_t threads = (n + BI_CH - 1) / BI_CH;
  batch_invert_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, stream>>>(d_a, (uint32_t)n);
  return hipGetLastError();
}
size_t prefix_product_tmp_elems(size_t n) { return (n + PP_BLOCK - 1) / PP_BLOCK + 1; }
hipError_t poly_prefix_product(const fp_words* d_a, size_t n, fp_words* d_tmp, fp_words* d_out, size_t count_out,
                               const words8* init, hipStream_t stream) {
  const uint32_t nblk = (uint32_t)((n + 1 + PP_BLOCK - 1) / PP_BLOCK);  // covers out[0..n]
  if (nblk > 1024 || count_out > n + 1) return hipErrorInvalidValue;    // n <= 2^21
  prefix_product_blocks<<<nblk, PP_THREADS, 0, stream>>>(d_a, (uint32_t)n, d_tmp);
  prefix_product_scan_blocks<<<1, 1024, 0, stream>>>(d_tmp, nblk);
  words8 one{};
  prefix_product_write<<<nblk, PP_THREADS, 0, stream>>>(d_a, (uint32_t)n, d_tmp, init ? *init : one, init ? 1u : 0u,
                                                        (uint32_t)count_out, d_out);
  return hipGetLastError();
}

// ---- grand-product fractions (halo2 permutation::prover::commit / lookup::prover::commit_product)
// permutation chunk: den[i] = prod_c (beta * sigma_c[i] + gamma + v_c[i])
//                    num[i] = prod_c (delta^(j0+c) * omega^i * beta + gamma + v_c[i])
// pow_tab (optional): omega^i for i < n as 2^261-domain words (NttEngine::local_twiddles(omega, k + 1), cached per domain):
// one product instead of an exponentiation per row
__global__ void __launch_bounds__(256) perm_fraction_kernel(PermCols cols, uint32_t ncols, words8 beta_w, words8 gamma_w, words8 dstart_w,
                                                            words8 delta_w, words8 omega_w, uint32_t n, uint32_t numer,
                                                            const fp_words* __restrict__ pow_tab, fp_words* __restrict__ io) {
  side_kernel_prio();
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const f29 beta = f29_words_to_r261<P>(beta_w.l), gamma = f29_words_to_r261<P>(gamma_w.l);
  f29 acc = numer ? load_hat(io + i) : f29_one<P>();          // numerators multiply the inverted denominators
  f29 dw = f29_one<P>();
  if (numer && pow_tab) dw = f29_mul<P>(f29_words_to_r261<P>(dstart_w.l), f29_load_r256<P>(pow_tab + i));   // hat * hat * 2^-261 = hat
  else if (numer) dw = f29_mul<P>(f29_words_to_r261<P>(dstart_w.l), f29_pow_u64<P>(f29_words_to_r261<P>(omega_w.l), i));
  const f29 delta = f29_words_to_r261<P>(delta_w.l);
  for (uint32_t c = 0; c < ncols; c++) {
    f29 v = load_hat(cols.values[c] + i);                      // < 2
    f29 t = numer ? f29_mul<P>(dw, beta) : f29_mul<P>(load_hat(cols.sigma[c] + i), beta);
    t = f29_add(f29_add(t, gamma), v);                         // < 6
    acc = f29_mul<P>(acc, t);                                  // 12
    if (numer) dw = f29_mul<P>(dw, delta);
  }
  store_hat(io + i, acc);
}
// lookup: den[i] = (a'[i] + beta)(s'[i] + gamma);  num[i] = (a[i] + beta)(s[i] + gamma)
__global__ void lookup_fraction_kernel(const fp_words* __restrict__ x, const fp_words* __restrict__ y, words8 beta_w,
                                       words8 gamma_w, uint32_t n, uint32_t numer, fp_words* __restrict__ io) {
  side_kernel_prio();
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const f29 beta = f29_words_to_r261<P>(beta_w.l), gamma = f29_words_to_r261<P>(gamma_w.l);
  f29 t = f29_mul<P>(f29_add(load_hat(x + i), beta), f29_add(load_hat(y + i), gamma));   // 4 * 4
  if (numer) t = f29_mul<P>(t, load_hat(io + i));
  store_hat(io + i, t);
}
hipError_t poly_perm_fraction(const PermCols& cols, uint32_t ncols, const words8& beta, const words8& gamma,
                              const words8& delta_start, const words8& delta, const words8& omega, size_t n,
                              int numer, fp_words* d_io, hipStream_t stream, const fp_words* d_pow_tab) {
  perm_fraction_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(cols, ncols, beta, gamma, delta_start, delta,
                                                                        omega, (uint32_t)n, (uint32_t)numer, d_pow_tab, d_io);
  return hipGetLastError();
}
hipError_t poly_lookup_fraction(const fp_words* d_x, const fp_words* d_y, const words8& beta, const words8& gamma,
                                size_t n, int numer, fp_words* d_io, hipStream_t stream) {
  lookup_fraction_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(d_x, d_y, beta, gamma, (uint32_t)n,
                                                                          (uint32_t)numer, d_io);
  return hipGetLastError();
}
// ---- all grand products of a proof in batched launches ------------------------------------------------------
// halo2 builds the permutation argument's z per chunk and each lookup's z one after the other; every one of them is
// "denominators -> batch inversion -> numerators -> running product", and the inversion is ONE division-step chain per
// lane (~65 us) whatever the size.  Here the P products of a proof share the launches: blockIdx.y = product, one
// inversion pass over P * n elements, one three-launch running product with grid.y = P; a chunk's z continues from the
// previous chunk's last usable value through a device-side scalar (no host round trip).
__global__ void __launch_bounds__(256) grand_fraction_kernel(GrandProducts g, words8 beta_w, words8 gamma_w, words8 delta_w,
                                                             uint32_t n, uint32_t numer, const fp_words* __restrict__ pow_tab,
                                                             fp_words* __restrict__ io) {
  side_kernel_prio();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (i >= n) return;
  io += (size_t)y * n;
  const f29 beta = f29_words_to_r261<P>(beta_w.l), gamma = f29_words_to_r261<P>(gamma_w.l);
  if (y < g.n_perm) {
    f29 acc = numer ? load_hat(io + i) : f29_one<P>();
    f29 dw = f29_one<P>();
    if (numer) dw = f29_mul<P>(f29_words_to_r261<P>(g.delta_start[y].l), f29_load_r256<P>(pow_tab + i));   // hat * hat * 2^-261 = hat
    const f29 delta = f29_words_to_r261<P>(delta_w.l);
    const uint32_t ncols = g.ncols[y];
    for (uint32_t c = 0; c < ncols; c++) {
      f29 v = load_hat(g.perm[y].values[c] + i);                 // < 2
      f29 t = numer ? f29_mul<P>(dw, beta) : f29_mul<P>(load_hat(g.perm[y].sigma[c] + i), beta);
      t = f29_add(f29_add(t, gamma), v);                         // < 6
      acc = f29_mul<P>(acc, t);                                  // 12
      if (numer) dw = f29_mul<P>(dw, delta);
    }
    store_hat(io + i, acc);
  } else {
    const uint32_t l = y - g.n_perm;
    const fp_words* x = g.lookup[l][numer ? 0 : 2];
    const fp_words* t_ = g.lookup[l][numer ? 1 : 3];
    f29 t = f29_mul<P>(f29_add(load_hat(x + i), beta), f29_add(load_hat(t_ + i), gamma));   // 4 * 4
    if (numer) t = f29_mul<P>(t, load_hat(io + i));
    store_hat(io + i, t);
  }
}
__global__ void __launch_bounds__(256) prefix_product_blocks_batch(const fp_words* __restrict__ a, uint32_t n, uint32_t nblk,
                                                                   fp_words* __restrict__ bprod) {
  side_kernel_prio();
  __shared__ uint32_t sh[PP_THREADS][9];
  a += (size_t)blockIdx.y * n;
  const uint32_t tid = threadIdx.x, first = (blockIdx.x * PP_THREADS + tid) * PP_CH;
  f29 acc = f29_one<P>();
  for (uint32_t i = 0; i < PP_CH; i++)
    if (first + i < n) acc = f29_mul<P>(acc, load_hat(a + first + i));
  f29 total;
  block_exclusive_scan_mul(acc, sh, tid, PP_THREADS, &total);
  if (tid == 0) store_hat(bprod + (size_t)blockIdx.y * nblk + blockIdx.x, total);
}
__global__ void __launch_bounds__(1024) prefix_product_scan_blocks_batch(fp_words* __restrict__ bprod, uint32_t nblk) {
  side_kernel_prio();
  __shared__ uint32_t sh[1024][9];
  bprod += (size_t)blockIdx.y * nblk;
  const uint32_t tid = threadIdx.x;
  f29 mine = tid < nblk ? load_hat(bprod + tid) : f29_one<P>();
  f29 ex = block_exclusive_scan_mul(mine, sh, tid, 1024, nullptr);
  if (tid < nblk) store_hat(bprod + tid, ex);
}
__global__ void __launch_bounds__(256) prefix_product_write_batch(const fp_words* __restrict__ a, uint32_t n, uint32_t nblk,
                                                                  const fp_words* __restrict__ bprod, uint32_t count_out,
                                                                  GrandOut outs) {
  side_kernel_prio();
  __shared__ uint32_t sh[PP_THREADS][9];
  a += (size_t)blockIdx.y * n;
  fp_words* __restrict__ out = outs.z[blockIdx.y];
  const uint32_t tid = threadIdx.x, first = (blockIdx.x * PP_THREADS + tid) * PP_CH;
  f29 v[PP_CH];
  f29 acc = f29_one<P>();
#pragma unroll
  for (uint32_t i = 0; i < PP_CH; i++) {
    v[i] = (first + i < n) ? load_hat(a + first + i) : f29_one<P>();
    acc = f29_mul<P>(acc, v[i]);
  }
  f29 run = f29_mul<P>(block_exclusive_scan_mul(acc, sh, tid, PP_THREADS, nullptr), load_hat(bprod + (size_t)blockIdx.y * nblk + blockIdx.x));
#pragma unroll
  for (uint32_t i = 0; i < PP_CH; i++) {
    if (first + i < count_out) store_hat(out + first + i, run);
    if (outs.closing && first + i == outs.closing_row) store_hat(outs.closing + blockIdx.y, run);
    run = f29_mul<P>(run, v[i]);
  }
}
// z[i] *= *scalar (a value another kernel of the stream has just written: the previous chunk's z at its last usable row)
__global__ void __launch_bounds__(256) scale_by_device_scalar_kernel(fp_words* __restrict__ z, uint32_t n, const fp_words* __restrict__ scalar,
                                                                     fp_words* __restrict__ closing, uint32_t closing_row) {
  side_kernel_prio();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const f29 v = f29_mul<P>(load_hat(z + i), load_hat(scalar));
  store_hat(z + i, v);
  if (closing && i == closing_row) store_hat(closing, v);   // (the unscaled value prefix_product_write_batch left there is replaced)
}
size_t grand_products_mod_elems(size_t n, uint32_t products) { return (size_t)products * n; }
size_t grand_products_tmp_elems(size_t n, uint32_t products) { return (size_t)products * ((n + 1 + PP_BLOCK - 1) / PP_BLOCK) + 1; }
hipError_t poly_grand_products(const GrandProducts& g, const words8& beta, const words8& gamma, const words8& delta, size_t n,
                               size_t usable, const fp_words* d_pow_tab, fp_words* d_mod, fp_words* d_tmp, const GrandOut& outs,
                               hipStream_t stream) {
  const uint32_t Pn = g.n_perm + g.n_lookup;
  if (Pn == 0) return hipSuccess;
  if (Pn > GRAND_MAX || n == 0 || usable >= n || (g.n_perm && !d_pow_tab)) return hipErrorInvalidValue;
  const uint32_t nblk = (uint32_t)((n + 1 + PP_BLOCK - 1) / PP_BLOCK);   // covers out[0..n]
  if (nblk > 1024) return hipErrorInvalidValue;                          // n <= 2^21
  const dim3 rows((unsigned)((n + 255) / 256), Pn);
  grand_fraction_kernel<<<rows, 256, 0, stream>>>(g, beta, gamma, delta, (uint32_t)n, 0u, d_pow_tab, d_mod);
  batch_invert_kernel<<<(unsigned)(((size_t)Pn * n / BI_CH + 255) / 256 + 1), 256, 0, stream>>>(d_mod, (uint32_t)(Pn * n));
  grand_fraction_kernel<<<rows, 256, 0, stream>>>(g, beta, gamma, delta, (uint32_t)n, 1u, d_pow_tab, d_mod);
  prefix_product_blocks_batch<<<dim3(nblk, Pn), PP_THREADS, 0, stream>>>(d_mod, (uint32_t)n, nblk, d_tmp);
  prefix_product_scan_blocks_batch<<<dim3(1, Pn), 1024, 0, stream>>>(d_tmp, nblk);
  prefix_product_write_batch<<<dim3(nblk, Pn), PP_THREADS, 0, stream>>>(d_mod, (uint32_t)n, nblk, d_tmp, (uint32_t)n, outs);
  // chunk j of the permutation argument starts where chunk j - 1 ended: z_j = z_{j-1}[usable] * (its own running product)
  for (uint32_t j = 1; j < g.n_perm; j++)
    scale_by_device_scalar_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(outs.z[j], (uint32_t)n, outs.z[j - 1] + usable,
                                                                                   outs.closing ? outs.closing + j : nullptr, outs.closing_row);
  return hipGetLastError();
}

hipError_t poly_mul_elementwise(const fp_words* d_a, const fp_words* d_b, size_t n, fp_words* d_out,
                                hipStream_t stream) {
  if (!n) return hipSuccess;
  mul_elementwise_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(d_a, d_b, (uint32_t)n, d_out);
  return hipGetLastError();
}

size_t poly_eval_batch_blocks(size_t n) {   // partials per polynomial (what d_partial holds m times)
  const size_t per = (size_t)(n <= ((size_t)1 << 24) ? 16 : 32) * EV_THREADS;
  return (n + per - 1) / per;
}
hipError_t poly_eval_batch(const fp_words* const* d_polys, const words8* xs, uint32_t m, size_t n, fp_words* d_partial,
                           fp_words* d_out, hipStream_t stream) {
  if (m == 0 || m > EVAL_BATCH_MAX) return hipErrorInvalidValue;
  const bool small = n <= ((size_t)1 << 24);
  const uint32_t blocks = (uint32_t)poly_eval_batch_blocks(n);
  if (n == 0 || blocks > (small ? 16u : 32u) * EV_THREADS) return hipErrorInvalidValue;
  EvalBatchArgs a;
  for (uint32_t j = 0; j < m; j++) {
    a.polys[j] = d_polys[j];
    a.x[j] = xs[j];
  }
  if (small) {
    eval_poly_batch_kernel<16><<<dim3(blocks, m), EV_THREADS, 0, stream>>>(a, (uint32_t)n, 0, blocks, d_partial, d_out);
    eval_poly_batch_kernel<16><<<dim3(1, m), EV_THREADS, 0, stream>>>(a, blocks, 1, blocks, d_partial, d_out);
  } else {
    eval_poly_batch_kernel<32><<<dim3(blocks, m), EV_THREADS, 0, stream>>>(a, (uint32_t)n, 0, blocks, d_partial, d_out);
    eval_poly_batch_kernel<32><<<dim3(1, m), EV_THREADS, 0, stream>>>(a, blocks, 1, blocks, d_partial, d_out);
  }
  return hipGetLastError();
}
hipError_t poly_kate_division(const fp_words* d_a, size_t n, const words8& b, fp_words* d_tmp, fp_words* d_q,
                              fp_words* d_rem, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const uint32_t nblk = (uint32_t)((n + KD_BLOCK - 1) / KD_BLOCK);
  if (nblk > 1024) return hipErrorInvalidValue;
  const KatePowers pw = kate_powers(b);
  if (nblk == 1) {  // n <= 2048: the block's suffix scan is the whole division, one launch instead of three
    kate_write<<<1, KD_THREADS, 0, stream>>>(d_a, (uint32_t)n, pw, nullptr, d_q, d_rem);
    return hipGetLastError();
  }
  kate_blocks<<<nblk, KD_THREADS, 0, stream>>>(d_a, (uint32_t)n, pw, d_tmp);
  uint32_t scan_threads = 64;
  while (scan_threads < nblk) scan_threads <<= 1;
  kate_scan_blocks<<<1, scan_threads, 0, stream>>>(d_tmp, nblk, pw);
  kate_write<<<nblk, KD_THREADS, 0, stream>>>(d_a, (uint32_t)n, pw, d_tmp, d_q, d_rem);
  return hipGetLastError();
}
size_t kate_batch_powers_bytes(uint32_t m) { return (size_t)m * sizeof(KatePowers); }
size_t kate_batch_tmp_elems(size_t n, uint32_t m) { return (size_t)m * ((n + KD_BLOCK - 1) / KD_BLOCK) + 1; }
// q[j] = a[j] / (X - b[j]) for m <= 16 polynomials of n coefficients (n written per quotient, the last one 0); h_pw: m
// KatePowers on the host (scratch), d_pw / d_tmp: device scratch (kate_batch_powers_bytes / kate_batch_tmp_elems)
hipError_t poly_kate_division_batch(const fp_words* const* d_a, size_t n, const words8* b, uint32_t m, fp_words* const* d_q,
                                    uint8_t* h_pw, uint8_t* d_pw, fp_words* d_tmp, hipStream_t stream) {
  if (m == 0 || n == 0) return hipSuccess;
  if (m > KATE_BATCH_MAX) return hipErrorInvalidValue;
  const uint32_t nblk = (uint32_t)((n + KD_BLOCK - 1) / KD_BLOCK);
  if (nblk > 1024) return hipErrorInvalidValue;
  KatePowers* hp = reinterpret_cast<KatePowers*>(h_pw);
  KateBatch bt{};
  for (uint32_t j = 0; j < m; j++) {
    hp[j] = kate_powers(b[j]);
    bt.a[j] = d_a[j];
    bt.q[j] = d_q[j];
  }
  hipError_t e = hipMemcpyAsync(d_pw, h_pw, kate_batch_powers_bytes(m), hipMemcpyHostToDevice, stream);
  if (e != hipSuccess) return e;
  const KatePowers* dp = reinterpret_cast<const KatePowers*>(d_pw);
  if (nblk > 1) {
    kate_blocks_batch<<<dim3(nblk, m), KD_THREADS, 0, stream>>>(bt, (uint32_t)n, dp, d_tmp, nblk);
    uint32_t scan_threads = 64;
    while (scan_threads < nblk) scan_threads <<= 1;
    kate_scan_blocks_batch<<<dim3(1, m), scan_threads, 0, stream>>>(d_tmp, nblk, dp);
  }
  kate_write_batch<<<dim3(nblk, m), KD_THREADS, 0, stream>>>(bt, (uint32_t)n, dp, d_tmp, nblk);
  return hipGetLastError();
}
hipError_t poly_lincomb(const fp_words* const* d_polys, const words8* coeffs, uint32_t m, size_t n, fp_words* d_out,
                        hipStream_t stream, const words8* low, uint32_t n_low) {
  if (m > LINCOMB_MAX || n_low > LINCOMB_LOW_MAX || n_low > n) return hipErrorInvalidValue;   // m = 0: the low polynomial alone
  if (n == 0) return hipSuccess;
  LinCombArgs a;
  for (uint32_t j = 0; j < m; j++) {
    a.polys[j] = d_polys[j];
    a.coeff[j] = coeffs[j];
  }
  a.n_low = n_low;
  for (uint32_t t = 0; t < n_low; t++) a.low[t] = low[t];
  lincomb_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(a, m, (uint32_t)n, d_out);
  return hipGetLastError();
}

hipError_t poly_lincomb_sets(const fp_words* const* d_polys, const words8* coeffs, const uint32_t* first, uint32_t n_sets, size_t n,
                             const words8* low, const uint32_t* n_low, fp_words* const* d_out, hipStream_t stream) {
  if (n_sets == 0 || n_sets > LINCOMB_SETS_MAX || first[0] != 0 || first[n_sets] > LINCOMB_SETS_POLYS) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  LinCombSetsArgs a{};
  for (uint32_t s = 0; s < n_sets; s++) {
    if (first[s + 1] < first[s] || first[s + 1] - first[s] > LINCOMB_MAX || n_low[s] > LINCOMB_SETS_LOW || n_low[s] > n) return hipErrorInvalidValue;
    a.first[s] = first[s];
    a.n_low[s] = n_low[s];
    a.out[s] = d_out[s];
    for (uint32_t t = 0; t < n_low[s]; t++) a.low[s][t] = low[s * LINCOMB_SETS_LOW + t];
  }
  a.first[n_sets] = first[n_sets];
  for (uint32_t j = 0; j < first[n_sets]; j++) {
    a.polys[j] = d_polys[j];
    a.coeff[j] = coeffs[j];
  }
  lincomb_sets_kernel<<<dim3((unsigned)((n + 255) / 256), n_sets), 256, 0, stream>>>(a, (uint32_t)n);
  return hipGetLastError();
}

// ------------------------------------------------------------------ ChaCha20-keyed uniform field elements
struct ChaChaKey {
  uint32_t w[8];
};
__device__ __forceinline__ uint32_t rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
#define SG_QR(a, b, c, d)        \
  a += b; d ^= a; d = rotl32(d, 16); \
  c += d; b ^= c; b = rotl32(b, 12); \
  a += b; d ^= a; d = rotl32(d, 8);  \
  c += d; b ^= c; b = rotl32(b, 7);
// blockIdx.y = draw d of a batch: stream id `stream + d`, its own length and output (one launch for the blinding rows of
// several columns; a single draw is a batch of one)
struct RandomBatch {
  fp_words* out[RANDOM_BATCH_MAX];
  uint32_t n[RANDOM_BATCH_MAX];
};
__global__ void __launch_bounds__(256) fr_random_kernel(ChaChaKey key, uint32_t stream_lo0, uint32_t stream_hi0, RandomBatch rb) {
  side_kernel_prio();
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = rb.n[blockIdx.y];
  if (i >= n) return;
  fp_words* __restrict__ out = rb.out[blockIdx.y];
  const uint64_t stream_id = (((uint64_t)stream_hi0 << 32) | stream_lo0) + blockIdx.y;
  const uint32_t stream_lo = (uint32_t)stream_id, stream_hi = (uint32_t)(stream_id >> 32);
  // r as 8 LE words
  const uint32_t R[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  for (uint32_t attempt = 0;; attempt++) {
    uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key.w[0], key.w[1], key.w[2], key.w[3],
                       key.w[4],    key.w[5],    key.w[6],    key.w[7],    (uint32_t)i, attempt, stream_lo, stream_hi};
    uint32_t x[16];
#pragma unroll
    for (int t = 0; t < 16; t++) x[t] = in[t];
#pragma unroll 1
    for (int round = 0; round < 10; round++) {
      SG_QR(x[0], x[4], x[8], x[12]) SG_QR(x[1], x[5], x[9], x[13]) SG_QR(x[2], x[6], x[10], x[14]) SG_QR(x[3], x[7], x[11], x[15])
      SG_QR(x[0], x[5], x[10], x[15]) SG_QR(x[1], x[6], x[11], x[12]) SG_QR(x[2], x[7], x[8], x[13]) SG_QR(x[3], x[4], x[9], x[14])
    }
    uint32_t c[8];
#pragma unroll
    for (int t = 0; t < 8; t++) c[t] = x[t] + in[t];
    c[7] &= 0x3fffffffu;
    bool lt = false, eq = true;  // c < r ?
#pragma unroll
    for (int t = 7; t >= 0; t--) {
      lt = lt || (eq && c[t] < R[t]);
      eq = eq && c[t] == R[t];
    }
    if (lt) {
      fp_words w;
      w.q[0] = make_uint4(c[0], c[1], c[2], c[3]);
      w.q[1] = make_uint4(c[4], c[5], c[6], c[7]);
      out[i] = w;
      return;
    }
  }
}
#undef SG_QR
hipError_t poly_random_batch(const uint32_t key[8], uint64_t first_stream_id, uint32_t m, fp_words* const* d_out, const size_t* n,
                             hipStream_t stream) {
  if (!m) return hipSuccess;
  if (m > RANDOM_BATCH_MAX) return hipErrorInvalidValue;
  RandomBatch rb{};
  size_t longest = 0;
  for (uint32_t d = 0; d < m; d++) {
    if (n[d] >= ((size_t)1 << 32)) return hipErrorInvalidValue;  // the block counter is the element index
    rb.out[d] = d_out[d];
    rb.n[d] = (uint32_t)n[d];
    longest = std::max(longest, n[d]);
  }
  if (!longest) return hipSuccess;
  ChaChaKey k;
  for (int t = 0; t < 8; t++) k.w[t] = key[t];
  fr_random_kernel<<<dim3((unsigned)((longest + 255) / 256), m), 256, 0, stream>>>(k, (uint32_t)first_stream_id,
                                                                                   (uint32_t)(first_stream_id >> 32), rb);
  return hipGetLastError();
}
hipError_t poly_random(const uint32_t key[8], uint64_t stream_id, size_t n, fp_words* d_out, hipStream_t stream) {
  return poly_random_batch(key, stream_id, 1, &d_out, &n, stream);
}

// ------------------------------------------------------------------ lookup permutation for range tables
// work layout (u32): hist_a[B] | hist_t[B] | pre_a[B] | pre_rep[B] | pre_left[B] | left[B]
__device__ __forceinline__ bool small_canonical(const fp_words* p, uint32_t* v) {
  f29 k = f29_zero();
  k.l[0] = 32;  // canonical = x~ * 2^5 * 2^-261
  uint32_t w[8];
  f29_to_words(f29_cond_sub_p<Fr29>(f29_mul<Fr29>(f29_load_r256<Fr29>(p), k)), w);
  *v = w[0];
  return !(w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7]) && w[0] < LOOKUP_BINS;
}
// Round 5: values below LOOKUP_LDS_BINS (every value of an 8-bit range table) are counted in the workgroup's LDS first and reach
// the global bins with one atomic per value the workgroup has seen -- most rows of a range check hold the same value (unused
// rows: 0), and 2 048 waves adding to ONE global word, even with one atomic per wave, were 44 us at 4 % VALU busy
// (profiles/r05z_proof_budget.json)
static constexpr uint32_t LOOKUP_LDS_BINS = 256;
__global__ void __launch_bounds__(256) lookup_permute_hist(const fp_words* __restrict__ input, const fp_words* __restrict__ table,
                                                           size_t rows, uint32_t* __restrict__ work, uint32_t* __restrict__ flag) {
  side_kernel_prio();
  __shared__ uint32_t s_a[LOOKUP_LDS_BINS], s_t[LOOKUP_LDS_BINS], s_max;
  s_a[threadIdx.x] = 0;
  s_t[threadIdx.x] = 0;
  if (threadIdx.x == 0) s_max = 0;
  __syncthreads();
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows) {
    uint32_t a, t;
    if (!small_canonical(table + i, &t)) {
      atomicMax(flag, 2u);
    } else {
      if (t < LOOKUP_LDS_BINS) atomicAdd(&s_t[t], 1u);
      else atomicAdd(&work[LOOKUP_BINS + t], 1u);
      atomicMax(&s_max, t);   // highest table value: bounds the scans and searches below (a range table uses 256 of the 65536 bins)
      if (!small_canonical(input + i, &a)) {
        atomicMax(flag, 1u);
      } else {
        if (a < LOOKUP_LDS_BINS) atomicAdd(&s_a[a], 1u);
        else atomicAdd(&work[a], 1u);
      }
    }
  }
  __syncthreads();
  if (s_a[threadIdx.x]) atomicAdd(&work[threadIdx.x], s_a[threadIdx.x]);
  if (s_t[threadIdx.x]) atomicAdd(&work[LOOKUP_BINS + threadIdx.x], s_t[threadIdx.x]);
  if (threadIdx.x == 0 && s_max > *reinterpret_cast<volatile uint32_t*>(flag + 1)) atomicMax(flag + 1, s_max);
}
// one workgroup: three exclusive prefix sums over the bins (input counts, repeated rows, leftover table values)
__global__ void __launch_bounds__(1024) lookup_permute_scan(uint32_t* __restrict__ work, uint32_t* __restrict__ flag, uint32_t rows) {
  side_kernel_prio();
  __shared__ uint32_t s_sum[3][1024];
  const uint32_t bound = min(flag[1] + 1, LOOKUP_BINS), PER = (bound + 1023) / 1024;
  const uint32_t tid = threadIdx.x;
  uint32_t* hist_a = work;
  uint32_t* hist_t = work + LOOKUP_BINS;
  uint32_t* pre[3] = {work + 2 * LOOKUP_BINS, work + 3 * LOOKUP_BINS, work + 4 * LOOKUP_BINS};
  uint32_t* left = work + 5 * LOOKUP_BINS;
  uint32_t tot[3] = {0, 0, 0};
  bool missing = false;
  for (uint32_t j = 0; j < PER; j++) {
    const uint32_t v = tid * PER + j;
    if (v >= bound) break;
    const uint32_t ca = hist_a[v], ct = hist_t[v];
    const uint32_t used = ca ? 1u : 0u;
    missing = missing || ct < used;
    tot[0] += ca;
    tot[1] += ca - used;
    tot[2] += ct - min(ct, used);
  }
  if (missing) atomicMax(flag, 1u);
  for (int q = 0; q < 3; q++) s_sum[q][tid] = tot[q];
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    uint32_t add[3] = {0, 0, 0};
    if (tid >= d)
      for (int q = 0; q < 3; q++) add[q] = s_sum[q][tid - d];
    __syncthreads();
    for (int q = 0; q < 3; q++) s_sum[q][tid] += add[q];
    __syncthreads();
  }
  // an input above the table's maximum sits in a bin that was not scanned: the counts do not add up
  if (tid == 1023 && s_sum[0][1023] != rows) atomicMax(flag, 1u);
  uint32_t run[3];
  for (int q = 0; q < 3; q++) run[q] = s_sum[q][tid] - tot[q];
  for (uint32_t j = 0; j < PER; j++) {
    const uint32_t v = tid * PER + j;
    if (v >= bound) break;
    const uint32_t ca = hist_a[v], ct = hist_t[v];
    const uint32_t used = ca ? 1u : 0u, lf = ct - min(ct, used);
    pre[0][v] = run[0];
    pre[1][v] = run[1];
    pre[2][v] = run[2];
    left[v] = lf;
    run[0] += ca;
    run[1] += ca - used;
    run[2] += lf;
  }
}
// largest v with pre[v] <= x among the bins that own at least one element (count[v] > 0 and pre[v] <= x < pre[v] + count[v])
__device__ __forceinline__ uint32_t bin_of(const uint32_t* __restrict__ pre, const uint32_t* __restrict__ count, uint32_t x,
                                           uint32_t bound) {
  uint32_t lo = 0, hi = bound - 1;
  while (lo < hi) {   // last v with pre[v] <= x
    const uint32_t mid = (lo + hi + 1) >> 1;
    if (pre[mid] <= x) lo = mid; else hi = mid - 1;
  }
  while (count[lo] == 0 && lo > 0) lo--;   // empty bins share their prefix with the owner before them
  return lo;
}
// opts (all optional): `clean` = the work space of the NEXT call on this stream, whose histograms and flag words this launch
// zeroes (2 * LOOKUP_BINS + 2 words at `clean`, the flag words first in clean_flag): no memset launches; `status` = where the
// call's verdict goes (0 ok, 1 an input not in the table, 2 table not a range table), e.g. mapped host memory; mont: the
// outputs in Montgomery form (otherwise canonical small integers, which the caller converts)
struct LookupWriteOpts {
  uint32_t* clean;
  uint32_t* clean_flag;
  uint32_t* status;
  uint32_t mont;
};
__global__ void __launch_bounds__(256) lookup_permute_write(size_t rows, const uint32_t* __restrict__ work,
                                                            const uint32_t* __restrict__ flag, fp_words* __restrict__ out_a,
                                                            fp_words* __restrict__ out_s, LookupWriteOpts o) {
  side_kernel_prio();
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o.clean) {
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t j = i; j < 2 * (size_t)LOOKUP_BINS; j += step) o.clean[j] = 0u;
    if (i < 2) o.clean_flag[i] = 0u;
  }
  const uint32_t verdict = flag[0];
  if (o.status && i == 0) {
    *reinterpret_cast<volatile uint32_t*>(o.status) = verdict;
    __threadfence_system();
  }
  if (i >= rows || verdict) return;   // flagged inputs: the caller discards the outputs
  const uint32_t bound = min(flag[1] + 1, LOOKUP_BINS);
  const uint32_t* hist_a = work;
  const uint32_t *pre_a = work + 2 * LOOKUP_BINS, *pre_rep = work + 3 * LOOKUP_BINS, *pre_left = work + 4 * LOOKUP_BINS,
                 *left = work + 5 * LOOKUP_BINS;
  const uint32_t v = bin_of(pre_a, hist_a, (uint32_t)i, bound);
  uint32_t s = v;
  const uint32_t within = (uint32_t)i - pre_a[v];
  if (within) s = bin_of(pre_left, left, pre_rep[v] + within - 1, bound);
  if (o.mont) {   // v * 2^256 mod r: what fr_montgomery(.., to_mont) makes of the canonical words
    f29 a = f29_zero(), b = f29_zero();
    a.l[0] = v;   // v, s < 2^16: one limb
    b.l[0] = s;
    const f29 k = f29_const<P>(P::r517);
    f29_store_canonical<P>(out_a + i, f29_mul<P>(a, k));
    f29_store_canonical<P>(out_s + i, f29_mul<P>(b, k));
    return;
  }
  fp_words w;   // canonical small integers; the caller converts both columns to Montgomery form
  w.q[0] = make_uint4(v, 0, 0, 0);
  w.q[1] = make_uint4(0, 0, 0, 0);
  out_a[i] = w;
  w.q[0].x = s;
  out_s[i] = w;
}
hipError_t poly_lookup_permute_small(const fp_words* d_input, const fp_words* d_table, size_t rows, uint32_t* d_work,
                                     fp_words* d_permuted_input, fp_words* d_permuted_table, uint32_t* d_flag,
                                     hipStream_t stream) {
  if (!rows) return hipSuccess;
  if (rows >= ((size_t)1 << 31)) return hipErrorInvalidValue;
  hipError_t e = hipMemsetAsync(d_work, 0, 2 * (size_t)LOOKUP_BINS * sizeof(uint32_t), stream);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(d_flag, 0, 2 * sizeof(uint32_t), stream);
  if (e != hipSuccess) return e;
  const unsigned blocks = (unsigned)((rows + 255) / 256);
  lookup_permute_hist<<<blocks, 256, 0, stream>>>(d_input, d_table, rows, d_work, d_flag);
  lookup_permute_scan<<<1, 1024, 0, stream>>>(d_work, d_flag, (uint32_t)rows);
  lookup_permute_write<<<blocks, 256, 0, stream>>>(rows, d_work, d_flag, d_permuted_input, d_permuted_table, LookupWriteOpts{nullptr, nullptr, nullptr, 0u});
  return hipGetLastError();
}
// the same without memsets, conversions or a copy back: d_work / d_flag must arrive zeroed (the previous call's write pass did it,
// or the allocation), d_next_work / d_next_flag are zeroed for the next call, *d_status receives the verdict, the outputs are
// Montgomery words
hipError_t poly_lookup_permute_small_chained(const fp_words* d_input, const fp_words* d_table, size_t rows, uint32_t* d_work,
                                             uint32_t* d_flag, uint32_t* d_next_work, uint32_t* d_next_flag, fp_words* d_permuted_input,
                                             fp_words* d_permuted_table, uint32_t* d_status, hipStream_t stream) {
  if (!rows) return hipSuccess;
  if (rows >= ((size_t)1 << 31)) return hipErrorInvalidValue;
  const unsigned blocks = (unsigned)((rows + 255) / 256);
  lookup_permute_hist<<<blocks, 256, 0, stream>>>(d_input, d_table, rows, d_work, d_flag);
  lookup_permute_scan<<<1, 1024, 0, stream>>>(d_work, d_flag, (uint32_t)rows);
  lookup_permute_write<<<blocks, 256, 0, stream>>>(rows, d_work, d_flag, d_permuted_input, d_permuted_table,
                                                   LookupWriteOpts{d_next_work, d_next_flag, d_status, 1u});
  return hipGetLastError();
}

}  // namespace sg
