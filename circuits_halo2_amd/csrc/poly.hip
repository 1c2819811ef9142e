// Device-resident polynomial helpers around the MSM/NTT hot path (the
// first "next" row that needs no circuit knowledge): halo2's `eval_polynomial` (the 35
// evaluations of create_proof step 11), `BatchInvert::batch_invert` and the exclusive prefix
// product the permutation / lookup grand products are built from (steps 5-6).  With these the
// vectors that feed the commitments never leave HBM between an NTT and an MSM.
//
// All vectors: n x 32 B Fr, Montgomery-2^256 words (halo2curves layout); arithmetic on
// 9 x 29-bit limbs (bn254_f29.cuh) in the 2^261 domain.
#include "poly.h"

namespace sg {

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
static constexpr uint32_t EV_CH = 32, EV_THREADS = 256;
__global__ void __launch_bounds__(256) eval_poly_kernel(const fp_words* __restrict__ c, uint32_t n, words8 xw,
                                                        uint32_t log_stride, fp_words* __restrict__ out) {
  // element i of this level has weight x^(i << log_stride)
  __shared__ uint32_t sh[EV_THREADS][9];
  const uint32_t tid = threadIdx.x;
  f29 x = f29_words_to_r261<P>(xw.l);
  for (uint32_t k = 0; k < log_stride; k++) x = f29_sqr<P>(x);   // x^(2^log_stride)
  const uint32_t first = (blockIdx.x * EV_THREADS + tid) * EV_CH;
  f29 acc = f29_zero();
  if (first < n) {
    uint32_t last = min(n, first + EV_CH);
    acc = load_hat(c + last - 1);
    for (uint32_t i = last - 1; i-- > first;) {
      acc = f29_mul<P>(acc, x);                         // bound 2*2 (mul out <2, x<2)
      acc = f29_add(acc, load_hat(c + i));              // < 4
    }
    acc = f29_mul<P>(acc, f29_one<P>());                // back below 2p
  }
#pragma unroll
  for (int q = 0; q < 9; q++) sh[tid][q] = acc.l[q];
  __syncthreads();
  f29 xp = x;                                            // x^(CH * 2^l) per level
  for (uint32_t k = 1; k < EV_CH; k <<= 1) xp = f29_sqr<P>(xp);
  for (uint32_t s = 1; s < EV_THREADS; s <<= 1) {
    if ((tid & (2 * s - 1)) == 0) {
      f29 lo, hi;
#pragma unroll
      for (int q = 0; q < 9; q++) { lo.l[q] = sh[tid][q]; hi.l[q] = sh[tid + s][q]; }
      lo = f29_mul<P>(f29_add(lo, f29_mul<P>(hi, xp)), f29_one<P>());   // (<2 + <2) -> <2
#pragma unroll
      for (int q = 0; q < 9; q++) sh[tid][q] = lo.l[q];
    }
    xp = f29_sqr<P>(xp);
    __syncthreads();
  }
  if (tid == 0) {
    f29 r;
#pragma unroll
    for (int q = 0; q < 9; q++) r.l[q] = sh[0][q];
    store_hat(out + blockIdx.x, r);
  }
}

// ---- batch inversion (zeros stay zero, like ff::BatchInvert) --------------------------------
static constexpr uint32_t BI_CH = 8;
__global__ void __launch_bounds__(256) batch_invert_kernel(fp_words* __restrict__ a, uint32_t n) {
  const uint32_t first = (blockIdx.x * blockDim.x + threadIdx.x) * BI_CH;
  if (first >= n) return;
  const uint32_t cnt = min(BI_CH, n - first);
  f29 pre[BI_CH];           // prefix products over the non-zero elements
  f29 acc = f29_one<P>();
#pragma unroll
  for (uint32_t i = 0; i < BI_CH; i++) {
    pre[i] = acc;
    if (i < cnt) {
      f29 v = load_hat(a + first + i);
      if (!f29_is_zero_mod_p<P>(v)) acc = f29_mul<P>(acc, v);
    }
  }
  f29 inv = f29_inv<P>(acc);
#pragma unroll
  for (uint32_t k = 0; k < BI_CH; k++) {
    const uint32_t i = BI_CH - 1 - k;
    if (i < cnt) {
      f29 v = load_hat(a + first + i);
      if (!f29_is_zero_mod_p<P>(v)) {
        store_hat(a + first + i, f29_mul<P>(inv, pre[i]));
        inv = f29_mul<P>(inv, v);
      }
    }
  }
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
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // a~ * b^ * 2^-261 = (ab)~
  f29_store_canonical<P>(out + i, f29_mul<P>(f29_load_r256<P>(a + i), load_hat(b + i)));
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
    log_stride += 13;  // log2(EV_CH * EV_THREADS)
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
__global__ void perm_fraction_kernel(PermCols cols, uint32_t ncols, words8 beta_w, words8 gamma_w, words8 dstart_w,
                                     words8 delta_w, words8 omega_w, uint32_t n, uint32_t numer,
                                     fp_words* __restrict__ io) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const f29 beta = f29_words_to_r261<P>(beta_w.l), gamma = f29_words_to_r261<P>(gamma_w.l);
  f29 acc = numer ? load_hat(io + i) : f29_one<P>();          // numerators multiply the inverted denominators
  f29 dw = f29_one<P>();
  if (numer) dw = f29_mul<P>(f29_words_to_r261<P>(dstart_w.l), f29_pow_u64<P>(f29_words_to_r261<P>(omega_w.l), i));
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
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const f29 beta = f29_words_to_r261<P>(beta_w.l), gamma = f29_words_to_r261<P>(gamma_w.l);
  f29 t = f29_mul<P>(f29_add(load_hat(x + i), beta), f29_add(load_hat(y + i), gamma));   // 4 * 4
  if (numer) t = f29_mul<P>(t, load_hat(io + i));
  store_hat(io + i, t);
}
hipError_t poly_perm_fraction(const PermCols& cols, uint32_t ncols, const words8& beta, const words8& gamma,
                              const words8& delta_start, const words8& delta, const words8& omega, size_t n,
                              int numer, fp_words* d_io, hipStream_t stream) {
  perm_fraction_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(cols, ncols, beta, gamma, delta_start, delta,
                                                                        omega, (uint32_t)n, (uint32_t)numer, d_io);
  return hipGetLastError();
}
hipError_t poly_lookup_fraction(const fp_words* d_x, const fp_words* d_y, const words8& beta, const words8& gamma,
                                size_t n, int numer, fp_words* d_io, hipStream_t stream) {
  lookup_fraction_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(d_x, d_y, beta, gamma, (uint32_t)n,
                                                                          (uint32_t)numer, d_io);
  return hipGetLastError();
}
hipError_t poly_mul_elementwise(const fp_words* d_a, const fp_words* d_b, size_t n, fp_words* d_out,
                                hipStream_t stream) {
  if (!n) return hipSuccess;
  mul_elementwise_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(d_a, d_b, (uint32_t)n, d_out);
  return hipGetLastError();
}

}  // namespace sg
