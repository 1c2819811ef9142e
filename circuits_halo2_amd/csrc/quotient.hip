// Permutation- and lookup-argument contributions to the quotient numerator h(X) over the
// extended coset (the generic two thirds of halo2's `evaluate_h`; the custom
// gates are circuit specific and stay with the caller).  Term order and formulas are those of
// halo2's evaluator and can be read off the generated verifier, which folds the same terms at
// one point: contracts/src/InclusionVerifier.sol:903-997.
//
//   values[i] <- values[i] * y + term(i)   for every term, in halo2's order:
//   permutation:  l0 (1 - z_0);  l_last (z_l^2 - z_l);  l0 (z_s - z_{s-1}(w^last X)) for s >= 1;
//                 l_active (z_s(wX) prod_j (v_j + beta sigma_j + gamma)
//                           - z_s(X) prod_j (v_j + beta zeta delta^j X + gamma))   per set s
//   lookup:       l0 (1 - z);  l_last (z^2 - z);
//                 l_active (z(wX)(a' + beta)(s' + gamma) - z(X)(a + beta)(s + gamma));
//                 l0 (a' - s');  l_active (a' - s')(a' - a'(w^-1 X))
// Row i of the extended domain is the point zeta * omega_ext^i; a rotation by omega^r is an
// index shift of r * 2^(ext_k - k).
//
// Arithmetic: all arrays stay in the memory (2^256) domain.  A product of j memory-domain
// values comes out scaled by 2^-5j; instead of converting operands, every term is folded with
// ONE fused two-product reduction  values * y^ + raw_term * 2^(261+5j)  (f29_mul2), which both
// applies the y-fold and undoes the drift.  ~45 products per row: VALU-bound like the rest.
#include "quotient.h"
#include "side_prio.cuh"

#include <cstring>

namespace sg {
SG_DEFINE_SIDE_PRIO_SETTER(quotient_set_side_prio)

typedef Fr29 P;
__device__ __forceinline__ f29 ld(const fp_words* p, size_t i) { return f29_load_r256<P>(p + i); }
__device__ __forceinline__ f29 fold(const f29& acc, const f29& y_hat, const f29& raw, int j) {
  return f29_mul2<P>(acc, y_hat, raw, f29_const<P>(P::p2[j]));
}

struct QuotConsts {  // per-launch constants, converted once per workgroup
  f29 beta_hat, gamma_t, y_hat, delta_hat, bz_t, w_base_hat, one_t;
};

__global__ void __launch_bounds__(256) quot_perm_kernel(QuotPermArgs a) {
  side_kernel_prio();
  __shared__ uint32_t sc[7][9];
  const uint32_t tid = threadIdx.x;
  const size_t n_blk = (size_t)1 << a.ext_k;                       // rows per block: the whole domain, or one coset
  const size_t n_ext = a.cosets ? (size_t)a.cosets << a.k : n_blk;
  const size_t first = (size_t)blockIdx.x * blockDim.x;
  // a workgroup lies inside one block when blocks are at least a workgroup long; otherwise (tiny domains) every thread
  // takes its own block's shift and power
  const bool uniform = !a.cosets || n_blk >= blockDim.x;
  if (tid < 7) {
    f29 v;
    if (tid == 0) v = f29_words_to_r261<P>(a.beta);
    else if (tid == 1) v = f29_from_words<0>(a.gamma);
    else if (tid == 2) v = f29_words_to_r261<P>(a.y);
    else if (tid == 3) v = f29_words_to_r261<P>(a.delta);
    else if (tid == 4) v = f29_mul<P>(f29_from_words<0>(a.beta), f29_words_to_r261<P>(a.cosets ? a.shift[min((size_t)QUOT_MAX_COSETS - 1, first >> a.k)] : a.zeta));   // (beta shift)~
    else if (tid == 5) v = f29_pow_u64<P>(f29_words_to_r261<P>(a.omega_ext), (uint64_t)(first & (n_blk - 1)));
    else v = f29_const<P>(P::r256);                                                                 // 1~
#pragma unroll
    for (int q = 0; q < 9; q++) sc[tid][q] = v.l[q];
  }
  __syncthreads();
  const size_t i = (size_t)blockIdx.x * blockDim.x + tid;
  if (i >= n_ext) return;
  auto cst = [&](int k) { f29 r; for (int q = 0; q < 9; q++) r.l[q] = sc[k][q]; return r; };
  const f29 beta_hat = cst(0), gamma_t = cst(1), y_hat = cst(2), delta_hat = cst(3), one_t = cst(6);
  const size_t mask = n_blk - 1, base = i & ~mask;
  const size_t rot = (size_t)1 << (a.ext_k - a.k);
  const size_t i_next = base | ((i + rot) & mask);
  const size_t i_last = base | ((i + n_blk - (size_t)a.last_rot_abs * rot) & mask);

  f29 acc = ld(a.values, i);
  const f29 l0 = ld(a.l0, i);
  // l0 (1 - z_0)
  acc = fold(acc, y_hat, f29_mul<P>(f29_sub<P, 0>(one_t, ld(a.z[0], i)), l0), 1);
  // l_last (z_l^2 - z_l) = l_last * z_l * (z_l - 1)
  {
    f29 zl = ld(a.z[a.nsets - 1], i);
    f29 t = f29_mul<P>(f29_mul<P>(f29_sub<P, 0>(zl, one_t), zl), ld(a.l_last, i));
    acc = fold(acc, y_hat, t, 2);
  }
  // l0 (z_s - z_{s-1}(w^last X))
  for (uint32_t s = 1; s < a.nsets; s++)
    acc = fold(acc, y_hat, f29_mul<P>(f29_sub<P, 0>(ld(a.z[s], i), ld(a.z[s - 1], i_last)), l0), 1);
  // product terms; current_delta~ = (beta zeta)~ * omega_ext^i, times delta per column
  f29 cd;
  if (uniform) {
    cd = f29_mul<P>(cst(4), f29_mul<P>(cst(5), ld(a.pow_lo, tid)));   // tilde * hat = tilde; pow_lo holds hats
  } else {
    const f29 bs = f29_mul<P>(f29_from_words<0>(a.beta), f29_words_to_r261<P>(a.shift[i >> a.k]));
    cd = f29_mul<P>(bs, f29_pow_u64<P>(f29_words_to_r261<P>(a.omega_ext), (uint64_t)(i & mask)));
  }
  const f29 l_active = ld(a.l_active, i);
  uint32_t col = 0;
  for (uint32_t s = 0; s < a.nsets; s++) {
    const uint32_t m = min(a.chunk_len, a.ncols - col);
    f29 left = ld(a.z[s], i_next), right = ld(a.z[s], i);
    for (uint32_t j = 0; j < m; j++, col++) {
      const f29 v = ld(a.cols[col], i);
      f29 fl = f29_add(f29_add(v, f29_mul<P>(ld(a.sigma[col], i), beta_hat)), gamma_t);    // < 4
      f29 fr = f29_add(f29_add(v, cd), gamma_t);                                         // < 4
      left = f29_mul<P>(left, fl);
      right = f29_mul<P>(right, fr);
      cd = f29_mul<P>(cd, delta_hat);
    }
    acc = fold(acc, y_hat, f29_mul<P>(f29_sub<P, 0>(left, right), l_active), (int)m + 1);
  }
  f29_store_canonical<P>(a.values + i, acc);
}

__global__ void __launch_bounds__(256) quot_lookup_kernel(QuotLookupArgs a) {
  side_kernel_prio();
  __shared__ uint32_t sc[4][9];
  const uint32_t tid = threadIdx.x;
  const size_t n_blk = (size_t)1 << a.ext_k;
  const size_t n_ext = a.cosets ? (size_t)a.cosets << a.k : n_blk;
  if (tid < 4) {
    f29 v;
    if (tid == 0) v = f29_from_words<0>(a.beta);
    else if (tid == 1) v = f29_from_words<0>(a.gamma);
    else if (tid == 2) v = f29_words_to_r261<P>(a.y);
    else v = f29_const<P>(P::r256);
#pragma unroll
    for (int q = 0; q < 9; q++) sc[tid][q] = v.l[q];
  }
  __syncthreads();
  const size_t i = (size_t)blockIdx.x * blockDim.x + tid;
  if (i >= n_ext) return;
  auto cst = [&](int k) { f29 r; for (int q = 0; q < 9; q++) r.l[q] = sc[k][q]; return r; };
  const f29 beta_t = cst(0), gamma_t = cst(1), y_hat = cst(2), one_t = cst(3);
  const size_t mask = n_blk - 1, base = i & ~mask;
  const size_t rot = (size_t)1 << (a.ext_k - a.k);
  const size_t i_next = base | ((i + rot) & mask), i_prev = base | ((i + n_blk - rot) & mask);

  f29 acc = ld(a.values, i);
  const f29 l0 = ld(a.l0, i), l_active = ld(a.l_active, i);
  const f29 z = ld(a.z, i), ap = ld(a.permuted_input, i), sp = ld(a.permuted_table, i);
  acc = fold(acc, y_hat, f29_mul<P>(f29_sub<P, 0>(one_t, z), l0), 1);
  acc = fold(acc, y_hat, f29_mul<P>(f29_mul<P>(f29_sub<P, 0>(z, one_t), z), ld(a.l_last, i)), 2);
  {
    f29 lhs = f29_mul<P>(f29_mul<P>(ld(a.z, i_next), f29_add(ap, beta_t)), f29_add(sp, gamma_t));
    f29 rhs = f29_mul<P>(f29_mul<P>(z, f29_add(ld(a.input, i), beta_t)), f29_add(ld(a.table, i), gamma_t));
    acc = fold(acc, y_hat, f29_mul<P>(f29_sub<P, 0>(lhs, rhs), l_active), 3);
  }
  const f29 d = f29_sub<P, 0>(ap, sp);                                   // a' - s' (+2r), bound 3
  acc = fold(acc, y_hat, f29_mul<P>(d, l0), 1);
  acc = fold(acc, y_hat, f29_mul<P>(f29_mul<P>(d, f29_sub<P, 0>(ap, ld(a.permuted_input, i_prev))), l_active), 2);
  f29_store_canonical<P>(a.values + i, acc);
}

// ------------------------------------------------------------------ cosets (see quotient.h)
struct CosetShifts {
  uint32_t c[MAX_COSETS][8];
};
__global__ void __launch_bounds__(256) coset_fill_powers_kernel(fp_words* __restrict__ table, CosetShifts c, uint32_t log_n) {
  side_kernel_prio();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (i >> log_n) return;
  f29_store_canonical<P>(table + ((size_t)b << log_n) + i, f29_pow_u64<P>(f29_words_to_r261<P>(c.c[b]), i));
}
hipError_t coset_fill_powers(fp_words* table, const words8* c, uint32_t nc, uint32_t log_n, hipStream_t stream) {
  if (nc == 0 || nc > MAX_COSETS) return hipErrorInvalidValue;
  CosetShifts sh{};
  for (uint32_t b = 0; b < nc; b++) std::memcpy(sh.c[b], c[b].l, 32);
  const uint32_t n = 1u << log_n;
  coset_fill_powers_kernel<<<dim3((n + 255) / 256, nc), 256, 0, stream>>>(table, sh, log_n);
  return hipGetLastError();
}
__global__ void __launch_bounds__(256) coset_scale_kernel(CosetScaleArgs a) {
  side_kernel_prio();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >> a.log_n) return;
  const f29 x = ld(a.in[blockIdx.y], i);                       // x~, bound < 6 for any 256-bit word value
  fp_words* __restrict__ out = a.out[blockIdx.y];
  for (uint32_t b = 0; b < a.nc; b++) {
    const size_t at = ((size_t)b << a.log_n) + i;
    f29_store_canonical<P>(out + at, f29_mul<P>(x, ld(a.table, at)));   // x~ * (c^i)^ * 2^-261 = (x c^i)~
  }
}
hipError_t coset_scale(const CosetScaleArgs& a, uint32_t count, hipStream_t stream) {
  if (count == 0) return hipSuccess;
  if (count > COSET_BATCH_MAX || a.nc == 0 || a.nc > MAX_COSETS) return hipErrorInvalidValue;
  const uint32_t n = 1u << a.log_n;
  coset_scale_kernel<<<dim3((n + 255) / 256, count), 256, 0, stream>>>(a);
  return hipGetLastError();
}
// the 5 x 5 (or nc x nc) solve per coefficient: stages the matrix in LDS, p_b = raw_b c_b^-i
__device__ __forceinline__ void coset_matrix_to_lds(const CosetCombineArgs& a, uint32_t nc, uint32_t (*sm)[9]) {
  const uint32_t tid = threadIdx.x;
  if (tid < nc * nc) {
    const f29 v = f29_words_to_r261<P>(a.m[(tid / nc) * MAX_COSETS + tid % nc]);
#pragma unroll
    for (int q = 0; q < 9; q++) sm[tid][q] = v.l[q];
  }
  __syncthreads();
}
__device__ __forceinline__ f29 coset_matrix_entry(const uint32_t (*sm)[9], uint32_t at) {
  f29 r;
#pragma unroll
  for (int q = 0; q < 9; q++) r.l[q] = sm[at][q];
  return r;
}
// NC <= 5 cosets, known at compile time: the loops unroll and p[] stays in registers
template <uint32_t NC>
__global__ void __launch_bounds__(256) coset_combine_kernel(CosetCombineArgs a) {
  side_kernel_prio();
  __shared__ uint32_t sm[MAX_COSETS * MAX_COSETS][9];
  coset_matrix_to_lds(a, NC, sm);
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >> a.log_n) return;
  f29 p[NC];
#pragma unroll
  for (uint32_t b = 0; b < NC; b++) {
    const size_t at = ((size_t)b << a.log_n) + i;
    p[b] = f29_mul<P>(ld(a.raw, at), ld(a.table_inv, at));     // (raw c^-i)~, < 2
  }
  for (uint32_t t = 0; t < NC; t++) {   // (not unrolled: p[] is indexed by b only)
    f29 row[NC];
#pragma unroll
    for (uint32_t b = 0; b < NC; b++) row[b] = coset_matrix_entry(sm, t * NC + b);
    f29_store_canonical<P>(a.pieces[t] + i, f29_dot<P, NC>(p, row));   // sum_b p_b m_tb under ONE reduction (bound NC * 2 * 2)
  }
}
// any a.nc <= MAX_COSETS (p[] indexed at run time)
__global__ void __launch_bounds__(256) coset_combine_any_kernel(CosetCombineArgs a) {
  side_kernel_prio();
  __shared__ uint32_t sm[MAX_COSETS * MAX_COSETS][9];
  coset_matrix_to_lds(a, a.nc, sm);
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >> a.log_n) return;
  f29 p[MAX_COSETS];
  for (uint32_t b = 0; b < a.nc; b++) {
    const size_t at = ((size_t)b << a.log_n) + i;
    p[b] = f29_mul<P>(ld(a.raw, at), ld(a.table_inv, at));
  }
  const f29 one_hat = f29_one<P>();
  for (uint32_t t = 0; t < a.nc; t++) {
    f29 acc = f29_mul<P>(p[0], coset_matrix_entry(sm, t * a.nc));
    for (uint32_t b = 1; b < a.nc; b++) acc = f29_mul2<P>(p[b], coset_matrix_entry(sm, t * a.nc + b), acc, one_hat);
    f29_store_canonical<P>(a.pieces[t] + i, acc);
  }
}
hipError_t coset_combine(const CosetCombineArgs& a, hipStream_t stream) {
  if (a.nc == 0 || a.nc > MAX_COSETS) return hipErrorInvalidValue;
  const uint32_t n = 1u << a.log_n;
  if (a.nc == 5) coset_combine_kernel<5><<<(n + 255) / 256, 256, 0, stream>>>(a);   // the reference circuit's degree
  else coset_combine_any_kernel<<<(n + 255) / 256, 256, 0, stream>>>(a);
  return hipGetLastError();
}

hipError_t quotient_permutation(const QuotPermArgs& a, hipStream_t stream) {
  const size_t n_ext = a.cosets ? (size_t)a.cosets << a.k : (size_t)1 << a.ext_k;
  quot_perm_kernel<<<(unsigned)((n_ext + 255) / 256), 256, 0, stream>>>(a);
  return hipGetLastError();
}
hipError_t quotient_lookup(const QuotLookupArgs& a, hipStream_t stream) {
  const size_t n_ext = a.cosets ? (size_t)a.cosets << a.k : (size_t)1 << a.ext_k;
  quot_lookup_kernel<<<(unsigned)((n_ext + 255) / 256), 256, 0, stream>>>(a);
  return hipGetLastError();
}

}  // namespace sg
