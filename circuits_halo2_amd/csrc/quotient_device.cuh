// Device side of the permutation- and lookup-argument terms of the quotient numerator, shared by quotient.hip (one kernel
// per argument, values read and written in between) and numerator.hip (gates, permutation and lookup of a row in ONE pass).
// Formulas, order and domains: the header comment of quotient.hip.
#pragma once
#include "quotient.h"

namespace sg {
typedef Fr29 P;
__device__ __forceinline__ f29 ld(const fp_words* p, size_t i) { return f29_load_r256<P>(p + i); }
__device__ __forceinline__ f29 fold(const f29& acc, const f29& y_hat, const f29& raw, int j) {
  return f29_mul2<P>(acc, y_hat, raw, f29_const<P>(P::p2[j]));
}
__device__ __forceinline__ f29 lds_const(const uint32_t (*sc)[9], int k) {
  f29 r;
#pragma unroll
  for (int q = 0; q < 9; q++) r.l[q] = sc[k][q];
  return r;
}

// ---- permutation argument.  sc[0..6]: beta^, gamma~, y^, delta^, (beta shift)~, omega_ext^(first row of the workgroup)^, 1~
static constexpr int QUOT_PERM_CONSTS = 7;
// threads 0..6 of the workgroup convert the launch constants; `first`: the workgroup's first row.  __syncthreads() afterwards.
__device__ __forceinline__ void quot_perm_setup(const QuotPermArgs& a, uint32_t (*sc)[9], size_t first) {
  const uint32_t tid = threadIdx.x;
  const size_t n_blk = (size_t)1 << a.ext_k;
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
}
// acc (the row's running value, memory domain, bound <= 6) folded with the permutation argument's terms of row i
__device__ __forceinline__ f29 quot_perm_terms(const QuotPermArgs& a, const uint32_t (*sc)[9], f29 acc, size_t i) {
  const uint32_t tid = threadIdx.x;
  const size_t n_blk = (size_t)1 << a.ext_k;
  // a workgroup lies inside one block when blocks are at least a workgroup long; otherwise (tiny domains) every thread
  // takes its own block's shift and power
  const bool uniform = !a.cosets || n_blk >= blockDim.x;
  const f29 beta_hat = lds_const(sc, 0), gamma_t = lds_const(sc, 1), y_hat = lds_const(sc, 2), delta_hat = lds_const(sc, 3), one_t = lds_const(sc, 6);
  const size_t mask = n_blk - 1, base = i & ~mask;
  const size_t rot = (size_t)1 << (a.ext_k - a.k);
  const size_t i_next = base | ((i + rot) & mask);
  const size_t i_last = base | ((i + n_blk - (size_t)a.last_rot_abs * rot) & mask);

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
    cd = f29_mul<P>(lds_const(sc, 4), f29_mul<P>(lds_const(sc, 5), ld(a.pow_lo, tid)));   // tilde * hat = tilde; pow_lo holds hats
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
  return acc;
}

// ---- lookup argument.  sc[0..3]: beta~, gamma~, y^, 1~
static constexpr int QUOT_LOOKUP_CONSTS = 4;
__device__ __forceinline__ void quot_lookup_setup(const QuotLookupArgs& a, uint32_t (*sc)[9]) {
  const uint32_t tid = threadIdx.x;
  if (tid < 4) {
    f29 v;
    if (tid == 0) v = f29_from_words<0>(a.beta);
    else if (tid == 1) v = f29_from_words<0>(a.gamma);
    else if (tid == 2) v = f29_words_to_r261<P>(a.y);
    else v = f29_const<P>(P::r256);
#pragma unroll
    for (int q = 0; q < 9; q++) sc[tid][q] = v.l[q];
  }
}
// `input`: the (theta-compressed) input expression at row i, memory domain, bound <= 6
__device__ __forceinline__ f29 quot_lookup_terms(const QuotLookupArgs& a, const uint32_t (*sc)[9], f29 acc, const f29& input, size_t i) {
  const size_t n_blk = (size_t)1 << a.ext_k;
  const f29 beta_t = lds_const(sc, 0), gamma_t = lds_const(sc, 1), y_hat = lds_const(sc, 2), one_t = lds_const(sc, 3);
  const size_t mask = n_blk - 1, base = i & ~mask;
  const size_t rot = (size_t)1 << (a.ext_k - a.k);
  const size_t i_next = base | ((i + rot) & mask), i_prev = base | ((i + n_blk - rot) & mask);
  const f29 l0 = ld(a.l0, i), l_active = ld(a.l_active, i);
  const f29 z = ld(a.z, i), ap = ld(a.permuted_input, i), sp = ld(a.permuted_table, i);
  acc = fold(acc, y_hat, f29_mul<P>(f29_sub<P, 0>(one_t, z), l0), 1);
  acc = fold(acc, y_hat, f29_mul<P>(f29_mul<P>(f29_sub<P, 0>(z, one_t), z), ld(a.l_last, i)), 2);
  {
    f29 lhs = f29_mul<P>(f29_mul<P>(ld(a.z, i_next), f29_add(ap, beta_t)), f29_add(sp, gamma_t));
    f29 rhs = f29_mul<P>(f29_mul<P>(z, f29_add(input, beta_t)), f29_add(ld(a.table, i), gamma_t));
    acc = fold(acc, y_hat, f29_mul<P>(f29_sub<P, 0>(lhs, rhs), l_active), 3);
  }
  const f29 d = f29_sub<P, 0>(ap, sp);                                   // a' - s' (+2r), bound 3
  acc = fold(acc, y_hat, f29_mul<P>(d, l0), 1);
  acc = fold(acc, y_hat, f29_mul<P>(f29_mul<P>(d, f29_sub<P, 0>(ap, ld(a.permuted_input, i_prev))), l_active), 2);
  return acc;
}
}  // namespace sg
