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
#include "quotient_device.cuh"
#include "side_prio.cuh"

#include <cstring>

namespace sg {
SG_DEFINE_SIDE_PRIO_SETTER(quotient_set_side_prio)

__global__ void __launch_bounds__(256) quot_perm_kernel(QuotPermArgs a) {
  side_kernel_prio();
  __shared__ uint32_t sc[QUOT_PERM_CONSTS][9];
  const size_t n_blk = (size_t)1 << a.ext_k;                       // rows per block: the whole domain, or one coset
  const size_t n_ext = a.cosets ? (size_t)a.cosets << a.k : n_blk;
  quot_perm_setup(a, sc, (size_t)blockIdx.x * blockDim.x);
  __syncthreads();
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_ext) return;
  f29_store_canonical<P>(a.values + i, quot_perm_terms(a, sc, ld(a.values, i), i));
}

__global__ void __launch_bounds__(256) quot_lookup_kernel(QuotLookupArgs a) {
  side_kernel_prio();
  __shared__ uint32_t sc[QUOT_LOOKUP_CONSTS][9];
  const size_t n_blk = (size_t)1 << a.ext_k;
  const size_t n_ext = a.cosets ? (size_t)a.cosets << a.k : n_blk;
  quot_lookup_setup(a, sc);
  __syncthreads();
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_ext) return;
  f29_store_canonical<P>(a.values + i, quot_lookup_terms(a, sc, ld(a.values, i), ld(a.input, i), i));
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
