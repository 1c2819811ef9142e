// Host-side interface of the quotient-numerator kernels (see quotient.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "msm.h"

namespace sg {
static constexpr uint32_t QUOT_MAX_SETS = 8, QUOT_MAX_COLS = 16, QUOT_MAX_COSETS = 8;
struct QuotPermArgs {  // kernel argument; all arrays 2^ext_k x 32 B (cosets = 0) or cosets x 2^k x 32 B, memory (2^256) domain
  fp_words* values;
  const fp_words* z[QUOT_MAX_SETS];
  const fp_words* cols[QUOT_MAX_COLS];
  const fp_words* sigma[QUOT_MAX_COLS];
  const fp_words* l0;
  const fp_words* l_last;
  const fp_words* l_active;
  const fp_words* pow_lo;  // omega_ext^t, t < 256, 2^261-domain words (NttEngine local twiddles)
  uint32_t nsets, ncols, chunk_len, k, ext_k, last_rot_abs;
  uint32_t beta[8], gamma[8], y[8], delta[8], zeta[8], omega_ext[8];
  // coset-major layout (cosets > 0): block b of 2^k rows is the coset shift[b] * H; ext_k = k, omega_ext = omega, zeta unused
  uint32_t cosets;
  uint32_t shift[QUOT_MAX_COSETS][8];
};
struct QuotLookupArgs {
  fp_words* values;
  const fp_words *z, *permuted_input, *permuted_table, *input, *table, *l0, *l_last, *l_active;
  uint32_t k, ext_k;
  uint32_t beta[8], gamma[8], y[8];
  uint32_t cosets;   // > 0: coset-major arrays of cosets * 2^k rows (ext_k = k)
};
// ---- the quotient on quotient-degree many cosets of the 2^k domain instead of the whole extended domain
// deg h < d n (d = cs.degree() - 1 = 5 here), so h is determined by its values on d cosets c_b H of the 2^k domain H:
// halo2 evaluates on the 2^(k+3) extended domain = 8 such cosets, c_b = zeta omega_ext^b; the first d of them suffice.
// On c_b H the polynomial X^n is the constant g_b = c_b^n, therefore
//   * the vanishing polynomial X^n - 1 is the constant g_b - 1 there, and
//   * h restricted to the coset is P_b = h mod (X^n - g_b) = sum_t g_b^t h_t, h_t the pieces of n coefficients,
// so the pieces follow from d inverse transforms of size n and one d x d Vandermonde solve per coefficient index:
//   h_t[i] = sum_b M[t][b] c_b^-i iNTT_n(values_b)[i],   M = V^-1 diag(1 / (g_b - 1)),  V[b][t] = g_b^t.
// The same h as halo2's extended_to_coeff (it is unique), from 5/8 of the rows.
static constexpr uint32_t MAX_COSETS = 8, COSET_BATCH_MAX = 16;
struct CosetScaleArgs {  // out[j][b * n + i] = in[j][i] * c_b^i
  const fp_words* in[COSET_BATCH_MAX];
  fp_words* out[COSET_BATCH_MAX];
  const fp_words* table;   // [nc][n]: c_b^i as 2^261-domain words
  uint32_t log_n, nc;
};
hipError_t coset_scale(const CosetScaleArgs& a, uint32_t count, hipStream_t stream);
struct CosetCombineArgs {  // pieces[t][i] = sum_b m[t][b] * table_inv[b][i] * raw[b * n + i]
  const fp_words* raw;
  fp_words* pieces[MAX_COSETS];
  const fp_words* table_inv;   // [nc][n]: c_b^-i as 2^261-domain words
  uint32_t log_n, nc;
  uint32_t m[MAX_COSETS * MAX_COSETS][8];   // Montgomery-2^256 words, row-major [t][b]
};
hipError_t coset_combine(const CosetCombineArgs& a, hipStream_t stream);
// table[b * n + i] = (c[b]^i) as 2^261-domain words, b < nc, i < 2^log_n
hipError_t coset_fill_powers(fp_words* table, const words8* c, uint32_t nc, uint32_t log_n, hipStream_t stream);
hipError_t quotient_permutation(const QuotPermArgs& a, hipStream_t stream);
hipError_t quotient_lookup(const QuotLookupArgs& a, hipStream_t stream);
}  // namespace sg
