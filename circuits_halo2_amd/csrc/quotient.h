// Host-side interface of the quotient-numerator kernels (see quotient.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "msm.h"

namespace sg {
static constexpr uint32_t QUOT_MAX_SETS = 8, QUOT_MAX_COLS = 16;
struct QuotPermArgs {  // kernel argument; all arrays 2^ext_k x 32 B, memory (2^256) domain
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
};
struct QuotLookupArgs {
  fp_words* values;
  const fp_words *z, *permuted_input, *permuted_table, *input, *table, *l0, *l_last, *l_active;
  uint32_t k, ext_k;
  uint32_t beta[8], gamma[8], y[8];
};
hipError_t quotient_permutation(const QuotPermArgs& a, hipStream_t stream);
hipError_t quotient_lookup(const QuotLookupArgs& a, hipStream_t stream);
}  // namespace sg
