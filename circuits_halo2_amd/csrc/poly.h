// Host-side interface of the polynomial helpers (see poly.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "msm.h"

namespace sg {
// *d_out = sum_i coeffs[i] x^i; tmp buffers of poly_eval_tmp_elems(n) elements each
hipError_t poly_eval(const fp_words* d_coeffs, size_t n, const words8& x, fp_words* d_tmp_a, fp_words* d_tmp_b,
                     fp_words* d_out, hipStream_t stream);
size_t poly_eval_tmp_elems(size_t n);
// in place; zeros stay zero
hipError_t poly_batch_invert(fp_words* d_a, size_t n, hipStream_t stream);
// out[0] = 1, out[i] = a[0] * ... * a[i-1], i <= n (n + 1 outputs); n <= 2^21
hipError_t poly_prefix_product(const fp_words* d_a, size_t n, fp_words* d_tmp, fp_words* d_out, hipStream_t stream);
size_t prefix_product_tmp_elems(size_t n);
hipError_t poly_mul_elementwise(const fp_words* d_a, const fp_words* d_b, size_t n, fp_words* d_out,
                                hipStream_t stream);
}  // namespace sg
