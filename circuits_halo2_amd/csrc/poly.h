// Host-side interface of the polynomial helpers (see poly.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "msm.h"

namespace sg {
// *d_out = sum_i coeffs[i] x^i; tmp buffers of poly_eval_tmp_elems(n) elements each
hipError_t poly_eval(const fp_words* d_coeffs, size_t n, const words8& x, fp_words* d_tmp_a, fp_words* d_tmp_b,
                     fp_words* d_out, hipStream_t stream);
size_t poly_eval_tmp_elems(size_t n);
// out[j] = polys[j](xs[j]) for m <= EVAL_BATCH_MAX polynomials of n <= 2^26 coefficients each, two launches;
// d_partial: m * poly_eval_batch_blocks(n) elements
size_t poly_eval_batch_blocks(size_t n);
static constexpr uint32_t EVAL_BATCH_MAX = 40;
hipError_t poly_eval_batch(const fp_words* const* d_polys, const words8* xs, uint32_t m, size_t n, fp_words* d_partial,
                           fp_words* d_out, hipStream_t stream);
// in place; zeros stay zero
hipError_t poly_batch_invert(fp_words* d_a, size_t n, hipStream_t stream);
// out[0] = 1, out[i] = a[0] * ... * a[i-1], i <= n (n + 1 outputs); n <= 2^21
// count_out <= n + 1 values are written; *init (optional) multiplies every output (z[0] = init)
hipError_t poly_prefix_product(const fp_words* d_a, size_t n, fp_words* d_tmp, fp_words* d_out, size_t count_out,
                               const words8* init, hipStream_t stream);
static constexpr uint32_t PERM_MAX_COLS = 8;
struct PermCols {  // one chunk of the permutation argument (kernel argument)
  const fp_words* values[PERM_MAX_COLS];
  const fp_words* sigma[PERM_MAX_COLS];
};
// numer = 0: io[i] = prod_c (beta sigma_c[i] + gamma + v_c[i]);  numer = 1: io[i] *= prod_c (delta_start
// delta^c omega^i beta + gamma + v_c[i])
hipError_t poly_perm_fraction(const PermCols& cols, uint32_t ncols, const words8& beta, const words8& gamma,
                              const words8& delta_start, const words8& delta, const words8& omega, size_t n,
                              int numer, fp_words* d_io, hipStream_t stream, const fp_words* d_pow_tab = nullptr);
// numer = 0: io[i] = (x[i] + beta)(y[i] + gamma);  numer = 1: io[i] *= (x[i] + beta)(y[i] + gamma)
hipError_t poly_lookup_fraction(const fp_words* d_x, const fp_words* d_y, const words8& beta, const words8& gamma,
                                size_t n, int numer, fp_words* d_io, hipStream_t stream);
size_t prefix_product_tmp_elems(size_t n);
// all grand products of one proof in batched launches (poly.hip): products 0 .. n_perm-1 are the chunks of the permutation
// argument in order (chunk j's z continues from chunk j-1's value at row `usable`), then n_lookup lookup products
static constexpr uint32_t GRAND_MAX = 8;
struct GrandProducts {   // kernel argument
  uint32_t n_perm, n_lookup;
  uint32_t ncols[GRAND_MAX];             // columns of permutation chunk p
  PermCols perm[GRAND_MAX];
  words8 delta_start[GRAND_MAX];         // delta^(index of the chunk's first column)
  const fp_words* lookup[GRAND_MAX][4];  // input, table, permuted input, permuted table (compressed expressions, Lagrange)
};
struct GrandOut {
  fp_words* z[GRAND_MAX];
  // optional: closing[p] <- z_p[closing_row] (the value every satisfied argument ends on, 1) for every product, written by the
  // kernels that produce the row -- device-visible memory, e.g. page-locked host memory mapped into the device
  fp_words* closing = nullptr;
  uint32_t closing_row = 0;
};
size_t grand_products_mod_elems(size_t n, uint32_t products);
size_t grand_products_tmp_elems(size_t n, uint32_t products);
// d_pow_tab: omega^i, i < n, as 2^261-domain words (NttEngine::local_twiddles); d_mod / d_tmp: work space of the sizes above
hipError_t poly_grand_products(const GrandProducts& g, const words8& beta, const words8& gamma, const words8& delta, size_t n,
                               size_t usable, const fp_words* d_pow_tab, fp_words* d_mod, fp_words* d_tmp, const GrandOut& outs,
                               hipStream_t stream);
// Kate division a(X) = q(X)(X - b) + a(b): q_out gets n slots (q_0..q_{n-2}, then a zero) and must not alias a,
// rem_out (optional) a(b); n <= 2^21; d_tmp: 1024 elements
size_t kate_batch_powers_bytes(uint32_t m);
size_t kate_batch_tmp_elems(size_t n, uint32_t m);
hipError_t poly_kate_division_batch(const fp_words* const* d_a, size_t n, const words8* b, uint32_t m, fp_words* const* d_q,
                                    uint8_t* h_pw, uint8_t* d_pw, fp_words* d_tmp, hipStream_t stream);
hipError_t poly_kate_division(const fp_words* d_a, size_t n, const words8& b, fp_words* d_tmp, fp_words* d_q,
                              fp_words* d_rem, hipStream_t stream);
// *d_count (device u32, zeroed here) = number of elements of the m <= 16 columns (n each) whose 256-bit word value is >= r
hipError_t poly_count_noncanonical(const fp_words* const* d_cols, uint32_t m, size_t n, uint32_t* d_count, hipStream_t stream);
// the same without the clearing and without atomics: *d_flag = 1 when any such element exists, untouched otherwise (the caller
// clears it; it may live in page-locked host memory mapped into the device: no memset launch, no copy back)
hipError_t poly_flag_noncanonical(const fp_words* const* d_cols, uint32_t m, size_t n, uint32_t* d_flag, hipStream_t stream);
// out[i] = sum_j coeffs[j] * polys[j][i], m <= LINCOMB_MAX
static constexpr uint32_t LINCOMB_MAX = 32;
// optionally + low[i] for i < n_low <= LINCOMB_LOW_MAX (a polynomial of a few coefficients, passed by value)
static constexpr uint32_t LINCOMB_LOW_MAX = 8;
hipError_t poly_lincomb(const fp_words* const* d_polys, const words8* coeffs, uint32_t m, size_t n, fp_words* d_out,
                        hipStream_t stream, const words8* low = nullptr, uint32_t n_low = 0);
// the same for up to LINCOMB_SETS_MAX independent combinations of one length in ONE launch (grid.y = combination): combination s
// takes polys / coeffs [first[s], first[s + 1]) (at most LINCOMB_MAX of LINCOMB_SETS_POLYS in all) and n_low[s] <= LINCOMB_SETS_LOW
// low coefficients -- the rotation sets of the multi-open
static constexpr uint32_t LINCOMB_SETS_MAX = 8, LINCOMB_SETS_POLYS = 48, LINCOMB_SETS_LOW = 4;
hipError_t poly_lincomb_sets(const fp_words* const* d_polys, const words8* coeffs, const uint32_t* first, uint32_t n_sets, size_t n,
                             const words8* low, const uint32_t* n_low, fp_words* const* d_out, hipStream_t stream);
// halo2 lookup::prover::permute_expression_pair for range tables (every table value < 2^16), on the device:
// A' = the input rows sorted, S' = the table rearranged so that every row has A'[i] == S'[i] or A'[i] == A'[i-1]
// (first occurrences take their value from the table, the leftover table values fill the repeated rows in
// increasing order).  d_work: LOOKUP_PERMUTE_WORK u32.  d_flag (u32, device): 0 ok, 1 an input value that is
// not in the table, 2 a table value >= 2^16 (the caller takes the general path; wins over 1).
static constexpr uint32_t LOOKUP_BINS = 1u << 16;
static constexpr size_t LOOKUP_PERMUTE_WORK = 6 * (size_t)LOOKUP_BINS + 16;
hipError_t poly_lookup_permute_small(const fp_words* d_input, const fp_words* d_table, size_t rows, uint32_t* d_work,
                                     fp_words* d_permuted_input, fp_words* d_permuted_table, uint32_t* d_flag,
                                     hipStream_t stream);
hipError_t poly_lookup_permute_small_chained(const fp_words* d_input, const fp_words* d_table, size_t rows, uint32_t* d_work,
                                             uint32_t* d_flag, uint32_t* d_next_work, uint32_t* d_next_flag, fp_words* d_permuted_input,
                                             fp_words* d_permuted_table, uint32_t* d_status, hipStream_t stream);
// n uniform field elements from ChaCha20 (RFC 8439 block function) keyed by `key` (8 LE words): element i takes the
// first 32 bytes of block (counter = i, nonce = (attempt, stream_lo, stream_hi)), top two bits cleared, and is
// redrawn with attempt + 1 while >= r (~24 %); the accepted limbs are written as they are (a uniform value in any
// fixed representation is uniform).  Blinding rows / the random polynomial of a proof, without host traffic.
hipError_t poly_random(const uint32_t key[8], uint64_t stream_id, size_t n, fp_words* d_out, hipStream_t stream);
// `m` <= RANDOM_BATCH_MAX draws in one launch: draw d fills d_out[d] with n[d] elements of stream `first_stream_id + d`
static constexpr uint32_t RANDOM_BATCH_MAX = 8;
hipError_t poly_random_batch(const uint32_t key[8], uint64_t first_stream_id, uint32_t m, fp_words* const* d_out, const size_t* n,
                             hipStream_t stream);
hipError_t poly_mul_elementwise(const fp_words* d_a, const fp_words* d_b, size_t n, fp_words* d_out,
                                hipStream_t stream);
}  // namespace sg
