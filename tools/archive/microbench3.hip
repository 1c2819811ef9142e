// A/B beside f29_mul (bn254_f29.cuh): a 254-bit Montgomery product on 5 x 52-bit limbs with FP64 FMAs (the
// "hi/lo two-FMA split": hi = fma(a, b, C) - C is the product rounded to a multiple of 2^52, lo = fma(a, b, -hi) is
// the exact remainder; both are moved to 64-bit integer columns through the mantissa of a magic constant).  Same
// structure as the integer code: 25 partial products a_i b_j interleaved with 25 reduction products m_i p_j, column
// accumulation in int64, one carry sweep.  Measured per product, dependent chain per lane, several occupancies.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Icirculits... tools/microbench3.hip -o tools/microbench3
// The FP64 kernel is an INSTRUCTION-MIX replica (every FMA / add / integer accumulation of the real algorithm is there and
// data-dependent, the bookkeeping of the magic constants is simplified): it bounds what a correct version could reach; its
// numerical output is not checked.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../circuits_halo2_amd/csrc/bn254_f29.cuh"
using namespace sg;

// ---- 5 x 52-bit limbs, Montgomery radix 2^260, modulus = BN254 Fq
struct f52 { double l[5]; };   // exact integers < 2^52 (lazy: < 2^53)
__device__ __constant__ double P52[5];
__device__ __constant__ double PINV52_LO, PINV52_HI;   // -p^-1 mod 2^52 split in 26-bit halves (for the low product)
static constexpr double C1 = 30423614405477505635920876929024.0;   // 1.5 * 2^104: ulp = 2^52
static constexpr double C3 = 6755399441055744.0;                   // 1.5 * 2^52:  ulp = 1

__device__ __forceinline__ void pp(double a, double b, int64_t& col_lo, int64_t& col_hi) {
  const double t = __builtin_fma(a, b, C1);           // C1 + round(a b / 2^52) 2^52
  const double hi = t - C1;                            // exact
  const double lo = __builtin_fma(a, b, -hi);         // exact, |lo| <= 2^51
  col_hi += __double_as_longlong(t);                   // mantissa arithmetic: the constants' bit patterns are removed per column
  col_lo += __double_as_longlong(lo + C3);
}
// low 52 bits of x * pinv (x < 2^52): 26-bit halves through exact FP64 products
__device__ __forceinline__ double mul_lo52(double x) {
  const double two26 = 67108864.0, two52 = 4503599627370496.0;
  const double xh = __builtin_floor(x * (1.0 / two26)), xl = x - xh * two26;
  double r = xl * PINV52_LO + __builtin_fma(xl, PINV52_HI, xh * PINV52_LO) * two26;   // < 2^54 + 2^79: reduce mod 2^52 in two steps
  r -= __builtin_floor(r * (1.0 / (two52 * two26))) * (two52 * two26);
  r -= __builtin_floor(r * (1.0 / two52)) * two52;
  return r;
}
__device__ __forceinline__ f52 f52_mul(const f52& a, const f52& b) {
  int64_t col[11];
#pragma unroll
  for (int k = 0; k < 11; k++) col[k] = 0;
  const int64_t K1 = __double_as_longlong(C1), K3 = __double_as_longlong(C3);
  f52 r;
#pragma unroll
  for (int i = 0; i < 5; i++) {
#pragma unroll
    for (int j = 0; j < 5; j++) pp(a.l[i], b.l[j], col[i + j], col[i + j + 1]);
    // column i is complete up to carries: remove the magic constants it received so far and take its low 52 bits
    int64_t c = col[i] - (int64_t)(i < 5 ? (i + 1) : 5) * K3 * 2 - (int64_t)(i ? (i < 5 ? i : 5) : 0) * K1 * 2 + (i ? 0 : 0);
    const double low = (double)(c & 0xfffffffffffffLL);
    const double m = mul_lo52(low);
#pragma unroll
    for (int j = 0; j < 5; j++) pp(m, P52[j], col[i + j], col[i + j + 1]);
    col[i + 1] += col[i] >> 52;   // (the exact bookkeeping of the constants is folded into the final sweep below)
  }
#pragma unroll
  for (int k = 0; k < 5; k++) {
    int64_t c = col[5 + k] - 10 * (K1 + K3);
    r.l[k] = (double)(c & 0xfffffffffffffLL);
    col[6 + k] += c >> 52;
  }
  return r;
}

__global__ void k_f29(uint32_t* out, int iters) {
  f29 x, y;
  for (int i = 0; i < 9; i++) { x.l[i] = (threadIdx.x * 2654435761u + i * 40503u) & M29; y.l[i] = (blockIdx.x * 97u + i * 7919u + 5) & M29; }
  x.l[8] &= 0xfffff; y.l[8] &= 0xfffff;
  for (int it = 0; it < iters; it++) x = f29_mul<Fq29>(x, y);
  uint32_t s = 0;
  for (int i = 0; i < 9; i++) s ^= x.l[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_f52(uint32_t* out, int iters) {
  f52 x, y;
  for (int i = 0; i < 5; i++) { x.l[i] = (double)((threadIdx.x * 2654435761ull + i * 40503ull) & 0xfffffffffffffull); y.l[i] = (double)((blockIdx.x * 97ull + i * 7919ull + 5) & 0xfffffffffffffull); }
  for (int it = 0; it < iters; it++) x = f52_mul(x, y);
  double s = 0;
  for (int i = 0; i < 5; i++) s += x.l[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(int64_t)s;
}

template <class K>
static void bench(const char* name, K kernel, uint32_t* d_out, int cus) {
  const int iters = 512;
  for (int waves_per_cu : {4, 8, 16, 32}) {
    const int blocks = cus * waves_per_cu / 4;   // 256 threads = 4 waves
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    kernel<<<blocks, 256>>>(d_out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    kernel<<<blocks, 256>>>(d_out, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double products = (double)blocks * 256 * iters;
    const double clk_per_wave_product = ms * 1e-3 * 2.4e9 / (products / 64 / (cus * 4.0));
    printf("%-28s waves/CU %2d  %8.3f ms  %8.1f G products/s  %7.0f clk per wave-product per SIMD (@2.4 GHz)\n", name, waves_per_cu, ms,
           products / ms * 1e-6, clk_per_wave_product);
  }
}
int main() {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  // BN254 Fq in 52-bit limbs, -p^-1 mod 2^52 in 26-bit halves
  const unsigned __int128 lo = ((unsigned __int128)0x97816a916871ca8dULL << 64) | 0x3c208c16d87cfd47ULL;
  const unsigned __int128 hi = ((unsigned __int128)0x30644e72e131a029ULL << 64) | 0xb85045b68181585dULL;
  double p52[5];
  const uint64_t M52 = (1ull << 52) - 1;
  p52[0] = (double)((uint64_t)lo & M52);
  p52[1] = (double)((uint64_t)(lo >> 52) & M52);
  p52[2] = (double)((uint64_t)((lo >> 104) | (hi << 24)) & M52);
  p52[3] = (double)((uint64_t)(hi >> 28) & M52);
  p52[4] = (double)((uint64_t)(hi >> 80) & M52);
  uint64_t inv = 1, p0 = (uint64_t)lo;   // p0^-1 mod 2^64 by Newton, then negate
  for (int i = 0; i < 6; i++) inv *= 2 - p0 * inv;
  const uint64_t ninv = (0 - inv) & M52;
  const double inv_lo = (double)(ninv & ((1u << 26) - 1)), inv_hi = (double)(ninv >> 26);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(P52), p52, sizeof p52);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(PINV52_LO), &inv_lo, 8);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(PINV52_HI), &inv_hi, 8);
  uint32_t* d_out;
  (void)hipMalloc(&d_out, (size_t)cus * 8 * 256 * 4 * 2);
  printf("device  CUs %d\n", cus);
  bench("f29_mul  (9 x 29-bit, int)", k_f29, d_out, cus);
  bench("f52_mul  (5 x 52-bit, FP64)", k_f52, d_out, cus);
  return 0;
}
