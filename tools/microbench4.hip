// Rate of the field products of bn254_f29.cuh as the device sees them, dependent chains per lane at several occupancies.
// Built twice -- as is (column chains in inline asm) and with -DSG_F29_ROW_SCAN (the plain C++ definition, what the
// compiler makes of it) -- the two binaries print the same checksums: the two forms agree limb for limb.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench4.hip -o tools/microbench4
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSG_F29_ROW_SCAN tools/microbench4.hip -o tools/microbench4_rows
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../circuits_halo2_amd/csrc/bn254_f29.cuh"
using namespace sg;

__device__ __forceinline__ void seed(f29& x, f29& y, f29& z) {
  for (int i = 0; i < 9; i++) {
    x.l[i] = (threadIdx.x * 2654435761u + i * 40503u) & M29;
    y.l[i] = (blockIdx.x * 97u + i * 7919u + 5) & M29;
    z.l[i] = (threadIdx.x * 40503u + blockIdx.x * 31u + i * 2654435761u + 11) & M29;
  }
  x.l[8] &= 0xfffff; y.l[8] &= 0xfffff; z.l[8] &= 0xfffff;
}
__device__ __forceinline__ uint32_t fold(const f29& x) { uint32_t s = 0; for (int i = 0; i < 9; i++) s = s * 31 + x.l[i]; return s; }

template <class P> __global__ void k_mul(uint32_t* out, int iters) {
  f29 x, y, z; seed(x, y, z);
  for (int it = 0; it < iters; it++) x = f29_mul<P>(x, y);
  out[blockIdx.x * blockDim.x + threadIdx.x] = fold(x);
}
template <class P> __global__ void k_sqr(uint32_t* out, int iters) {
  f29 x, y, z; seed(x, y, z);
  for (int it = 0; it < iters; it++) x = f29_sqr<P>(x);
  out[blockIdx.x * blockDim.x + threadIdx.x] = fold(x);
}
// operands that the compiler cannot relate to each other (with a shared operand the plain C++ form factors a sum of products)
__device__ __forceinline__ f29 other(const f29& s, int k) {
  f29 r;
  for (int i = 0; i < 9; i++) r.l[i] = (s.l[(i + k) % 9] * 2654435761u + k) & (i == 8 ? 0xfffffu : M29);
  return r;
}
// (every operand changes from one iteration to the next: the plain C++ form hoists a product of two loop invariants)
template <class P> __global__ void k_mul2(uint32_t* out, int iters) {   // x y + z w
  f29 x, y, z; seed(x, y, z);
  f29 w = other(z, 3);
  for (int it = 0; it < iters; it++) {
    const f29 t = f29_mul2<P>(x, y, z, w);
    w = z; z = y; y = x; x = t;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = fold(x);
}
template <class P> __global__ void k_mul_add(uint32_t* out, int iters) {   // Horner: x = x y + z
  f29 x, y, z; seed(x, y, z);
  for (int it = 0; it < iters; it++) x = f29_mul_add<P>(x, y, z);
  out[blockIdx.x * blockDim.x + threadIdx.x] = fold(x);
}
template <class P> __global__ void k_dot5(uint32_t* out, int iters) {      // iters / 5 dot products of five terms
  f29 x, y, z; seed(x, y, z);
  f29 a[5] = {x, other(x, 1), other(y, 2), other(z, 3), other(x, 4)}, b[5] = {z, other(y, 5), other(z, 6), other(x, 7), other(y, 8)};
  for (int it = 0; it < iters / 5; it++) {
    const f29 t = f29_dot<P, 5>(a, b);
    b[4] = b[3]; b[3] = b[2]; b[2] = b[1]; b[1] = b[0]; b[0] = a[4];
    a[4] = a[3]; a[3] = a[2]; a[2] = a[1]; a[1] = a[0]; a[0] = t;
    x = t;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = fold(x);
}
template <class P> __global__ void k_red(uint32_t* out, int iters) {       // lazy sum, then the product-free reduction
  f29 x, y, z; seed(x, y, z);
  for (int it = 0; it < iters; it++) x = f29_reduce_small<P>(f29_add(f29_add(x, y), z));
  out[blockIdx.x * blockDim.x + threadIdx.x] = fold(x);
}
#if !defined(SG_F29_ROW_SCAN)
// ---- experiment: TWO independent products as chains that alternate instruction by instruction inside one asm block (the
// second chain fills the first one's dependency gaps within the wave instead of leaving that to the other waves)
#define DT(i) "v_mad_u64_u32 %[t], vcc, %[x" #i "], %[y" #i "], %[t]\nv_mad_u64_u32 %[u], vcc, %[p" #i "], %[q" #i "], %[u]\n"
#define DV(i) [x##i] "v"(x[i]), [y##i] "v"(y[i]), [p##i] "v"(pp[i]), [q##i] "v"(q[i])
#define DS(i) [x##i] "v"(x[i]), [y##i] "s"(y[i]), [p##i] "v"(pp[i]), [q##i] "s"(y[i])
template <int N>
__device__ __forceinline__ void dual_vv(uint64_t& t, uint64_t& u, const uint32_t* x, const uint32_t* y, const uint32_t* pp, const uint32_t* q) {
  if constexpr (N == 1) asm(DT(0) : [t] "+v"(t), [u] "+v"(u) : DV(0) : "vcc");
  if constexpr (N == 2) asm(DT(0) DT(1) : [t] "+v"(t), [u] "+v"(u) : DV(0), DV(1) : "vcc");
  if constexpr (N == 3) asm(DT(0) DT(1) DT(2) : [t] "+v"(t), [u] "+v"(u) : DV(0), DV(1), DV(2) : "vcc");
  if constexpr (N == 4) asm(DT(0) DT(1) DT(2) DT(3) : [t] "+v"(t), [u] "+v"(u) : DV(0), DV(1), DV(2), DV(3) : "vcc");
  if constexpr (N > 4) { dual_vv<4>(t, u, x, y, pp, q); dual_vv<N - 4>(t, u, x + 4, y + 4, pp + 4, q + 4); }
}
#define DTS(i) "v_mad_u64_u32 %[t], vcc, %[x" #i "], %[y" #i "], %[t]\nv_mad_u64_u32 %[u], vcc, %[p" #i "], %[y" #i "], %[u]\n"
#define DSS(i) [x##i] "v"(x[i]), [y##i] "s"(y[i]), [p##i] "v"(pp[i])
template <int N>
__device__ __forceinline__ void dual_vs(uint64_t& t, uint64_t& u, const uint32_t* x, const uint32_t* pp, const uint32_t* y) {
  if constexpr (N == 1) asm(DTS(0) : [t] "+v"(t), [u] "+v"(u) : DSS(0) : "vcc");
  if constexpr (N == 2) asm(DTS(0) DTS(1) : [t] "+v"(t), [u] "+v"(u) : DSS(0), DSS(1) : "vcc");
  if constexpr (N == 3) asm(DTS(0) DTS(1) DTS(2) : [t] "+v"(t), [u] "+v"(u) : DSS(0), DSS(1), DSS(2) : "vcc");
  if constexpr (N == 4) asm(DTS(0) DTS(1) DTS(2) DTS(3) : [t] "+v"(t), [u] "+v"(u) : DSS(0), DSS(1), DSS(2), DSS(3) : "vcc");
  if constexpr (N > 4) { dual_vs<4>(t, u, x, pp, y); dual_vs<N - 4>(t, u, x + 4, pp + 4, y + 4); }
}
template <class P, int K>
__device__ __forceinline__ void pair_columns(uint64_t& t, uint64_t& u, uint32_t (&m)[9], uint32_t (&n)[9], f29& r, f29& s,
                                             const f29& a, const f29& b, const f29& c, const f29& d) {
  {
    constexpr int LO = K < 9 ? 0 : K - 8, N = (K < 9 ? K : 8) - LO + 1;
    uint32_t x[N], y[N], pp[N], q[N];
#pragma unroll
    for (int i = 0; i < N; i++) { x[i] = a.l[LO + i]; y[i] = b.l[K - LO - i]; pp[i] = c.l[LO + i]; q[i] = d.l[K - LO - i]; }
    dual_vv<N>(t, u, x, y, pp, q);
  }
  {
    constexpr int LO = K < 9 ? 0 : K - 8, HI = K < 9 ? K - 1 : 8, N = HI - LO + 1;
    if constexpr (N >= 1) {
      uint32_t x[N], pp[N], y[N];
#pragma unroll
      for (int i = 0; i < N; i++) { x[i] = m[LO + i]; pp[i] = n[LO + i]; y[i] = P::p[K - LO - i]; }
      dual_vs<N>(t, u, x, pp, y);
    }
  }
  if constexpr (K < 9) {
    m[K] = ((uint32_t)t * P::inv) & M29;
    n[K] = ((uint32_t)u * P::inv) & M29;
    t += (uint64_t)m[K] * P::p[0];
    u += (uint64_t)n[K] * P::p[0];
    t >>= 29;
    u >>= 29;
  } else {
    r.l[K - 9] = (uint32_t)t & M29;
    s.l[K - 9] = (uint32_t)u & M29;
    t >>= 29;
    u >>= 29;
  }
  if constexpr (K < 16) pair_columns<P, K + 1>(t, u, m, n, r, s, a, b, c, d);
}
template <class P>
__device__ __forceinline__ void f29_mul_pair(const f29& a, const f29& b, const f29& c, const f29& d, f29& r, f29& s) {
  uint32_t m[9], n[9];
  uint64_t t = 0, u = 0;
  pair_columns<P, 0>(t, u, m, n, r, s, a, b, c, d);
  r.l[8] = (uint32_t)t;
  s.l[8] = (uint32_t)u;
}
template <class P> __global__ void k_mul_pair(uint32_t* out, int iters) {
  f29 x, y, z; seed(x, y, z);
  for (int it = 0; it < iters / 2; it++) { f29 r, s; f29_mul_pair<P>(x, y, z, y, r, s); x = r; z = s; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = fold(x) ^ (fold(z) * 7);
}
#endif
// two independent chains of products per lane (iters / 2 rounds: the same number of products)
template <class P> __global__ void k_mul_x2(uint32_t* out, int iters) {
  f29 x, y, z; seed(x, y, z);
  for (int it = 0; it < iters / 2; it++) { x = f29_mul<P>(x, y); z = f29_mul<P>(z, y); }
  out[blockIdx.x * blockDim.x + threadIdx.x] = fold(x) ^ (fold(z) * 7);
}
// lazy sums between the products, as the curve formulas have them: x = (x + z) * (y + x), z = z * y
template <class P> __global__ void k_mixed(uint32_t* out, int iters) {
  f29 x, y, z; seed(x, y, z);
  for (int it = 0; it < iters / 2; it++) { f29 s = f29_add(x, z), u = f29_add(y, x); x = f29_mul<P>(s, u); z = f29_mul<P>(z, y); }
  out[blockIdx.x * blockDim.x + threadIdx.x] = fold(x) ^ (fold(z) * 7);
}

template <class K>
static void bench(const char* name, K kernel, uint32_t* d_out, int cus) {
  const int iters = 512;
  uint64_t sum = 0;
  for (int waves_per_cu : {4, 8, 12, 16, 20, 32}) {
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
    const double clk = ms * 1e-3 * 2.4e9 / (products / 64 / (cus * 4.0));
    printf("time      %-22s waves/CU %2d  %8.3f ms  %8.1f G products/s  %7.0f clk per wave-product per SIMD (@2.4 GHz)\n", name,
           waves_per_cu, ms, products / ms * 1e-6, clk);
    if (waves_per_cu == 4) {
      uint32_t* h = (uint32_t*)malloc((size_t)blocks * 256 * 4);
      (void)hipMemcpy(h, d_out, (size_t)blocks * 256 * 4, hipMemcpyDeviceToHost);
      for (int i = 0; i < blocks * 256; i++) sum = sum * 1000003 + h[i];
      free(h);
    }
  }
  printf("checksum  %-22s %016llx\n", name, (unsigned long long)sum);
}
int main() {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  uint32_t* d_out;
  (void)hipMalloc(&d_out, (size_t)cus * 8 * 256 * 4 * 2);
  for (int i = 0; i < 200; i++) k_mul<Fq29><<<cus * 8, 256>>>(d_out, 512);   // clocks up before anything is timed
  (void)hipDeviceSynchronize();
#if defined(SG_F29_ROW_SCAN)
  printf("device  CUs %d   products: plain C++ (row scanning)\n", cus);
#else
  printf("device  CUs %d   products: column chains\n", cus);
#endif
  bench("f29_mul<Fq>", k_mul<Fq29>, d_out, cus);
  bench("f29_mul<Fr>", k_mul<Fr29>, d_out, cus);
  bench("f29_sqr<Fq>", k_sqr<Fq29>, d_out, cus);
  bench("f29_mul2<Fr>", k_mul2<Fr29>, d_out, cus);
  bench("f29_mul_add<Fr>", k_mul_add<Fr29>, d_out, cus);
  bench("f29_dot<Fr, 5> (per term)", k_dot5<Fr29>, d_out, cus);
  bench("f29_reduce_small<Fr>", k_red<Fr29>, d_out, cus);
  bench("f29_mul<Fq> x 2", k_mul_x2<Fq29>, d_out, cus);
  bench("add, add, mul, mul <Fq>", k_mixed<Fq29>, d_out, cus);
#if !defined(SG_F29_ROW_SCAN)
  bench("f29_mul<Fq> x 2, one block", k_mul_pair<Fq29>, d_out, cus);   // same checksum as "f29_mul<Fq> x 2"
#endif
  return 0;
}
