// Does straight-line code larger than the instruction cache slow a dependent chain of point additions?
// One general XYZZ addition is ~66 KB of code; the reductions inline it at 4 call sites.  COPIES additions unrolled back
// to back inside a loop: 1 copy fits the 64 KB cache (shared by two CUs), 2 / 4 / 8 copies do not.  Same number of
// additions in every variant, one wave per SIMD (256 workgroups of 256) and 3 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I circuits_halo2_amd/csrc tools/microbench_icache.hip -o tools/microbench_icache
#include <hip/hip_runtime.h>
#include <cstdio>
#include "bn254_curve29.cuh"
using namespace sg;

__device__ xyzz29 mul_small(uint32_t k) {  // [k]G with G = (1, 2)
  typedef Fq29 P;
  uint32_t w[16];
  f29 one256 = f29_const<P>(P::r256);
  f29_to_words(one256, w);
  f29 two = f29_cond_sub_p<P>(f29_normalize(f29_add(one256, one256)));
  f29_to_words(two, w + 8);
  affine29 g = affine29_from_words(w);
  xyzz29 acc = xyzz29_identity();
  for (int bit = 31; bit >= 0; bit--) {
    acc = xyzz29_double(acc);
    if ((k >> bit) & 1) xyzz29_madd(acc, g);
  }
  return acc;
}
template <int COPIES>
__global__ void __launch_bounds__(256) chain(uint32_t* sink, int adds) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  xyzz29 A = mul_small(t * 7 + 1), B = mul_small(t * 13 + 3);
  for (int i = 0; i < adds; i += COPIES) {
#pragma unroll
    for (int c = 0; c < COPIES; c++) xyzz29_add(A, B);
  }
  uint32_t w[32];
  xyzz29_to_words(A, w);
  if (w[0] == 0x12345678u) sink[0] = w[1];
}
template <int COPIES>
static float run(uint32_t* sink, int blocks, int adds) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  chain<COPIES><<<blocks, 256>>>(sink, adds);
  hipEventRecord(a);
  chain<COPIES><<<blocks, 256>>>(sink, adds);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  return ms;
}
int main() {
  uint32_t* sink; hipMalloc(&sink, 64);
  const int adds = 64;
  for (int blocks : {256, 768}) {
    printf("%d workgroups of 256 (%d wave(s) per SIMD), %d dependent additions per lane\n", blocks, blocks / 256, adds);
    printf("  1 copy  in the loop: %7.1f us per addition\n", run<1>(sink, blocks, adds) * 1e3 / adds);
    printf("  2 copies           : %7.1f us\n", run<2>(sink, blocks, adds) * 1e3 / adds);
    printf("  4 copies           : %7.1f us\n", run<4>(sink, blocks, adds) * 1e3 / adds);
    printf("  8 copies           : %7.1f us\n", run<8>(sink, blocks, adds) * 1e3 / adds);
  }
  return 0;
}
