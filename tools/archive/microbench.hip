// Instruction-rate microbenchmark for gfx950: integer multiply-add flavours vs f64 FMA,
// used to set the integer-ALU roofline quoted in DESIGN.md (SURVEY.md §8d asks for a
// measured 32-bit-MAD/s peak next to the HBM roofline).
// build: hipcc --offload-arch=gfx950 -O3 -o microbench tools/microbench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int ILP = 8;

__global__ void k_mad_u64_u32(uint64_t* out, uint32_t a, uint32_t b) {
  uint64_t acc[ILP];
  uint32_t x = a + threadIdx.x, y = b + blockIdx.x;
  for (int i = 0; i < ILP; i++) acc[i] = i + threadIdx.x;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = (uint64_t)x * (uint32_t)(y + i) + acc[i];
  }
  uint64_t s = 0;
  for (int i = 0; i < ILP; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mul_lo(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[ILP];
  for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = acc[i] * (b + i);
  }
  uint32_t s = 0;
  for (int i = 0; i < ILP; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mul_hi(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[ILP];
  for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = __umulhi(acc[i], b + i) + 12345u;
  }
  uint32_t s = 0;
  for (int i = 0; i < ILP; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad_u24(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[ILP];
  for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = ((acc[i] & 0xffffffu) * ((b + i) & 0xffffffu)) + acc[i];
  }
  uint32_t s = 0;
  for (int i = 0; i < ILP; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_add_u64(uint64_t* out, uint32_t a, uint32_t b) {
  uint64_t acc[ILP];
  uint64_t y = ((uint64_t)b << 32) | a;
  for (int i = 0; i < ILP; i++) acc[i] = i + threadIdx.x;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = acc[i] + (y ^ acc[(i + 1) % ILP]);
  }
  uint64_t s = 0;
  for (int i = 0; i < ILP; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_add_u32(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[ILP];
  for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = (acc[i] + b) ^ acc[(i + 1) % ILP];
  }
  uint32_t s = 0;
  for (int i = 0; i < ILP; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma_f64(uint64_t* out, uint32_t a, uint32_t b) {
  double acc[ILP];
  double x = 1.0 + 1e-9 * a, y = 1e-9 * b;
  for (int i = 0; i < ILP; i++) acc[i] = i + threadIdx.x;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = __builtin_fma(acc[i], x, y);
  }
  double s = 0;
  for (int i = 0; i < ILP; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}
__global__ void k_fma_f32(uint64_t* out, uint32_t a, uint32_t b) {
  float acc[ILP];
  float x = 1.0f + 1e-7f * a, y = 1e-7f * b;
  for (int i = 0; i < ILP; i++) acc[i] = i + threadIdx.x;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = __builtin_fmaf(acc[i], x, y);
  }
  float s = 0;
  for (int i = 0; i < ILP; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}

// 8x32-bit-limb Montgomery product (the compiler-scheduled CIOS the kernels use)
struct alignas(16) fp_t { uint32_t l[8]; };
struct FrP {
  static constexpr uint32_t p[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr uint32_t inv = 0xefffffffu;
};
template <class P> __device__ __forceinline__ fp_t mont_mul(const fp_t& a, const fp_t& b) {
  uint32_t t[9];
#pragma unroll
  for (int k = 0; k < 9; k++) t[k] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { c = (uint64_t)a.l[j] * b.l[i] + t[j] + c; t[j] = (uint32_t)c; c >>= 32; }
    c += t[8]; t[8] = (uint32_t)c;
    uint32_t m = t[0] * P::inv;
    c = (uint64_t)m * P::p[0] + t[0]; c >>= 32;
#pragma unroll
    for (int j = 1; j < 8; j++) { c = (uint64_t)m * P::p[j] + t[j] + c; t[j - 1] = (uint32_t)c; c >>= 32; }
    c += t[8]; t[7] = (uint32_t)c; t[8] = (uint32_t)(c >> 32);
  }
  uint32_t s[8]; uint64_t br = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) { uint64_t d = (uint64_t)t[j] - P::p[j] - br; s[j] = (uint32_t)d; br = (d >> 32) & 1; }
  fp_t r;
#pragma unroll
  for (int j = 0; j < 8; j++) r.l[j] = br ? t[j] : s[j];
  return r;
}
constexpr int MUL_ITERS = 512;
__global__ void k_montmul(uint64_t* out, uint32_t a, uint32_t b) {
  fp_t x, y;
  for (int i = 0; i < 8; i++) { x.l[i] = a * (i + 1) + threadIdx.x; y.l[i] = b * (i + 3) + blockIdx.x; }
  x.l[7] &= 0x0fffffff; y.l[7] &= 0x0fffffff;
  for (int it = 0; it < MUL_ITERS; it++) { x = mont_mul<FrP>(x, y); y = mont_mul<FrP>(y, x); }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x.l[0] ^ y.l[3];
}

template <typename K> float timeit(K kern, int blocks, int threads, uint64_t* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<blocks, threads>>>(out, 3, 5);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; r++) kern<<<blocks, threads>>>(out, 3, 5);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device %s CUs %d clock %d kHz\n", prop.name, cus, prop.clockRate);
  uint64_t* out; CK(hipMalloc(&out, sizeof(uint64_t) * cus * 64 * 1024));
  struct { const char* name; void (*k)(uint64_t*, uint32_t, uint32_t); double ops_per_thread; } tests[] = {
    {"v_mad_u64_u32", k_mad_u64_u32, (double)ITERS * ILP},
    {"v_mul_lo_u32", k_mul_lo, (double)ITERS * ILP},
    {"v_mul_hi_u32(+add)", k_mul_hi, (double)ITERS * ILP},
    {"v_mul_u24+add", k_mad_u24, (double)ITERS * ILP},
    {"u64 add(+xor)", k_add_u64, (double)ITERS * ILP},
    {"u32 add(+xor)", k_add_u32, (double)ITERS * ILP},
    {"v_fma_f64", k_fma_f64, (double)ITERS * ILP},
    {"v_fma_f32", k_fma_f32, (double)ITERS * ILP},
    {"montmul8x32", k_montmul, (double)MUL_ITERS * 2},
  };
  for (auto& t : tests) {
    for (int wpc : {4, 8, 16, 32}) {  // waves per CU
      int threads = 256, blocks = cus * wpc / 4;
      float ms = timeit(t.k, blocks, threads, out);
      double total = t.ops_per_thread * (double)threads * blocks;
      double per_cu_clk = total / (ms * 1e-3) / cus / 2.4e9;
      printf("%-20s waves/CU %2d  %8.3f ms  %10.3f Gop/s  %7.2f op/clk/CU@2.4GHz  (wave-instr every %.2f clk/SIMD)\n", t.name, wpc, ms,
             total / ms * 1e-6, per_cu_clk, 64.0 * 4 / per_cu_clk);
    }
  }
  return 0;
}
