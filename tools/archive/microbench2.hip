// Inline-asm instruction-rate microbenchmark (gfx950): true issue cost of the integer
// instructions a 32-bit-limb Montgomery product is made of.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
constexpr int ITERS = 2048;
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

__global__ void k_mad64(uint64_t* out, uint32_t a, uint32_t b) {
  uint64_t acc[8]; uint32_t x = a + threadIdx.x, y = b + blockIdx.x;
  for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x;
  for (int it = 0; it < ITERS; it++) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y) : "vcc");
    REP8(X)
#undef X
  }
  uint64_t s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad64_sgprcarry(uint64_t* out, uint32_t a, uint32_t b) {
  uint64_t acc[8]; uint32_t x = a + threadIdx.x, y = b + blockIdx.x;
  for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x;
  for (int it = 0; it < ITERS; it++) {
#define X(i) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y) : "s20", "s21");
    REP8(X)
#undef X
  }
  uint64_t s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mullo(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[8]; uint32_t y = b + blockIdx.x;
  for (int i = 0; i < 8; i++) acc[i] = a + i + threadIdx.x;
  for (int it = 0; it < ITERS; it++) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(y));
    REP8(X)
#undef X
  }
  uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mulhi(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[8]; uint32_t y = b + blockIdx.x;
  for (int i = 0; i < 8; i++) acc[i] = a + i + threadIdx.x;
  for (int it = 0; it < ITERS; it++) {
#define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(y));
    REP8(X)
#undef X
  }
  uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad24(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[8]; uint32_t y = b + blockIdx.x;
  for (int i = 0; i < 8; i++) acc[i] = a + i + threadIdx.x;
  for (int it = 0; it < ITERS; it++) {
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(acc[i]) : "v"(y));
    REP8(X)
#undef X
  }
  uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_lshladd64(uint64_t* out, uint32_t a, uint32_t b) {
  uint64_t acc[8]; uint64_t y = ((uint64_t)b << 32) | (a + threadIdx.x);
  for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x;
  for (int it = 0; it < ITERS; it++) {
#define X(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(y));
    REP8(X)
#undef X
  }
  uint64_t s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_addco(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[8]; uint32_t y = b + blockIdx.x;
  for (int i = 0; i < 8; i++) acc[i] = a + i + threadIdx.x;
  for (int it = 0; it < ITERS; it++) {
#define X(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(acc[i]) : "v"(y) : "vcc");
    REP8(X)
#undef X
  }
  uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mov(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[8], t[8];
  for (int i = 0; i < 8; i++) { acc[i] = a + i + threadIdx.x; t[i] = b; }
  for (int it = 0; it < ITERS; it++) {
#define X(i) asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %0" : "+v"(acc[i]), "+v"(t[i]));
    REP8(X)
#undef X
  }
  uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= acc[i] ^ t[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_add3(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[8]; uint32_t y = b + blockIdx.x;
  for (int i = 0; i < 8; i++) acc[i] = a + i + threadIdx.x;
  for (int it = 0; it < ITERS; it++) {
#define X(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(acc[i]) : "v"(y));
    REP8(X)
#undef X
  }
  uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma64(uint64_t* out, uint32_t a, uint32_t b) {
  double acc[8]; double x = 1.0 + 1e-9 * a, y = 1e-9 * b;
  for (int i = 0; i < 8; i++) acc[i] = i + threadIdx.x;
  for (int it = 0; it < ITERS; it++) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(x), "v"(y));
    REP8(X)
#undef X
  }
  double s = 0; for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}
__global__ void k_mul_u64(uint64_t* out, uint32_t a, uint32_t b) {  // dependent mad chain latency (1 chain)
  uint64_t acc = threadIdx.x; uint32_t x = a + threadIdx.x, y = b;
  for (int it = 0; it < ITERS * 8; it++) {
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y) : "vcc");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <typename K> float timeit(K kern, int blocks, int threads, uint64_t* out) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  kern<<<blocks, threads>>>(out, 3, 5); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; r++) kern<<<blocks, threads>>>(out, 3, 5);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}
int main() {
  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
  int cus = prop.multiProcessorCount;
  printf("device %s CUs %d\n", prop.name, cus);
  uint64_t* out; (void)hipMalloc(&out, sizeof(uint64_t) * cus * 64 * 1024);
  struct { const char* name; void (*k)(uint64_t*, uint32_t, uint32_t); double ops; } tests[] = {
    {"v_mad_u64_u32 (vcc)", k_mad64, 8.0 * ITERS}, {"v_mad_u64_u32 (sgpr)", k_mad64_sgprcarry, 8.0 * ITERS},
    {"v_mad_u64_u32 dep-chain", k_mul_u64, 8.0 * ITERS},
    {"v_mul_lo_u32", k_mullo, 8.0 * ITERS}, {"v_mul_hi_u32", k_mulhi, 8.0 * ITERS}, {"v_mad_u32_u24", k_mad24, 8.0 * ITERS},
    {"v_lshl_add_u64", k_lshladd64, 8.0 * ITERS}, {"v_add_co+v_addc_co (pair)", k_addco, 8.0 * ITERS},
    {"v_mov_b32 x2", k_mov, 8.0 * ITERS}, {"v_add3_u32", k_add3, 8.0 * ITERS}, {"v_fma_f64", k_fma64, 8.0 * ITERS},
  };
  for (auto& t : tests) for (int wpc : {4, 8, 16, 32}) {
    int threads = 256, blocks = cus * wpc / 4;
    float ms = timeit(t.k, blocks, threads, out);
    double total = t.ops * (double)threads * blocks;
    double per_cu_clk = total / (ms * 1e-3) / cus / 2.4e9;
    printf("%-28s waves/CU %2d %8.3f ms %10.1f Gop/s  wave-instr every %6.2f clk/SIMD (@2.4GHz)\n", t.name, wpc, ms, total / ms * 1e-6, 256.0 / per_cu_clk);
  }
  return 0;
}
