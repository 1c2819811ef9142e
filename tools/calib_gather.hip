// Calibration of rocprofv3's FETCH_SIZE for the access pattern of msm_accumulate: every lane reads one 64-byte affine
// point (4 x 16 B) at an unrelated index of a table.  /opt/skills/guides/MI355X_MICROARCH.md (HBM section) gives the
// counter's behaviour for wide coalesced streams only (x 1/2 on gfx950) and asks for a calibration "on a known byte count
// in your own access pattern" for anything else.  This program issues a KNOWN number of such gathers:
//   calib_gather <log2 table points> <loads per lane>     (run under: rocprofv3 --pmc FETCH_SIZE -- ./calib_gather 24 64)
// and prints the bytes it asked for; tools/calib_gather.sh divides FETCH_SIZE by them.  Two table sizes: 2^20 points
// (64 MiB: resident in the 256 MiB Infinity Cache, as the MSM's base table is) and 2^24 (1 GiB: every gather misses it).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(256) gather64(const uint4* __restrict__ table, uint32_t mask, uint32_t loads, uint32_t* sink) {
  uint32_t idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (uint32_t k = 0; k < loads; k++) {
    idx = idx * 1664525u + 1013904223u;
    const uint4* p = table + (size_t)((idx >> 4) & mask) * 4;
    const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    acc.x ^= a.x ^ b.y ^ c.z ^ d.w;
    acc.y += a.y + b.z + c.w + d.x;
  }
  if (acc.x == 0x9e3779b9u && acc.y == 77u) sink[0] = 1;   // never true in practice: keeps the loads alive
}
__global__ void fill(uint4* t, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) t[i] = make_uint4((uint32_t)i, (uint32_t)(i >> 7), (uint32_t)(i * 3), 1u);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
  const int log_points = argc > 1 ? std::atoi(argv[1]) : 24;
  const uint32_t loads = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 64;
  if (log_points < 10 || log_points > 26) return 2;
  const size_t points = (size_t)1 << log_points, words = points * 4;
  uint4* table = nullptr;
  uint32_t* sink = nullptr;
  CK(hipMalloc(&table, words * sizeof(uint4)));
  CK(hipMalloc(&sink, 4));
  fill<<<(unsigned)((words + 255) / 256), 256>>>(table, words);
  CK(hipDeviceSynchronize());
  const unsigned threads = 1u << 20;
  for (int rep = 0; rep < 3; rep++) gather64<<<threads / 256, 256>>>(table, (uint32_t)(points - 1), loads, sink);
  CK(hipDeviceSynchronize());
  std::printf("{\"kernel\": \"gather64\", \"table_MiB\": %zu, \"launches\": 3, \"threads\": %u, \"loads_per_lane\": %u, \"bytes_per_launch\": %llu}\n",
              points * 64 >> 20, threads, loads, (unsigned long long)threads * loads * 64ull);
  return 0;
}
