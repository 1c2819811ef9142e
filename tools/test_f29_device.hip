// Device self-test of the field products (bn254_f29.cuh): the device code spells every product as column chains of
// v_mad_u64_u32 in inline asm, the host build of the same header is the plain C++ definition (row scanning), which
// tests/checks/limb_f29_check.py pins against Python integers.  This program runs f29_mul, f29_sqr, f29_mul2,
// f29_mul_add and f29_dot<2..5> on the GPU and on the host over the same operands -- random, at the top of their lazy
// bounds (limbs up to 2^29 + 3, values up to the bound the contract allows), and the edge values 0, 1, p - 1, p, 2^29 - 1
// in every limb -- for both fields, and compares all nine limbs.  tests/test_gpu_parity.py runs it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Icircuits_halo2_amd/csrc tools/test_f29_device.hip -o tools/test_f29_device
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "bn254_f29.cuh"
using namespace sg;

static constexpr int OUTS = 9;   // mul, sqr, mul2, mul_add, dot2, dot3, dot4, dot5, mul(b, a)
struct Case {
  f29 v[10];   // operands (which of them an operation reads: see run_case)
};

template <class P>
SG_HD void run_case(const Case& c, f29* out) {
  out[0] = f29_mul<P>(c.v[0], c.v[1]);                       // bounds 13 x 13
  out[1] = f29_sqr<P>(c.v[0]);
  out[2] = f29_mul2<P>(c.v[2], c.v[3], c.v[4], c.v[5]);      // 9 x 9 + 9 x 9
  out[3] = f29_mul_add<P>(c.v[0], c.v[1], c.v[6]);           // 13 x 13, z up to bound 100
  {
    const f29 a[2] = {c.v[2], c.v[4]}, b[2] = {c.v[3], c.v[5]};
    out[4] = f29_dot<P, 2>(a, b);
  }
  {
    const f29 a[3] = {c.v[7], c.v[8], c.v[9]}, b[3] = {c.v[8], c.v[9], c.v[7]};   // 5 x 5 each
    out[5] = f29_dot<P, 3>(a, b);
  }
  {
    const f29 a[4] = {c.v[7], c.v[8], c.v[9], c.v[7]}, b[4] = {c.v[8], c.v[9], c.v[7], c.v[7]};
    out[6] = f29_dot<P, 4>(a, b);
  }
  {
    const f29 a[5] = {c.v[7], c.v[8], c.v[9], c.v[7], c.v[8]}, b[5] = {c.v[8], c.v[9], c.v[7], c.v[7], c.v[8]};
    out[7] = f29_dot<P, 5>(a, b);
  }
  out[8] = f29_mul<P>(c.v[1], c.v[0]);
}
template <class P>
__global__ void k_run(const Case* cases, uint32_t n, f29* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  run_case<P>(cases[i], out + (size_t)i * OUTS);
}

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 16);
}
// a normalised value below bound * p: limbs 0..7 below 2^29 (+ `extra` <= 3 when lazy), top limb below bound * (p[8] + 1) - 1
template <class P>
static f29 random_value(uint32_t bound, int mode) {
  f29 r;
  for (int i = 0; i < 8; i++) {
    uint32_t v = mode == 2 ? M29 : (rnd() & M29);
    if (mode >= 1) v += rnd() & 3;          // lazy limbs: up to 2^29 + 2 after a carry step (the contract allows + 3)
    if (mode == 2) v = M29 + 3;
    r.l[i] = v;
  }
  const uint32_t top_max = bound * P::p[8] - 1;   // value < bound * p
  r.l[8] = mode == 2 ? top_max : rnd() % (top_max + 1);
  return r;
}
template <class P>
static f29 edge_value(int which) {
  f29 r = f29_zero();
  switch (which) {
    case 0: break;                                              // 0
    case 1: r.l[0] = 1; break;                                  // 1
    case 2: for (int i = 0; i < 9; i++) r.l[i] = P::p[i]; break;                     // p
    case 3: for (int i = 0; i < 9; i++) r.l[i] = P::p[i]; r.l[0] -= 1; break;        // p - 1
    case 4: for (int i = 0; i < 8; i++) r.l[i] = M29; r.l[8] = P::p[8] - 1; break;   // all ones below p's top limb
    default: for (int i = 0; i < 9; i++) r.l[i] = P::one[i]; break;                  // 2^261 mod p
  }
  return r;
}
template <class P>
static int run_field(const char* name) {
  std::vector<Case> cases;
  const uint32_t bounds[10] = {13, 13, 9, 9, 9, 9, 100, 5, 5, 5};
  for (int mode = 0; mode < 3; mode++)
    for (int rep = 0; rep < (mode == 2 ? 1 : 20000); rep++) {
      Case c;
      for (int j = 0; j < 10; j++) c.v[j] = random_value<P>(bounds[j], mode);
      cases.push_back(c);
    }
  for (int e0 = 0; e0 < 6; e0++)
    for (int e1 = 0; e1 < 6; e1++) {
      Case c;
      for (int j = 0; j < 10; j++) c.v[j] = edge_value<P>((j & 1) ? e1 : e0);
      cases.push_back(c);
      for (int j = 0; j < 10; j++) c.v[j] = (j % 3 == 0) ? edge_value<P>(e0) : (j % 3 == 1) ? edge_value<P>(e1) : random_value<P>(bounds[j], 1);
      cases.push_back(c);
    }
  const uint32_t n = (uint32_t)cases.size();
  Case* d_cases;
  f29* d_out;
  if (hipMalloc(&d_cases, sizeof(Case) * n) != hipSuccess || hipMalloc(&d_out, sizeof(f29) * OUTS * n) != hipSuccess) return 2;
  (void)hipMemcpy(d_cases, cases.data(), sizeof(Case) * n, hipMemcpyHostToDevice);
  k_run<P><<<(n + 63) / 64, 64>>>(d_cases, n, d_out);
  if (hipDeviceSynchronize() != hipSuccess) { printf("%s: kernel failed\n", name); return 2; }
  std::vector<f29> dev((size_t)n * OUTS);
  (void)hipMemcpy(dev.data(), d_out, sizeof(f29) * OUTS * n, hipMemcpyDeviceToHost);
  (void)hipFree(d_cases); (void)hipFree(d_out);
  size_t bad = 0;
  for (uint32_t i = 0; i < n; i++) {
    f29 ref[OUTS];
    run_case<P>(cases[i], ref);
    for (int o = 0; o < OUTS; o++)
      for (int q = 0; q < 9; q++)
        if (ref[o].l[q] != dev[(size_t)i * OUTS + o].l[q]) {
          if (bad < 5) printf("%s: case %u output %d limb %d: host %08x device %08x\n", name, i, o, q, ref[o].l[q], dev[(size_t)i * OUTS + o].l[q]);
          bad++;
        }
    // a * b == b * a, limb for limb (the column sums are symmetric)
    for (int q = 0; q < 9; q++)
      if (dev[(size_t)i * OUTS].l[q] != dev[(size_t)i * OUTS + 8].l[q]) bad++;
  }
  printf("%s: %u cases x %d products, mismatching limbs: %zu\n", name, n, OUTS, bad);
  return bad ? 1 : 0;
}
int main() {
  int rc = run_field<Fq29>("Fq");
  rc |= run_field<Fr29>("Fr");
  printf(rc ? "FAILED\n" : "all products agree\n");
  return rc;
}
