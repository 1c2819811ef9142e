// The create_proof-shaped op schedule of tools/proof_flow.py, driven from C++ over the C ABI: the number a
// compiled host (the reference's is Rust) would see, without the Python / ctypes overhead of the other driver.
// Same shape (MstInclusion column / argument counts, 16 commitments in the six groups the Fiat-Shamir order
// allows, 9 + 9 + 1 transforms, evaluate_h = gates + permutation + lookup, 35 evaluations, multi-open), random
// inputs (timing only; every op is parity-tested on its own), a host sync wherever a challenge is derived.
//   build: hipcc -O2 -std=c++17 -Iinclude tools/proof_flow.cpp -o tools/proof_flow_cpp -Lcircuits_halo2_amd -lsumma_gpu
//   usage: proof_flow_cpp [k = 17] [n_gates = 12] [reps = 5] [multi_stream = 1]   -> one JSON line
// multi_stream: the three grand products and the three rotation sets of the multi-open are independent latency
// chains; each gets its own stream (the library keeps its work space per stream).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <vector>

#include "summa_gpu.h"

#define CK(x)                                                                         \
  do {                                                                                \
    if ((x) != SG_OK) {                                                               \
      std::fprintf(stderr, "%s failed: %s (%s:%d)\n", #x, sg_last_error(), __FILE__, __LINE__); \
      std::exit(1);                                                                   \
    }                                                                                 \
  } while (0)
#define HK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
      std::exit(1);                                                           \
    }                                                                         \
  } while (0)

static std::mt19937_64 rng(0x53554d4d41ull);
struct Fr { uint64_t l[4]; };
static Fr rand_fr() {  // any value < 2^253 is a valid (Montgomery) residue: uniformity is irrelevant for timing
  Fr f;
  for (auto& w : f.l) w = rng();
  f.l[3] &= (1ull << 61) - 1;
  return f;
}
static void* dev_random(size_t n) {
  std::vector<Fr> h(n);
  for (auto& f : h) f = rand_fr();
  void* d;
  HK(hipMalloc(&d, 32 * n));
  HK(hipMemcpy(d, h.data(), 32 * n, hipMemcpyHostToDevice));
  return d;
}
static void* dev_alloc(size_t bytes) {
  void* d;
  HK(hipMalloc(&d, bytes));
  return d;
}
using clk = std::chrono::steady_clock;
static double ms_since(clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); }

int main(int argc, char** argv) {
  const uint32_t k = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 17, ext_k = k + 3;
  const int n_gates = argc > 2 ? std::atoi(argv[2]) : 12, reps = argc > 3 ? std::atoi(argv[3]) : 5;
  const bool multi = argc > 4 ? std::atoi(argv[4]) != 0 : true;
  const size_t n = (size_t)1 << k, ne = (size_t)1 << ext_k;
  CK(sg_init(0));
  // SRS: valid points from random scalars; both bases resident, fixed-base tables built once
  void* d_sc = dev_random(n);
  void* d_g = dev_alloc(64 * n);
  CK(sg_g1_fixed_base_mul_dev(d_sc, n, d_g, nullptr));
  std::vector<uint8_t> hg(64 * n);
  HK(hipMemcpy(hg.data(), d_g, 64 * n, hipMemcpyDeviceToHost));
  uint64_t srs;
  CK(sg_srs_upload(k, hg.data(), hg.data(), &srs));
  CK(sg_srs_precompute(srs, 0, 0));
  CK(sg_srs_precompute(srs, 1, 0));
  CK(sg_srs_precompute(srs, 2, 0));
  // proving-key side (extended basis), witness columns (Lagrange basis)
  std::vector<void*> fixed_ext(11), sigma_lag(6), sigma_ext(6), advice(3);
  for (auto& p : fixed_ext) p = dev_random(ne);
  for (auto& p : sigma_lag) p = dev_random(n);
  for (auto& p : sigma_ext) p = dev_random(ne);
  for (auto& p : advice) p = dev_random(n);
  void *l0 = dev_random(ne), *l_last = dev_random(ne), *l_active = dev_random(ne), *instance = dev_random(n);
  void* perm_cols[6] = {advice[0], advice[1], advice[2], instance, dev_random(n), dev_random(n)};
  // work buffers
  void *z0 = dev_alloc(32 * n), *z1 = dev_alloc(32 * n), *zl = dev_alloc(32 * n);
  std::vector<void*> coeff(9), ext(9);
  for (auto& p : coeff) p = dev_alloc(32 * n);
  for (auto& p : ext) p = dev_alloc(32 * ne);
  void* values = dev_alloc(32 * ne);
  std::vector<void*> quot(3);
  for (auto& p : quot) p = dev_alloc(32 * n);
  void* tmps[6];
  for (auto& p : tmps) p = dev_alloc(32 * n);
  void *hx = dev_alloc(32 * n), *lx = dev_alloc(32 * n),
       *wq = dev_alloc(32 * n);
  // stand-in gate program: n_gates Poseidon-round-shaped gates folded with y (see tools/proof_flow.py)
  std::vector<Fr> consts;
  std::vector<int32_t> rotations = {0, 1};
  std::vector<sg_calculation> calcs;
  std::vector<sg_value_source> parts;
  auto vs = [](uint32_t kind, uint32_t index, uint32_t rot = 0) { return sg_value_source{kind, index, rot}; };
  auto add_calc = [&](uint32_t op, sg_value_source a, sg_value_source b = sg_value_source{0, 0, 0}) {
    sg_calculation c{};
    c.op = op; c.a = a; c.b = b;
    calcs.push_back(c);
    return vs(SG_VS_INTERMEDIATE, (uint32_t)calcs.size() - 1);
  };
  auto add_const = [&]() { consts.push_back(rand_fr()); return vs(SG_VS_CONSTANT, (uint32_t)consts.size() - 1); };
  for (int t = 0; t < n_gates; t++) {
    sg_value_source terms[2];
    for (uint32_t j = 0; j < 2; j++) {
      auto x = add_calc(SG_OP_ADD, vs(SG_VS_ADVICE, j, 0), add_const());
      auto x2 = add_calc(SG_OP_SQUARE, x), x4 = add_calc(SG_OP_SQUARE, x2);
      terms[j] = add_calc(SG_OP_MUL, add_calc(SG_OP_MUL, x4, x), add_const());
    }
    auto d = add_calc(SG_OP_SUB, add_calc(SG_OP_ADD, terms[0], terms[1]), vs(SG_VS_ADVICE, (uint32_t)t % 3, 1));
    parts.push_back(add_calc(SG_OP_MUL, vs(SG_VS_FIXED, (uint32_t)t % 11, 0), d));
  }
  {
    sg_calculation h{};
    h.op = SG_OP_HORNER; h.a = vs(SG_VS_PREVIOUS_VALUE, 0); h.b = vs(SG_VS_Y, 0);
    h.parts_offset = 0; h.parts_len = (uint32_t)parts.size();
    calcs.push_back(h);
  }
  sg_graph graph{reinterpret_cast<const uint8_t*>(consts.data()), (uint32_t)consts.size(), rotations.data(),
                 (uint32_t)rotations.size(), calcs.data(), (uint32_t)calcs.size(), parts.data(), (uint32_t)parts.size()};

  auto challenge = [&](const uint8_t* from) {  // stands for the transcript: depends on the bytes just read back
    Fr f = rand_fr();
    f.l[0] ^= from[0];
    return f;
  };
  auto b8 = [](const Fr& f) { return reinterpret_cast<const uint8_t*>(&f); };
  uint8_t omega_inv[32], n_inv[32];
  CK(sg_domain_constant(k, 1, omega_inv));
  CK(sg_domain_constant(k, 2, n_inv));
  hipStream_t st[2];
  hipEvent_t ev_fork, ev_join[2];
  for (auto& x : st) HK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
  HK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
  for (auto& x : ev_join) HK(hipEventCreateWithFlags(&x, hipEventDisableTiming));
  // fork: side streams wait for everything enqueued on the default stream; join: the default stream waits for them
  auto fork = [&]() {
    HK(hipEventRecord(ev_fork, nullptr));
    for (auto& x : st) HK(hipStreamWaitEvent(x, ev_fork, 0));
  };
  auto join = [&]() {
    for (int i = 0; i < 2; i++) {
      HK(hipEventRecord(ev_join[i], st[i]));
      HK(hipStreamWaitEvent(nullptr, ev_join[i], 0));
    }
  };
  std::map<std::string, double> best;
  for (int rep = 0; rep < reps + 1; rep++) {
    std::map<std::string, double> t;
    HK(hipDeviceSynchronize());
    auto t0 = clk::now(), t1 = t0;
    uint8_t out[16 * 64];
    // 1: advice commitments
    CK(sg_commit_batch_dev(srs, 1, advice.data(), 3, n, nullptr, out));
    t["1_advice_commit"] = ms_since(t1); t1 = clk::now();
    Fr theta = challenge(out);
    // 2: lookup permuted columns
    void* perm2[2] = {advice[2], instance};
    CK(sg_commit_batch_dev(srs, 1, perm2, 2, n, nullptr, out));
    t["2_lookup_permuted_commit"] = ms_since(t1); t1 = clk::now();
    Fr beta = challenge(out), gamma = challenge(out + 64), one = rand_fr(), delta4 = rand_fr();
    // 3: grand products, one fused commitment job (three Lagrange-form, one coefficient-form)
    if (multi) {
      // z1 continues from z0's last usable value: compute it from 1 concurrently, scale afterwards
      fork();
      CK(sg_permutation_product_dev(perm_cols, sigma_lag.data(), 4, b8(beta), b8(gamma), b8(one), k, nullptr, z0, nullptr));
      CK(sg_permutation_product_dev(perm_cols + 4, sigma_lag.data() + 4, 2, b8(beta), b8(gamma), b8(delta4), k, nullptr, z1, st[0]));
      CK(sg_lookup_product_dev(advice[0], advice[1], advice[2], instance, b8(beta), b8(gamma), n, zl, st[1]));
      uint8_t z0_last[32];   // waits for z0 only (default stream); z1 and zl keep running on their streams
      HK(hipMemcpyAsync(z0_last, static_cast<uint8_t*>(z0) + 32 * (n - 6), 32, hipMemcpyDeviceToHost, nullptr));
      HK(hipStreamSynchronize(nullptr));
      join();
      void* z1p[1] = {z1};
      CK(sg_fr_lincomb_dev(z1p, z0_last, 1, n, z1, nullptr));   // z1 *= z0[n - 6]
    } else {
      CK(sg_permutation_product_dev(perm_cols, sigma_lag.data(), 4, b8(beta), b8(gamma), b8(one), k, nullptr, z0, nullptr));
      uint8_t z0_last[32];
      HK(hipMemcpy(z0_last, static_cast<uint8_t*>(z0) + 32 * (n - 6), 32, hipMemcpyDeviceToHost));
      CK(sg_permutation_product_dev(perm_cols + 4, sigma_lag.data() + 4, 2, b8(beta), b8(gamma), b8(delta4), k, z0_last, z1, nullptr));
      CK(sg_lookup_product_dev(advice[0], advice[1], advice[2], instance, b8(beta), b8(gamma), n, zl, nullptr));
    }
    void* ph3[4] = {z0, z1, zl, advice[0]};
    int basis3[4] = {1, 1, 1, 0};
    CK(sg_commit_batch_mixed_dev(srs, basis3, ph3, 4, n, nullptr, out));
    t["3_grand_products_commit"] = ms_since(t1); t1 = clk::now();
    Fr y = challenge(out);
    // 4a: 9 x iNTT(2^k), 9 x coset NTT(2^(k+3))
    void* lag[9] = {advice[0], advice[1], advice[2], instance, advice[2], instance, z0, z1, zl};
    for (int i = 0; i < 9; i++) HK(hipMemcpyAsync(coeff[i], lag[i], 32 * n, hipMemcpyDeviceToDevice, nullptr));
    CK(sg_ntt_fr_batch_dev(coeff.data(), 9, omega_inv, n_inv, k, nullptr));      // 9 iNTTs: one launch per pass
    CK(sg_coeff_to_extended_batch_dev(coeff.data(), ext.data(), 9, k, ext_k, nullptr));
    HK(hipDeviceSynchronize());
    t["4a_ntts"] = ms_since(t1); t1 = clk::now();
    // 4b: evaluate_h
    HK(hipMemsetAsync(values, 0, 32 * ne, nullptr));
    void* e_adv[3] = {ext[0], ext[1], ext[2]};
    void* e_inst[1] = {ext[3]};
    CK(sg_quotient_gates_dev(values, &graph, fixed_ext.data(), 11, e_adv, 3, e_inst, 1, nullptr, 0, b8(beta), b8(gamma),
                             b8(theta), b8(y), k, ext_k, nullptr));
    void* e_z[2] = {ext[6], ext[7]};
    void* e_cols[6] = {ext[0], ext[1], ext[2], ext[3], fixed_ext[2], fixed_ext[3]};
    CK(sg_quotient_permutation_dev(values, e_z, 2, e_cols, sigma_ext.data(), 6, 4, l0, l_last, l_active, b8(beta), b8(gamma),
                                   b8(y), k, ext_k, 6, nullptr));
    CK(sg_quotient_lookup_dev(values, ext[8], ext[4], ext[5], ext[0], ext[1], l0, l_last, l_active, b8(beta), b8(gamma),
                              b8(y), k, ext_k, nullptr));
    HK(hipDeviceSynchronize());
    t["4b_evaluate_h"] = ms_since(t1); t1 = clk::now();
    // 4c: quotient polynomial and its 5 pieces
    CK(sg_divide_by_vanishing_poly_dev(values, k, ext_k, nullptr));
    CK(sg_extended_to_coeff_dev(values, k, ext_k, nullptr));
    void* pieces[5];
    for (int i = 0; i < 5; i++) pieces[i] = static_cast<uint8_t*>(values) + 32 * n * i;
    CK(sg_commit_batch_dev(srs, 0, pieces, 5, n, nullptr, out));
    t["4c_quotient_commit"] = ms_since(t1); t1 = clk::now();
    Fr x = challenge(out);
    // 5: 35 evaluations
    void* polys[35];
    std::vector<Fr> pts(35, x);
    for (int i = 0; i < 35; i++) polys[i] = i % 11 < 9 ? coeff[i % 11] : pieces[i % 11 - 9];
    uint8_t evals[35 * 32];
    CK(sg_fr_eval_poly_batch_dev(polys, n, b8(pts[0]), 35, nullptr, evals));
    t["5_evaluations"] = ms_since(t1); t1 = clk::now();
    Fr v = challenge(evals);
    // 6: multi-open: per rotation set a linear combination and one division per point, two commitments
    std::vector<Fr> vsv(10, v);
    struct Set { std::vector<void*> polys; int points; };
    std::vector<Set> sets = {{{coeff.begin(), coeff.end()}, 1},
                             {{coeff[0], coeff[1], coeff[2], coeff[6], coeff[7], coeff[8]}, 2},
                             {{coeff[6]}, 3}};
    if (multi) fork();
    for (size_t s = 0; s < sets.size(); s++) {
      hipStream_t q = (multi && s > 0) ? st[s - 1] : nullptr;   // one rotation set per stream
      void *src = tmps[2 * s], *dst = tmps[2 * s + 1];
      CK(sg_fr_lincomb_dev(sets[s].polys.data(), b8(vsv[0]), (uint32_t)sets[s].polys.size(), n, src, q));
      for (int p = 0; p < sets[s].points; p++) {
        CK(sg_fr_kate_division_dev(src, n, b8(x), p + 1 == sets[s].points ? quot[s] : dst, nullptr, q));
        std::swap(src, dst);
      }
    }
    if (multi) join();
    CK(sg_fr_lincomb_dev(quot.data(), b8(vsv[0]), 3, n, hx, nullptr));
    CK(sg_commit_dev(srs, 0, hx, n, nullptr, out));
    Fr u = challenge(out);
    void* lxs[4] = {quot[0], quot[1], quot[2], hx};
    CK(sg_fr_lincomb_dev(lxs, b8(vsv[0]), 4, n, lx, nullptr));
    CK(sg_fr_kate_division_dev(lx, n, b8(u), wq, nullptr, nullptr));
    CK(sg_commit_dev(srs, 0, wq, n, nullptr, out));
    t["6_multiopen"] = ms_since(t1);
    t["total"] = ms_since(t0);
    if (rep >= 1 && (best.empty() || t["total"] < best["total"])) best = t;  // rep 0 warms work spaces, plans, tables
  }
  std::printf("{\"k\": %u, \"n_gates\": %d, \"driver\": \"c++\", \"multi_stream\": %d", k, n_gates, multi ? 1 : 0);
  for (auto& kv : best) std::printf(", \"%s\": %.4f", kv.first.c_str(), kv.second);
  std::printf("}\n");
  sg_srs_free(srs);
  sg_shutdown();
  return 0;
}
