// Parity test of the C++ host mirror (include/summa_gpu.hpp) against the CPU oracle, written the way the
// reference's own prover tests use halo2 (zk_prover/src/circuits/tests.rs:45-88: read params, commit,
// transform, compare).  Test infrastructure: links liboracle.so as the checker.  Needs an MI355X.
//   usage: parity_main <path to tests/golden/hermez-raw-11>
#include <cstdio>
#include <fstream>

#include "summa_gpu.hpp"

extern "C" {
int orc_best_multiexp(const uint8_t* scalars, const uint8_t* bases, size_t n, int threads, uint8_t out_affine[64]);
void orc_best_fft(uint8_t* a, const uint8_t omega[32], uint32_t log_n, int threads);
void orc_omega(uint32_t k, uint8_t out[32]);
void orc_lagrange_to_coeff(uint8_t* a, uint32_t k, int threads);
void orc_coeff_to_extended(const uint8_t* coeffs, uint32_t k, uint32_t ext_k, uint8_t* out, int threads);
void orc_extended_to_coeff(uint8_t* ext, uint32_t k, uint32_t ext_k, int threads);
void orc_divide_by_vanishing_poly(uint8_t* ext, uint32_t k, uint32_t ext_k);
void orc_random_fr(uint64_t seed, size_t n, uint8_t* out);
void orc_fixed_base_mul(const uint8_t* scalars, size_t n, int threads, uint8_t* out);
void orc_fr_eval_poly(const uint8_t* coeffs, size_t n, const uint8_t x[32], uint8_t o[32]);
}

using namespace summa;

static int failures = 0;
#define EXPECT(cond)                                            \
  do {                                                          \
    if (!(cond)) {                                              \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      failures++;                                               \
    }                                                           \
  } while (0)

static std::vector<Fr> random_fr(uint64_t seed, size_t n) {
  std::vector<Fr> v(n);
  orc_random_fr(seed, n, bytes(v.data()));
  return v;
}
static G1Affine oracle_msm(const std::vector<Fr>& s, const G1Affine* bases) {
  G1Affine out;
  orc_best_multiexp(bytes(s.data()), bytes(bases), s.size(), 4, bytes(&out));
  return out;
}

int main(int argc, char** argv) {
  if (argc < 2) {
    std::printf("usage: %s <hermez-raw-11>\n", argv[0]);
    return 2;
  }
  try {
    init(0);
    // --- params: the reference's SRS container, RawBytes
    std::ifstream f(argv[1], std::ios::binary);
    ParamsKZG params = ParamsKZG::read(f);
    EXPECT(params.k() == 11);
    {
      std::ifstream again(argv[1], std::ios::binary);
      std::string raw((std::istreambuf_iterator<char>(again)), std::istreambuf_iterator<char>());
      std::vector<uint8_t> w = params.write();
      EXPECT(w.size() == raw.size() && std::memcmp(w.data(), raw.data(), w.size()) == 0);
    }
    // --- commit / commit_lagrange vs the oracle's best_multiexp; the two bases agree on one polynomial
    EvaluationDomain dom(6, 11);
    EXPECT(dom.extended_k() == 14);
    std::vector<Fr> evals = random_fr(1, 2048);
    G1Affine c_l = params.commit_lagrange(evals);
    EXPECT(c_l == oracle_msm(evals, params.get_g_lagrange().data()));
    std::vector<Fr> coeffs = dom.lagrange_to_coeff(evals);
    EXPECT(params.commit(coeffs) == c_l);
    params.precompute();                                   // fixed-base tables: same bits
    EXPECT(params.commit_lagrange(evals) == c_l && params.commit(coeffs) == c_l);
    std::vector<Fr> shorter(evals.begin(), evals.begin() + 700);
    EXPECT(params.commit(shorter) == oracle_msm(shorter, params.get_g().data()));
    EXPECT(best_multiexp(evals, params.get_g()) == oracle_msm(evals, params.get_g().data()));
    EXPECT(best_multiexp({}, {}).is_identity());
    // --- best_fft and the EvaluationDomain methods vs the oracle
    {
      std::vector<Fr> a = random_fr(2, 1 << 12), want = a;
      Fr w;
      orc_omega(12, bytes(&w));
      orc_best_fft(bytes(want.data()), bytes(&w), 12, 4);
      best_fft(a, w, 12);
      EXPECT(a == want);
    }
    {
      std::vector<Fr> want = evals;
      orc_lagrange_to_coeff(bytes(want.data()), 11, 4);
      EXPECT(coeffs == want);
      std::vector<Fr> ext = dom.coeff_to_extended(coeffs), want_ext(dom.extended_len());
      orc_coeff_to_extended(bytes(coeffs.data()), 11, 14, bytes(want_ext.data()), 4);
      EXPECT(ext == want_ext);
      std::vector<Fr> q = dom.divide_by_vanishing_poly(ext);
      orc_divide_by_vanishing_poly(bytes(want_ext.data()), 11, 14);
      EXPECT(q == want_ext);
      std::vector<Fr> back = dom.extended_to_coeff(dom.coeff_to_extended(coeffs));
      EXPECT(back.size() == 5 * 2048);
      EXPECT(std::equal(coeffs.begin(), coeffs.end(), back.begin()));
      for (size_t i = 2048; i < back.size(); i++) EXPECT(back[i] == Fr{});
    }
    // --- setup / downsize: commit(f) == f(tau) * G on a synthetic SRS; write/read round trip
    {
      Fr tau = random_fr(3, 1)[0];
      ParamsKZG p = ParamsKZG::setup(8, tau);
      std::vector<Fr> poly = random_fr(4, 256);
      Fr at_tau;
      orc_fr_eval_poly(bytes(poly.data()), 256, bytes(&tau), bytes(&at_tau));
      G1Affine want;
      orc_fixed_base_mul(bytes(&at_tau), 1, 1, bytes(&want));
      EXPECT(p.commit(poly) == want);
      std::vector<Fr> ev = poly;                       // evaluations over the domain = NTT of the coefficients
      best_fft(ev, EvaluationDomain(2, 8).get_omega(), 8);
      EXPECT(p.commit_lagrange(ev) == want);
      std::vector<uint8_t> raw = p.write();
      ParamsKZG q = ParamsKZG::read(raw.data(), raw.size());
      EXPECT(q.get_g() == p.get_g() && q.get_g_lagrange() == p.get_g_lagrange());
      p.downsize(6);
      std::vector<Fr> small(poly.begin(), poly.begin() + 64);
      orc_fr_eval_poly(bytes(small.data()), 64, bytes(&tau), bytes(&at_tau));
      orc_fixed_base_mul(bytes(&at_tau), 1, 1, bytes(&want));
      EXPECT(p.commit(small) == want);
      std::vector<Fr> ev6 = small;                     // the recomputed g_lagrange commits to the same polynomial
      best_fft(ev6, EvaluationDomain(2, 6).get_omega(), 6);
      EXPECT(p.commit_lagrange(ev6) == want);
      bool threw = false;
      try { p.downsize(9); } catch (const std::invalid_argument&) { threw = true; }
      EXPECT(threw);
    }
    bool threw = false;
    try { best_multiexp(evals, std::vector<G1Affine>(3)); } catch (const std::invalid_argument&) { threw = true; }
    EXPECT(threw);
  } catch (const std::exception& e) {
    std::printf("exception: %s\n", e.what());
    return 1;
  }
  std::printf(failures ? "parity_main: %d failure(s)\n" : "parity_main: all checks passed\n", failures);
  return failures ? 1 : 0;
}
