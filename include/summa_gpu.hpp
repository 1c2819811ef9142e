// summa_gpu.hpp -- C++ host mirror of the halo2 interface this back-end stands behind, header-only over
// the C ABI of summa_gpu.h.  The reference's host language is Rust (no toolchain in the build image), so
// the compiled-language host side is C++: the same names, argument meaning and failure behaviour as
//   halo2_proofs::arithmetic::{best_multiexp, best_fft}
//   halo2_proofs::poly::EvaluationDomain::{new, lagrange_to_coeff, coeff_to_extended, extended_to_coeff,
//                                          divide_by_vanishing_poly, get_omega, get_omega_inv, extended_len}
//   halo2_proofs::poly::kzg::commitment::ParamsKZG::{read, write, setup, downsize, commit, commit_lagrange, k}
// as the reference reaches them from zk_prover/src/circuits/utils.rs:55-76,94-101.  Types are layout-compatible
// with halo2curves (Fr: 4 x u64 Montgomery limbs; G1Affine: x || y, identity = all zero).  Errors of the
// library surface as summa::Error (upstream panics / io::Error in the same places); there is no CPU path.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <istream>
#include <stdexcept>
#include <string>
#include <vector>

#include "summa_gpu.h"

namespace summa {

struct Fr {
  std::array<uint64_t, 4> limbs{};  // Montgomery form, little-endian limbs
  bool operator==(const Fr& o) const { return limbs == o.limbs; }
};
struct G1Affine {
  std::array<uint64_t, 4> x{}, y{};  // Montgomery Fq; (0, 0) = identity
  bool operator==(const G1Affine& o) const { return x == o.x && y == o.y; }
  bool is_identity() const { return *this == G1Affine{}; }
};
static_assert(sizeof(Fr) == 32 && sizeof(G1Affine) == 64, "layout contract with halo2curves");

class Error : public std::runtime_error {
 public:
  Error(int code, const std::string& msg) : std::runtime_error("summa_gpu error " + std::to_string(code) + ": " + msg), code_(code) {}
  int code() const { return code_; }

 private:
  int code_;
};
inline void check(int rc) {
  if (rc != SG_OK) throw Error(rc, sg_last_error() ? sg_last_error() : "");
}
inline void init(int device = 0) { check(sg_init(device)); }

inline const uint8_t* bytes(const Fr* p) { return reinterpret_cast<const uint8_t*>(p); }
inline uint8_t* bytes(Fr* p) { return reinterpret_cast<uint8_t*>(p); }
inline const uint8_t* bytes(const G1Affine* p) { return reinterpret_cast<const uint8_t*>(p); }
inline uint8_t* bytes(G1Affine* p) { return reinterpret_cast<uint8_t*>(p); }

/// halo2_proofs::arithmetic::best_multiexp(coeffs, bases) for bn256::G1Affine; result affine-normalised
inline G1Affine best_multiexp(const std::vector<Fr>& coeffs, const std::vector<G1Affine>& bases) {
  if (coeffs.size() != bases.size()) throw std::invalid_argument("best_multiexp: coeffs.len() != bases.len()");  // upstream assert_eq!
  G1Affine out;
  check(sg_msm_g1(bytes(coeffs.data()), bytes(bases.data()), coeffs.size(), bytes(&out)));
  return out;
}
/// halo2_proofs::arithmetic::best_fft(a, omega, log_n): in place, natural order in and out
inline void best_fft(std::vector<Fr>& a, const Fr& omega, uint32_t log_n) {
  if (a.size() != (size_t(1) << log_n)) throw std::invalid_argument("best_fft: a.len() != 1 << log_n");  // upstream assert_eq!
  check(sg_ntt_fr(bytes(a.data()), bytes(&omega), log_n));
}

/// halo2_proofs::poly::EvaluationDomain<Fr>
class EvaluationDomain {
 public:
  /// EvaluationDomain::new(j, k): quotient_poly_degree = j - 1, extended_k = k + ceil(log2(j - 1))
  EvaluationDomain(uint32_t j, uint32_t k) : k_(k), quotient_poly_degree_(j - 1) {
    if (j < 2) throw std::invalid_argument("degree must be at least 2");
    uint32_t e = 0;
    while ((1u << e) < quotient_poly_degree_) e++;
    extended_k_ = k + e;
    if (extended_k_ > 28) throw std::invalid_argument("extended_k exceeds the 2-adicity of the field");
  }
  uint32_t k() const { return k_; }
  uint32_t extended_k() const { return extended_k_; }
  size_t extended_len() const { return size_t(1) << extended_k_; }
  uint32_t get_quotient_poly_degree() const { return quotient_poly_degree_; }
  Fr get_omega() const { return constant(k_, 0); }
  Fr get_omega_inv() const { return constant(k_, 1); }
  Fr get_extended_omega() const { return constant(extended_k_, 0); }

  /// Lagrange -> coefficient basis (ifft with omega^-1 and 1/n)
  std::vector<Fr> lagrange_to_coeff(std::vector<Fr> a) const {
    if (a.size() != (size_t(1) << k_)) throw std::invalid_argument("lagrange_to_coeff: wrong length");
    check(sg_lagrange_to_coeff(bytes(a.data()), k_));
    return a;
  }
  /// coefficients -> evaluations over the extended coset zeta * <omega_ext>
  std::vector<Fr> coeff_to_extended(const std::vector<Fr>& a) const {
    if (a.size() != (size_t(1) << k_)) throw std::invalid_argument("coeff_to_extended: wrong length");
    std::vector<Fr> out(extended_len());
    check(sg_coeff_to_extended(bytes(a.data()), k_, extended_k_, bytes(out.data())));
    return out;
  }
  /// inverse of coeff_to_extended, truncated to n * quotient_poly_degree coefficients (as upstream)
  std::vector<Fr> extended_to_coeff(std::vector<Fr> a) const {
    if (a.size() != extended_len()) throw std::invalid_argument("extended_to_coeff: wrong length");
    check(sg_extended_to_coeff(bytes(a.data()), k_, extended_k_));
    a.resize((size_t(1) << k_) * quotient_poly_degree_);
    return a;
  }
  std::vector<Fr> divide_by_vanishing_poly(std::vector<Fr> a) const {
    if (a.size() != extended_len()) throw std::invalid_argument("divide_by_vanishing_poly: wrong length");
    check(sg_divide_by_vanishing_poly(bytes(a.data()), k_, extended_k_));
    return a;
  }

 private:
  static Fr constant(uint32_t k, int which) {
    Fr out;
    check(sg_domain_constant(k, which, bytes(&out)));
    return out;
  }
  uint32_t k_, quotient_poly_degree_, extended_k_ = 0;
};

/// halo2_proofs::poly::kzg::commitment::ParamsKZG<Bn256>: the bases stay resident in HBM (SRS cache)
class ParamsKZG {
 public:
  ParamsKZG(uint32_t k, std::vector<G1Affine> g, std::vector<G1Affine> g_lagrange, std::array<uint8_t, 128> g2 = {},
            std::array<uint8_t, 128> s_g2 = {})
      : k_(k), g_(std::move(g)), g_lagrange_(std::move(g_lagrange)), g2_(g2), s_g2_(s_g2) {
    if (g_.size() != n() || g_lagrange_.size() != n()) throw std::invalid_argument("ParamsKZG: basis length != 2^k");
  }
  ParamsKZG(const ParamsKZG&) = delete;
  ParamsKZG& operator=(const ParamsKZG&) = delete;
  ParamsKZG(ParamsKZG&& o) noexcept { *this = std::move(o); }
  ParamsKZG& operator=(ParamsKZG&& o) noexcept {
    release();
    k_ = o.k_; g_ = std::move(o.g_); g_lagrange_ = std::move(o.g_lagrange_); g2_ = o.g2_; s_g2_ = o.s_g2_;
    handle_ = o.handle_; has_handle_ = o.has_handle_;
    o.has_handle_ = false;
    return *this;
  }
  ~ParamsKZG() { release(); }

  /// ParamsKZG::read (SerdeFormat::RawBytes): k:u32 LE || g[2^k] || g_lagrange[2^k] || g2 || s_g2
  static ParamsKZG read(std::istream& in) {
    std::string raw((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    return read(reinterpret_cast<const uint8_t*>(raw.data()), raw.size());
  }
  static ParamsKZG read(const uint8_t* raw, size_t len) {
    if (len < 4) throw std::runtime_error("Failed to read params");
    uint32_t k;
    std::memcpy(&k, raw, 4);
    const size_t n = size_t(1) << (k & 31);
    if (k > 28 || len != 4 + 2 * n * 64 + 256) throw std::runtime_error("Failed to read params");
    std::vector<G1Affine> g(n), gl(n);
    std::memcpy(g.data(), raw + 4, 64 * n);
    std::memcpy(gl.data(), raw + 4 + 64 * n, 64 * n);
    std::array<uint8_t, 128> g2, s_g2;
    std::memcpy(g2.data(), raw + 4 + 128 * n, 128);
    std::memcpy(s_g2.data(), raw + 4 + 128 * n + 128, 128);
    return ParamsKZG(k, std::move(g), std::move(gl), g2, s_g2);
  }
  std::vector<uint8_t> write() const {
    std::vector<uint8_t> out(4 + 128 * n() + 256);
    std::memcpy(out.data(), &k_, 4);
    std::memcpy(out.data() + 4, g_.data(), 64 * n());
    std::memcpy(out.data() + 4 + 64 * n(), g_lagrange_.data(), 64 * n());
    std::memcpy(out.data() + 4 + 128 * n(), g2_.data(), 128);
    std::memcpy(out.data() + 4 + 128 * n() + 128, s_g2_.data(), 128);
    return out;
  }
  /// ParamsKZG::setup(k, rng) with the secret drawn by the caller (upstream: OsRng)
  static ParamsKZG setup(uint32_t k, const Fr& tau) {
    const size_t n = size_t(1) << k;
    std::vector<G1Affine> g(n), gl(n);
    check(sg_kzg_setup(k, bytes(&tau), bytes(g.data()), bytes(gl.data())));
    const Fr one{{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}};
    std::array<uint8_t, 128> g2, s_g2;
    check(sg_g2_generator_mul(bytes(&one), g2.data()));
    check(sg_g2_generator_mul(bytes(&tau), s_g2.data()));
    return ParamsKZG(k, std::move(g), std::move(gl), g2, s_g2);
  }
  /// ParamsKZG::downsize(k): keep g[0..2^k), recompute g_lagrange by an inverse FFT over G1
  void downsize(uint32_t k) {
    if (k > k_) throw std::invalid_argument("k is too large for the given params");  // utils.rs:58-60
    if (k == k_) return;
    release();
    g_.resize(size_t(1) << k);
    std::vector<G1Affine> gl(size_t(1) << k);
    check(sg_g1_to_lagrange(bytes(g_.data()), k, bytes(gl.data())));
    g_lagrange_ = std::move(gl);
    k_ = k;
  }
  uint32_t k() const { return k_; }
  size_t n() const { return size_t(1) << k_; }
  const std::vector<G1Affine>& get_g() const { return g_; }
  const std::vector<G1Affine>& get_g_lagrange() const { return g_lagrange_; }

  /// fixed-base window tables for both bases (optional; same results, faster commits)
  void precompute() {
    check(sg_srs_precompute(handle(), 0, 0));
    check(sg_srs_precompute(handle(), 1, 0));
    check(sg_srs_precompute(handle(), 2, 0));
  }
  /// commit to a polynomial in coefficient form
  G1Affine commit(const std::vector<Fr>& poly) { return commit_impl(0, poly); }
  /// commit to a polynomial in Lagrange form
  G1Affine commit_lagrange(const std::vector<Fr>& poly) { return commit_impl(1, poly); }

 private:
  uint64_t handle() {
    if (!has_handle_) {
      check(sg_srs_upload(k_, bytes(g_.data()), bytes(g_lagrange_.data()), &handle_));
      has_handle_ = true;
    }
    return handle_;
  }
  void release() {
    if (has_handle_) (void)sg_srs_free(handle_);
    has_handle_ = false;
  }
  G1Affine commit_impl(int basis, const std::vector<Fr>& poly) {
    if (poly.size() > n()) throw std::invalid_argument("polynomial longer than the SRS");
    G1Affine out;
    check(sg_commit(handle(), basis, bytes(poly.data()), poly.size(), bytes(&out)));
    return out;
  }
  uint32_t k_ = 0;
  std::vector<G1Affine> g_, g_lagrange_;
  std::array<uint8_t, 128> g2_{}, s_g2_{};
  uint64_t handle_ = 0;
  bool has_handle_ = false;
};

}  // namespace summa
