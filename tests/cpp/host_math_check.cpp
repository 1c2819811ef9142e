// Host-only exercise of the product's host arithmetic for the sanitizer leg (tests/test_sanitizers_cpu.py builds this
// with -fsanitize=address,undefined): csrc/host_curve.h (Fq, Fq2, G1 / G2 group laws), csrc/host_pairing.h (Miller loop,
// both final exponentiations), and the host parts of include/summa_prover.hpp / summa_circuit.hpp (Fr, Keccak-256,
// Blake2b, both transcripts, the lookup permutation, the floor plan and the pinned verifying-key text).  Every check is
// self-contained (algebraic identities, published test vectors); the program prints "host math ok" and exits 0.
//   build: g++ -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -Icircuits_halo2_amd/csrc tests/cpp/host_math_check.cpp -L/opt/rocm/lib -lamdhip64
#include <cstdio>
#include <cstdlib>
#include <string>

#include "summa_circuit.hpp"
#include "host_pairing.h"

using namespace sg::host;
using summa::prover::Fr;

static int failures = 0;
#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) {                                                         \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      failures++;                                                          \
    }                                                                      \
  } while (0)

static uint64_t rng_state = 0x9e3779b97f4a7c15ULL;
static uint64_t rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return rng_state;
}
static Fq random_fq() {
  Fq a{{rnd(), rnd(), rnd(), rnd() >> 3}};
  return a * Fq::one();   // any 253-bit word value times R mod q: a field element in Montgomery form
}
static Jac g1_mul(uint64_t k) {
  const Jac g{Fq::one(), Fq::one() + Fq::one(), Fq::one()};
  Jac acc = Jac::identity();
  for (int bit = 63; bit >= 0; bit--) {
    acc = jac_double(acc);
    if ((k >> bit) & 1) acc = jac_add(acc, g);
  }
  return acc;
}
static Affine g1_affine(const Jac& p) {
  uint8_t b[64];
  jac_to_affine_bytes(p, b);
  Affine a;
  std::memcpy(a.x.v, b, 32);
  std::memcpy(a.y.v, b + 32, 32);
  return a;
}
static G2AffinePt g2_mul(uint64_t k) {
  const Fr s = Fr::from_u64(k);
  uint8_t b[128];
  g2_generator_mul(s.bytes(), b);
  G2AffinePt q;
  std::memcpy(q.x.c0.v, b, 32); std::memcpy(q.x.c1.v, b + 32, 32);
  std::memcpy(q.y.c0.v, b + 64, 32); std::memcpy(q.y.c1.v, b + 96, 32);
  q.inf = q.x.is_zero() && q.y.is_zero();
  return q;
}
static std::string hex(const uint8_t* p, size_t n) {
  static const char* d = "0123456789abcdef";
  std::string s;
  for (size_t i = 0; i < n; i++) { s += d[p[i] >> 4]; s += d[p[i] & 15]; }
  return s;
}

int main() {
  // ---- Fq / Fq2
  for (int i = 0; i < 50; i++) {
    const Fq a = random_fq(), b = random_fq(), c = random_fq();
    CHECK((a + b) * c == a * c + b * c);
    CHECK(a.is_zero() || a * a.inv() == Fq::one());
    CHECK(a.inv() == a.inv_fermat());                               // the binary inverse against its definition x^(p-2)
    CHECK(a - a == Fq::zero() && a.dbl() == a + a && a.sqr() == a * a);
    const Fq2 x{a, b}, y{c, a};
    CHECK(x * y == y * x && x.sqr() == x * x);
    CHECK(x.is_zero() || x * x.inv() == Fq2::one());
  }
  {   // edge values of the binary inverse: 0 -> 0, 1, 2, p - 1, p - 2, (p + 1) / 2 and 2000 more random values
    CHECK(Fq::zero().inv() == Fq::zero() && Fq::one().inv() == Fq::one());
    const Fq two = Fq::one() + Fq::one(), m1 = Fq::zero() - Fq::one(), m2 = m1 - Fq::one();
    for (const Fq& e : {two, m1, m2, two.inv_fermat()}) CHECK(e.inv() == e.inv_fermat() && e * e.inv() == Fq::one());
    for (int i = 0; i < 2000; i++) {
      Fq a = random_fq();
      if (i % 7 == 0) a.v[3] = a.v[2] = 0;                          // short values
      if (i % 11 == 0) a.v[0] &= ~(uint64_t)0xffff;                 // many trailing zero bits
      if (Fq::geq_p(a.v)) Fq::sub_p(a.v);
      CHECK(a.inv() == a.inv_fermat());
    }
  }
  // ---- G1: (a + b) G = a G + b G, doubling and cancellation paths
  for (int i = 0; i < 6; i++) {
    const uint64_t a = rnd() >> 2, b = rnd() >> 2;
    uint8_t s[64], t[64];
    jac_to_affine_bytes(jac_add(g1_mul(a), g1_mul(b)), s);
    jac_to_affine_bytes(g1_mul(a + b), t);
    CHECK(!std::memcmp(s, t, 64));
    jac_to_affine_bytes(jac_add(g1_mul(a), g1_mul(a)), s);
    jac_to_affine_bytes(g1_mul(2 * a), t);
    CHECK(!std::memcmp(s, t, 64));
    Jac neg = g1_mul(a);
    neg.y = Fq::zero() - neg.y;
    CHECK(jac_add(g1_mul(a), neg).is_identity());
  }
  // ---- G2 and the pairing: e(a G1, b G2) e(-(ab) G1, G2) = 1, and not for a wrong product
  {
    const uint64_t a = 0x1234567, b = 0x89abcd;
    const G2AffinePt qb = g2_mul(b), q1 = g2_mul(1), q_sum = g2_mul(b + 1);
    CHECK(g2_on_curve(qb) && g2_on_curve(q1) && !qb.inf);
    {   // (b + 1) G2 = b G2 + G2 through the Jacobian law
      G2Jac j = g2_add(G2Jac{qb.x, qb.y, Fq2::one()}, G2Jac{q1.x, q1.y, Fq2::one()});
      const Fq2 zi = j.z.inv(), zi2 = zi.sqr();
      CHECK(j.x * zi2 == q_sum.x && j.y * zi2 * zi == q_sum.y);
    }
    const PreparedG2 pb = prepare_g2(qb), p1 = prepare_g2(q1);
    Affine pa = g1_affine(g1_mul(a));
    Affine pab = g1_affine(g1_mul(a * b));
    pab.y = Fq::zero() - pab.y;
    Fq12 ml = multi_miller_loop({pa, pab}, {&pb, &p1});
    CHECK(final_exponentiation(ml).is_one());
    CHECK(final_exponentiation_plain(ml).is_one());
    Affine wrong = g1_affine(g1_mul(a * b + 1));
    wrong.y = Fq::zero() - wrong.y;
    ml = multi_miller_loop({pa, wrong}, {&pb, &p1});
    CHECK(!final_exponentiation(ml).is_one());
    {
      const Fq12 fast = final_exponentiation(ml), plain = final_exponentiation_plain(ml);
      // the two differ by a fixed power (the addition chain computes a multiple of the exponent): both are 1 or neither is
      CHECK(fast.is_one() == plain.is_one());
    }
  }
  // ---- Fq12 as a tower over its flat coefficients: every fast form against the coefficient-by-coefficient definition
  {
    auto random_fq12 = [&]() {
      Fq12 r;
      for (auto& c : r.c) c = Fq2{random_fq(), random_fq()};
      return r;
    };
    for (int i = 0; i < 20; i++) {
      const Fq12 a = random_fq12(), b = random_fq12();
      CHECK(a * b == a.mul_schoolbook(b));
      CHECK(a.sqr() == a.mul_schoolbook(a));
      CHECK((a * fq12_inv(a)).is_one());
      CHECK(fq12_inv(a) == fq12_inv_by_norm(a));
      CHECK(a * Fq12::one() == a && (Fq12::one() * b) == b);
      // the easy part of the final exponentiation lands in the cyclotomic subgroup, where the nine-squaring form holds
      Fq12 a6 = a;
      for (int k = 0; k < 6; k++) a6 = frobenius(a6);
      Fq12 g = a6 * fq12_inv(a);
      g = frobenius(frobenius(g)) * g;
      CHECK(g.cyclotomic_sqr() == g.sqr());
      CHECK((g * fq12_conj(g)).is_one());                   // there the inverse is the conjugate
      Fq12 p = g;                                            // g^u by cyclotomic squarings == by plain squarings
      constexpr uint64_t U = 0x44e992b44a6909f1ULL;
      for (int k = 61; k >= 0; k--) {
        p = p.sqr();
        if ((U >> k) & 1) p = p.mul_schoolbook(g);
      }
      CHECK(fq12_pow_u(g) == p);
      // a sparse line through the general product
      const Fq a0 = random_fq();
      const Fq2 a1{random_fq(), random_fq()}, a3{random_fq(), random_fq()};
      Fq12 line = Fq12::one();
      line.c[0] = Fq2{a0, Fq::zero()};
      line.c[1] = a1;
      line.c[3] = a3;
      CHECK(a.mul_line(a0, a1, a3) == a.mul_schoolbook(line));
    }
  }
  // ---- Fr
  for (int i = 0; i < 30; i++) {
    const uint64_t c[4] = {rnd(), rnd(), rnd(), rnd() >> 4};
    const Fr a = Fr::from_canonical_limbs(c);
    uint64_t back[4];
    a.to_canonical_limbs(back);
    CHECK(!std::memcmp(back, c, 32));
    CHECK(a.is_zero() || a * a.inv() == Fr::one());
    CHECK(a.pow((uint64_t)5) == a * a * a * a * a && (a - a).is_zero() && -a + a == Fr::zero());
    uint8_t be[32];
    a.to_be_bytes(be);
    CHECK(Fr::from_be_bytes_reduced(be) == a);
  }
  // ---- Keccak-256, Blake2b-512 (published vectors), the two transcripts
  {
    auto h = summa::prover::keccak256(reinterpret_cast<const uint8_t*>(""), 0);
    CHECK(hex(h.data(), 32) == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470");
    h = summa::prover::keccak256(reinterpret_cast<const uint8_t*>("abc"), 3);
    CHECK(hex(h.data(), 32) == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45");
    std::string big(200, 'a');
    h = summa::prover::keccak256(reinterpret_cast<const uint8_t*>(big.data()), big.size());   // more than one rate block
    CHECK(h[0] != 0 || h[1] != 0);
    summa::prover::Blake2b b(64, nullptr);   // unkeyed, no personalisation: RFC 7693 appendix A
    b.update(reinterpret_cast<const uint8_t*>("abc"), 3);
    uint8_t d[64];
    b.finalize(d, 64);
    CHECK(hex(d, 64) == "ba80a53f981c4d0d6a2797b69f12f6e94c212f14685ac4b74b12bb6fdbffa2d17d87c5392aab792dc252d5de4533cc9518d38aa8dbf1925ab92386edd4009923");
    summa::prover::EvmTranscript e1, e2;
    summa::prover::Blake2bTranscript t1, t2;
    uint8_t pt[64];
    jac_to_affine_bytes(g1_mul(7), pt);
    for (auto* tr : {&e1, &e2}) {
      tr->common_scalar(Fr::from_u64(9));
      tr->write_point(pt);
      tr->write_scalar(Fr::from_u64(11));
    }
    CHECK(e1.squeeze() == e2.squeeze() && e1.squeeze_again() == e2.squeeze_again() && e1.proof.size() == 96);
    for (auto* tr : {&t1, &t2}) {
      tr->common_scalar(Fr::from_u64(9));
      tr->write_point(pt);
      tr->write_scalar(Fr::from_u64(11));
    }
    CHECK(t1.squeeze() == t2.squeeze() && t1.proof.size() == 64);
  }
  // ---- the lookup permutation (halo2 permute_expression_pair) on a small table
  {
    const size_t rows = 64;
    std::vector<uint64_t> inp(4 * rows, 0), tab(4 * rows, 0), a(4 * rows), s(4 * rows);
    for (size_t i = 0; i < rows; i++) {
      tab[4 * i] = i;
      inp[4 * i] = (i * 7) % 13;
    }
    summa::prover::permute_expression_pair(inp.data(), tab.data(), rows, a.data(), s.data());
    for (size_t i = 0; i < rows; i++) {
      CHECK(i == 0 || a[4 * i] >= a[4 * (i - 1)]);
      CHECK(a[4 * i] == s[4 * i] || (i > 0 && a[4 * i] == a[4 * (i - 1)]));
    }
  }
  // ---- the floor plan and the pinned verifying-key text (summa_circuit.hpp)
  {
    using namespace summa::circuit;
    FloorPlan fp(11, 4, 2, 8);
    CHECK(fp.rows_used > 1000 && fp.rows_used < 2048 - 6 && fp.n_items > 0 && fp.n_absorbs > 0);
    const auto g = gates(2);
    CHECK(g.size() == 19);
    const summa::prover::Graph prog = gate_graph(2);
    CHECK(!prog.calculations.empty() && gate_challenge_exponents(2).size() == 12);
    std::vector<std::array<uint8_t, 64>> comms(17);
    for (size_t i = 0; i < comms.size(); i++) jac_to_affine_bytes(g1_mul(i + 2), comms[i].data());
    const auto d1 = verifying_key_digest(11, 2, comms), d2 = verifying_key_digest(11, 3, comms);
    CHECK(d1 != d2);
  }
  if (failures) {
    std::fprintf(stderr, "%d checks failed\n", failures);
    return 1;
  }
  std::printf("host math ok\n");
  return 0;
}
