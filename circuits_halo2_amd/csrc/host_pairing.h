// Host-side BN254 optimal-ate pairing check: the last step of the KZG verifier (`verify_proof` ->
// SingleStrategy -> multi_miller_loop + final_exponentiation in halo2curves; the EVM's precompile 0x08 for the
// generated Solidity verifier [REF contracts/src/InclusionVerifier.sol:185-202, 1395-1402]).  Two pairings per
// verified proof, a serial chain of a few thousand Fq multiplications: host work, as in the reference.
//
// Representation: Fq12 = Fq2[w] / (w^6 - xi), xi = 9 + u, as six Fq2 coefficients; the sextic twist
// E': y^2 = x^3 + 3/xi maps to E by (x, y) -> (x w^2, y w^3).  A line through twisted points with slope lambda,
// evaluated at P = (xP, yP) in G1, is  yP - lambda xP w + (lambda xT - yT) w^3  (sparse: w^0, w^1, w^3).
// The G2 side of a KZG check is fixed per SRS (g2, -s g2), so the slopes are computed once (`prepare`) and cached.
// Final exponentiation: easy part by Frobenius and a norm-based inversion, hard part by the three-exponentiation
// addition chain in the curve parameter u (a plain square-and-multiply over (q^4 - q^2 + 1) / r is kept as its check).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

#include "host_curve.h"

namespace sg {
namespace host {

inline Fq fq_from_u64(uint64_t v) {  // small integer -> Montgomery form (v * R mod q by repeated doubling of one())
  Fq acc = Fq::zero(), base = Fq::one();
  while (v) {
    if (v & 1) acc = acc + base;
    base = base.dbl();
    v >>= 1;
  }
  return acc;
}
inline Fq2 fq2_conj(const Fq2& a) { return Fq2{a.c0, Fq::zero() - a.c1}; }
inline Fq2 fq2_scale(const Fq2& a, const Fq& s) { return Fq2{a.c0 * s, a.c1 * s}; }
inline Fq2 fq2_mul_xi(const Fq2& a) {  // (c0 + c1 u)(9 + u) = (9 c0 - c1) + (9 c1 + c0) u
  Fq n0 = a.c0.dbl().dbl().dbl() + a.c0, n1 = a.c1.dbl().dbl().dbl() + a.c1;
  return Fq2{n0 - a.c1, n1 + a.c0};
}
inline Fq2 fq2_pow(const Fq2& a, const uint64_t* e, int words) {
  Fq2 acc = Fq2::one();
  for (int i = 64 * words - 1; i >= 0; i--) {
    acc = acc.sqr();
    if ((e[i >> 6] >> (i & 63)) & 1) acc = acc * a;
  }
  return acc;
}

struct Fq12 {
  Fq2 c[6];
  static Fq12 one() {
    Fq12 r;
    for (int i = 0; i < 6; i++) r.c[i] = Fq2::zero();
    r.c[0] = Fq2::one();
    return r;
  }
  bool is_one() const {
    if (!(c[0] == Fq2::one())) return false;
    for (int i = 1; i < 6; i++)
      if (!c[i].is_zero()) return false;
    return true;
  }
  Fq12 operator*(const Fq12& o) const {
    Fq2 t[11];
    for (int i = 0; i < 11; i++) t[i] = Fq2::zero();
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) t[i + j] = t[i + j] + c[i] * o.c[j];
    Fq12 r;
    for (int i = 0; i < 6; i++) r.c[i] = i < 5 ? t[i] + fq2_mul_xi(t[i + 6]) : t[i];
    return r;
  }
  Fq12 sqr() const { return *this * *this; }
  // multiply by the sparse line  a0 + a1 w + a3 w^3  (a0 in Fq)
  Fq12 mul_line(const Fq& a0, const Fq2& a1, const Fq2& a3) const {
    Fq2 t[9];
    for (int i = 0; i < 9; i++) t[i] = Fq2::zero();
    for (int i = 0; i < 6; i++) {
      t[i] = t[i] + fq2_scale(c[i], a0);
      t[i + 1] = t[i + 1] + c[i] * a1;
      t[i + 3] = t[i + 3] + c[i] * a3;
    }
    Fq12 r;
    for (int i = 0; i < 6; i++) r.c[i] = i < 3 ? t[i] + fq2_mul_xi(t[i + 6]) : t[i];
    return r;
  }
};

struct PairingConstants {
  Fq2 gamma[6];   // gamma[k] = xi^(k (q - 1) / 6):  (a w^k)^q = conj(a) gamma[k] w^k
  Fq2 twist_b;    // 3 / xi
  PairingConstants() {
    static constexpr uint64_t E[4] = {0x34b017592414d4e1ULL, 0xee9591c2e6bda1c2ULL, 0xf40d60f3c0403964ULL, 0x0810b7bdd032f006ULL};  // (q - 1) / 6
    Fq2 xi{fq_from_u64(9), Fq::one()};
    gamma[0] = Fq2::one();
    gamma[1] = fq2_pow(xi, E, 4);
    for (int k = 2; k < 6; k++) gamma[k] = gamma[k - 1] * gamma[1];
    twist_b = fq2_scale(xi.inv(), fq_from_u64(3));
  }
};
inline const PairingConstants& pairing_constants() {
  static const PairingConstants k;
  return k;
}
inline Fq12 frobenius(const Fq12& a) {
  const PairingConstants& k = pairing_constants();
  Fq12 r;
  for (int i = 0; i < 6; i++) r.c[i] = fq2_conj(a.c[i]) * k.gamma[i];
  return r;
}
inline Fq12 fq12_inv(const Fq12& a) {  // a^-1 = (a^q a^(q^2) .. a^(q^11)) / Norm(a), Norm(a) in Fq
  Fq12 f = frobenius(a), rest = f;
  for (int i = 2; i < 12; i++) {
    f = frobenius(f);
    rest = rest * f;
  }
  Fq12 norm = rest * a;  // lies in Fq: coefficient c[0].c0
  Fq ninv = norm.c[0].c0.inv();
  Fq12 r;
  for (int i = 0; i < 6; i++) r.c[i] = fq2_scale(rest.c[i], ninv);
  return r;
}
inline Fq12 fq12_conj(const Fq12& a) {   // a^(q^6): fixes the even powers of w (Fq6), maps w -> -w
  Fq12 r = a;
  for (int i = 1; i < 6; i += 2) r.c[i] = Fq2::zero() - a.c[i];
  return r;
}
inline Fq12 fq12_pow_u(const Fq12& a) {   // a^u, u = 4965661367192848881 (the BN254 parameter), 63 bits
  constexpr uint64_t U = 0x44e992b44a6909f1ULL;
  Fq12 acc = a;
  for (int i = 61; i >= 0; i--) {
    acc = acc.sqr();
    if ((U >> i) & 1) acc = acc * a;
  }
  return acc;
}
// f^((q^12 - 1) / r * c) with the fixed factor c = 2u (6u^2 + 3u + 1), coprime to r: equal to 1 exactly when the reduced
// pairing value is.  Easy part by Frobenius and a norm-based inversion; hard part by the addition chain of
// Fuentes-Castaneda, Knapp and Rodriguez-Henriquez in three exponentiations by u (after the easy part the inverse is
// the conjugate).  ~290 Fq12 products instead of the ~1140 of a plain square-and-multiply over (q^4 - q^2 + 1) / r.
inline Fq12 final_exponentiation(const Fq12& f) {
  Fq12 f6 = f;
  for (int i = 0; i < 6; i++) f6 = frobenius(f6);
  Fq12 r = f6 * fq12_inv(f);
  r = frobenius(frobenius(r)) * r;
  const Fq12 y0 = fq12_conj(fq12_pow_u(r));            // r^-u
  const Fq12 y1 = y0.sqr();
  const Fq12 y2 = y1.sqr();
  Fq12 y3 = y2 * y1;
  const Fq12 y4 = fq12_conj(fq12_pow_u(y3));
  const Fq12 y5 = y4.sqr();
  Fq12 y6 = fq12_conj(fq12_pow_u(y5));
  y3 = fq12_conj(y3);
  y6 = fq12_conj(y6);
  const Fq12 y7 = y6 * y4;
  Fq12 y8 = y7 * y3;
  const Fq12 y9 = y8 * y1;
  const Fq12 y10 = y8 * y4;
  const Fq12 y11 = y10 * r;
  const Fq12 y13 = frobenius(y9) * y11;
  y8 = frobenius(frobenius(y8));
  const Fq12 y14 = y8 * y13;
  Fq12 y15 = fq12_conj(r) * y9;
  y15 = frobenius(frobenius(frobenius(y15)));
  return y15 * y14;
}
// the same by definition: f^((q^12 - 1) / r) with a plain square-and-multiply over the 761-bit hard part (kept as the
// cross-check of the chain above: both are 1 on the same inputs)
inline Fq12 final_exponentiation_plain(const Fq12& f) {
  Fq12 f6 = f;
  for (int i = 0; i < 6; i++) f6 = frobenius(f6);
  Fq12 g = f6 * fq12_inv(f);
  g = frobenius(frobenius(g)) * g;
  static constexpr uint64_t H[12] = {0xe81bb482ccdf42b1ULL, 0x5abf5cc4f49c36d4ULL, 0xf1154e7e1da014fdULL, 0xdcc7b44c87cdbacfULL,
                                     0xaaa441e3954bcf8aULL, 0x6b887d56d5095f23ULL, 0x79581e16f3fd90c6ULL, 0x3b1b1355d189227dULL,
                                     0x4e529a5861876f6bULL, 0x6c0eb522d5b12278ULL, 0x331ec15183177fafULL, 0x01baaa710b0759adULL};
  Fq12 acc = Fq12::one();
  for (int i = 760; i >= 0; i--) {
    acc = acc.sqr();
    if ((H[i >> 6] >> (i & 63)) & 1) acc = acc * g;
  }
  return acc;
}

struct G2AffinePt {
  Fq2 x, y;
  bool inf;
};
// per Miller-loop step: slope and (slope * xT - yT); the doubling steps and the addition steps in loop order
struct PreparedG2 {
  std::vector<Fq2> lambda, c;
  bool inf = false;
};
inline bool g2_on_curve(const G2AffinePt& p) {
  if (p.inf) return true;
  return p.y.sqr() == p.x.sqr() * p.x + pairing_constants().twist_b;
}
inline void g2_step(G2AffinePt& t, const G2AffinePt& q, bool dbl, PreparedG2& out) {
  Fq2 lam;
  if (dbl) {
    Fq2 x2 = t.x.sqr();
    lam = (x2.dbl() + x2) * t.y.dbl().inv();
  } else {
    lam = (q.y - t.y) * (q.x - t.x).inv();
  }
  out.lambda.push_back(lam);
  out.c.push_back(lam * t.x - t.y);
  Fq2 nx = lam.sqr() - t.x - (dbl ? t.x : q.x);
  Fq2 ny = lam * (t.x - nx) - t.y;
  t.x = nx;
  t.y = ny;
}
static constexpr unsigned __int128 kAteLoop = ((unsigned __int128)0x1ULL << 64) | 0x9d797039be763ba8ULL;  // 6u + 2
inline PreparedG2 prepare_g2(const G2AffinePt& q) {
  PreparedG2 out;
  if (q.inf) {
    out.inf = true;
    return out;
  }
  const PairingConstants& k = pairing_constants();
  G2AffinePt t = q;
  for (int i = 63; i >= 0; i--) {
    g2_step(t, q, true, out);
    if ((kAteLoop >> i) & 1) g2_step(t, q, false, out);
  }
  // Frobenius corrections: Q1 = pi(Q), Q2 = -pi^2(Q) in twisted coordinates:
  // (x w^2, y w^3)^q = conj(x) gamma[2] w^2, conj(y) gamma[3] w^3
  G2AffinePt q1{fq2_conj(q.x) * k.gamma[2], fq2_conj(q.y) * k.gamma[3], false};
  G2AffinePt q2{fq2_conj(q1.x) * k.gamma[2], Fq2::zero() - fq2_conj(q1.y) * k.gamma[3], false};
  g2_step(t, q1, false, out);
  g2_step(t, q2, false, out);
  return out;
}
// f *= Miller function of (P, prepared Q) interleaved over all pairs: one shared squaring per loop step
inline Fq12 multi_miller_loop(const std::vector<Affine>& ps, const std::vector<const PreparedG2*>& qs) {
  Fq12 f = Fq12::one();
  std::vector<size_t> at(ps.size(), 0);
  auto line = [&](size_t j) {
    const PreparedG2& q = *qs[j];
    size_t s = at[j]++;
    f = f.mul_line(ps[j].y, fq2_scale(q.lambda[s], Fq::zero() - ps[j].x), q.c[s]);
  };
  for (int i = 63; i >= 0; i--) {
    f = f.sqr();
    for (size_t j = 0; j < ps.size(); j++) line(j);
    if ((kAteLoop >> i) & 1)
      for (size_t j = 0; j < ps.size(); j++) line(j);
  }
  for (size_t j = 0; j < ps.size(); j++) {
    line(j);
    line(j);
  }
  return f;
}

}  // namespace host
}  // namespace sg
