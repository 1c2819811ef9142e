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

// Fq6 = Fq2[v] / (v^3 - xi) as three Fq2 coefficients: the even (and, separately, the odd) powers of w of an Fq12 element,
// v = w^2.  Karatsuba: 6 Fq2 products per product.
struct Fq6 {
  Fq2 a, b, c;   // a + b v + c v^2
  Fq6 operator+(const Fq6& o) const { return Fq6{a + o.a, b + o.b, c + o.c}; }
  Fq6 operator-(const Fq6& o) const { return Fq6{a - o.a, b - o.b, c - o.c}; }
  Fq6 operator*(const Fq6& o) const {
    const Fq2 v0 = a * o.a, v1 = b * o.b, v2 = c * o.c;
    return Fq6{v0 + fq2_mul_xi((b + c) * (o.b + o.c) - v1 - v2), (a + b) * (o.a + o.b) - v0 - v1 + fq2_mul_xi(v2),
               (a + c) * (o.a + o.c) - v0 - v2 + v1};
  }
  Fq6 mul_v() const { return Fq6{fq2_mul_xi(c), a, b}; }   // times v
  Fq6 inv() const {
    const Fq2 A = a.sqr() - fq2_mul_xi(b * c), B = fq2_mul_xi(c.sqr()) - a * b, C = b.sqr() - a * c;
    const Fq2 f = (a * A + fq2_mul_xi(c * B + b * C)).inv();
    return Fq6{A * f, B * f, C * f};
  }
};

struct Fq12 {
  Fq2 c[6];   // sum c[k] w^k, w^6 = xi; as a tower: (c[0], c[2], c[4]) + (c[1], c[3], c[5]) w over Fq6, w^2 = v
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
  bool operator==(const Fq12& o) const {
    for (int i = 0; i < 6; i++)
      if (!(c[i] == o.c[i])) return false;
    return true;
  }
  Fq6 even() const { return Fq6{c[0], c[2], c[4]}; }
  Fq6 odd() const { return Fq6{c[1], c[3], c[5]}; }
  static Fq12 from_halves(const Fq6& e, const Fq6& o) {
    Fq12 r;
    r.c[0] = e.a; r.c[2] = e.b; r.c[4] = e.c;
    r.c[1] = o.a; r.c[3] = o.b; r.c[5] = o.c;
    return r;
  }
  // (e + o w)(e' + o' w) = (e e' + v o o') + ((e + o)(e' + o') - e e' - o o') w: three Fq6 products = 18 Fq2 products
  Fq12 operator*(const Fq12& x) const {
    const Fq6 e = even(), o = odd(), xe = x.even(), xo = x.odd();
    const Fq6 ee = e * xe, oo = o * xo;
    return from_halves(ee + oo.mul_v(), (e + o) * (xe + xo) - ee - oo);
  }
  // the definition, coefficient by coefficient (36 Fq2 products): what the tower forms are checked against
  Fq12 mul_schoolbook(const Fq12& o) const {
    Fq2 t[11];
    for (int i = 0; i < 11; i++) t[i] = Fq2::zero();
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) t[i + j] = t[i + j] + c[i] * o.c[j];
    Fq12 r;
    for (int i = 0; i < 6; i++) r.c[i] = i < 5 ? t[i] + fq2_mul_xi(t[i + 6]) : t[i];
    return r;
  }
  // complex squaring over Fq6: (e + o w)^2 = ((e + o)(e + v o) - eo - v eo) + 2 eo w: two Fq6 products
  Fq12 sqr() const {
    const Fq6 e = even(), o = odd(), eo = e * o;
    return from_halves((e + o) * (e + o.mul_v()) - eo - eo.mul_v(), eo + eo);
  }
  // squaring of an element of the cyclotomic subgroup (x^(q^6 + 1) = 1 and x^(q^4 - q^2 + 1) = 1: everything after the easy
  // part of the final exponentiation), Granger-Scott: nine Fq2 squarings.  With g0..g5 = c[0], c[1], .. c[5]:
  Fq12 cyclotomic_sqr() const {
    const Fq2 &x0 = c[0], &x1 = c[2], &x2 = c[4], &x3 = c[1], &x4 = c[3], &x5 = c[5];   // (C0.B0, C0.B1, C0.B2, C1.B0, C1.B1, C1.B2)
    Fq2 t0 = x4.sqr(), t1 = x0.sqr();
    const Fq2 t6 = (x4 + x0).sqr() - t0 - t1;                 // 2 x4 x0
    Fq2 t2 = x2.sqr(), t3 = x3.sqr();
    const Fq2 t7 = (x2 + x3).sqr() - t2 - t3;                 // 2 x2 x3
    Fq2 t4 = x5.sqr(), t5 = x1.sqr();
    const Fq2 t8 = fq2_mul_xi((x5 + x1).sqr() - t4 - t5);     // 2 x5 x1 xi
    t0 = fq2_mul_xi(t0) + t1;                                 // x4^2 xi + x0^2
    t2 = fq2_mul_xi(t2) + t3;                                 // x2^2 xi + x3^2
    t4 = fq2_mul_xi(t4) + t5;                                 // x5^2 xi + x1^2
    Fq12 r;
    r.c[0] = (t0 - x0).dbl() + t0;
    r.c[2] = (t2 - x1).dbl() + t2;
    r.c[4] = (t4 - x2).dbl() + t4;
    r.c[1] = (t8 + x3).dbl() + t8;
    r.c[3] = (t6 + x4).dbl() + t6;
    r.c[5] = (t7 + x5).dbl() + t7;
    return r;
  }
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
inline Fq12 fq12_inv(const Fq12& a) {  // (e - o w) / (e^2 - v o^2): one inversion in Fq6 (norm to Fq2 inside)
  const Fq6 e = a.even(), o = a.odd();
  const Fq6 d = (e * e - (o * o).mul_v()).inv();
  return Fq12::from_halves(e * d, Fq6{Fq2::zero(), Fq2::zero(), Fq2::zero()} - o * d);
}
// the same through the Frobenius orbit: a^-1 = (a^q a^(q^2) .. a^(q^11)) / Norm(a), Norm(a) in Fq (the check of the form above)
inline Fq12 fq12_inv_by_norm(const Fq12& a) {
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
// a^u, u = 4965661367192848881 (the BN254 parameter), 63 bits; `a` in the cyclotomic subgroup (the hard part's operands)
inline Fq12 fq12_pow_u(const Fq12& a) {
  constexpr uint64_t U = 0x44e992b44a6909f1ULL;
  Fq12 acc = a;
  for (int i = 61; i >= 0; i--) {
    acc = acc.cyclotomic_sqr();
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
  const Fq12 y1 = y0.cyclotomic_sqr();
  const Fq12 y2 = y1.cyclotomic_sqr();
  Fq12 y3 = y2 * y1;
  const Fq12 y4 = fq12_conj(fq12_pow_u(y3));
  const Fq12 y5 = y4.cyclotomic_sqr();
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
