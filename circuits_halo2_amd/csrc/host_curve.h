// Host-side BN254 arithmetic used only for the O(windows) tail of an MSM: the Horner
// combination of the per-window sums the GPU returns (c doublings per window) and the final
// affine normalisation.  A serial chain of ~250 point doublings plus one field inversion is
// a few tens of microseconds on a CPU core and several milliseconds on a single GPU lane,
// so this tail runs where halo2 itself runs it.  4 x 64-bit limbs, Montgomery R = 2^256.
#pragma once
#include <cstdint>
#include <cstring>

namespace sg {
namespace host {

typedef unsigned __int128 u128;

struct Fq {
  uint64_t v[4];

  static constexpr uint64_t P[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL,
                                    0x30644e72e131a029ULL};
  static constexpr uint64_t INV = 0x87d20782e4866389ULL;
  static constexpr uint64_t ONE[4] = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL,
                                      0x0e0a77c19a07df2fULL};

  static Fq zero() { return Fq{{0, 0, 0, 0}}; }
  static Fq one() { return Fq{{ONE[0], ONE[1], ONE[2], ONE[3]}}; }
  bool is_zero() const { return (v[0] | v[1] | v[2] | v[3]) == 0; }
  bool operator==(const Fq& o) const { return std::memcmp(v, o.v, 32) == 0; }

  static bool geq_p(const uint64_t a[4]) {
    for (int i = 3; i >= 0; i--) {
      if (a[i] != P[i]) return a[i] > P[i];
    }
    return true;
  }
  static void sub_p(uint64_t a[4]) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
      u128 d = (u128)a[i] - P[i] - borrow;
      a[i] = (uint64_t)d;
      borrow = (uint64_t)(d >> 64) & 1;
    }
  }
  Fq operator+(const Fq& o) const {
    Fq r;
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (u128)v[i] + o.v[i];
      r.v[i] = (uint64_t)c;
      c >>= 64;
    }
    if (geq_p(r.v)) sub_p(r.v);
    return r;
  }
  Fq operator-(const Fq& o) const {
    Fq r;
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
      u128 d = (u128)v[i] - o.v[i] - borrow;
      r.v[i] = (uint64_t)d;
      borrow = (uint64_t)(d >> 64) & 1;
    }
    if (borrow) {
      u128 c = 0;
      for (int i = 0; i < 4; i++) {
        c += (u128)r.v[i] + P[i];
        r.v[i] = (uint64_t)c;
        c >>= 64;
      }
    }
    return r;
  }
  // separated operand scanning: full 512-bit product, then word-by-word Montgomery reduction
  Fq operator*(const Fq& o) const {
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
      u128 c = 0;
      for (int j = 0; j < 4; j++) {
        c += (u128)v[i] * o.v[j] + t[i + j];
        t[i + j] = (uint64_t)c;
        c >>= 64;
      }
      t[i + 4] = (uint64_t)c;
    }
    uint64_t top = 0;
    for (int i = 0; i < 4; i++) {
      uint64_t m = t[i] * INV;
      u128 c = 0;
      for (int j = 0; j < 4; j++) {
        c += (u128)m * P[j] + t[i + j];
        t[i + j] = (uint64_t)c;
        c >>= 64;
      }
      for (int k = i + 4; k < 8 && c; k++) {
        c += t[k];
        t[k] = (uint64_t)c;
        c >>= 64;
      }
      top += (uint64_t)c;
    }
    Fq r{{t[4], t[5], t[6], t[7]}};
    if (top || geq_p(r.v)) sub_p(r.v);
    return r;
  }
  Fq sqr() const { return *this * *this; }
  Fq dbl() const { return *this + *this; }
  Fq inv_fermat() const {  // x^(p-2): 256 squarings + ~128 products, ~12 us (the definition the fast one is checked against)
    uint64_t e[4] = {P[0] - 2, P[1], P[2], P[3]};
    Fq acc = one();
    for (int i = 255; i >= 0; i--) {
      acc = acc.sqr();
      if ((e[i >> 6] >> (i & 63)) & 1) acc = acc * *this;
    }
    return acc;
  }
  // R^3 mod p: the binary inverse below works on the stored integer x~ = x R and yields (x R)^-1; one Montgomery product by
  // R^3 turns that into x^-1 R
  static constexpr uint64_t R3[4] = {0xb1cd6dafda1530dfULL, 0x62f210e6a7283db6ULL, 0xef7f0b0c0ada0afbULL, 0x20fd6e902d592544ULL};
  // Binary extended Euclid (variable time: the values inverted here are public -- commitments on their way to affine form): every
  // host tail of an MSM ends in one inversion, and at ~12 us Fermat's was as much as the rest of a fixed-base tail.  0 -> 0.
  Fq inv() const {
    if (is_zero()) return zero();
    auto is_one = [](const uint64_t a[4]) { return a[0] == 1 && (a[1] | a[2] | a[3]) == 0; };
    auto geq = [](const uint64_t a[4], const uint64_t b[4]) {
      for (int i = 3; i >= 0; i--)
        if (a[i] != b[i]) return a[i] > b[i];
      return true;
    };
    auto sub = [](uint64_t a[4], const uint64_t b[4]) {   // a -= b (a >= b)
      uint64_t borrow = 0;
      for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - b[i] - borrow;
        a[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
      }
    };
    auto halve_mod = [](uint64_t x[4]) {                  // x = x / 2 mod p, x < p
      uint64_t carry = 0;
      if (x[0] & 1) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
          c += (u128)x[i] + P[i];
          x[i] = (uint64_t)c;
          c >>= 64;
        }
        carry = (uint64_t)c;
      }
      for (int i = 0; i < 3; i++) x[i] = (x[i] >> 1) | (x[i + 1] << 63);
      x[3] = (x[3] >> 1) | (carry << 63);
    };
    auto sub_mod = [&](uint64_t a[4], const uint64_t b[4]) {   // a = a - b mod p, both < p
      if (geq(a, b)) {
        sub(a, b);
      } else {
        uint64_t t[4] = {b[0], b[1], b[2], b[3]};
        sub(t, a);                                           // b - a in (0, p)
        uint64_t r[4] = {P[0], P[1], P[2], P[3]};
        sub(r, t);
        for (int i = 0; i < 4; i++) a[i] = r[i];
      }
    };
    uint64_t u[4] = {v[0], v[1], v[2], v[3]}, w[4] = {P[0], P[1], P[2], P[3]};
    uint64_t x1[4] = {1, 0, 0, 0}, x2[4] = {0, 0, 0, 0};
    while (!is_one(u) && !is_one(w)) {
      while (!(u[0] & 1)) {
        for (int i = 0; i < 3; i++) u[i] = (u[i] >> 1) | (u[i + 1] << 63);
        u[3] >>= 1;
        halve_mod(x1);
      }
      while (!(w[0] & 1)) {
        for (int i = 0; i < 3; i++) w[i] = (w[i] >> 1) | (w[i + 1] << 63);
        w[3] >>= 1;
        halve_mod(x2);
      }
      if (geq(u, w)) {
        sub(u, w);
        sub_mod(x1, x2);
      } else {
        sub(w, u);
        sub_mod(x2, x1);
      }
    }
    const uint64_t* y = is_one(u) ? x1 : x2;
    return Fq{{y[0], y[1], y[2], y[3]}} * Fq{{R3[0], R3[1], R3[2], R3[3]}};
  }
};

struct Affine {
  Fq x, y;
};
// Jacobian (X/Z^2, Y/Z^3); identity: Z = 0
struct Jac {
  Fq x, y, z;
  static Jac identity() { return Jac{Fq::one(), Fq::one(), Fq::zero()}; }
  bool is_identity() const { return z.is_zero(); }
};

inline Jac jac_double(const Jac& p) {
  if (p.is_identity()) return p;
  Fq a = p.x.sqr(), b = p.y.sqr(), c = b.sqr();
  Fq d = ((p.x + b).sqr() - a - c).dbl();
  Fq e = a.dbl() + a, f = e.sqr();
  Jac r;
  r.x = f - d.dbl();
  r.y = e * (d - r.x) - c.dbl().dbl().dbl();
  r.z = (p.y * p.z).dbl();
  return r;
}
inline Jac jac_add(const Jac& p, const Jac& q) {
  if (p.is_identity()) return q;
  if (q.is_identity()) return p;
  Fq z1z1 = p.z.sqr(), z2z2 = q.z.sqr();
  Fq u1 = p.x * z2z2, u2 = q.x * z1z1;
  Fq s1 = p.y * q.z * z2z2, s2 = q.y * p.z * z1z1;
  if (u1 == u2) {
    if (s1 == s2) return jac_double(p);
    return Jac::identity();
  }
  Fq h = u2 - u1, hh = h.sqr(), hhh = h * hh, r = s2 - s1, v = u1 * hh;
  Jac o;
  o.x = r.sqr() - hhh - v.dbl();
  o.y = r * (v - o.x) - s1 * hhh;
  o.z = p.z * q.z * h;
  return o;
}
// XYZZ (x = X/ZZ, y = Y/ZZZ) -> Jacobian with Z = ZZZ/ZZ:  X' = X*Z^2/ZZ... use the
// relation ZZ^3 = ZZZ^2: take Z = ZZZ * ZZ^-1 is costly; instead scale to the equivalent
// Jacobian triple (X*ZZ, Y*ZZZ, ZZ): Z^2 = ZZ^2 -> x = X*ZZ/ZZ^2 = X/ZZ, Z^3 = ZZ^3 = ZZZ^2
// -> y = Y*ZZZ/ZZZ^2 = Y/ZZZ.
inline Jac jac_from_xyzz(const Fq& x, const Fq& y, const Fq& zz, const Fq& zzz) {
  if (zz.is_zero()) return Jac::identity();
  return Jac{x * zz, y * zzz, zz};
}
inline void jac_to_affine_bytes(const Jac& p, uint8_t out[64]) {
  if (p.is_identity()) {
    std::memset(out, 0, 64);
    return;
  }
  Fq zi = p.z.inv(), zi2 = zi.sqr();
  Fq ax = p.x * zi2, ay = p.y * zi2 * zi;
  std::memcpy(out, ax.v, 32);
  std::memcpy(out + 32, ay.v, 32);
}

// ---- G2 (verifier side of ParamsKZG: g2 and s_g2 = tau * g2).  Two scalar multiplications per SRS, on the
// host: Fq2 = Fq[u] / (u^2 + 1), Jacobian coordinates, a = 0 formulas as above.
struct Fq2 {
  Fq c0, c1;
  static Fq2 zero() { return Fq2{Fq::zero(), Fq::zero()}; }
  static Fq2 one() { return Fq2{Fq::one(), Fq::zero()}; }
  bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  bool operator==(const Fq2& o) const { return c0 == o.c0 && c1 == o.c1; }
  Fq2 operator+(const Fq2& o) const { return Fq2{c0 + o.c0, c1 + o.c1}; }
  Fq2 operator-(const Fq2& o) const { return Fq2{c0 - o.c0, c1 - o.c1}; }
  Fq2 operator*(const Fq2& o) const {  // Karatsuba, u^2 = -1
    Fq a = c0 * o.c0, b = c1 * o.c1, m = (c0 + c1) * (o.c0 + o.c1);
    return Fq2{a - b, m - a - b};
  }
  Fq2 sqr() const { return Fq2{(c0 + c1) * (c0 - c1), (c0 * c1).dbl()}; }
  Fq2 dbl() const { return *this + *this; }
  Fq2 inv() const {  // (c0 - c1 u) / (c0^2 + c1^2)
    Fq n = (c0.sqr() + c1.sqr()).inv();
    return Fq2{c0 * n, Fq::zero() - c1 * n};
  }
};
struct G2Jac {
  Fq2 x, y, z;
  static G2Jac identity() { return G2Jac{Fq2::one(), Fq2::one(), Fq2::zero()}; }
  bool is_identity() const { return z.is_zero(); }
};
inline G2Jac g2_double(const G2Jac& p) {
  if (p.is_identity()) return p;
  Fq2 a = p.x.sqr(), b = p.y.sqr(), c = b.sqr();
  Fq2 d = ((p.x + b).sqr() - a - c).dbl();
  Fq2 e = a.dbl() + a, f = e.sqr();
  G2Jac r;
  r.x = f - d.dbl();
  r.y = e * (d - r.x) - c.dbl().dbl().dbl();
  r.z = (p.y * p.z).dbl();
  return r;
}
inline G2Jac g2_add(const G2Jac& p, const G2Jac& q) {
  if (p.is_identity()) return q;
  if (q.is_identity()) return p;
  Fq2 z1z1 = p.z.sqr(), z2z2 = q.z.sqr();
  Fq2 u1 = p.x * z2z2, u2 = q.x * z1z1;
  Fq2 s1 = p.y * q.z * z2z2, s2 = q.y * p.z * z1z1;
  if (u1 == u2) return s1 == s2 ? g2_double(p) : G2Jac::identity();
  Fq2 h = u2 - u1, hh = h.sqr(), hhh = h * hh, r = s2 - s1, v = u1 * hh;
  G2Jac o;
  o.x = r.sqr() - hhh - v.dbl();
  o.y = r * (v - o.x) - s1 * hhh;
  o.z = p.z * q.z * h;
  return o;
}
// the standard BN254 G2 generator (the g2 of every halo2 ParamsKZG, e.g. the reference's hermez-raw-11),
// Montgomery form
inline G2Jac g2_generator() {
  static constexpr uint64_t X0[4] = {0x8e83b5d102bc2026ULL, 0xdceb1935497b0172ULL, 0xfbb8264797811adfULL, 0x19573841af96503bULL};
  static constexpr uint64_t X1[4] = {0xafb4737da84c6140ULL, 0x6043dd5a5802d8c4ULL, 0x09e950fc52a02f86ULL, 0x14fef0833aea7b6bULL};
  static constexpr uint64_t Y0[4] = {0x619dfa9d886be9f6ULL, 0xfe7fd297f59e9b78ULL, 0xff9e1a62231b7dfeULL, 0x28fd7eebae9e4206ULL};
  static constexpr uint64_t Y1[4] = {0x64095b56c71856eeULL, 0xdc57f922327d3cbbULL, 0x55f935be33351076ULL, 0x0da4a0e693fd6482ULL};
  G2Jac g;
  std::memcpy(g.x.c0.v, X0, 32); std::memcpy(g.x.c1.v, X1, 32);
  std::memcpy(g.y.c0.v, Y0, 32); std::memcpy(g.y.c1.v, Y1, 32);
  g.z = Fq2::one();
  return g;
}
// Montgomery Fr scalar (32 B) -> canonical integer words: one Montgomery product with 1
inline void fr_from_montgomery(const uint8_t in[32], uint64_t out[4]) {
  static constexpr uint64_t R[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
  static constexpr uint64_t RINV = 0xc2e1f593efffffffULL;
  uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  std::memcpy(t, in, 32);
  uint64_t top = 0;
  for (int i = 0; i < 4; i++) {
    uint64_t m = t[i] * RINV;
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)m * R[j] + t[i + j];
      t[i + j] = (uint64_t)c;
      c >>= 64;
    }
    for (int k = i + 4; k < 8 && c; k++) {
      c += t[k];
      t[k] = (uint64_t)c;
      c >>= 64;
    }
    top += (uint64_t)c;
  }
  (void)top;  // t[4..8) < 2r after the reduction of a value < r * 2^256
  uint64_t r[4] = {t[4], t[5], t[6], t[7]};
  bool ge = true;
  for (int i = 3; i >= 0; i--) {
    if (r[i] != R[i]) { ge = r[i] > R[i]; break; }
  }
  if (ge) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
      u128 d = (u128)r[i] - R[i] - borrow;
      r[i] = (uint64_t)d;
      borrow = (uint64_t)(d >> 64) & 1;
    }
  }
  std::memcpy(out, r, 32);
}
// out = scalar * G2 generator as halo2curves G2Affine bytes (x.c0 || x.c1 || y.c0 || y.c1, Montgomery);
// identity = 128 zero bytes
inline void g2_generator_mul(const uint8_t scalar_mont[32], uint8_t out[128]) {
  uint64_t k[4];
  fr_from_montgomery(scalar_mont, k);
  const G2Jac g = g2_generator();
  G2Jac acc = G2Jac::identity();
  for (int bit = 255; bit >= 0; bit--) {
    acc = g2_double(acc);
    if ((k[bit >> 6] >> (bit & 63)) & 1) acc = g2_add(acc, g);
  }
  if (acc.is_identity()) {
    std::memset(out, 0, 128);
    return;
  }
  Fq2 zi = acc.z.inv(), zi2 = zi.sqr();
  Fq2 ax = acc.x * zi2, ay = acc.y * zi2 * zi;
  std::memcpy(out, ax.c0.v, 32); std::memcpy(out + 32, ax.c1.v, 32);
  std::memcpy(out + 64, ay.c0.v, 32); std::memcpy(out + 96, ay.c1.v, 32);
}

}  // namespace host
}  // namespace sg
