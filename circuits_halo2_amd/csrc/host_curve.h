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
  Fq inv() const {  // x^(p-2)
    uint64_t e[4] = {P[0] - 2, P[1], P[2], P[3]};
    Fq acc = one();
    for (int i = 255; i >= 0; i--) {
      acc = acc.sqr();
      if ((e[i >> 6] >> (i & 63)) & 1) acc = acc * *this;
    }
    return acc;
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

}  // namespace host
}  // namespace sg
