// BN254 G1 (y^2 = x^3 + 3) point arithmetic for gfx950.
//
// Memory formats follow halo2curves: affine = 64 B (x || y, Montgomery
// Fq, identity = 64 zero bytes).  Buckets use extended Jacobian "XYZZ" coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; identity: ZZ = 0): a bucket += affine point costs
// 8M + 2S and bucket + bucket 12M + 2S.  Every exceptional case (either operand the
// identity, P = Q, P = -Q) is handled exactly -- KZG SRS points are distinct, but test data
// and selector-like scalar vectors are not required to be.
#pragma once
#include "bn254_field.cuh"

namespace sg {

struct alignas(16) g1_affine {
  fp_t x, y;
};
struct alignas(16) g1_xyzz {
  fp_t x, y, zz, zzz;
};

__device__ __forceinline__ bool affine_is_identity(const g1_affine& p) { return fp_is_zero(p.x) && fp_is_zero(p.y); }
__device__ __forceinline__ bool xyzz_is_identity(const g1_xyzz& p) { return fp_is_zero(p.zz); }

__device__ __forceinline__ g1_xyzz xyzz_identity() {
  g1_xyzz r;
  r.x = fp_zero<FqP>();
  r.y = fp_zero<FqP>();
  r.zz = fp_zero<FqP>();
  r.zzz = fp_zero<FqP>();
  return r;
}
__device__ __forceinline__ g1_xyzz xyzz_from_affine(const g1_affine& p) {
  g1_xyzz r;
  if (affine_is_identity(p)) return xyzz_identity();
  r.x = p.x;
  r.y = p.y;
  r.zz = fp_one<FqP>();
  r.zzz = fp_one<FqP>();
  return r;
}

__device__ __forceinline__ g1_affine affine_load(const g1_affine* p) {
  g1_affine r;
  r.x = fp_load(&p->x);
  r.y = fp_load(&p->y);
  return r;
}
__device__ __forceinline__ g1_xyzz xyzz_load(const g1_xyzz* p) {
  g1_xyzz r;
  r.x = fp_load(&p->x);
  r.y = fp_load(&p->y);
  r.zz = fp_load(&p->zz);
  r.zzz = fp_load(&p->zzz);
  return r;
}
__device__ __forceinline__ void xyzz_store(g1_xyzz* p, const g1_xyzz& v) {
  fp_store(&p->x, v.x);
  fp_store(&p->y, v.y);
  fp_store(&p->zz, v.zz);
  fp_store(&p->zzz, v.zzz);
}

// 2 * (affine point), a = 0.  (mdbl-2008-s-1)
__device__ __forceinline__ g1_xyzz xyzz_double_affine(const g1_affine& p) {
  if (affine_is_identity(p) || fp_is_zero(p.y)) return xyzz_identity();
  g1_xyzz r;
  fp_t u = fp_dbl<FqP>(p.y);
  fp_t v = fp_sqr<FqP>(u);
  fp_t w = fp_mul<FqP>(u, v);
  fp_t s = fp_mul<FqP>(p.x, v);
  fp_t xx = fp_sqr<FqP>(p.x);
  fp_t m = fp_add<FqP>(fp_dbl<FqP>(xx), xx);
  r.x = fp_sub<FqP>(fp_sqr<FqP>(m), fp_dbl<FqP>(s));
  r.y = fp_sub<FqP>(fp_mul<FqP>(m, fp_sub<FqP>(s, r.x)), fp_mul<FqP>(w, p.y));
  r.zz = v;
  r.zzz = w;
  return r;
}
// 2 * P  (dbl-2008-s-1)
__device__ __forceinline__ g1_xyzz xyzz_double(const g1_xyzz& p) {
  if (xyzz_is_identity(p) || fp_is_zero(p.y)) return xyzz_identity();
  g1_xyzz r;
  fp_t u = fp_dbl<FqP>(p.y);
  fp_t v = fp_sqr<FqP>(u);
  fp_t w = fp_mul<FqP>(u, v);
  fp_t s = fp_mul<FqP>(p.x, v);
  fp_t xx = fp_sqr<FqP>(p.x);
  fp_t m = fp_add<FqP>(fp_dbl<FqP>(xx), xx);
  r.x = fp_sub<FqP>(fp_sqr<FqP>(m), fp_dbl<FqP>(s));
  r.y = fp_sub<FqP>(fp_mul<FqP>(m, fp_sub<FqP>(s, r.x)), fp_mul<FqP>(w, p.y));
  r.zz = fp_mul<FqP>(v, p.zz);
  r.zzz = fp_mul<FqP>(w, p.zzz);
  return r;
}

// acc += q   (q affine; madd-2008-s)
__device__ __forceinline__ void xyzz_madd(g1_xyzz& acc, const g1_affine& q) {
  if (affine_is_identity(q)) return;
  if (xyzz_is_identity(acc)) {
    acc.x = q.x;
    acc.y = q.y;
    acc.zz = fp_one<FqP>();
    acc.zzz = fp_one<FqP>();
    return;
  }
  fp_t u2 = fp_mul<FqP>(q.x, acc.zz);
  fp_t s2 = fp_mul<FqP>(q.y, acc.zzz);
  fp_t p = fp_sub<FqP>(u2, acc.x);
  fp_t r = fp_sub<FqP>(s2, acc.y);
  if (fp_is_zero(p)) {
    if (fp_is_zero(r)) acc = xyzz_double_affine(q);
    else acc = xyzz_identity();
    return;
  }
  fp_t pp = fp_sqr<FqP>(p);
  fp_t ppp = fp_mul<FqP>(p, pp);
  fp_t qq = fp_mul<FqP>(acc.x, pp);
  fp_t x3 = fp_sub<FqP>(fp_sub<FqP>(fp_sqr<FqP>(r), ppp), fp_dbl<FqP>(qq));
  fp_t y3 = fp_sub<FqP>(fp_mul<FqP>(r, fp_sub<FqP>(qq, x3)), fp_mul<FqP>(acc.y, ppp));
  acc.zz = fp_mul<FqP>(acc.zz, pp);
  acc.zzz = fp_mul<FqP>(acc.zzz, ppp);
  acc.x = x3;
  acc.y = y3;
}

// acc += q   (both XYZZ; add-2008-s)
__device__ __forceinline__ void xyzz_add(g1_xyzz& acc, const g1_xyzz& q) {
  if (xyzz_is_identity(q)) return;
  if (xyzz_is_identity(acc)) {
    acc = q;
    return;
  }
  fp_t u1 = fp_mul<FqP>(acc.x, q.zz);
  fp_t u2 = fp_mul<FqP>(q.x, acc.zz);
  fp_t s1 = fp_mul<FqP>(acc.y, q.zzz);
  fp_t s2 = fp_mul<FqP>(q.y, acc.zzz);
  fp_t p = fp_sub<FqP>(u2, u1);
  fp_t r = fp_sub<FqP>(s2, s1);
  if (fp_is_zero(p)) {
    if (fp_is_zero(r)) acc = xyzz_double(acc);
    else acc = xyzz_identity();
    return;
  }
  fp_t pp = fp_sqr<FqP>(p);
  fp_t ppp = fp_mul<FqP>(p, pp);
  fp_t qq = fp_mul<FqP>(u1, pp);
  fp_t x3 = fp_sub<FqP>(fp_sub<FqP>(fp_sqr<FqP>(r), ppp), fp_dbl<FqP>(qq));
  fp_t y3 = fp_sub<FqP>(fp_mul<FqP>(r, fp_sub<FqP>(qq, x3)), fp_mul<FqP>(s1, ppp));
  acc.zz = fp_mul<FqP>(fp_mul<FqP>(acc.zz, q.zz), pp);
  acc.zzz = fp_mul<FqP>(fp_mul<FqP>(acc.zzz, q.zzz), ppp);
  acc.x = x3;
  acc.y = y3;
}

}  // namespace sg
