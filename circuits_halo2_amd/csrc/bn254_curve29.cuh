// BN254 G1 (y^2 = x^3 + 3) point arithmetic on 9 x 29-bit limbs (see bn254_f29.cuh).
//
// Memory formats follow halo2curves: affine = 64 B (x || y as 8 x u32
// Montgomery-2^256 words, identity = 64 zero bytes).  In registers coordinates live in the
// Montgomery-2^261 domain: loading the stored words shifted left by 5 bits yields
// x^ = 32 * x~ = x * 2^261 (mod p) with bound 32 at zero cost.  Buckets use extended Jacobian
// "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; identity: all limbs of ZZ zero):
// bucket += affine costs 8M + 2S, bucket + bucket 12M + 2S.
//
// Lazy-reduction invariants of a stored/accumulated XYZZ point (value < bound * p, limbs
// normalised): X < 8, Y < 4, ZZ < 2, ZZZ < 2.  Every product below is annotated with the
// bound product of its operands (must stay <= 170, see f29_mul).  All exceptional cases
// (identity operands, P = Q, P = -Q) are decided exactly (f29_is_zero_mod_p).
#pragma once
#include "bn254_f29.cuh"

// y3 = a*b - c*d: one fused reduction (f29_mul2) or two products; measured per kernel
#ifndef SG_FUSED_Y3
#define SG_FUSED_Y3 1
#endif

namespace sg {

struct affine29 {
  f29 x, y;  // x^, y^ with bound 32 (33 after negation)
  bool inf;
};
struct xyzz29 {
  f29 x, y, zz, zzz;
};

SG_HD bool xyzz29_is_identity(const xyzz29& p) { return f29_all_zero(p.zz); }
SG_HD xyzz29 xyzz29_identity() {
  xyzz29 r;
  r.x = f29_zero();
  r.y = f29_zero();
  r.zz = f29_zero();
  r.zzz = f29_zero();
  return r;
}

// 64-byte affine point (16 LE words) -> registers
SG_HD affine29 affine29_from_words(const uint32_t w[16]) {
  affine29 r;
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) o |= w[i];
  r.inf = (o == 0);
  r.x = f29_from_words<5>(w);
  r.y = f29_from_words<5>(w + 8);
  return r;
}
// y -> -y  (32p - y, bound 33)
SG_HD void affine29_negate(affine29& q) { q.y = f29_sub<Fq29, 4>(f29_zero(), q.y); }

// 2 * (affine point), a = 0  (mdbl-2008-s-1)
SG_HD xyzz29 xyzz29_double_affine(const affine29& q) {
  typedef Fq29 P;
  if (q.inf) return xyzz29_identity();
  xyzz29 r;
  f29 x = f29_mul<P>(q.x, f29_one<P>());                       // 33*1 -> < 2
  f29 y = f29_mul<P>(q.y, f29_one<P>());                       // < 2
  f29 u = f29_dbl(y);                                          // < 4
  f29 v = f29_sqr<P>(u);                                       // 16 -> < 2
  f29 w = f29_mul<P>(u, v);                                    // 8
  f29 s = f29_mul<P>(x, v);                                    // 4
  f29 xx = f29_sqr<P>(x);                                      // 4
  f29 m = f29_add(f29_dbl(xx), xx);                            // < 6
  r.x = f29_sub<P, 1>(f29_sqr<P>(m), f29_dbl(s));              // 36 -> <2 ; - <4 + 4p  => < 6
  f29 t = f29_sub<P, 2>(s, r.x);                               // < 2 + 8 = 10
#if SG_FUSED_Y3
  r.y = f29_mul2<P>(m, t, w, f29_sub<P, 0>(f29_zero(), y));    // m t + w (2p - y): 60 + 4 -> < 2
#else
  r.y = f29_sub<P, 0>(f29_mul<P>(m, t), f29_mul<P>(w, y));     // 60, 4 ; <2 - <2 + 2p => < 4
#endif
  r.zz = v;
  r.zzz = w;
  return r;
}
// 2 * P  (dbl-2008-s-1)
SG_HD xyzz29 xyzz29_double(const xyzz29& p) {
  typedef Fq29 P;
  if (xyzz29_is_identity(p)) return p;
  xyzz29 r;
  f29 u = f29_dbl(p.y);                                        // < 8
  f29 v = f29_sqr<P>(u);                                       // 64
  f29 w = f29_mul<P>(u, v);                                    // 16
  f29 s = f29_mul<P>(p.x, v);                                  // 16
  f29 xx = f29_sqr<P>(p.x);                                    // 64
  f29 m = f29_add(f29_dbl(xx), xx);                            // < 6
  r.x = f29_sub<P, 1>(f29_sqr<P>(m), f29_dbl(s));              // < 6
  f29 t = f29_sub<P, 2>(s, r.x);                               // < 10
#if SG_FUSED_Y3
  r.y = f29_mul2<P>(m, t, w, f29_sub<P, 1>(f29_zero(), p.y));  // m t + w (4p - Y): 60 + 8 -> < 2
#else
  r.y = f29_sub<P, 0>(f29_mul<P>(m, t), f29_mul<P>(w, p.y));   // 60, 8 => < 4
#endif
  r.zz = f29_mul<P>(v, p.zz);                                  // 4
  r.zzz = f29_mul<P>(w, p.zzz);                                // 4
  return r;
}

// acc += q   (q affine; madd-2008-s)
SG_HD void xyzz29_madd(xyzz29& acc, const affine29& q) {
  typedef Fq29 P;
  if (q.inf) return;
  if (xyzz29_is_identity(acc)) {
    acc.x = f29_mul<P>(q.x, f29_one<P>());                     // 33 -> < 2
    acc.y = f29_mul<P>(q.y, f29_one<P>());
    acc.zz = f29_one<P>();
    acc.zzz = f29_one<P>();
    return;
  }
  f29 u2 = f29_mul<P>(q.x, acc.zz);                            // 33*2 = 66
  f29 s2 = f29_mul<P>(q.y, acc.zzz);                           // 66
  f29 p = f29_sub<P, 2>(u2, acc.x);                            // < 2 + 8 = 10
  f29 r = f29_sub<P, 1>(s2, acc.y);                            // < 2 + 4 = 6
  if (f29_is_zero_mod_p<P>(p)) {
    if (f29_is_zero_mod_p<P>(r)) acc = xyzz29_double_affine(q);
    else acc = xyzz29_identity();
    return;
  }
  f29 pp = f29_sqr<P>(p);                                      // 100
  f29 ppp = f29_mul<P>(p, pp);                                 // 20
  f29 qq = f29_mul<P>(acc.x, pp);                              // 16
  f29 rr = f29_sqr<P>(r);                                      // 36
  // x3 = rr - ppp - 2 qq: (<4) - (<4) + 4p => < 8.  rr, ppp, qq come out of products (limbs < 2^29 exactly), so the two
  // differences and the doubling share ONE carry step: limbs stay in [2^29, 2^29 + 2^30.6 + 2^30.5) < 2^32 on the way, and the
  // value stays above p (ppp and qq are products of bounds <= 20: below 1.12 p each), so the carried top limb is the true one
  f29 x3 = f29_carry(f29_sub_nc<P, 1>(f29_sub_nc<P, 0>(rr, ppp), f29_add_nc(qq, qq)));
  // t = qq + 8p - x3 (< 10; above 0.75 p because x3 < rr + 6p < 7.25 p) goes straight into the product t r as its first operand:
  // limbs < 2^29 + 2^30.5, only the top limb carried
  f29 t = f29_carry_top(f29_sub_nc<P, 2>(qq, x3));
#if SG_FUSED_Y3
  f29 y3 = f29_mul2<P>(t, r, f29_sub<P, 1>(f29_zero(), acc.y), ppp);  // t r + (4p - Y1) ppp: 60 + 8 -> < 2
#else
  f29 y3 = f29_sub<P, 0>(f29_mul<P>(t, r), f29_mul<P>(acc.y, ppp));   // 60, 8 => < 4
#endif
  acc.zz = f29_mul<P>(acc.zz, pp);                             // 4
  acc.zzz = f29_mul<P>(acc.zzz, ppp);                          // 4
  acc.x = x3;
  acc.y = y3;
}

// acc += q   (both XYZZ; add-2008-s)
SG_HD void xyzz29_add(xyzz29& acc, const xyzz29& q) {
  typedef Fq29 P;
  if (xyzz29_is_identity(q)) return;
  if (xyzz29_is_identity(acc)) {
    acc = q;
    return;
  }
  f29 u1 = f29_mul<P>(acc.x, q.zz);                            // 16
  f29 u2 = f29_mul<P>(q.x, acc.zz);                            // 16
  f29 s1 = f29_mul<P>(acc.y, q.zzz);                           // 8
  f29 s2 = f29_mul<P>(q.y, acc.zzz);                           // 8
  f29 p = f29_sub<P, 0>(u2, u1);                               // < 4
  f29 r = f29_sub<P, 0>(s2, s1);                               // < 4
  if (f29_is_zero_mod_p<P>(p)) {
    if (f29_is_zero_mod_p<P>(r)) acc = xyzz29_double(acc);
    else acc = xyzz29_identity();
    return;
  }
  f29 pp = f29_sqr<P>(p);                                      // 16
  f29 ppp = f29_mul<P>(p, pp);                                 // 8
  f29 qq = f29_mul<P>(u1, pp);                                 // 4
  f29 rr = f29_sqr<P>(r);                                      // 16
  // one carry step for rr - ppp - 2 qq (< 8), none but the top limb's for t = qq - x3 (< 10): see xyzz29_madd
  f29 x3 = f29_carry(f29_sub_nc<P, 1>(f29_sub_nc<P, 0>(rr, ppp), f29_add_nc(qq, qq)));
  f29 t = f29_carry_top(f29_sub_nc<P, 2>(qq, x3));
#if SG_FUSED_Y3
  f29 y3 = f29_mul2<P>(t, r, f29_sub<P, 0>(f29_zero(), s1), ppp);  // t r + (2p - S1) ppp: 40 + 4 -> < 2
#else
  f29 y3 = f29_sub<P, 0>(f29_mul<P>(t, r), f29_mul<P>(s1, ppp));   // 40, 4 => < 4
#endif
  acc.zz = f29_mul<P>(f29_mul<P>(acc.zz, q.zz), pp);           // 4, 4
  acc.zzz = f29_mul<P>(f29_mul<P>(acc.zzz, q.zzz), ppp);       // 4, 4
  acc.x = x3;
  acc.y = y3;
}

// XYZZ in the 2^261 domain -> 32 canonical LE words (X, Y, ZZ, ZZZ as 8 x u32 Montgomery-2^256
// each), the format the host tail (host_curve.h) consumes
SG_HD void xyzz29_to_words(const xyzz29& p, uint32_t w[32]) {
  typedef Fq29 P;
  f29_to_words(f29_reduce_with<P>(p.x, P::r256), w);           // x^ * 2^256 * 2^-261 = x~
  f29_to_words(f29_reduce_with<P>(p.y, P::r256), w + 8);
  f29_to_words(f29_reduce_with<P>(p.zz, P::r256), w + 16);
  f29_to_words(f29_reduce_with<P>(p.zzz, P::r256), w + 24);
}

}  // namespace sg

#if defined(__HIPCC__)
namespace sg {
// ---- memory forms (device only) ----------------------------------------------------------
struct alignas(16) g1_affine_mem {  // halo2curves G1Affine: x || y, 8 LE words each
  uint4 q[4];
};
struct alignas(16) xyzz29_mem {  // 36 limbs: X[9] Y[9] ZZ[9] ZZZ[9]
  uint4 q[9];
};
struct alignas(16) fp_words {  // one field element in the 8 x u32 memory format
  uint4 q[2];
};

__device__ __forceinline__ affine29 affine29_load(const g1_affine_mem* p) {
  uint32_t w[16];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint4 v = p->q[i];
    w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
  }
  return affine29_from_words(w);
}
__device__ __forceinline__ xyzz29 xyzz29_load(const xyzz29_mem* p) {
  uint32_t w[36];
#pragma unroll
  for (int i = 0; i < 9; i++) {
    uint4 v = p->q[i];
    w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
  }
  xyzz29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) { r.x.l[i] = w[i]; r.y.l[i] = w[9 + i]; r.zz.l[i] = w[18 + i]; r.zzz.l[i] = w[27 + i]; }
  return r;
}
__device__ __forceinline__ void xyzz29_store(xyzz29_mem* p, const xyzz29& v) {
  uint32_t w[36];
#pragma unroll
  for (int i = 0; i < 9; i++) { w[i] = v.x.l[i]; w[9 + i] = v.y.l[i]; w[18 + i] = v.zz.l[i]; w[27 + i] = v.zzz.l[i]; }
#pragma unroll
  for (int i = 0; i < 9; i++) p->q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
__device__ __forceinline__ void fp_words_load(const fp_words* p, uint32_t w[8]) {
  uint4 a = p->q[0], b = p->q[1];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
__device__ __forceinline__ void fp_words_store(fp_words* p, const uint32_t w[8]) {
  p->q[0] = make_uint4(w[0], w[1], w[2], w[3]);
  p->q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
// (x^)^-1 in the 2^261 domain, 0 -> 0: division steps instead of a^(p-2) (bn254_f29.cuh: f29_inv_safegcd)
template <class P>
__device__ inline f29 f29_inv(const f29& x) {
  return f29_inv_safegcd<P>(x);
}
template <class P>
__device__ inline f29 f29_pow_u64(f29 x, uint64_t e) {
  f29 acc = f29_one<P>();
  while (e) {
    if (e & 1) acc = f29_mul<P>(acc, x);
    x = f29_sqr<P>(x);
    e >>= 1;
  }
  return acc;
}
// memory word formats <-> register domains for one field element
//   load_r256:  x~ words -> limbs, same domain, bound 1
//   load_r261:  x~ words -> x^ canonical-ish (< 2p) via one product
template <class P>
__device__ __forceinline__ f29 f29_load_r256(const fp_words* p) {
  uint32_t w[8];
  fp_words_load(p, w);
  return f29_from_words<0>(w);
}
template <class P>
__device__ __forceinline__ f29 f29_words_to_r261(const uint32_t w[8]) {
  f29 k;
#pragma unroll
  for (int i = 0; i < 9; i++) k.l[i] = P::r266[i];
  return f29_mul<P>(f29_from_words<0>(w), k);  // x~ * 2^266 * 2^-261 = x * 2^261
}
// value (bound <= 170, any domain) -> canonical words of the same residue
template <class P>
__device__ __forceinline__ void f29_store_canonical(fp_words* p, const f29& a_lt2p_normalised) {
  uint32_t w[8];
  f29_to_words(f29_cond_sub_p<P>(a_lt2p_normalised), w);
  fp_words_store(p, w);
}
// ---- quad-cooperative XYZZ addition -----------------------------------------------------------
// The latency-bound phases of an MSM (bucket reduction, merge rounds) run a few dependent point
// additions per lane with most of the chip idle; one lane needs ~6.5 us per addition (14 dependent
// field products).  Here the 4 lanes of a quad hold IDENTICAL copies of both operands and split the
// products: 4 rounds of one product per lane instead of 14 products, results exchanged with DPP
// quad broadcasts (full-rate VALU moves, no LDS).  Same formulas and bounds as xyzz29_add
// (add-2008-s); the rare special cases (P = +-Q) fall back to the serial code in all 4 lanes.
template <int SRC>
__device__ __forceinline__ f29 quad_bcast(const f29& v) {
  f29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    int t = __builtin_amdgcn_mov_dpp((int)v.l[i], SRC * 0x55, 0xf, 0xf, true);   // every source lane is live
    // keep the broadcast a plain v_mov_b32_dpp: the compiler's DPP combine folds it into the consuming
    // VOP2 (v_subrev_u32_dpp ...), which on gfx950 returned the lane's own value (tools/test_quad.hip)
    asm volatile("" : "+v"(t));
    r.l[i] = (uint32_t)t;
  }
  return r;
}
__device__ __forceinline__ f29 quad_sel(uint32_t role, const f29& a0, const f29& a1, const f29& a2, const f29& a3) {
  f29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    uint32_t lo = (role & 1) ? a1.l[i] : a0.l[i];
    uint32_t hi = (role & 1) ? a3.l[i] : a2.l[i];
    r.l[i] = (role & 2) ? hi : lo;
  }
  return r;
}
// acc += q; every lane of the quad passes the same acc / q and receives the same result
__device__ __forceinline__ void xyzz29_add_quad(xyzz29& acc, const xyzz29& q, uint32_t role) {
  typedef Fq29 P;
  if (xyzz29_is_identity(q)) return;
  if (xyzz29_is_identity(acc)) {
    acc = q;
    return;
  }
  // round 1: U1 = X1 ZZ2 | U2 = X2 ZZ1 | S1 = Y1 ZZZ2 | S2 = Y2 ZZZ1
  f29 m = f29_mul<P>(quad_sel(role, acc.x, q.x, acc.y, q.y), quad_sel(role, q.zz, acc.zz, q.zzz, acc.zzz));
  const f29 u1 = quad_bcast<0>(m), u2 = quad_bcast<1>(m), s1 = quad_bcast<2>(m), s2 = quad_bcast<3>(m);
  const f29 p = f29_sub<P, 0>(u2, u1);                         // < 4
  const f29 r = f29_sub<P, 0>(s2, s1);                         // < 4
  if (f29_is_zero_mod_p<P>(p)) {
    if (f29_is_zero_mod_p<P>(r)) acc = xyzz29_double(acc);
    else acc = xyzz29_identity();
    return;
  }
  // round 2: PP = P^2 | RR = R^2 | ZZ12 = ZZ1 ZZ2 | ZZZ12 = ZZZ1 ZZZ2
  m = f29_mul<P>(quad_sel(role, p, r, acc.zz, acc.zzz), quad_sel(role, p, r, q.zz, q.zzz));   // 16, 16, 4, 4
  const f29 pp = quad_bcast<0>(m), rr = quad_bcast<1>(m);
  // round 3: PPP = P PP | Q = U1 PP | ZZ3 = ZZ12 PP | V = ZZZ12 P      (lanes 2, 3 reuse their own m)
  m = f29_mul<P>(quad_sel(role, p, u1, m, m), quad_sel(role, pp, pp, pp, p));                  // 8, 4, 4, 8
  const f29 ppp = quad_bcast<0>(m), qq = quad_bcast<1>(m), zz3 = quad_bcast<2>(m);
  const f29 x3 = f29_carry(f29_sub_nc<P, 1>(f29_sub_nc<P, 0>(rr, ppp), f29_add_nc(qq, qq)));   // < 8, one carry step (xyzz29_madd)
  const f29 t = f29_carry_top(f29_sub_nc<P, 2>(qq, x3));                                       // < 10, a product's first operand
  // round 4: T R | S1 PPP | (idle: repeats lane 1) | ZZZ3 = V PP
  m = f29_mul<P>(quad_sel(role, t, s1, s1, m), quad_sel(role, r, ppp, ppp, pp));               // 40, 4, 4, 4
  acc.y = f29_sub<P, 0>(quad_bcast<0>(m), quad_bcast<1>(m));                                   // < 4
  acc.zzz = quad_bcast<3>(m);
  acc.zz = zz3;
  acc.x = x3;
}
}  // namespace sg
#endif
