// BN254 base/scalar field arithmetic for gfx950, 8 x 32-bit limbs, Montgomery form R = 2^256.
//
// Byte-compatible with halo2curves' `Fr`/`Fq` (4 x u64 little-endian limbs in Montgomery
// form): 8 x u32 LE limbs are the same 32 bytes, so `&[Fr]` buffers are
// consumed with zero conversion.  Values are kept fully reduced (< p) at every function
// boundary so results are bit-identical to the CPU prover's.
//
// The 32x32->64 multiply-add (`v_mad_u64_u32`) is the workhorse; the modulus and its
// Montgomery constants are compile-time literals (no VGPRs, no constant loads).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sg {

struct alignas(16) fp_t {
  uint32_t l[8];
};

// SURVEY.md §8 "Montgomery constants" table, re-derived in 32-bit limbs.
struct FqP {
  static constexpr uint32_t p[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                    0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr uint32_t r1[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                     0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr uint32_t r2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                     0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
  static constexpr uint32_t inv = 0xe4866389u;  // -p^-1 mod 2^32
};
struct FrP {
  static constexpr uint32_t p[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                    0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr uint32_t r1[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                     0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr uint32_t r2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                     0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
  static constexpr uint32_t inv = 0xefffffffu;
};

template <class P>
__device__ __forceinline__ fp_t fp_zero() {
  fp_t r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = 0;
  return r;
}
template <class P>
__device__ __forceinline__ fp_t fp_one() {
  fp_t r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = P::r1[i];
  return r;
}
__device__ __forceinline__ bool fp_is_zero(const fp_t& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.l[i];
  return o == 0;
}
__device__ __forceinline__ bool fp_eq(const fp_t& a, const fp_t& b) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.l[i] ^ b.l[i];
  return o == 0;
}

// r = a - p if a >= p else a   (a < 2p)
template <class P>
__device__ __forceinline__ fp_t fp_reduce_once(const fp_t& a) {
  uint32_t s[8];
  uint32_t br = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    uint64_t d = (uint64_t)a.l[j] - P::p[j] - br;
    s[j] = (uint32_t)d;
    br = (uint32_t)(d >> 63);
  }
  fp_t r;
#pragma unroll
  for (int j = 0; j < 8; j++) r.l[j] = br ? a.l[j] : s[j];
  return r;
}

template <class P>
__device__ __forceinline__ fp_t fp_add(const fp_t& a, const fp_t& b) {
  fp_t t;
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    uint64_t s = (uint64_t)a.l[j] + b.l[j] + c;
    t.l[j] = (uint32_t)s;
    c = (uint32_t)(s >> 32);
  }
  // p < 2^254: the sum of two reduced values never carries out of 256 bits
  return fp_reduce_once<P>(t);
}

template <class P>
__device__ __forceinline__ fp_t fp_sub(const fp_t& a, const fp_t& b) {
  uint32_t t[8];
  uint32_t br = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    uint64_t d = (uint64_t)a.l[j] - b.l[j] - br;
    t[j] = (uint32_t)d;
    br = (uint32_t)(d >> 63);
  }
  fp_t r;
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    uint64_t s = (uint64_t)t[j] + (br ? P::p[j] : 0u) + c;
    r.l[j] = (uint32_t)s;
    c = (uint32_t)(s >> 32);
  }
  return r;
}

template <class P>
__device__ __forceinline__ fp_t fp_neg(const fp_t& a) {
  return fp_is_zero(a) ? a : fp_sub<P>(fp_zero<P>(), a);
}
template <class P>
__device__ __forceinline__ fp_t fp_dbl(const fp_t& a) {
  return fp_add<P>(a, a);
}

// Montgomery product a*b*2^-256 mod p (CIOS, 8 x 32-bit limbs).  With p < 2^254 and
// a, b < p the running value stays below 2^288, so a single extra limb t[8] suffices and
// the result before the final conditional subtraction is < 2p.
template <class P>
__device__ __forceinline__ fp_t fp_mul(const fp_t& a, const fp_t& b) {
  uint32_t t[9];
#pragma unroll
  for (int k = 0; k < 9; k++) t[k] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      c = (uint64_t)a.l[j] * b.l[i] + t[j] + c;
      t[j] = (uint32_t)c;
      c >>= 32;
    }
    c += t[8];
    t[8] = (uint32_t)c;
    uint32_t m = t[0] * P::inv;
    c = (uint64_t)m * P::p[0] + t[0];
    c >>= 32;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      c = (uint64_t)m * P::p[j] + t[j] + c;
      t[j - 1] = (uint32_t)c;
      c >>= 32;
    }
    c += t[8];
    t[7] = (uint32_t)c;
    t[8] = (uint32_t)(c >> 32);
  }
  fp_t r;
#pragma unroll
  for (int j = 0; j < 8; j++) r.l[j] = t[j];
  return fp_reduce_once<P>(r);
}
template <class P>
__device__ __forceinline__ fp_t fp_sqr(const fp_t& a) {
  return fp_mul<P>(a, a);
}
template <class P>
__device__ __forceinline__ fp_t fp_from_mont(const fp_t& a) {
  fp_t one = fp_zero<P>();
  one.l[0] = 1;
  return fp_mul<P>(a, one);
}
template <class P>
__device__ __forceinline__ fp_t fp_to_mont(const fp_t& a) {
  fp_t r2;
#pragma unroll
  for (int i = 0; i < 8; i++) r2.l[i] = P::r2[i];
  return fp_mul<P>(a, r2);
}
// x^(p-2); ~380 products, used only off the hot path (normalisation of a handful of points)
template <class P>
__device__ inline fp_t fp_inv(const fp_t& x) {
  fp_t acc = fp_one<P>();
  for (int i = 255; i >= 0; i--) {
    acc = fp_sqr<P>(acc);
    uint32_t e = P::p[i >> 5];
    if ((i >> 5) == 0) e -= 2;  // p[0] >= 2 for both fields
    if ((e >> (i & 31)) & 1) acc = fp_mul<P>(acc, x);
  }
  return acc;
}
template <class P>
__device__ inline fp_t fp_pow_u64(fp_t x, uint64_t e) {
  fp_t acc = fp_one<P>();
  while (e) {
    if (e & 1) acc = fp_mul<P>(acc, x);
    x = fp_sqr<P>(x);
    e >>= 1;
  }
  return acc;
}

// 32-byte element <-> two 16-byte vector accesses
__device__ __forceinline__ fp_t fp_load(const fp_t* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 lo = q[0], hi = q[1];
  fp_t r;
  r.l[0] = lo.x; r.l[1] = lo.y; r.l[2] = lo.z; r.l[3] = lo.w;
  r.l[4] = hi.x; r.l[5] = hi.y; r.l[6] = hi.z; r.l[7] = hi.w;
  return r;
}
__device__ __forceinline__ void fp_store(fp_t* p, const fp_t& v) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

}  // namespace sg
