// BN254 field arithmetic in 9 x 29-bit limbs for gfx950 (Montgomery radix R' = 2^261).
//
// Why 29-bit limbs: on gfx950 a 32x32+64 multiply-add (v_mad_u64_u32, 5.1 clk/wave) costs
// about the same as ANY other integer instruction, and it has no carry-in.  With full 32-bit
// limbs every product needs extra add-with-carry / move instructions (a compiler-scheduled
// 8x32 CIOS product is 128 mads + ~420 other instructions).  With 29-bit limbs a column of
// 9 + 9 products (a_i*b_j and m*p_j, each < 2^58) fits a 64-bit accumulator with room to
// spare, so a Montgomery product is a pure chain of 162 mads into 64-bit accumulators plus
// one shift/add per column -- and additions / subtractions become 9 independent 32-bit adds
// with no carry chain ("lazy": values stay below a small multiple of p, limbs below
// 2^29 + 4; the product tolerates both).
//
// Conventions
//   f29           limbs l[0..8], value = sum l[k] * 2^(29k); "normalised" = every limb
//                 < 2^29 + 4 (l[8] carries the excess of the value over 2^232).
//   bound B       value < B * p.  f29_mul(a, b) requires Ba * Bb <= 170 and returns a value
//                 < 2p with exactly normalised limbs (< 2^29).
//   domains       memory holds x~ = x * 2^256 mod p (halo2curves).  f29_mul(a, b) = a*b*2^-261.
//                 Curve code works on x^ = x * 2^261 mod p (loading x~ shifted left by 5 bits
//                 gives x^ lazily, bound 32); NTT data stays x~ and only the twiddles are
//                 stored as w^ (then x~ * w^ * 2^-261 = (xw)~).
// The header also compiles as plain C++ (tools/test_f29.cpp checks it against big integers).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SG_HD __host__ __device__ __forceinline__
#else
#define SG_HD inline
#endif

namespace sg {

struct f29 {
  uint32_t l[9];
};
static constexpr uint32_t M29 = (1u << 29) - 1;

struct Fq29 {
  static constexpr int32_t p30[9] = {0x187cfd47, 0x3082305b, 0x071ca8d3, 0x205aa45a, 0x01585d97, 0x0116da06, 0x1a029b85, 0x139cb84c, 0x00003064};
  static constexpr uint32_t pinv30 = 0x1b799c77u;
  static constexpr uint32_t r783[9] = {0x0e2312b2u, 0x16c05ca2u, 0x0bc84389u, 0x1cdf310bu, 0x11adafddu, 0x032e568eu, 0x1d6ae48cu, 0x10d4cd1fu, 0x0026c2d2u};
  static constexpr uint32_t p[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u,
                                    0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
  static constexpr uint32_t inv = 0x04866389u;      // -p^-1 mod 2^29
  static constexpr uint32_t pinv = 0x1b799c77u;     //  p^-1 mod 2^29
  static constexpr uint32_t one[9] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u,
                                      0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};  // 2^261 mod p
  static constexpr uint32_t r256[9] = {0x058f0d9du, 0x1aea1c6eu, 0x11c2cf74u, 0x11d651ebu, 0x1462c0a7u,
                                       0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u};  // 2^256 mod p
  static constexpr uint32_t r266[9] = {0x13349ca1u, 0x1a5d84a8u, 0x0a3e5cacu, 0x100249e0u, 0x12b951e8u,
                                       0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};  // 2^266 mod p
  static constexpr uint32_t r517[9] = {0x0f6b5c04u, 0x08ead878u, 0x1645525du, 0x1aefe9cdu, 0x09d605edu,
                                       0x0483a115u, 0x0d08508bu, 0x0dba4804u, 0x001982b4u};  // 2^517 mod p
  // k*p with limbs 0..7 raised by 2^30 (borrowed from the limb above): a - b + subc never
  // goes negative limb-wise for normalised b < k*p.  Rows: k = 2, 4, 8, 16, 32, 64.
  static constexpr uint32_t subc[6][9] = {
      {0x50f9fa8eu, 0x4208c16bu, 0x58e5469cu, 0x45aa459fu, 0x4b0bb2eeu, 0x45b6817fu, 0x414dc280u, 0x5cb84c66u, 0x0060c89au},
      {0x41f3f51cu, 0x441182d9u, 0x51ca8d3au, 0x4b548b41u, 0x561765deu, 0x4b6d0300u, 0x429b8502u, 0x597098ceu, 0x00c19137u},
      {0x43e7ea38u, 0x482305b4u, 0x43951a76u, 0x56a91685u, 0x4c2ecbbeu, 0x56da0603u, 0x45370a06u, 0x52e1319eu, 0x01832271u},
      {0x47cfd470u, 0x50460b6au, 0x472a34eeu, 0x4d522d0cu, 0x585d977fu, 0x4db40c08u, 0x4a6e140fu, 0x45c2633eu, 0x030644e5u},
      {0x4f9fa8e0u, 0x408c16d6u, 0x4e5469dfu, 0x5aa45a1au, 0x50bb2f00u, 0x5b681813u, 0x54dc2820u, 0x4b84c67eu, 0x060c89ccu},
      {0x5f3f51c0u, 0x41182daeu, 0x5ca8d3c0u, 0x5548b436u, 0x41765e03u, 0x56d03029u, 0x49b85043u, 0x57098cffu, 0x0c19139au}};
};
struct Fr29 {
  // modulus in signed 30-bit limbs, p^-1 mod 2^30 and 2^783 mod p (29-bit limbs): f29_inv
  static constexpr int32_t p30[9] = {0x30000001, 0x0f87d64f, 0x1b970914, 0x0cfa121e, 0x01585d28, 0x0116da06, 0x1a029b85, 0x139cb84c, 0x00003064};
  static constexpr uint32_t pinv30 = 0x10000001u;
  static constexpr uint32_t r783[9] = {0x001fddb2u, 0x17d30b63u, 0x1a2600eeu, 0x09507c47u, 0x1496b29bu, 0x0b00a268u, 0x15b645ebu, 0x1f9fcb3du, 0x001baa96u};
  static constexpr uint32_t p[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u,
                                    0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
  static constexpr uint32_t inv = 0x0fffffffu;
  static constexpr uint32_t pinv = 0x10000001u;
  static constexpr uint32_t one[9] = {0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu,
                                      0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
  static constexpr uint32_t r256[9] = {0x0ffffffbu, 0x04b1a0e2u, 0x18334a6bu, 0x18ed2b3eu, 0x1462e36fu,
                                       0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u};
  static constexpr uint32_t r266[9] = {0x0fffead7u, 0x1d5444f4u, 0x04438aa5u, 0x03b4d096u, 0x134c84dau,
                                       0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};
  static constexpr uint32_t r517[9] = {0x142db4dfu, 0x19d6990eu, 0x1472f48cu, 0x06dbe7e3u, 0x0b84d579u,
                                       0x10f9faf7u, 0x121f4380u, 0x17a112deu, 0x001275c7u};
  // 2^(261 + 5j) mod r, j = 0..12: undoes the 2^-5 drift of j products of memory-domain values
  static constexpr uint32_t p2[13][9] = {
      {0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u},
      {0x0fffead7u, 0x1d5444f4u, 0x04438aa5u, 0x03b4d096u, 0x134c84dau, 0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u},
      {0x0ffd5addu, 0x0d5998b1u, 0x1d5ce7cau, 0x1f285fe9u, 0x08ff23b9u, 0x09c89e51u, 0x177e12abu, 0x185f3518u, 0x001ed378u},
      {0x1fab5b8cu, 0x1df999b9u, 0x0c6a22f9u, 0x08c0aa38u, 0x117004feu, 0x1ff2bb1bu, 0x02b8bc53u, 0x0cb3a707u, 0x001298f7u},
      {0x156b7174u, 0x0a771fc5u, 0x00f2ab72u, 0x1a4e7ba3u, 0x0bbac1a1u, 0x1c105a69u, 0x0f44fb72u, 0x0a231672u, 0x000e6b3fu},
      {0x1d6e2e77u, 0x1756e719u, 0x1d182771u, 0x037a5bd1u, 0x15a3cd9du, 0x08560665u, 0x02c18312u, 0x0325767bu, 0x0019e128u},
      {0x1dc5cecfu, 0x1ad26ca5u, 0x0ee684d6u, 0x0a71da12u, 0x0696b8ecu, 0x1a317feau, 0x0d1b6cf5u, 0x109045eeu, 0x00057bcdu},
      {0x08b9d9ddu, 0x1d1e8edfu, 0x11bc2de7u, 0x16c98f74u, 0x1245a600u, 0x1d9e3b04u, 0x0178faf6u, 0x06f44b31u, 0x001e4cc5u},
      {0x173b3b8cu, 0x16985f72u, 0x1852e6a9u, 0x1ce69b8cu, 0x1a404dd5u, 0x1aa65184u, 0x0215c5c7u, 0x1f566a11u, 0x0001c285u},
      {0x1767717fu, 0x13fc41b7u, 0x1c00b0e5u, 0x1502e0a4u, 0x1283e839u, 0x11eeefd9u, 0x0211d7b9u, 0x1c711beeu, 0x0007ec70u},
      {0x1cee2fdbu, 0x0439d7d9u, 0x1849671fu, 0x19493fdau, 0x04dfeaa4u, 0x0f95b76fu, 0x1ef890ecu, 0x0656bebdu, 0x000b9894u},
      {0x0dc5fb59u, 0x0dcd42e0u, 0x04a7e5adu, 0x127404b0u, 0x05549302u, 0x1eb828a0u, 0x1a81f4c7u, 0x0652cc52u, 0x00205461u},
      {0x08bf6b0bu, 0x0d5f32f8u, 0x076dbb09u, 0x0a64b20cu, 0x06981b8eu, 0x1b08c437u, 0x028e1ea7u, 0x1cca6816u, 0x001251b6u}};
  static constexpr uint32_t subc[6][9] = {
      {0x40000002u, 0x5e1f593du, 0x5cb8489fu, 0x4fa121e4u, 0x4b0ba504u, 0x45b6817fu, 0x414dc280u, 0x5cb84c66u, 0x0060c89au},
      {0x40000004u, 0x5c3eb27cu, 0x59709141u, 0x5f4243cbu, 0x56174a0au, 0x4b6d0300u, 0x429b8502u, 0x597098ceu, 0x00c19137u},
      {0x40000008u, 0x587d64fau, 0x52e12285u, 0x5e848799u, 0x4c2e9417u, 0x56da0603u, 0x45370a06u, 0x52e1319eu, 0x01832271u},
      {0x40000010u, 0x50fac9f6u, 0x45c2450du, 0x5d090f35u, 0x585d2831u, 0x4db40c08u, 0x4a6e140fu, 0x45c2633eu, 0x030644e5u},
      {0x40000020u, 0x41f593eeu, 0x4b848a1du, 0x5a121e6cu, 0x50ba5065u, 0x5b681813u, 0x54dc2820u, 0x4b84c67eu, 0x060c89ccu},
      {0x40000040u, 0x43eb27deu, 0x5709143cu, 0x54243cdau, 0x4174a0cdu, 0x56d03029u, 0x49b85043u, 0x57098cffu, 0x0c19139au}};
};

template <class P>
SG_HD f29 f29_const(const uint32_t (&c)[9]) {
  f29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = c[i];
  return r;
}
SG_HD f29 f29_zero() {
  f29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = 0;
  return r;
}
template <class P>
SG_HD f29 f29_one() {
  f29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = P::one[i];
  return r;
}
SG_HD bool f29_all_zero(const f29& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) o |= a.l[i];
  return o == 0;
}

// one parallel carry step: limbs < 2^31 in -> limbs < 2^29 + 4 out (l[0] exactly < 2^29)
SG_HD f29 f29_carry(const f29& a) {
  f29 r;
  r.l[0] = a.l[0] & M29;
#pragma unroll
  for (int i = 1; i < 8; i++) r.l[i] = (a.l[i] & M29) + (a.l[i - 1] >> 29);
  r.l[8] = a.l[8] + (a.l[7] >> 29);
  return r;
}
// exact sequential normalisation: every limb 0..7 < 2^29
SG_HD f29 f29_normalize(const f29& a) {
  f29 r;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint32_t t = a.l[i] + c;
    r.l[i] = t & M29;
    c = t >> 29;
  }
  r.l[8] = a.l[8] + c;
  return r;
}

// lazy addition: bound Ba + Bb
SG_HD f29 f29_add(const f29& a, const f29& b) {
  f29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
  return f29_carry(r);
}
// lazy subtraction a - b + K*p, K = 2 << LOGK_MINUS1 must be >= bound(b); bound Ba + K
template <class P, int KIDX>
SG_HD f29 f29_sub(const f29& a, const f29& b) {
  f29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + P::subc[KIDX][i] - b.l[i];
  return f29_carry(r);
}
SG_HD f29 f29_dbl(const f29& a) { return f29_add(a, a); }
// ---- the same sums WITHOUT their carry step, for chains of sums that end in ONE carry step and for values that go straight
// into a product as its FIRST operand.  Limbs are plain 32-bit sums here: every limb of an intermediate must stay below 2^32
// and, limb by limb, non-negative (the constants of f29_sub_nc have their limbs 0..7 raised by 2^30, so a subtrahend with
// limbs < 2^30 is fine: a normalised value, or the double of an exactly normalised one); the TOP limb may wrap below zero
// (it is short by the two units the limb below borrowed) until a carry step or f29_carry_top has added them back.
// f29_mul(a, b) takes an `a` with limbs up to 2^31 when b is exactly normalised (nine products of < 2^60 and nine of < 2^58
// per column) -- but its top limb must be the true one: f29_carry_top.
SG_HD f29 f29_add_nc(const f29& a, const f29& b) {
  f29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
  return r;
}
template <class P, int KIDX>
SG_HD f29 f29_sub_nc(const f29& a, const f29& b) {
  f29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + P::subc[KIDX][i] - b.l[i];
  return r;
}
// the carry of limb 7 into the top limb only (3 instructions instead of 24): the top limb is then what a full carry step
// would leave, limbs 0..6 stay as they are
SG_HD f29 f29_carry_top(const f29& a) {
  f29 r = a;
  r.l[8] = a.l[8] + (a.l[7] >> 29);
  r.l[7] = a.l[7] & M29;
  return r;
}

#if defined(__HIP_DEVICE_COMPILE__) && !defined(SG_F29_ROW_SCAN)
// ---- device: column scanning, one chain of v_mad_u64_u32 per column ------------------------------------------------
// Column k of a Montgomery product is (carry of column k-1) + sum a_i b_(k-i) + sum m_i p_(k-i).  v_mad_u64_u32 has a
// 64-bit addend, so the carry can enter as the addend of the column's first multiply-add and the whole column is ONE
// chain: 163 multiply-adds + 9 mul_lo + 17 and + 16 shifts = 205 VALU instructions.  Written as C++ (either row- or
// column-wise) the compiler re-associates every column into "sum of products" + carry and spends a v_lshl_add_u64 per
// column on joining them (229 instructions, and 18 live 64-bit accumulators in the row-wise form), so the chains are
// spelled in inline asm, one block per run of products; the integer result is the same, limb for limb
// (tools/microbench4.hip measures both and compares their outputs).  A block is opaque to the scheduler: the latency of a
// dependent chain is covered by the other waves of the SIMD (>= 2), which every kernel here has.
#define SG_CH_T(i) "v_mad_u64_u32 %[t], vcc, %[x" #i "], %[y" #i "], %[t]\n"
#define SG_CH_V(i) [x##i] "v"(x[i]), [y##i] "v"(y[i])
#define SG_CH_S(i) [x##i] "v"(x[i]), [y##i] "s"(y[i])
// t += sum_{i < N} x[i] * y[i]; y in VGPRs
template <int N>
__device__ __forceinline__ void f29_chain_vv(uint64_t& t, const uint32_t (&x)[N], const uint32_t (&y)[N]) {
  static_assert(N >= 1 && N <= 9, "");
  if constexpr (N == 1) asm(SG_CH_T(0) : [t] "+v"(t) : SG_CH_V(0) : "vcc");
  if constexpr (N == 2) asm(SG_CH_T(0) SG_CH_T(1) : [t] "+v"(t) : SG_CH_V(0), SG_CH_V(1) : "vcc");
  if constexpr (N == 3) asm(SG_CH_T(0) SG_CH_T(1) SG_CH_T(2) : [t] "+v"(t) : SG_CH_V(0), SG_CH_V(1), SG_CH_V(2) : "vcc");
  if constexpr (N == 4)
    asm(SG_CH_T(0) SG_CH_T(1) SG_CH_T(2) SG_CH_T(3) : [t] "+v"(t) : SG_CH_V(0), SG_CH_V(1), SG_CH_V(2), SG_CH_V(3) : "vcc");
  if constexpr (N == 5)
    asm(SG_CH_T(0) SG_CH_T(1) SG_CH_T(2) SG_CH_T(3) SG_CH_T(4)
        : [t] "+v"(t) : SG_CH_V(0), SG_CH_V(1), SG_CH_V(2), SG_CH_V(3), SG_CH_V(4) : "vcc");
  if constexpr (N == 6)
    asm(SG_CH_T(0) SG_CH_T(1) SG_CH_T(2) SG_CH_T(3) SG_CH_T(4) SG_CH_T(5)
        : [t] "+v"(t) : SG_CH_V(0), SG_CH_V(1), SG_CH_V(2), SG_CH_V(3), SG_CH_V(4), SG_CH_V(5) : "vcc");
  if constexpr (N == 7)
    asm(SG_CH_T(0) SG_CH_T(1) SG_CH_T(2) SG_CH_T(3) SG_CH_T(4) SG_CH_T(5) SG_CH_T(6)
        : [t] "+v"(t) : SG_CH_V(0), SG_CH_V(1), SG_CH_V(2), SG_CH_V(3), SG_CH_V(4), SG_CH_V(5), SG_CH_V(6) : "vcc");
  if constexpr (N == 8)
    asm(SG_CH_T(0) SG_CH_T(1) SG_CH_T(2) SG_CH_T(3) SG_CH_T(4) SG_CH_T(5) SG_CH_T(6) SG_CH_T(7)
        : [t] "+v"(t) : SG_CH_V(0), SG_CH_V(1), SG_CH_V(2), SG_CH_V(3), SG_CH_V(4), SG_CH_V(5), SG_CH_V(6), SG_CH_V(7) : "vcc");
  if constexpr (N == 9)
    asm(SG_CH_T(0) SG_CH_T(1) SG_CH_T(2) SG_CH_T(3) SG_CH_T(4) SG_CH_T(5) SG_CH_T(6) SG_CH_T(7) SG_CH_T(8)
        : [t] "+v"(t)
        : SG_CH_V(0), SG_CH_V(1), SG_CH_V(2), SG_CH_V(3), SG_CH_V(4), SG_CH_V(5), SG_CH_V(6), SG_CH_V(7), SG_CH_V(8) : "vcc");
}
// the same with y in SGPRs (limbs of the modulus: compile-time constants)
template <int N>
__device__ __forceinline__ void f29_chain_vs(uint64_t& t, const uint32_t (&x)[N], const uint32_t (&y)[N]) {
  static_assert(N >= 1 && N <= 8, "");
  if constexpr (N == 1) asm(SG_CH_T(0) : [t] "+v"(t) : SG_CH_S(0) : "vcc");
  if constexpr (N == 2) asm(SG_CH_T(0) SG_CH_T(1) : [t] "+v"(t) : SG_CH_S(0), SG_CH_S(1) : "vcc");
  if constexpr (N == 3) asm(SG_CH_T(0) SG_CH_T(1) SG_CH_T(2) : [t] "+v"(t) : SG_CH_S(0), SG_CH_S(1), SG_CH_S(2) : "vcc");
  if constexpr (N == 4)
    asm(SG_CH_T(0) SG_CH_T(1) SG_CH_T(2) SG_CH_T(3) : [t] "+v"(t) : SG_CH_S(0), SG_CH_S(1), SG_CH_S(2), SG_CH_S(3) : "vcc");
  if constexpr (N == 5)
    asm(SG_CH_T(0) SG_CH_T(1) SG_CH_T(2) SG_CH_T(3) SG_CH_T(4)
        : [t] "+v"(t) : SG_CH_S(0), SG_CH_S(1), SG_CH_S(2), SG_CH_S(3), SG_CH_S(4) : "vcc");
  if constexpr (N == 6)
    asm(SG_CH_T(0) SG_CH_T(1) SG_CH_T(2) SG_CH_T(3) SG_CH_T(4) SG_CH_T(5)
        : [t] "+v"(t) : SG_CH_S(0), SG_CH_S(1), SG_CH_S(2), SG_CH_S(3), SG_CH_S(4), SG_CH_S(5) : "vcc");
  if constexpr (N == 7)
    asm(SG_CH_T(0) SG_CH_T(1) SG_CH_T(2) SG_CH_T(3) SG_CH_T(4) SG_CH_T(5) SG_CH_T(6)
        : [t] "+v"(t) : SG_CH_S(0), SG_CH_S(1), SG_CH_S(2), SG_CH_S(3), SG_CH_S(4), SG_CH_S(5), SG_CH_S(6) : "vcc");
  if constexpr (N == 8)
    asm(SG_CH_T(0) SG_CH_T(1) SG_CH_T(2) SG_CH_T(3) SG_CH_T(4) SG_CH_T(5) SG_CH_T(6) SG_CH_T(7)
        : [t] "+v"(t) : SG_CH_S(0), SG_CH_S(1), SG_CH_S(2), SG_CH_S(3), SG_CH_S(4), SG_CH_S(5), SG_CH_S(6), SG_CH_S(7) : "vcc");
}
#undef SG_CH_T
#undef SG_CH_V
#undef SG_CH_S
// t += sum over i + j = K of a_i b_j
template <int K>
__device__ __forceinline__ void f29_column_ab(uint64_t& t, const uint32_t (&a)[9], const uint32_t (&b)[9]) {
  constexpr int LO = K < 9 ? 0 : K - 8, N = (K < 9 ? K : 8) - LO + 1;
  uint32_t x[N], y[N];
#pragma unroll
  for (int i = 0; i < N; i++) { x[i] = a[LO + i]; y[i] = b[K - LO - i]; }
  f29_chain_vv<N>(t, x, y);
}
// t += sum over i + j = K, i < min(K, 9), of m_i p_j  (for K < 9 the term m_K p_0 follows once m_K is known)
template <class P, int K>
__device__ __forceinline__ void f29_column_mp(uint64_t& t, const uint32_t (&m)[9]) {
  constexpr int LO = K < 9 ? 0 : K - 8, HI = K < 9 ? K - 1 : 8, N = HI - LO + 1;
  if constexpr (N >= 1) {
    uint32_t x[N], y[N];
#pragma unroll
    for (int i = 0; i < N; i++) { x[i] = m[LO + i]; y[i] = P::p[K - LO - i]; }
    f29_chain_vs<N>(t, x, y);
  }
}
// closes column K < 9: the Montgomery digit m_K, its product with p_0, the carry into column K + 1
template <class P, int K>
__device__ __forceinline__ void f29_column_close_lo(uint64_t& t, uint32_t (&m)[9]) {
  m[K] = ((uint32_t)t * P::inv) & M29;
  t += (uint64_t)m[K] * P::p[0];
  t >>= 29;
}
template <class P, int K, class AB>
__device__ __forceinline__ void f29_columns(uint64_t& t, uint32_t (&m)[9], f29& r, const AB& ab) {
  ab.template column<K>(t);
  f29_column_mp<P, K>(t, m);
  if constexpr (K < 9) {
    f29_column_close_lo<P, K>(t, m);
  } else {
    r.l[K - 9] = (uint32_t)t & M29;
    t >>= 29;
  }
  if constexpr (K < 16) f29_columns<P, K + 1, AB>(t, m, r, ab);
}
struct f29_ab_mul {
  const f29 &a, &b;
  template <int K> __device__ __forceinline__ void column(uint64_t& t) const { f29_column_ab<K>(t, a.l, b.l); }
};
struct f29_ab_mul2 {
  const f29 &a, &b, &c, &d;
  template <int K> __device__ __forceinline__ void column(uint64_t& t) const {
    f29_column_ab<K>(t, a.l, b.l);
    f29_column_ab<K>(t, c.l, d.l);
  }
};
struct f29_ab_sqr {
  const f29& a;
  const uint32_t (&a2)[9];   // the doubled limbs
  // pairs i < j, i + j = K against the doubled operand, the square a_(K/2)^2 on even columns
  template <int K> __device__ __forceinline__ void column(uint64_t& t) const {
    constexpr int LO = K < 9 ? 0 : K - 8, PAIRS = (K + 1) / 2 - LO, N = PAIRS + (K % 2 == 0 ? 1 : 0);
    uint32_t x[N], y[N];
#pragma unroll
    for (int i = 0; i < PAIRS; i++) { x[i] = a2[LO + i]; y[i] = a.l[K - LO - i]; }
    if constexpr (K % 2 == 0) { x[N - 1] = a.l[K / 2]; y[N - 1] = a.l[K / 2]; }
    f29_chain_vv<N>(t, x, y);
  }
};

// Montgomery product a*b*2^-261 mod p.  Requires normalised limbs and Ba*Bb <= 170;
// returns exactly normalised limbs, value < (Ba*Bb/170.7 + 1) p < 2p.
template <class P>
SG_HD f29 f29_mul(const f29& a, const f29& b) {
  uint32_t m[9];
  uint64_t t = 0;
  f29 r;
  f29_columns<P, 0>(t, m, r, f29_ab_mul{a, b});
  r.l[8] = (uint32_t)t;
  return r;
}
// a^2 * 2^-261: the off-diagonal products are taken once against the doubled operand
// (45 multiply-adds instead of 81 in front of the reduction).  Same contract as f29_mul.
template <class P>
SG_HD f29 f29_sqr(const f29& a) {
  uint32_t a2[9];
#pragma unroll
  for (int k = 0; k < 9; k++) a2[k] = a.l[k] << 1;
  uint32_t m[9];
  uint64_t t = 0;
  f29 r;
  f29_columns<P, 0>(t, m, r, f29_ab_sqr{a, a2});
  r.l[8] = (uint32_t)t;
  return r;
}
// (a*b + c*d) * 2^-261 with ONE reduction.  Requires Ba*Bb + Bc*Bd <= 170; 27 products of
// < 2^58 per column still fit the 64-bit accumulator.
template <class P>
SG_HD f29 f29_mul2(const f29& a, const f29& b, const f29& c, const f29& d) {
  uint32_t m[9];
  uint64_t t = 0;
  f29 r;
  f29_columns<P, 0>(t, m, r, f29_ab_mul2{a, b, c, d});
  r.l[8] = (uint32_t)t;
  return r;
}
// a*b*2^-261 + z: since 261 = 9 * 29, z * 2^261 is z moved up by nine limbs, so z enters the HIGH columns of the
// product as one more addend each (9 instructions) instead of a second product z * 1^ under the same reduction (81
// multiply-adds).  Requires Ba*Bb <= 170 and z normalised; returns exactly normalised limbs, value
// < (Ba*Bb/170.7 + 1 + Bz) p.  The Horner step of a polynomial evaluation / division, a running sum of products.
__device__ __forceinline__ void f29_chain_add(uint64_t& t, uint32_t z) {
  asm("v_mad_u64_u32 %[t], vcc, %[z], 1, %[t]" : [t] "+v"(t) : [z] "v"(z) : "vcc");
}
template <class AB>
struct f29_ab_plus {
  AB ab;
  const f29& z;
  template <int K> __device__ __forceinline__ void column(uint64_t& t) const {
    if constexpr (K >= 9) f29_chain_add(t, z.l[K - 9]);
    ab.template column<K>(t);
  }
};
template <class P>
SG_HD f29 f29_mul_add(const f29& a, const f29& b, const f29& z) {
  uint32_t m[9];
  uint64_t t = 0;
  f29 r;
  f29_columns<P, 0>(t, m, r, f29_ab_plus<f29_ab_mul>{f29_ab_mul{a, b}, z});
  r.l[8] = (uint32_t)t + z.l[8];
  return r;
}
// (a_0 b_0 + ... + a_(N-1) b_(N-1)) * 2^-261 with ONE reduction, N <= 5: 9 N + 9 products of < 2^58 per column still
// fit the 64-bit accumulator (54 * 2^58 < 2^64).  Requires sum Ba_i*Bb_i <= 170; returns exactly normalised limbs, < 2p.
template <int N>
struct f29_ab_dot {
  const f29 (&a)[N];
  const f29 (&b)[N];
  template <int K> __device__ __forceinline__ void column(uint64_t& t) const {
#pragma unroll
    for (int i = 0; i < N; i++) f29_column_ab<K>(t, a[i].l, b[i].l);
  }
};
template <class P, int N>
SG_HD f29 f29_dot(const f29 (&a)[N], const f29 (&b)[N]) {
  static_assert(N >= 1 && N <= 5, "column sums must stay below 2^64");
  uint32_t m[9];
  uint64_t t = 0;
  f29 r;
  f29_columns<P, 0>(t, m, r, f29_ab_dot<N>{a, b});
  r.l[8] = (uint32_t)t;
  return r;
}
#else
// ---- host (and -DSG_F29_ROW_SCAN): row scanning in plain C++, the definition the device code is checked against ----
// Montgomery product a*b*2^-261 mod p.  Requires normalised limbs and Ba*Bb <= 170;
// returns exactly normalised limbs, value < (Ba*Bb/170.7 + 1) p < 2p.
template <class P>
SG_HD f29 f29_mul(const f29& a, const f29& b) {
  uint64_t acc[18];
#pragma unroll
  for (int k = 0; k < 18; k++) acc[k] = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
#pragma unroll
    for (int j = 0; j < 9; j++) acc[i + j] += (uint64_t)a.l[i] * b.l[j];
    uint32_t m = ((uint32_t)acc[i] * P::inv) & M29;
#pragma unroll
    for (int j = 0; j < 9; j++) acc[i + j] += (uint64_t)m * P::p[j];
    acc[i + 1] += acc[i] >> 29;
  }
  f29 r;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    r.l[k] = (uint32_t)acc[9 + k] & M29;
    acc[10 + k] += acc[9 + k] >> 29;
  }
  r.l[8] = (uint32_t)acc[17];
  return r;
}
// a^2 * 2^-261: the off-diagonal products are taken once against the doubled operand
// (45 multiply-adds instead of 81 in front of the reduction).  Same contract as f29_mul.
template <class P>
SG_HD f29 f29_sqr(const f29& a) {
  uint64_t acc[18];
#pragma unroll
  for (int k = 0; k < 18; k++) acc[k] = 0;
  uint32_t a2[9];
#pragma unroll
  for (int k = 0; k < 9; k++) a2[k] = a.l[k] << 1;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    // every product p + q = i with p <= q has p <= i, so column i is complete after row i
    acc[2 * i] += (uint64_t)a.l[i] * a.l[i];
#pragma unroll
    for (int j = i + 1; j < 9; j++) acc[i + j] += (uint64_t)a2[i] * a.l[j];
    uint32_t m = ((uint32_t)acc[i] * P::inv) & M29;
#pragma unroll
    for (int j = 0; j < 9; j++) acc[i + j] += (uint64_t)m * P::p[j];
    acc[i + 1] += acc[i] >> 29;
  }
  f29 r;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    r.l[k] = (uint32_t)acc[9 + k] & M29;
    acc[10 + k] += acc[9 + k] >> 29;
  }
  r.l[8] = (uint32_t)acc[17];
  return r;
}
// (a*b + c*d) * 2^-261 with ONE reduction.  Requires Ba*Bb + Bc*Bd <= 170; 27 products of
// < 2^58 per column still fit the 64-bit accumulators.
template <class P>
SG_HD f29 f29_mul2(const f29& a, const f29& b, const f29& c, const f29& d) {
  uint64_t acc[18];
#pragma unroll
  for (int k = 0; k < 18; k++) acc[k] = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
#pragma unroll
    for (int j = 0; j < 9; j++) acc[i + j] += (uint64_t)a.l[i] * b.l[j];
#pragma unroll
    for (int j = 0; j < 9; j++) acc[i + j] += (uint64_t)c.l[i] * d.l[j];
    uint32_t m = ((uint32_t)acc[i] * P::inv) & M29;
#pragma unroll
    for (int j = 0; j < 9; j++) acc[i + j] += (uint64_t)m * P::p[j];
    acc[i + 1] += acc[i] >> 29;
  }
  f29 r;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    r.l[k] = (uint32_t)acc[9 + k] & M29;
    acc[10 + k] += acc[9 + k] >> 29;
  }
  r.l[8] = (uint32_t)acc[17];
  return r;
}

// a*b*2^-261 + z (see the device form above): z joins the high columns
template <class P>
SG_HD f29 f29_mul_add(const f29& a, const f29& b, const f29& z) {
  uint64_t acc[18];
#pragma unroll
  for (int k = 0; k < 18; k++) acc[k] = k >= 9 ? z.l[k - 9] : 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
#pragma unroll
    for (int j = 0; j < 9; j++) acc[i + j] += (uint64_t)a.l[i] * b.l[j];
    uint32_t m = ((uint32_t)acc[i] * P::inv) & M29;
#pragma unroll
    for (int j = 0; j < 9; j++) acc[i + j] += (uint64_t)m * P::p[j];
    acc[i + 1] += acc[i] >> 29;
  }
  f29 r;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    r.l[k] = (uint32_t)acc[9 + k] & M29;
    acc[10 + k] += acc[9 + k] >> 29;
  }
  r.l[8] = (uint32_t)acc[17];
  return r;
}
// (sum_i a_i b_i) * 2^-261 with ONE reduction, N <= 5
template <class P, int N>
SG_HD f29 f29_dot(const f29 (&a)[N], const f29 (&b)[N]) {
  static_assert(N >= 1 && N <= 5, "column sums must stay below 2^64");
  uint64_t acc[18];
#pragma unroll
  for (int k = 0; k < 18; k++) acc[k] = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
#pragma unroll
    for (int t = 0; t < N; t++) {
#pragma unroll
      for (int j = 0; j < 9; j++) acc[i + j] += (uint64_t)a[t].l[i] * b[t].l[j];
    }
    uint32_t m = ((uint32_t)acc[i] * P::inv) & M29;
#pragma unroll
    for (int j = 0; j < 9; j++) acc[i + j] += (uint64_t)m * P::p[j];
    acc[i + 1] += acc[i] >> 29;
  }
  f29 r;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    r.l[k] = (uint32_t)acc[9 + k] & M29;
    acc[10 + k] += acc[9 + k] >> 29;
  }
  r.l[8] = (uint32_t)acc[17];
  return r;
}
#endif

// ONE Montgomery limb step: a * 2^-29 mod p.  a normalised with bound <= 170; returns exactly normalised limbs,
// value < (170 / 2^29 + 1) p < 2p.  9 multiply-adds instead of the 162 of a product: the closing reduction of an NTT
// pass whose data carries a factor 2^29 on purpose (ntt.hip: the last inter-pass twiddle table is scaled by it).
template <class P>
SG_HD f29 f29_mont_step(const f29& a) {
  const uint32_t m = (a.l[0] * P::inv) & M29;
  uint64_t acc = (uint64_t)a.l[0] + (uint64_t)m * P::p[0];   // == 0 mod 2^29
  f29 r;
#pragma unroll
  for (int j = 1; j < 9; j++) {
    acc = (acc >> 29) + (uint64_t)a.l[j] + (uint64_t)m * P::p[j];
    r.l[j - 1] = (uint32_t)acc & M29;
  }
  r.l[8] = (uint32_t)(acc >> 29);
  return r;
}

// Same residue, below 2p, exactly normalised limbs -- WITHOUT a product: for a normalised value with bound <= 170 the top
// limb alone estimates the quotient q' = floor(l8 * floor(2^40 / (p8 + 1)) / 2^40) <= floor(v / p), short by at most one
// (v / p < l8 / (p8 + 1) + 2^-12, q' > l8 / (p8 + 1) - 1 - 2^-10), so v - q' p lies in [0, 2p): one multiply for q', then nine
// multiply-subtracts with a borrow chain (~45 instructions against the 206 of f29_mul(v, 1^)).  The domain is unchanged.
template <class P>
SG_HD f29 f29_reduce_small(const f29& a) {
  constexpr uint64_t RECIP = ((uint64_t)1 << 40) / ((uint64_t)P::p[8] + 1);
  const uint32_t q = (uint32_t)(((uint64_t)a.l[8] * RECIP) >> 40);   // l8 < 2^30, RECIP < 2^19
  f29 r;
  int64_t c = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    c += (int64_t)a.l[i] - (int64_t)((uint64_t)q * P::p[i]);         // q p_i < 2^8 * 2^29
    r.l[i] = i < 8 ? ((uint32_t)c & M29) : (uint32_t)c;              // the last carry is the (non-negative) top limb
    c >>= 29;                                                        // arithmetic: the borrow
  }
  return r;
}

// value < 2p with exactly normalised limbs -> canonical [0, p)
template <class P>
SG_HD f29 f29_cond_sub_p(const f29& a) {
  // d = a - p with borrow chain over 29-bit limbs
  f29 d;
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    uint32_t t = a.l[i] - P::p[i] - borrow;
    borrow = t >> 31;  // limbs < 2^30: a negative difference sets bit 31
    d.l[i] = (i < 8) ? (t & M29) : t;
  }
  f29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = borrow ? a.l[i] : d.l[i];
  return r;
}
// any normalised value with bound <= 170 -> canonical representative of the same residue
// TIMES 2^-261 * c ... helper: full reduction through one product with `c`
template <class P>
SG_HD f29 f29_reduce_with(const f29& a, const uint32_t (&c)[9]) {
  f29 k;
#pragma unroll
  for (int i = 0; i < 9; i++) k.l[i] = c[i];
  return f29_cond_sub_p<P>(f29_mul<P>(a, k));
}
// canonical representative of a (same Montgomery domain); a normalised, bound <= 170
template <class P>
SG_HD f29 f29_canonical(const f29& a) {
  return f29_cond_sub_p<P>(f29_reduce_small<P>(a));
}
// is a == 0 (mod p)?  a normalised (l[0] < 2^29 exactly), bound <= 64.
// If a = k*p then (a.l[0] * p^-1) mod 2^29 == k; anything else passes the filter with
// probability 64 / 2^29 and is then decided exactly.
template <class P>
SG_HD bool f29_is_zero_mod_p(const f29& a) {
  uint32_t t = (a.l[0] * P::pinv) & M29;
  if (t > 64) return false;
  return f29_all_zero(f29_canonical<P>(a));
}

// ---- conversions with the 8 x 32-bit memory format -------------------------------------
// x (8 LE words, < 2^256) shifted left by SH bits (SH < 29) -> limbs (exactly normalised)
template <int SH>
SG_HD f29 f29_from_words(const uint32_t w[8]) {
  f29 r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    // bits [29k - SH, 29k - SH + 29) of x
    int lo = 29 * k - SH;
    uint32_t v;
    if (lo < 0) {
      v = (w[0] << (-lo)) & M29;  // only k = 0 with SH > 0
    } else {
      int wi = lo >> 5, sh = lo & 31;
      uint64_t two = (wi < 8 ? (uint64_t)w[wi] : 0) | (wi + 1 < 8 ? (uint64_t)w[wi + 1] << 32 : 0);
      v = (uint32_t)(two >> sh);
      if (k < 8) v &= M29;
    }
    r.l[k] = v;
  }
  return r;
}
// canonical limbs (value < 2^256) -> 8 LE words
SG_HD void f29_to_words(const f29& a, uint32_t w[8]) {
#pragma unroll
  for (int i = 0; i < 8; i++) {
    // bits [32i, 32i+32) of the value
    int k = (32 * i) / 29, sh = (32 * i) % 29;
    uint64_t two = (uint64_t)a.l[k] | ((uint64_t)a.l[k + 1] << 29);
    uint32_t v = (uint32_t)(two >> sh);
    if (sh + 32 > 58 && k + 2 < 9) v |= a.l[k + 2] << (58 - sh);
    w[i] = v;
  }
}

// ---- modular inversion by Bernstein-Yang division steps ("safegcd") ---------------------------
// (x^)^-1 in the 2^261 domain: (x^-1)^.  Fermat's a^(p-2) is 254 squarings + ~127 products, ~200 us as a
// dependent chain on one lane; the division-step iteration works on 30-bit signed limbs with small 2x2 transition
// matrices: 20 rounds of 30 branch-free steps (600 >= the 590 steps that suffice for 256-bit moduli), each round
// 30 x ~14 scalar operations plus ~90 multiply-adds to apply the matrix to (d, e) and (f, g) -- about a tenth of the
// instructions.  Constant iteration count: no divergence between lanes.  Inverse of 0 is 0.
struct s30 {
  int32_t v[9];  // value = sum v[i] * 2^(30 i), limbs in (-2^30, 2^30)
};
template <class P>
SG_HD f29 f29_inv_safegcd(const f29& x) {
  constexpr int32_t M30 = (int32_t)(0xffffffffu >> 2);
  // canonical integer of x^ as 30-bit limbs
  uint32_t w[9];
  f29_to_words(f29_canonical<P>(x), w);
  w[8] = 0;
  s30 d{}, e{}, f, g;
  e.v[0] = 1;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    f.v[i] = P::p30[i];
    const int bit = 30 * i, wi = bit >> 5, sh = bit & 31;
    const uint64_t two = (uint64_t)w[wi] | (wi + 1 < 9 ? (uint64_t)w[wi + 1] << 32 : 0);
    g.v[i] = (int32_t)((uint32_t)(two >> sh) & (uint32_t)M30);
  }
  int32_t zeta = -1;
  for (int round = 0; round < 20; round++) {
    // 30 division steps on the low limbs -> transition matrix t = [[u, v], [q, r]] (scaled by 2^30)
    uint32_t u = 1, v = 0, q = 0, r = 1, f0 = (uint32_t)f.v[0], g0 = (uint32_t)g.v[0];
    for (int i = 0; i < 30; i++) {
      uint32_t c1 = (uint32_t)(zeta >> 31), c2 = 0u - (g0 & 1u);
      const uint32_t xx = (f0 ^ c1) - c1, yy = (u ^ c1) - c1, zz = (v ^ c1) - c1;
      g0 += xx & c2;
      q += yy & c2;
      r += zz & c2;
      c1 &= c2;
      zeta = (int32_t)(((uint32_t)zeta ^ c1) - 1u);
      f0 += g0 & c1;
      u += q & c1;
      v += r & c1;
      g0 >>= 1;
      u <<= 1;
      v <<= 1;
    }
    const int32_t tu = (int32_t)u, tv = (int32_t)v, tq = (int32_t)q, tr = (int32_t)r;
    {  // (d, e) <- t * (d, e) / 2^30 mod p
      const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
      int32_t md = (tu & sd) + (tv & se), me = (tq & sd) + (tr & se);
      int64_t cd = (int64_t)tu * d.v[0] + (int64_t)tv * e.v[0];
      int64_t ce = (int64_t)tq * d.v[0] + (int64_t)tr * e.v[0];
      md -= (int32_t)((P::pinv30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
      me -= (int32_t)((P::pinv30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
      cd += (int64_t)P::p30[0] * md;
      ce += (int64_t)P::p30[0] * me;
      cd >>= 30;
      ce >>= 30;
#pragma unroll
      for (int i = 1; i < 9; i++) {
        cd += (int64_t)tu * d.v[i] + (int64_t)tv * e.v[i] + (int64_t)P::p30[i] * md;
        ce += (int64_t)tq * d.v[i] + (int64_t)tr * e.v[i] + (int64_t)P::p30[i] * me;
        d.v[i - 1] = (int32_t)cd & M30;
        e.v[i - 1] = (int32_t)ce & M30;
        cd >>= 30;
        ce >>= 30;
      }
      d.v[8] = (int32_t)cd;
      e.v[8] = (int32_t)ce;
    }
    {  // (f, g) <- t * (f, g) / 2^30 (exact)
      int64_t cf = (int64_t)tu * f.v[0] + (int64_t)tv * g.v[0];
      int64_t cg = (int64_t)tq * f.v[0] + (int64_t)tr * g.v[0];
      cf >>= 30;
      cg >>= 30;
#pragma unroll
      for (int i = 1; i < 9; i++) {
        cf += (int64_t)tu * f.v[i] + (int64_t)tv * g.v[i];
        cg += (int64_t)tq * f.v[i] + (int64_t)tr * g.v[i];
        f.v[i - 1] = (int32_t)cf & M30;
        g.v[i - 1] = (int32_t)cg & M30;
        cf >>= 30;
        cg >>= 30;
      }
      f.v[8] = (int32_t)cf;
      g.v[8] = (int32_t)cg;
    }
  }
  // g = 0 and f = +-1 (or f = +-p when x = 0): d = +-x^-1; bring it to [0, p)
  {
    int32_t cond_add = d.v[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) d.v[i] += P::p30[i] & cond_add;
    const int32_t cond_negate = f.v[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) d.v[i] = (d.v[i] ^ cond_negate) - cond_negate;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      d.v[i + 1] += d.v[i] >> 30;
      d.v[i] &= M30;
    }
    cond_add = d.v[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) d.v[i] += P::p30[i] & cond_add;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      d.v[i + 1] += d.v[i] >> 30;
      d.v[i] &= M30;
    }
  }
  // 30-bit limbs -> 29-bit limbs; X^-1 = x^-1 * 2^-261, times 2^783 * 2^-261 gives x^-1 * 2^261
  f29 out;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const int bit = 29 * k, li = bit / 30, sh = bit % 30;
    uint64_t two = (uint64_t)(uint32_t)d.v[li] | (li + 1 < 9 ? (uint64_t)(uint32_t)d.v[li + 1] << 30 : 0);
    out.l[k] = (uint32_t)(two >> sh) & M29;
  }
  return f29_mul<P>(out, f29_const<P>(P::r783));
}

}  // namespace sg
