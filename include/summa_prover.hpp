// create_proof for the constraint system of the reference's MstInclusionCircuit, driven from C++ over the C ABI
// (include/summa_gpu.h): the compiled-host counterpart of circuits_halo2_amd/prover.py, same steps, same transcript,
// same 2144-byte proof [REF zk_prover/src/circuits/utils.rs:94-101 -> halo2_proofs::plonk::create_proof with
// ProverSHPLONK; proof layout and Keccak transcript: contracts/src/InclusionVerifier.sol:85-110, 274-367].
// Every data-parallel step runs on the device; the host does what upstream also does serially (transcript, the
// lookup's sort, scalars of the multi-open, blinding factors from the OS entropy source).
// The circuit-specific inputs -- fixed / permutation columns, the GraphEvaluator programs of the gates and of the
// lookup input -- come from the proving key (here: a bundle file written by circuits_halo2_amd.prover.export_bundle).
// Header-only; needs the HIP runtime for device buffers.  Proofs are checked by tests/test_gpu_prover.py.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <functional>
#include <exception>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "summa_gpu.h"

namespace summa {
namespace prover {

// ------------------------------------------------------------------ BN254 Fr on the host (Montgomery, 4 x 64)
struct Fr {
  uint64_t l[4];
  static constexpr uint64_t P[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
  static constexpr uint64_t INV = 0xc2e1f593efffffffULL;  // -p^-1 mod 2^64
  static constexpr uint64_t R1[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
  static constexpr uint64_t R2[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL};
  static Fr zero() { return Fr{{0, 0, 0, 0}}; }
  static Fr one() { return Fr{{R1[0], R1[1], R1[2], R1[3]}}; }
  static bool geq_p(const uint64_t a[4]) {
    for (int i = 3; i >= 0; i--) {
      if (a[i] != P[i]) return a[i] > P[i];
    }
    return true;
  }
  static void sub_p(uint64_t a[4]) {
    unsigned __int128 borrow = 0;
    for (int i = 0; i < 4; i++) {
      unsigned __int128 t = (unsigned __int128)a[i] - P[i] - (uint64_t)borrow;
      a[i] = (uint64_t)t;
      borrow = (t >> 64) & 1;
    }
  }
  Fr operator+(const Fr& o) const {
    Fr r;
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (unsigned __int128)l[i] + o.l[i];
      r.l[i] = (uint64_t)c;
      c >>= 64;
    }
    if (c || geq_p(r.l)) sub_p(r.l);
    return r;
  }
  Fr operator-(const Fr& o) const {
    Fr r;
    unsigned __int128 borrow = 0;
    for (int i = 0; i < 4; i++) {
      unsigned __int128 t = (unsigned __int128)l[i] - o.l[i] - (uint64_t)borrow;
      r.l[i] = (uint64_t)t;
      borrow = (t >> 64) & 1;
    }
    if (borrow) {
      unsigned __int128 c = 0;
      for (int i = 0; i < 4; i++) {
        c += (unsigned __int128)r.l[i] + P[i];
        r.l[i] = (uint64_t)c;
        c >>= 64;
      }
    }
    return r;
  }
  Fr operator-() const { return zero() - *this; }
  Fr operator*(const Fr& o) const {  // CIOS Montgomery product
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
      unsigned __int128 c = 0;
      for (int j = 0; j < 4; j++) {
        c += (unsigned __int128)l[j] * o.l[i] + t[j];
        t[j] = (uint64_t)c;
        c >>= 64;
      }
      c += t[4];
      t[4] = (uint64_t)c;
      t[5] = (uint64_t)(c >> 64);
      const uint64_t m = t[0] * INV;
      c = (unsigned __int128)m * P[0] + t[0];
      c >>= 64;
      for (int j = 1; j < 4; j++) {
        c += (unsigned __int128)m * P[j] + t[j];
        t[j - 1] = (uint64_t)c;
        c >>= 64;
      }
      c += t[4];
      t[3] = (uint64_t)c;
      t[4] = t[5] + (uint64_t)(c >> 64);
    }
    Fr r{{t[0], t[1], t[2], t[3]}};
    if (t[4] || geq_p(r.l)) sub_p(r.l);
    return r;
  }
  bool operator==(const Fr& o) const { return !std::memcmp(l, o.l, 32); }
  bool operator!=(const Fr& o) const { return !(*this == o); }
  bool is_zero() const { return !(l[0] | l[1] | l[2] | l[3]); }
  Fr pow(const uint64_t e[4]) const {
    int top = 255;   // square-and-multiply from the highest set bit (most exponents here are rotations and small powers)
    while (top >= 0 && !((e[top / 64] >> (top % 64)) & 1)) top--;
    Fr r = one();
    for (int i = top; i >= 0; i--) {
      r = r * r;
      if ((e[i / 64] >> (i % 64)) & 1) r = r * *this;
    }
    return r;
  }
  Fr pow(uint64_t e) const {
    const uint64_t ee[4] = {e, 0, 0, 0};
    return pow(ee);
  }
  Fr inv() const {
    const uint64_t e[4] = {P[0] - 2, P[1], P[2], P[3]};
    return pow(e);
  }
  static Fr from_u64(uint64_t v) { return from_canonical_limbs(std::array<uint64_t, 4>{v, 0, 0, 0}.data()); }
  static Fr from_canonical_limbs(const uint64_t c[4]) {  // c < p
    Fr a{{c[0], c[1], c[2], c[3]}}, r2{{R2[0], R2[1], R2[2], R2[3]}};
    return a * r2;
  }
  // any 256-bit big-endian integer, reduced mod p (challenges: keccak output)
  static Fr from_be_bytes_reduced(const uint8_t b[32]) {
    uint64_t c[4];
    for (int i = 0; i < 4; i++) {
      uint64_t w = 0;
      for (int j = 0; j < 8; j++) w = (w << 8) | b[8 * (3 - i) + j];
      c[i] = w;
    }
    while (geq_p(c)) sub_p(c);
    return from_canonical_limbs(c);
  }
  void to_canonical_limbs(uint64_t out[4]) const {
    Fr o{{1, 0, 0, 0}};
    Fr c = *this * o;
    std::memcpy(out, c.l, 32);
  }
  void to_be_bytes(uint8_t out[32]) const {
    uint64_t c[4];
    to_canonical_limbs(c);
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 8; j++) out[8 * (3 - i) + j] = (uint8_t)(c[i] >> (8 * (7 - j)));
  }
  const uint8_t* bytes() const { return reinterpret_cast<const uint8_t*>(l); }  // Montgomery, as the ABI takes it
};

// Fq only appears as bytes to convert: Montgomery little-endian (ABI) -> canonical big-endian (proof / transcript)
inline void fq_mont_to_be(const uint8_t in[32], uint8_t out[32]) {
  static constexpr uint64_t Q[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
  static constexpr uint64_t QINV = 0x87d20782e4866389ULL;
  uint64_t t[5];
  std::memcpy(t, in, 32);
  t[4] = 0;
  for (int i = 0; i < 4; i++) {  // Montgomery reduction of (in * 1)
    const uint64_t m = t[0] * QINV;
    unsigned __int128 c = (unsigned __int128)m * Q[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; j++) {
      c += (unsigned __int128)m * Q[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = (uint64_t)(c >> 64);
  }
  bool ge = true;
  for (int i = 3; i >= 0; i--) {
    if (t[i] != Q[i]) {
      ge = t[i] > Q[i];
      break;
    }
  }
  if (ge) {
    unsigned __int128 borrow = 0;
    for (int i = 0; i < 4; i++) {
      unsigned __int128 d = (unsigned __int128)t[i] - Q[i] - (uint64_t)borrow;
      t[i] = (uint64_t)d;
      borrow = (d >> 64) & 1;
    }
  }
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 8; j++) out[8 * (3 - i) + j] = (uint8_t)(t[i] >> (8 * (7 - j)));
}

// ------------------------------------------------------------------ Keccak-256 (Ethereum's) and the EVM transcript
inline void keccak_f(uint64_t s[25]) {
  static constexpr uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
      0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
      0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
      0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  static constexpr int ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
  static constexpr int PIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
  for (int round = 0; round < 24; round++) {
    uint64_t bc[5];
    for (int i = 0; i < 5; i++) bc[i] = s[i] ^ s[i + 5] ^ s[i + 10] ^ s[i + 15] ^ s[i + 20];
    for (int i = 0; i < 5; i++) {
      const uint64_t t = bc[(i + 4) % 5] ^ ((bc[(i + 1) % 5] << 1) | (bc[(i + 1) % 5] >> 63));
      for (int j = 0; j < 25; j += 5) s[j + i] ^= t;
    }
    uint64_t t = s[1];
    for (int i = 0; i < 24; i++) {
      const int j = PIL[i];
      const uint64_t b = s[j];
      s[j] = (t << ROT[i]) | (t >> (64 - ROT[i]));
      t = b;
    }
    for (int j = 0; j < 25; j += 5) {
      for (int i = 0; i < 5; i++) bc[i] = s[j + i];
      for (int i = 0; i < 5; i++) s[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
    }
    s[0] ^= RC[round];
  }
}
inline std::array<uint8_t, 32> keccak256(const uint8_t* data, size_t len) {
  uint64_t s[25] = {0};
  constexpr size_t rate = 136;
  std::vector<uint8_t> buf(data, data + len);
  buf.push_back(0x01);
  while (buf.size() % rate) buf.push_back(0);
  buf.back() |= 0x80;
  for (size_t off = 0; off < buf.size(); off += rate) {
    for (size_t i = 0; i < rate / 8; i++) {
      uint64_t w;
      std::memcpy(&w, buf.data() + off + 8 * i, 8);
      s[i] ^= w;
    }
    keccak_f(s);
  }
  std::array<uint8_t, 32> out;
  std::memcpy(out.data(), s, 32);
  return out;
}

// Both transcripts of the reference [REF zk_prover/src/circuits/utils.rs:93 (Blake2bWrite / Challenge255, `full_prover`),
// :170 (Keccak256Transcript, `gen_proof_solidity_calldata`)]: absorb the verifying key's digest first (`vk.hash_into`),
// then the instances, commitments and evaluations as create_proof produces them.
struct EvmTranscript {
  std::vector<uint8_t> buf, proof;
  bool squeezed = false;
  void common_scalar(const Fr& v) {
    uint8_t b[32];
    v.to_be_bytes(b);
    buf.insert(buf.end(), b, b + 32);
    squeezed = false;
  }
  void write_scalar(const Fr& v) {
    uint8_t b[32];
    v.to_be_bytes(b);
    buf.insert(buf.end(), b, b + 32);
    proof.insert(proof.end(), b, b + 32);
    squeezed = false;
  }
  void write_point(const uint8_t affine_mont[64]) {  // as the ABI returns commitments
    uint8_t b[64];
    fq_mont_to_be(affine_mont, b);
    fq_mont_to_be(affine_mont + 32, b + 32);
    buf.insert(buf.end(), b, b + 64);
    proof.insert(proof.end(), b, b + 64);
    squeezed = false;
  }
  Fr squeeze() {   // keccak(buffer) mod r, the hash becomes the buffer; right after a squeeze: keccak(hash || 0x01)
    if (squeezed) {
      buf.resize(32);
      buf.push_back(0x01);
    }
    auto h = keccak256(buf.data(), buf.size());
    buf.assign(h.begin(), h.end());
    squeezed = true;
    return Fr::from_be_bytes_reduced(h.data());
  }
  Fr squeeze_again() { return squeeze(); }
};

// Blake2b (RFC 7693), unkeyed, with personalisation; `finalize` works on a copy, as the transcript needs it
struct Blake2b {
  uint64_t h[8];
  uint8_t buf[128];
  size_t buflen = 0;
  uint64_t t0 = 0, t1 = 0;
  static constexpr uint64_t IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                     0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
  Blake2b(size_t outlen, const char* personal16) {
    uint8_t param[64] = {0};
    param[0] = (uint8_t)outlen;
    param[2] = 1;
    param[3] = 1;
    if (personal16) std::memcpy(param + 48, personal16, std::min<size_t>(16, std::strlen(personal16)));
    for (int i = 0; i < 8; i++) {
      uint64_t w;
      std::memcpy(&w, param + 8 * i, 8);
      h[i] = IV[i] ^ w;
    }
  }
  static uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
  void compress(const uint8_t block[128], bool last) {
    static constexpr uint8_t SIGMA[12][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
    uint64_t m[16], v[16];
    std::memcpy(m, block, 128);
    for (int i = 0; i < 8; i++) {
      v[i] = h[i];
      v[i + 8] = IV[i];
    }
    v[12] ^= t0;
    v[13] ^= t1;
    if (last) v[14] = ~v[14];
    auto g = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
      v[a] = v[a] + v[b] + x; v[d] = rotr(v[d] ^ v[a], 32);
      v[c] = v[c] + v[d]; v[b] = rotr(v[b] ^ v[c], 24);
      v[a] = v[a] + v[b] + y; v[d] = rotr(v[d] ^ v[a], 16);
      v[c] = v[c] + v[d]; v[b] = rotr(v[b] ^ v[c], 63);
    };
    for (int r = 0; r < 12; r++) {
      const uint8_t* sg = SIGMA[r];
      g(0, 4, 8, 12, m[sg[0]], m[sg[1]]); g(1, 5, 9, 13, m[sg[2]], m[sg[3]]);
      g(2, 6, 10, 14, m[sg[4]], m[sg[5]]); g(3, 7, 11, 15, m[sg[6]], m[sg[7]]);
      g(0, 5, 10, 15, m[sg[8]], m[sg[9]]); g(1, 6, 11, 12, m[sg[10]], m[sg[11]]);
      g(2, 7, 8, 13, m[sg[12]], m[sg[13]]); g(3, 4, 9, 14, m[sg[14]], m[sg[15]]);
    }
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
  }
  void update(const uint8_t* data, size_t len) {
    while (len) {
      if (buflen == 128) {   // a full buffer is only compressed once more input follows (the last block is special)
        t0 += 128;
        if (t0 < 128) t1++;
        compress(buf, false);
        buflen = 0;
      }
      const size_t take = std::min(len, 128 - buflen);
      std::memcpy(buf + buflen, data, take);
      buflen += take;
      data += take;
      len -= take;
    }
  }
  void finalize(uint8_t* out, size_t outlen) const {   // on a copy: the state keeps absorbing afterwards
    Blake2b c = *this;
    c.t0 += c.buflen;
    if (c.t0 < c.buflen) c.t1++;
    std::memset(c.buf + c.buflen, 0, 128 - c.buflen);
    c.compress(c.buf, true);
    std::memcpy(out, c.h, outlen);
  }
};
struct Blake2bTranscript {   // Blake2bWrite<_, G1Affine, Challenge255<_>> (halo2_proofs transcript.rs)
  Blake2b state{64, "Halo2-Transcript"};
  std::vector<uint8_t> proof;
  static void reverse32(uint8_t b[32]) { std::reverse(b, b + 32); }
  void common_scalar(const Fr& v) {
    uint8_t b[33];
    b[0] = 2;
    v.to_be_bytes(b + 1);
    reverse32(b + 1);
    state.update(b, 33);
  }
  void write_scalar(const Fr& v) {
    uint8_t b[33];
    b[0] = 2;
    v.to_be_bytes(b + 1);
    reverse32(b + 1);
    state.update(b, 33);
    proof.insert(proof.end(), b + 1, b + 33);
  }
  void write_point(const uint8_t affine_mont[64]) {
    uint8_t b[65];
    b[0] = 1;
    fq_mont_to_be(affine_mont, b + 1);
    fq_mont_to_be(affine_mont + 32, b + 33);
    reverse32(b + 1);
    reverse32(b + 33);
    bool inf = true;
    for (int i = 0; i < 64; i++) inf = inf && !affine_mont[i];
    if (inf) throw std::runtime_error("cannot write points at infinity to the transcript");
    state.update(b, 65);
    uint8_t c[32];
    std::memcpy(c, b + 1, 32);                    // x little-endian, bit 6 of the last byte = parity of y
    c[31] |= (uint8_t)((b[33] & 1) << 6);
    proof.insert(proof.end(), c, c + 32);
  }
  Fr squeeze() {   // prefix 0, digest of a clone, 64 bytes as a little-endian integer mod r (from_uniform_bytes)
    const uint8_t zero = 0;
    state.update(&zero, 1);
    uint8_t d[64];
    state.finalize(d, 64);
    Fr lo, hi, r2;
    std::memcpy(lo.l, d, 32);
    std::memcpy(hi.l, d + 32, 32);
    std::memcpy(r2.l, Fr::R2, 32);
    return lo * r2 + (hi * r2) * r2;              // lo R + hi R^2: Montgomery form of lo + hi 2^256
  }
  Fr squeeze_again() { return squeeze(); }
};

struct Options {
  bool sanity_checks = true;   // refuse a witness whose permutation / lookup grand product does not close (upstream's cargo
                               // feature), or whose advice words are not canonical field elements
};
struct WitnessError : std::runtime_error {   // the assignment, not the machinery, is at fault
  using std::runtime_error::runtime_error;
};

// ------------------------------------------------------------------ inputs
struct Graph {  // a GraphEvaluator program as the C ABI takes it
  std::vector<uint8_t> constants;
  std::vector<int32_t> rotations;
  std::vector<sg_calculation> calculations;
  std::vector<sg_value_source> parts;
  sg_graph view() const {
    return sg_graph{constants.data(), (uint32_t)(constants.size() / 32), rotations.data(), (uint32_t)rotations.size(),
                    calculations.data(), (uint32_t)calculations.size(), parts.data(), (uint32_t)parts.size()};
  }
};

inline void ck(int rc, const char* what) {
  if (rc != SG_OK) throw std::runtime_error(std::string(what) + ": " + sg_last_error());
}
inline void hk(hipError_t e, const char* what) {
  if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
// Per-thread prover state: the stream proofs of this thread run on, two side streams with their fork / join events,
// the pool of freed device columns and the page-locked staging area.  Nothing is shared between host threads, so
// several proofs can be in flight on one GPU (one thread each, circuits_halo2_amd/batch.py); the C ABI underneath
// gives every concurrent call its own lane.
inline bool& session_gone() {   // trivially destructible, so still readable after this thread's Session has been destroyed
  static thread_local bool gone = false;
  return gone;
}
// What a host thread's session leaves behind when the thread ends.  A thread-local destructor must not call into HIP: worker
// threads are joined while the process shuts down, and hipFree from a thread's destructor then runs into the runtime's own
// teardown (seen as a crash in amd::Context::svmFree with more worker threads than lanes).  So a dying session hands its
// device columns, streams, events and pinned staging to this process-wide store; the next session on another thread takes
// them from here before asking the runtime, and release_orphans() (sg_shutdown, with the device idle) returns them.
struct Orphans {
  std::mutex mu;
  std::multimap<size_t, void*> pool;
  std::vector<hipStream_t> streams;
  std::vector<hipEvent_t> events;
  std::vector<std::pair<uint64_t*, size_t>> pinned;
};
inline Orphans& orphans() {
  static Orphans* o = new Orphans();   // never destroyed: sessions may end after static destruction has begun
  return *o;
}
inline void release_orphans() {
  Orphans& o = orphans();
  std::lock_guard<std::mutex> lk(o.mu);
  for (auto& kv : o.pool) (void)hipFree(kv.second);
  for (auto& p : o.pinned) (void)hipHostFree(p.first);
  for (auto& st : o.streams) (void)hipStreamDestroy(st);
  for (auto& e : o.events) (void)hipEventDestroy(e);
  o.pool.clear();
  o.pinned.clear();
  o.streams.clear();
  o.events.clear();
}
struct Session {
  hipStream_t main = nullptr;                 // NULL: HIP's default stream (single-threaded callers)
  hipStream_t side[2] = {nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
  std::multimap<size_t, void*> pool;          // freed columns are kept for the next proof (hipMalloc / hipFree synchronise the device)
  uint64_t* pinned = nullptr;
  size_t pinned_cap = 0;
  uint64_t* pinned_small = nullptr;           // 8 rows of page-locked memory, mapped into the device: small values the kernels write
                                              // there themselves (checked where the host waits anyway; no copy launches)
  uint8_t* pinned_small_dev = nullptr;        // ... its device address
  ~Session() {   // no HIP calls here (see Orphans)
    Orphans& o = orphans();
    std::lock_guard<std::mutex> lk(o.mu);
    for (auto& kv : pool) o.pool.emplace(kv.first, kv.second);
    if (pinned) o.pinned.emplace_back(pinned, pinned_cap);
    if (pinned_small) o.pinned.emplace_back(pinned_small, (size_t)0);   // (capacity 0: released, never handed on as row staging)
    for (auto& st : side)
      if (st) o.streams.push_back(st);
    if (ev_fork) o.events.push_back(ev_fork);
    for (auto& e : ev_join)
      if (e) o.events.push_back(e);
    session_gone() = true;
  }
};
inline Session& session() {
  static thread_local Session s;
  return s;
}
// a stream / an event for a new session: one an ended session left behind, else a new one
inline hipStream_t adopt_or_create_stream(int priority) {
  {
    Orphans& o = orphans();
    std::lock_guard<std::mutex> lk(o.mu);
    if (!o.streams.empty()) {
      hipStream_t st = o.streams.back();
      o.streams.pop_back();
      return st;
    }
  }
  hipStream_t st = nullptr;
  hk(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, priority), "stream");
  return st;
}
inline hipEvent_t adopt_or_create_event() {
  {
    Orphans& o = orphans();
    std::lock_guard<std::mutex> lk(o.mu);
    if (!o.events.empty()) {
      hipEvent_t e = o.events.back();
      o.events.pop_back();
      return e;
    }
  }
  hipEvent_t e = nullptr;
  hk(hipEventCreateWithFlags(&e, hipEventDisableTiming), "event");
  return e;
}
inline hipStream_t main_stream() { return session().main; }
struct StreamScope {   // run this thread's prover calls on `s` for the scope
  hipStream_t prev;
  explicit StreamScope(hipStream_t s) : prev(session().main) { session().main = s; }
  ~StreamScope() { session().main = prev; }
};
inline std::multimap<size_t, void*>& column_pool() { return session().pool; }
inline void release_column_pool() {
  for (auto& kv : column_pool()) (void)hipFree(kv.second);
  column_pool().clear();
}
inline void d2h(void* host, const void* dev, size_t bytes) {   // ordered on the thread's main stream, complete on return
  // (the stream is drained by the library's wait FIRST -- asleep when "host.wait_sleep_us" is set: a copy into pageable memory
  // waits for everything ahead of it inside hipMemcpyAsync, busily; csrc/host_wait.h, host_copy_d2h)
  ck(sg_stream_wait(main_stream()), "sync");
  hk(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, main_stream()), "D2H");
  ck(sg_stream_wait(main_stream()), "sync");
}
inline void h2d(void* dev, const void* host, size_t bytes) {
  hk(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, main_stream()), "H2D");
  ck(sg_stream_wait(main_stream()), "sync");   // the host buffer may be a temporary
}
struct DevCol {  // device column of Fr (Montgomery); owned unless borrowed from the caller
  void* p = nullptr;
  size_t rows = 0;
  bool owned = true;
  DevCol() = default;
  static DevCol borrow(void* ptr, size_t r) {
    DevCol c;
    c.p = ptr;
    c.rows = r;
    c.owned = false;
    return c;
  }
  explicit DevCol(size_t r) : rows(r) {
    auto it = column_pool().find(r);
    if (it != column_pool().end()) {
      p = it->second;
      column_pool().erase(it);
    } else {
      {
        Orphans& o = orphans();
        std::lock_guard<std::mutex> lk(o.mu);
        auto ot = o.pool.find(r);
        if (ot != o.pool.end()) {
          p = ot->second;
          o.pool.erase(ot);
        }
      }
      if (!p) hk(hipMalloc(&p, 32 * r), "hipMalloc");
    }
  }
  DevCol(const DevCol&) = delete;
  DevCol& operator=(const DevCol&) = delete;
  DevCol(DevCol&& o) noexcept : p(o.p), rows(o.rows), owned(o.owned) { o.p = nullptr; }
  void give_back() {   // to this thread's pool; objects that outlive the thread's session (process exit) just let go
    if (p && owned && !session_gone()) column_pool().emplace(rows, p);
    p = nullptr;
  }
  DevCol& operator=(DevCol&& o) noexcept {
    give_back();
    p = o.p;
    rows = o.rows;
    owned = o.owned;
    o.p = nullptr;
    return *this;
  }
  ~DevCol() { give_back(); }
  uint8_t* at(size_t row) const { return static_cast<uint8_t*>(p) + 32 * row; }
  void upload(const void* host, size_t first, size_t count) { h2d(at(first), host, 32 * count); }
  void zero() { hk(hipMemsetAsync(p, 0, 32 * rows, main_stream()), "memset"); }
};

// constraint-system constants of MstInclusionCircuit (circuits_halo2_amd/mst_inclusion.py)
constexpr uint32_t NUM_ADVICE = 3, NUM_FIXED = 11, NUM_SIGMA = 6, BLINDING = 5, CHUNK = 4, QUOTIENT_PIECES = 5;
constexpr int ROT_LAST = -(int)(BLINDING + 1);
enum Kind { A_, F_, SIGMA_, Z_, LZ_, PIN_, PTAB_, RANDOM_, H_ };
struct Key {
  Kind kind;
  uint32_t index;
  bool operator<(const Key& o) const { return kind != o.kind ? kind < o.kind : index < o.index; }
};
struct Query { Key key; int rot; };

inline std::vector<Query> eval_order() {  // [REF InclusionVerifier.sol:500-1000, calldata slots 0x03e4 ..]
  std::vector<Query> q = {{{A_, 0}, 0}, {{A_, 1}, 0}, {{A_, 0}, 1}, {{A_, 1}, 1}, {{A_, 2}, 0}, {{A_, 1}, -1}, {{A_, 0}, -1},
                          {{F_, 2}, 0}, {{F_, 3}, 0}, {{F_, 0}, 0}, {{F_, 1}, 0}};
  for (uint32_t j = 4; j < 11; j++) q.push_back({{F_, j}, 0});
  q.push_back({{RANDOM_, 0}, 0});
  for (uint32_t j = 0; j < 6; j++) q.push_back({{SIGMA_, j}, 0});
  for (Query x : std::vector<Query>{{{Z_, 0}, 0}, {{Z_, 0}, 1}, {{Z_, 0}, ROT_LAST}, {{Z_, 1}, 0}, {{Z_, 1}, 1}, {{LZ_, 0}, 0},
                                    {{LZ_, 0}, 1}, {{PIN_, 0}, 0}, {{PIN_, 0}, -1}, {{PTAB_, 0}, 0}})
    q.push_back(x);
  return q;
}
struct RotationSet { std::vector<int> rots; std::vector<Key> polys; };
inline std::vector<RotationSet> rotation_sets() {  // nu order; polynomials in increasing power of zeta [REF :1159-1340]
  std::vector<RotationSet> s(5);
  s[0] = {{-1, 0, 1}, {{A_, 0}, {A_, 1}}};
  s[1].rots = {0};
  s[1].polys = {{A_, 2}, {PTAB_, 0}, {F_, 2}, {F_, 3}, {F_, 0}, {F_, 1}};
  for (uint32_t j = 4; j < 11; j++) s[1].polys.push_back({F_, j});
  for (uint32_t j = 0; j < 6; j++) s[1].polys.push_back({SIGMA_, j});
  s[1].polys.push_back({H_, 0});
  s[1].polys.push_back({RANDOM_, 0});
  s[2] = {{ROT_LAST, 0, 1}, {{Z_, 0}}};
  s[3] = {{0, 1}, {{Z_, 1}, {LZ_, 0}}};
  s[4] = {{-1, 0}, {{PIN_, 0}}};
  return s;
}

struct ProvingKey {
  std::vector<uint64_t> table_rows;  // the lookup table column (fixed 4), canonical limbs of the usable rows (host copy)
  bool table_is_range = false;       // ... every usable row below 2^16: the device-side permutation applies (known with the key)
  uint32_t k = 0;
  size_t n = 0, usable = 0;
  uint64_t srs = 0;
  uint8_t vk_digest_be[32] = {0};
  Graph gates, lookup_input;
  // the challenges the gate program reads: challenge i = sum over e in gate_challenge_exps[i] of y^e
  std::vector<std::vector<uint32_t>> gate_challenge_exps;
  std::vector<DevCol> fixed_lag, sigma_lag, fixed_coeff, sigma_coeff, fixed_ext, sigma_ext;
  DevCol l0_ext, l_last_ext, l_active_ext;
  uint32_t ext_k() const { return k + 3; }  // degree 6: extended domain 2^(k + 3)
  // the quotient has degree < QUOTIENT_PIECES * n: it is evaluated on that many cosets of the 2^k domain (the first ones of
  // the extended domain, coset-major: sg_coeff_to_cosets_batch_dev), not on all 2^(k + 3) points; "ext" columns have this size
  size_t ext_rows() const { return n * QUOTIENT_PIECES; }

  // fixed / sigma: Lagrange-basis device columns (moved in); builds the coefficient and extended-coset forms
  void build(uint32_t k_, uint64_t srs_, std::vector<DevCol>&& fixed, std::vector<DevCol>&& sigma) {
    k = k_;
    n = (size_t)1 << k;
    usable = n - (BLINDING + 1);
    srs = srs_;
    fixed_lag = std::move(fixed);
    sigma_lag = std::move(sigma);
    const size_t ne = ext_rows();
    uint8_t omega_inv[32], n_inv[32];
    ck(sg_domain_constant(k, 1, omega_inv), "domain constant");
    ck(sg_domain_constant(k, 2, n_inv), "domain constant");
    std::vector<DevCol> sel;
    const Fr one = Fr::one();
    for (int s = 0; s < 3; s++) {
      sel.emplace_back(n);
      sel.back().zero();
    }
    sel[0].upload(one.l, 0, 1);
    sel[1].upload(one.l, usable, 1);
    {
      std::vector<Fr> ones(usable, one);
      sel[2].upload(ones.data(), 0, usable);
    }
    auto transform = [&](std::vector<DevCol>& lag, std::vector<DevCol>& coeff, std::vector<DevCol>& ext) {
      std::vector<void*> pc, pe;
      std::vector<const void*> pl;
      for (auto& c : lag) {
        coeff.emplace_back(n);
        ext.emplace_back(ne);
        pl.push_back(c.p);
        pc.push_back(coeff.back().p);
        pe.push_back(ext.back().p);
      }
      for (size_t i = 0; i < pc.size(); i += 16) {   // batched launches hold at most 16 vectors
        const size_t m = std::min<size_t>(16, pc.size() - i);
        ck(sg_ntt_fr_batch_oop_dev(pl.data() + i, pc.data() + i, m, omega_inv, n_inv, k, main_stream()), "iNTT batch");
        ck(sg_coeff_to_cosets_batch_dev(pc.data() + i, pe.data() + i, m, k, ext_k(), QUOTIENT_PIECES, main_stream()), "coset NTT batch");
      }
    };
    transform(fixed_lag, fixed_coeff, fixed_ext);
    transform(sigma_lag, sigma_coeff, sigma_ext);
    std::vector<DevCol> sel_coeff, sel_ext;
    transform(sel, sel_coeff, sel_ext);
    l0_ext = std::move(sel_ext[0]);
    l_last_ext = std::move(sel_ext[1]);
    l_active_ext = std::move(sel_ext[2]);
    {
      DevCol canon(n);
      table_rows.resize(4 * n);
      ck(sg_fr_from_montgomery_dev(fixed_lag[4].p, canon.p, n, main_stream()), "from_montgomery");
      d2h(table_rows.data(), canon.p, 32 * n);
    }
    ck(sg_stream_wait(main_stream()), "sync");
    table_is_range = true;
    for (size_t i = 0; i < usable; i++)
      table_is_range = table_is_range && table_rows[4 * i] < (1u << 16) && !(table_rows[4 * i + 1] | table_rows[4 * i + 2] | table_rows[4 * i + 3]);
  }
};

inline void os_random(uint8_t* out, size_t bytes) {
  FILE* f = std::fopen("/dev/urandom", "rb");
  if (!f || std::fread(out, 1, bytes, f) != bytes) {
    if (f) std::fclose(f);
    throw std::runtime_error("/dev/urandom");
  }
  std::fclose(f);
}

inline uint64_t* pinned_small_rows() {   // 8 rows, allocated once per session
  uint64_t*& p = session().pinned_small;
  if (!p) {
    hk(hipHostMalloc(reinterpret_cast<void**>(&p), 32 * 8, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc");
    hk(hipHostGetDevicePointer(reinterpret_cast<void**>(&session().pinned_small_dev), p, 0), "hipHostGetDevicePointer");
  }
  return p;
}
// rows of that block: 0 .. 2 the grand products' closing values, 3 the remainder of the final division, 4 two status words
// (range check of the advice columns, lookup permutation)
enum { MAIL_CLOSING = 0, MAIL_REMAINDER = 3, MAIL_STATUS = 4 };
inline uint8_t* pinned_small_dev_row(uint32_t row) {
  pinned_small_rows();
  return session().pinned_small_dev + 32 * row;
}
inline uint64_t* pinned_rows(size_t rows) {  // page-locked host staging, grown on demand, kept (per thread)
  uint64_t*& p = session().pinned;
  size_t& cap = session().pinned_cap;
  if (rows > cap) {
    if (p) (void)hipHostFree(p);
    hk(hipHostMalloc(reinterpret_cast<void**>(&p), 32 * rows), "hipHostMalloc");
    cap = rows;
  }
  return p;
}

// halo2 `permute_expression_pair` on the usable rows (canonical limbs, rows of 4): A' sorted; S' such that every row
// has A'[i] == S'[i] or A'[i] == A'[i-1].  One-limb tables (range checks) sort by the low limb only.
inline void permute_expression_pair(const uint64_t* inp, const uint64_t* table, size_t rows, uint64_t* a_out, uint64_t* s_out) {
  using Row = std::array<uint64_t, 4>;
  auto less = [](const Row& x, const Row& y) {
    for (int i = 3; i >= 0; i--)
      if (x[i] != y[i]) return x[i] < y[i];
    return false;
  };
  std::vector<Row> a(rows), t(rows);
  std::memcpy(a.data(), inp, 32 * rows);
  std::memcpy(t.data(), table, 32 * rows);
  bool small = true;
  for (auto& r : t) small = small && !(r[1] | r[2] | r[3]);
  if (small) {  // sort 8-byte keys instead of 32-byte rows
    std::vector<uint64_t> ka(rows), kt(rows);
    for (size_t i = 0; i < rows; i++) {
      if (a[i][1] | a[i][2] | a[i][3]) throw std::runtime_error("lookup input value not in the table");
      ka[i] = a[i][0];
      kt[i] = t[i][0];
    }
    const uint64_t top = *std::max_element(kt.begin(), kt.end());
    if (top < (1u << 20)) {  // range tables: counting sort
      std::vector<uint32_t> ca(top + 1, 0), ct(top + 1, 0);
      for (size_t i = 0; i < rows; i++) {
        if (ka[i] > top) throw std::runtime_error("lookup input value not in the table");
        ca[ka[i]]++;
        ct[kt[i]]++;
      }
      size_t ia = 0, it = 0;
      for (uint64_t v = 0; v <= top; v++) {
        for (uint32_t c = 0; c < ca[v]; c++) ka[ia++] = v;
        for (uint32_t c = 0; c < ct[v]; c++) kt[it++] = v;
      }
    } else {
      std::sort(ka.begin(), ka.end());
      std::sort(kt.begin(), kt.end());
    }
    for (size_t i = 0; i < rows; i++) {
      a[i] = Row{ka[i], 0, 0, 0};
      t[i] = Row{kt[i], 0, 0, 0};
    }
  } else {
    std::sort(a.begin(), a.end(), less);
    std::sort(t.begin(), t.end(), less);
  }
  std::vector<Row> s(rows);
  std::vector<size_t> repeated;
  std::vector<bool> used(rows, false);
  size_t ti = 0;
  for (size_t i = 0; i < rows; i++) {
    if (i && a[i] == a[i - 1]) {
      repeated.push_back(i);
      continue;
    }
    while (ti < rows && less(t[ti], a[i])) ti++;   // both sorted: one forward sweep
    if (ti == rows || t[ti] != a[i]) throw std::runtime_error("lookup input value not in the table");
    used[ti] = true;
    s[i] = t[ti++];
  }
  size_t ri = 0;
  for (size_t j = 0; j < rows; j++)
    if (!used[j]) s[repeated[ri++]] = t[j];
  std::memcpy(a_out, a.data(), 32 * rows);
  std::memcpy(s_out, s.data(), 32 * rows);
}

struct Timings { std::map<std::string, double> ms; };

// advice: 3 device columns (Lagrange, n rows; the last 6 rows are overwritten with blinding values)
template <class Transcript>
std::vector<uint8_t> create_proof_with(const ProvingKey& pk, std::vector<DevCol>& advice, const std::vector<Fr>& instances,
                                       Transcript& tr, const Options& opt = Options(), Timings* timings = nullptr) {
  if (advice.size() != NUM_ADVICE) throw std::invalid_argument("create_proof: three advice columns expected");
  for (auto& a : advice)
    if (a.rows != pk.n || !a.p) throw std::invalid_argument("create_proof: advice columns of 2^k rows expected");
  if (instances.size() > pk.usable) throw std::invalid_argument("create_proof: more instances than usable rows");
  const uint32_t k = pk.k, ext_k = pk.ext_k();
  const size_t n = pk.n, u = pk.usable, ne = pk.ext_rows();
  auto clock = std::chrono::steady_clock::now();
  auto lap = [&](const char* name) {  // per-phase wall clock with the device drained, only when asked for
    if (!timings) return;
    hk(hipDeviceSynchronize(), "sync");   // timing mode only
    const auto now = std::chrono::steady_clock::now();
    timings->ms[name] += std::chrono::duration<double, std::milli>(now - clock).count();
    clock = now;
  };
  // SG_PROVER_TRACE=1: the host's own timeline (no synchronisation added): when each step of the driver was reached
  const bool trace = std::getenv("SG_PROVER_TRACE") != nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  std::vector<std::pair<const char*, double>> marks;
  auto mark = [&](const char* what) {
    if (trace) marks.emplace_back(what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count());
  };
  struct TraceDump {
    const std::vector<std::pair<const char*, double>>& m;
    ~TraceDump() {
      double prev = 0;
      for (auto& kv : m) {
        std::fprintf(stderr, "  %9.1f us (+%7.1f)  %s\n", kv.second, kv.second - prev, kv.first);
        prev = kv.second;
      }
    }
  } trace_dump{marks};
  const Fr zero = Fr::zero();
  uint8_t omega_inv_b[32], n_inv_b[32], omega_b[32];
  ck(sg_domain_constant(k, 0, omega_b), "domain constant");
  ck(sg_domain_constant(k, 1, omega_inv_b), "domain constant");
  ck(sg_domain_constant(k, 2, n_inv_b), "domain constant");
  Fr omega, omega_inv;
  std::memcpy(omega.l, omega_b, 32);
  std::memcpy(omega_inv.l, omega_inv_b, 32);
  tr.common_scalar(Fr::from_be_bytes_reduced(pk.vk_digest_be));   // vk.hash_into(transcript)
  for (auto& v : instances) tr.common_scalar(v);

  // blinding values: a 32-byte key from the OS per proof, expanded by ChaCha20 on the device; one stream id per draw
  uint8_t key[32];
  os_random(key, 32);
  uint64_t draws = 0;
  struct Rows { DevCol* col; size_t first, count; };
  auto rand_rows = [&](std::initializer_list<Rows> list) {   // one launch; draw ids as if drawn one after the other
    void* outs[8];
    size_t counts[8];
    uint32_t m = 0;
    for (const Rows& r : list) {
      outs[m] = r.col->at(r.first);
      counts[m++] = r.count;
    }
    ck(sg_fr_random_batch_dev(key, draws + 1, outs, counts, m, main_stream()), "fr_random");
    draws += m;
  };
  // a draw out of turn: the values of draw number `id` (1-based, in the order upstream's prover draws) on stream `st`, for a
  // column that depends on nothing and can be filled while the device is busy with something else.  The draw numbers of
  // the others do not move, so the proof under a fixed key is the one the in-order schedule gives.
  auto rand_at = [&](uint64_t id, DevCol& col, size_t first, size_t count, hipStream_t st) {
    void* outs[1] = {col.at(first)};
    size_t counts[1] = {count};
    ck(sg_fr_random_batch_dev(key, id, outs, counts, 1, st), "fr_random");
  };
  // two side streams: independent latency chains (the transforms of a phase under its commitments, the three grand
  // products, the rotation sets of the multi-open) run next to the main (null) stream; the library keeps its work
  // space per stream.  fork: the side streams wait for everything enqueued on the main stream; join: the reverse
  hipStream_t(&side_streams)[2] = session().side;
  hipEvent_t& ev_fork = session().ev_fork;
  hipEvent_t(&ev_join)[2] = session().ev_join;
  const hipStream_t ms = main_stream();
  if (!side_streams[0]) {
    {
      // the side streams carry work that is needed a phase later (transforms under a commitment job): lowest priority, so
      // that the commitment job's latency-bound kernels on the other streams are dispatched first (SG_SIDE_PRIORITY=0: normal)
      int least = 0, greatest = 0;
      (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
      const char* v = std::getenv("SG_SIDE_PRIORITY");
      const int prio = (v && v[0] == '0') ? 0 : least;
      for (auto& st : side_streams) st = adopt_or_create_stream(prio);
    }
    ev_fork = adopt_or_create_event();
    for (auto& e : ev_join) e = adopt_or_create_event();
  }
  // Everything on the main stream: (a) a proof that is one of several in flight (between sg_commit_combine_begin / _end: the
  // batch driver) -- the other proofs are its concurrency, and every fork / join is three event records and four stream waits,
  // each a marker the runtime's completion thread has to retire (one thread per process: 3.6 ms of CPU per proof at 25 records
  // and 22 waits; without the side streams a batch of 1024 runs 2-7 % faster on a whole host and on a 1/8 share of it,
  // profiles/r04_sweeps/batch_host_cpu_profile.txt); (b) SG_PROVER_SERIAL (development aid: a kernel trace shows every kernel alone)
  const bool serial = std::getenv("SG_PROVER_SERIAL") != nullptr || sg_commit_combining() == 1;
  const hipStream_t side[2] = {serial ? ms : side_streams[0], serial ? ms : side_streams[1]};
  // A proof that ends in an exception (WitnessError, a failed call) may leave kernels behind on the side streams, and its
  // device columns go back to this thread's pool as the stack unwinds -- the next proof would take them while those
  // kernels still write to them.  On the way out by exception the three streams are drained first.
  struct DrainOnUnwind {
    hipStream_t s[3];
    int depth = std::uncaught_exceptions();
    ~DrainOnUnwind() {
      if (std::uncaught_exceptions() > depth)
        for (hipStream_t st : s) (void)hipStreamSynchronize(st);
    }
  } drain_on_unwind{{side[0], side[1], ms}};
  auto fork = [&]() {
    if (serial) return;                  // one stream: nothing to order (and no markers for the runtime to retire)
    hk(hipEventRecord(ev_fork, ms), "event");
    for (auto& st : side) hk(hipStreamWaitEvent(st, ev_fork, 0), "wait");
  };
  auto join = [&]() {
    if (serial) return;
    for (int i = 0; i < 2; i++) {
      hk(hipEventRecord(ev_join[i], side[i]), "event");
      hk(hipStreamWaitEvent(ms, ev_join[i], 0), "wait");
    }
  };
  auto to_coeff_ext = [&](const std::vector<void*>& lag, std::vector<DevCol>& coeff, std::vector<DevCol>& ext, hipStream_t st) {
    std::vector<void*> pc, pe;
    for (size_t i = 0; i < lag.size(); i++) {
      coeff.emplace_back(n);
      ext.emplace_back(ne);
      pc.push_back(coeff.back().p);
      pe.push_back(ext.back().p);
    }
    ck(sg_ntt_fr_batch_oop_dev(lag.data(), pc.data(), pc.size(), omega_inv_b, n_inv_b, k, st), "iNTT batch");   // Lagrange columns stay
    ck(sg_coeff_to_cosets_batch_dev(pc.data(), pe.data(), pc.size(), k, ext_k, QUOTIENT_PIECES, st), "coset NTT batch");
  };
  auto commit_batch = [&](std::vector<void*> cols, std::vector<int> basis) {
    std::vector<uint8_t> out(64 * cols.size());
    ck(sg_commit_batch_mixed_dev(pk.srs, basis.data(), cols.data(), cols.size(), n, main_stream(), out.data()), "commit");
    for (size_t i = 0; i < cols.size(); i++) tr.write_point(out.data() + 64 * i);
  };
  auto commit_points = [&](std::vector<void*> cols, std::vector<int> basis) {   // the same, the points handed back
    std::vector<uint8_t> out(64 * cols.size());
    ck(sg_commit_batch_mixed_dev(pk.srs, basis.data(), cols.data(), cols.size(), n, main_stream(), out.data()), "commit");
    return out;
  };

  // -- 1: advice
  // status words in mapped page-locked memory, written by the kernels themselves and looked at after the commitments (where the
  // host waits anyway): the range check of the advice columns, the verdict of the lookup permutation
  volatile uint32_t* status = reinterpret_cast<volatile uint32_t*>(pinned_small_rows() + 4 * MAIL_STATUS);
  uint8_t* status_dev = pinned_small_dev_row(MAIL_STATUS);
  status[0] = status[1] = 0;
  if (opt.sanity_checks) {
    const void* cols[3] = {advice[0].p, advice[1].p, advice[2].p};
    ck(sg_fr_flag_noncanonical_dev(cols, 3, n, status_dev, main_stream()), "range check of the advice columns");
  }
  // blinding rows of the three advice columns AND of the two permuted lookup columns (draws 1 .. 5, in upstream's order) in one
  // launch: the permutation kernels below write rows below `u` only
  DevCol pin(n), ptab(n);
  rand_rows({{&advice[0], u, n - u}, {&advice[1], u, n - u}, {&advice[2], u, n - u}, {&pin, u, n - u}, {&ptab, u, n - u}});
  DevCol instance_col(n);
  if (instances.size() <= 8) {   // the column = its few values then zeros: one launch, the values travel as kernel arguments
    ck(sg_fr_lincomb_low_dev(nullptr, nullptr, 0, n, instances.empty() ? nullptr : instances[0].bytes(), (uint32_t)instances.size(),
                             instance_col.p, main_stream()), "instance column");
  } else {
    instance_col.zero();
    instance_col.upload(instances.data(), 0, instances.size());
  }
  std::vector<DevCol> co1, ex1;
  fork();
  // under the commitments; issued FIRST: started later (after the lookup's kernels) these transforms ran into the commitment
  // job's latency-bound kernels and cost it more (0.70 -> 0.82 ms) than the earlier start of the lookup kernels gained (50 us)
  to_coeff_ext({advice[0].p, advice[1].p, advice[2].p, instance_col.p}, co1, ex1, side[0]);
  // the random polynomial of phase 3 (draw 9: after the three advice columns, the two permuted columns and the three grand
  // products) depends on nothing: 2^k values drawn here, under phase 1's commitment job, instead of on phase 3's critical path
  DevCol random_poly(n);
  rand_at(9, random_poly, 0, n, side[1]);
  // ... and is committed with phase 1's columns (below) instead of phase 3's: the one dense column among the commitments of phases
  // 1-3 leaves the job that sits on the critical path behind the grand products (phase 3: 1.34 -> 1.0 ms) and joins one that
  // is a latency chain of witness-like columns anyway (phase 1: 1.0 -> 1.24 ms); its point enters the transcript where upstream
  // writes it.  A proof is GPU-bound by now (5.4 ms of kernel time in 5.6 ms of wall clock), so the gain is the fused job's
  // saved front end and reduction, not the overlap: committing the column from a helper thread, concurrently with phase 1's
  // job, was measured and is slower (profiles/r04_sweeps/random_polynomial_early.txt).  SG_PROVER_RANDOM_EARLY=0: upstream's order.
  static const bool random_early = [] { const char* v = std::getenv("SG_PROVER_RANDOM_EARLY"); return !(v && v[0] == '0'); }();
  // -- 2 (computed ahead of its place in the transcript): the lookup's permuted columns.  This circuit's lookup has ONE input
  // and ONE table expression, so the theta-compression is the expression itself and nothing here waits for theta: the two
  // permuted columns are committed in the SAME fused job as the advice columns (one MSM group's latency instead of two) and
  // their points enter the transcript where upstream writes them, after theta has been squeezed.
  const sg_graph g_in = pk.lookup_input.view(), g_gates = pk.gates.view();
  std::vector<void*> fixed_lag_p, adv_lag_p = {advice[0].p, advice[1].p, advice[2].p}, inst_lag_p = {instance_col.p};
  for (auto& c : pk.fixed_lag) fixed_lag_p.push_back(c.p);
  DevCol inp(n);   // (not cleared: the input program writes every row and reads no previous value)
  ck(sg_quotient_gates_dev(inp.p, &g_in, fixed_lag_p.data(), NUM_FIXED, adv_lag_p.data(), NUM_ADVICE, inst_lag_p.data(), 1, nullptr, 0,
                           zero.bytes(), zero.bytes(), zero.bytes(), zero.bytes(), k, k, main_stream()), "lookup input");
  // range tables (a property of the key): on the device, nothing waited for -- the verdict lands in status[1]
  const int prc = pk.table_is_range ? sg_lookup_permute_small_async_dev(inp.p, pk.fixed_lag[4].p, u, pin.p, ptab.p, status_dev + 4, main_stream())
                                    : SG_ERR_UNSUPPORTED;
  if (prc == SG_ERR_UNSUPPORTED) {   // general tables: sort on the host, as upstream does
    DevCol canon(n);
    uint64_t* stage = pinned_rows(3 * n);   // page-locked staging: the three 32 n-byte transfers run at link speed
    uint64_t *h_inp = stage, *h_a = stage + 4 * n, *h_s = stage + 8 * n;
    ck(sg_fr_from_montgomery_dev(inp.p, canon.p, n, main_stream()), "from_montgomery");
    d2h(h_inp, canon.p, 32 * n);
    permute_expression_pair(h_inp, pk.table_rows.data(), u, h_a, h_s);
    pin.upload(h_a, 0, u);
    ptab.upload(h_s, 0, u);
    ck(sg_fr_to_montgomery_dev(pin.p, pin.p, u, main_stream()), "to_montgomery");
    ck(sg_fr_to_montgomery_dev(ptab.p, ptab.p, u, main_stream()), "to_montgomery");
  } else if (prc == SG_ERR_WITNESS) {
    throw WitnessError("lookup input value not in the table");
  } else {
    ck(prc, "lookup permutation");
  }
  // sorted columns: long constant runs -> difference form (sg_commit, basis 2)
  mark("1: lookup columns ready, commit [a0 a1 a2 a' s'] issued");
  // (the random polynomial rides along: see where it is drawn)
  // (advice and permuted-lookup columns of this circuit are witness-like: a few thousand used rows of small values)
  const int SP = SG_BASIS_SPARSE;
  std::vector<uint8_t> pts;
  if (random_early) {
    if (!serial) {
      hk(hipEventRecord(ev_join[1], side[1]), "event");          // the draw of the random polynomial (side stream 1)
      hk(hipStreamWaitEvent(ms, ev_join[1], 0), "wait");
    }
    pts = commit_points({advice[0].p, advice[1].p, advice[2].p, pin.p, ptab.p, random_poly.p}, {1 | SP, 1 | SP, 1 | SP, 2 | SP, 2 | SP, 0});
  } else {
    pts = commit_points({advice[0].p, advice[1].p, advice[2].p, pin.p, ptab.p}, {1 | SP, 1 | SP, 1 | SP, 2 | SP, 2 | SP});
  }
  // (the commitment job has waited for the stream: the status words are final)
  if (status[1] == 1) throw WitnessError("lookup input value not in the table");
  if (status[1]) throw std::runtime_error("lookup permutation: the key's table is not a range table after all");
  if (opt.sanity_checks && status[0]) throw WitnessError("advice words >= r (not canonical Montgomery field elements)");
  mark("1: commitments back");
  for (int i = 0; i < 3; i++) tr.write_point(pts.data() + 64 * i);
  const Fr theta = tr.squeeze();

  lap("1_advice");
  for (int i = 3; i < 5; i++) tr.write_point(pts.data() + 64 * i);
  const Fr beta = tr.squeeze(), gamma = tr.squeeze_again();

  lap("2_lookup");
  // -- 3: grand products, random polynomial
  auto lag_col = [&](uint32_t kind, uint32_t idx) -> void* {
    return kind == SG_VS_ADVICE ? advice[idx].p : kind == SG_VS_FIXED ? pk.fixed_lag[idx].p : instance_col.p;
  };
  const uint32_t perm_kind[6] = {SG_VS_FIXED, SG_VS_ADVICE, SG_VS_ADVICE, SG_VS_FIXED, SG_VS_ADVICE, SG_VS_INSTANCE};
  const uint32_t perm_idx[6] = {2, 0, 1, 3, 2, 0};
  // the three grand products are independent up to one scalar (z1 continues from z0's last usable value)
  std::vector<DevCol> zs;
  zs.emplace_back(n);
  zs.emplace_back(n);
  DevCol lz(n);
  {
    std::vector<void*> vals[2], sig[2];
    for (uint32_t c = 0; c < NUM_SIGMA; c++) {
      vals[c / CHUNK].push_back(lag_col(perm_kind[c], perm_idx[c]));
      sig[c / CHUNK].push_back(pk.sigma_lag[c].p);
    }
    // one batched call (sg_grand_products_dev): the three products share their launches -- one inversion pass instead of
    // three latency chains --, z1 continues from z0's last usable value on the device, and nothing here waits for the device
    std::vector<void*> all_vals, all_sig;
    uint32_t chunk_cols[2];
    for (int c = 0; c < 2; c++) {
      chunk_cols[c] = (uint32_t)vals[c].size();
      all_vals.insert(all_vals.end(), vals[c].begin(), vals[c].end());
      all_sig.insert(all_sig.end(), sig[c].begin(), sig[c].end());
    }
    void* lookup_cols[4] = {inp.p, pk.fixed_lag[4].p, pin.p, ptab.p};
    void* z_out[3] = {zs[0].p, zs[1].p, lz.p};
    // z0[u], z1[u], lz[u] land in mapped host memory, written by the kernels that produce the row: looked at when the commitments are back
    // (cleared first: a value left by an earlier proof of this thread must never pass for this proof's)
    std::memset(pinned_small_rows() + 4 * MAIL_CLOSING, 0, 3 * 32);
    ck(sg_grand_products_closing_dev(all_vals.data(), all_sig.data(), chunk_cols, 2, lookup_cols, 1, beta.bytes(), gamma.bytes(), k, u, z_out,
                                     opt.sanity_checks ? pinned_small_dev_row(MAIL_CLOSING) : nullptr, main_stream()), "grand products");
    mark("3: grand products enqueued");
  }
  uint64_t* closing = pinned_small_rows() + 4 * MAIL_CLOSING;
  rand_rows({{&zs[0], u + 1, n - u - 1}, {&zs[1], u + 1, n - u - 1}, {&lz, u + 1, n - u - 1}});
  draws += 1;                                // draw 9, the random polynomial, was made in phase 1
  std::vector<DevCol> co3, ex3;
  join();                                    // (the random polynomial's stream)
  fork();
  to_coeff_ext({pin.p, ptab.p, zs[0].p, zs[1].p, lz.p}, co3, ex3, side[0]);   // under the commitments
  // the grand products stay constant wherever the ratio is 1 -- all the unused rows: difference form
  mark("3: commit [z0 z1 lz random] issued");
  if (random_early) {   // three piecewise-constant columns in difference form: a sparse job
    commit_batch({zs[0].p, zs[1].p, lz.p}, {2 | SP, 2 | SP, 2 | SP});
    tr.write_point(pts.data() + 64 * 5);                         // the random polynomial's commitment, made in phase 1's job
  } else {
    commit_batch({zs[0].p, zs[1].p, lz.p, random_poly.p}, {2, 2, 2, 0});
  }
  mark("3: commitments back");
  if (opt.sanity_checks) {   // the kernels that wrote them precede the commitment job on the main stream: complete by now
    Fr last;
    std::memcpy(last.l, closing + 4, 32);         // z1[u]: the permutation's last chunk
    if (last != Fr::one()) throw WitnessError("permutation argument not satisfied by the assignment");
    std::memcpy(last.l, closing + 8, 32);         // lz[u]
    if (last != Fr::one()) throw WitnessError("lookup argument not satisfied by the assignment");
  }
  const Fr y = tr.squeeze();
  join();

  lap("3_grand_products");
  // -- 4: quotient
  // the kernels take the coset-major arrays whole (QUOTIENT_PIECES blocks of 2^k rows; a rotation is an index shift of 1 inside
  // a block): one launch each
  DevCol values(ne);
  std::vector<Fr> y_powers;   // the gate program's challenges: sums of powers of y (ProvingKey::gate_challenge_exps)
  {
    uint32_t top = 0;
    for (auto& group : pk.gate_challenge_exps)
      for (uint32_t e : group) top = std::max(top, e);
    std::vector<Fr> pw;   // y^0 .. y^top by one product each (an exponentiation per term was 60 us of host time with the device idle)
    if (top < (1u << 16)) {
      pw.resize((size_t)top + 1);
      pw[0] = Fr::one();
      for (uint32_t e = 1; e <= top; e++) pw[e] = pw[e - 1] * y;
    }
    for (auto& group : pk.gate_challenge_exps) {
      Fr v = Fr::zero();
      for (uint32_t e : group) v = v + (pw.empty() ? y.pow((uint64_t)e) : pw[e]);
      y_powers.push_back(v);
    }
  }
  if (y_powers.empty()) y_powers.push_back(Fr::zero());
  mark("4: challenges of the gate program ready");
  std::vector<DevCol> pieces_col;
  for (uint32_t i = 0; i < QUOTIENT_PIECES; i++) pieces_col.emplace_back(n);
  mark("4: buffers ready");
  {
    // evaluate_h in one call: gates, permutation argument, lookup argument (its input expression on the way) -- one pass over
    // the coset rows for this circuit's programs (sg_quotient_numerator_cosets_dev); `values` needs no clearing
    std::vector<void*> fixed_e, adv_e = {ex1[0].p, ex1[1].p, ex1[2].p}, inst_e = {ex1[3].p};
    for (auto& c : pk.fixed_ext) fixed_e.push_back(c.p);
    std::vector<void*> col_e, sig_e, z_e = {ex3[2].p, ex3[3].p};
    for (uint32_t c = 0; c < NUM_SIGMA; c++) {
      col_e.push_back(perm_kind[c] == SG_VS_ADVICE ? ex1[perm_idx[c]].p : perm_kind[c] == SG_VS_FIXED ? pk.fixed_ext[perm_idx[c]].p : ex1[3].p);
      sig_e.push_back(pk.sigma_ext[c].p);
    }
    ck(sg_quotient_numerator_cosets_dev(values.p, &g_gates, &g_in, fixed_e.data(), NUM_FIXED, adv_e.data(), NUM_ADVICE, inst_e.data(), 1,
                                        y_powers[0].bytes(), (uint32_t)pk.gate_challenge_exps.size(), z_e.data(), 2, col_e.data(), sig_e.data(),
                                        NUM_SIGMA, CHUNK, pk.l0_ext.p, pk.l_last_ext.p, pk.l_active_ext.p, ex3[4].p, ex3[0].p, ex3[1].p,
                                        pk.fixed_ext[4].p, nullptr, beta.bytes(), gamma.bytes(), theta.bytes(), y.bytes(), k, ext_k, QUOTIENT_PIECES,
                                        BLINDING + 1, main_stream()), "quotient numerator");
  }
  mark("4: numerator enqueued");
  std::vector<void*> pieces;
  for (auto& c : pieces_col) pieces.push_back(c.p);
  ck(sg_cosets_to_pieces_dev(values.p, pieces.data(), k, ext_k, QUOTIENT_PIECES, main_stream()), "cosets_to_pieces");
  mark("4: pieces enqueued, commit issued");
  commit_batch(pieces, std::vector<int>(QUOTIENT_PIECES, 0));
  mark("4: commitments back");
  const Fr x = tr.squeeze();
  const uint64_t n_limbs[4] = {n, 0, 0, 0};
  const Fr x_n = x.pow(n_limbs);

  lap("4_quotient");
  // -- 5: evaluations
  std::map<Key, void*> poly;
  for (uint32_t j = 0; j < NUM_ADVICE; j++) poly[{A_, j}] = co1[j].p;
  for (uint32_t j = 0; j < NUM_FIXED; j++) poly[{F_, j}] = pk.fixed_coeff[j].p;
  for (uint32_t j = 0; j < NUM_SIGMA; j++) poly[{SIGMA_, j}] = pk.sigma_coeff[j].p;
  poly[{PIN_, 0}] = co3[0].p;
  poly[{PTAB_, 0}] = co3[1].p;
  poly[{Z_, 0}] = co3[2].p;
  poly[{Z_, 1}] = co3[3].p;
  poly[{LZ_, 0}] = co3[4].p;
  poly[{RANDOM_, 0}] = random_poly.p;
  auto point = [&](int rot) { return rot >= 0 ? x * omega.pow((uint64_t)rot) : x * omega_inv.pow((uint64_t)(-rot)); };
  const auto order = eval_order();
  std::vector<void*> ev_polys;
  std::vector<Fr> ev_points;
  for (auto& q : order) {
    ev_polys.push_back(poly.at(q.key));
    ev_points.push_back(point(q.rot));
  }
  for (uint32_t i = 0; i < QUOTIENT_PIECES; i++) {   // the quotient pieces at x ride along: h(x) = sum_i x^(n i) h_i(x)
    ev_polys.push_back(pieces[i]);
    ev_points.push_back(x);
  }
  mark("5: evaluation list built");
  std::vector<Fr> ev(ev_polys.size());
  ck(sg_fr_eval_poly_batch_dev(ev_polys.data(), n, ev_points[0].bytes(), (uint32_t)ev_polys.size(), main_stream(),
                               reinterpret_cast<uint8_t*>(ev.data())), "evaluations");
  mark("5: evaluations back");
  std::map<std::pair<Key, int>, Fr> evals;
  for (size_t i = 0; i < order.size(); i++) {
    evals[{order[i].key, order[i].rot}] = ev[i];
    tr.write_scalar(ev[i]);
  }
  // h(X) = sum_i x^(n i) h_i(X) is never formed: its one use -- a term of rotation set 1's combination -- takes the five
  // pieces themselves with the weights zeta^j x^(n i) (below)
  std::vector<Fr> xn_pow(QUOTIENT_PIECES);
  xn_pow[0] = Fr::one();
  for (uint32_t i = 1; i < QUOTIENT_PIECES; i++) xn_pow[i] = xn_pow[i - 1] * x_n;
  Fr h_eval = Fr::zero();
  for (uint32_t i = QUOTIENT_PIECES; i-- > 0;) h_eval = h_eval * x_n + ev[order.size() + i];
  auto eval_of = [&](const Key& key, int rot) { return key.kind == H_ ? h_eval : evals.at({key, rot}); };

  lap("5_evaluations");
  // -- 6: SHPLONK
  const Fr zeta = tr.squeeze(), nu = tr.squeeze_again();
  mark("6: zeta, nu squeezed");
  const auto sets = rotation_sets();
  std::vector<DevCol> fs;
  std::vector<std::vector<Fr>> rs;
  // the Lagrange denominators prod_{j != i} (p_i - p_j) of every set depend only on x: one host inversion for all
  // of them (Montgomery's trick) instead of one 254-step exponentiation each
  std::vector<std::vector<Fr>> denom_inv(sets.size());
  {
    std::vector<Fr*> slots;
    for (size_t si = 0; si < sets.size(); si++) {
      const auto& rots = sets[si].rots;
      denom_inv[si].assign(rots.size(), Fr::one());
      for (size_t i = 0; i < rots.size(); i++) {
        for (size_t j = 0; j < rots.size(); j++)
          if (j != i) denom_inv[si][i] = denom_inv[si][i] * (point(rots[i]) - point(rots[j]));
        slots.push_back(&denom_inv[si][i]);
      }
    }
    std::vector<Fr> prefix(slots.size());
    Fr run = Fr::one();
    for (size_t t = 0; t < slots.size(); t++) {
      prefix[t] = run;
      run = run * *slots[t];
    }
    Fr inv = run.inv();
    for (size_t t = slots.size(); t-- > 0;) {
      const Fr v = *slots[t];
      *slots[t] = inv * prefix[t];
      inv = inv * v;
    }
  }
  mark("6: Lagrange denominators inverted");
  // r_i(X) through the set's (points, values) for every set first: host arithmetic on the evaluations alone.  r_i has at most
  // four coefficients: it enters the kernels BY VALUE (sg_fr_lincomb_low_dev), never as a column
  std::vector<std::vector<Fr>> zps(sets.size());
  for (size_t si = 0; si < sets.size(); si++) {
    const auto& set = sets[si];
    std::vector<Fr>& zp = zps[si];
    zp.resize(set.polys.size());
    for (size_t j = 0; j < set.polys.size(); j++) zp[j] = j ? zp[j - 1] * zeta : Fr::one();
    std::vector<Fr> pts, vals;
    for (int r : set.rots) {
      pts.push_back(point(r));
      Fr v = Fr::zero();
      for (size_t j = 0; j < set.polys.size(); j++) v = v + zp[j] * eval_of(set.polys[j], r);
      vals.push_back(v);
    }
    if (pts.size() > 4) throw std::runtime_error("rotation set of more than four points");
    std::vector<Fr> rc(pts.size(), Fr::zero());
    for (size_t i = 0; i < pts.size(); i++) {
      std::vector<Fr> basis = {Fr::one()};
      for (size_t j = 0; j < pts.size(); j++) {
        if (j == i) continue;
        std::vector<Fr> nb(basis.size() + 1, Fr::zero());
        for (size_t t = 0; t < basis.size(); t++) {
          nb[t + 1] = nb[t + 1] + basis[t];
          nb[t] = nb[t] - pts[j] * basis[t];
        }
        basis = nb;
      }
      const Fr scale = vals[i] * denom_inv[si][i];
      for (size_t t = 0; t < basis.size(); t++) rc[t] = rc[t] + scale * basis[t];
    }
    rs.push_back(rc);
  }
  mark("6: r_i interpolated");
  {
    // f_i = q_i - r_i for all five sets in ONE launch (grid.y = set): q_i = the zeta-combination of the set's polynomials, r_i by value
    std::vector<void*> ps, outs;
    std::vector<Fr> cs, lows(sets.size() * 4, Fr::zero());
    std::vector<uint32_t> sizes, n_lows;
    for (size_t si = 0; si < sets.size(); si++) {
      const auto& set = sets[si];
      uint32_t count = 0;
      for (size_t j = 0; j < set.polys.size(); j++) {
        if (set.polys[j].kind == H_) {
          for (uint32_t i = 0; i < QUOTIENT_PIECES; i++, count++) {
            ps.push_back(pieces[i]);
            cs.push_back(zps[si][j] * xn_pow[i]);
          }
        } else {
          ps.push_back(poly.at(set.polys[j]));
          cs.push_back(zps[si][j]);
          count++;
        }
      }
      sizes.push_back(count);
      for (size_t t = 0; t < rs[si].size(); t++) lows[4 * si + t] = -rs[si][t];
      n_lows.push_back((uint32_t)rs[si].size());
      fs.emplace_back(n);
      outs.push_back(fs.back().p);
    }
    ck(sg_fr_lincomb_sets_dev(ps.data(), cs[0].bytes(), sizes.data(), (uint32_t)sets.size(), n, lows[0].bytes(), n_lows.data(), outs.data(),
                              main_stream()), "set lincombs");
  }
  mark("6: set combinations enqueued");
  // f_i / Z_{S_i}: q_i - r_i vanishes on the whole set, and 1 / prod_j (X - p_j) = sum_j c_j / (X - p_j) with
  // c_j = 1 / prod_{t != j} (p_j - p_t) -- the Lagrange denominators already inverted above.  So every division of every
  // set is an independent exact Kate division: ONE batch (three launches for all eleven), and
  // f = sum_i nu^i f_i / Z_{S_i} is one linear combination of the eleven quotients
  DevCol f_all(n);
  std::vector<DevCol> quotients;   // (alive until the proof is done: returning them to the pool here would need a host sync)
  {
    std::vector<void*> div_in, div_out;
    std::vector<Fr> div_pts, weights;
    Fr nu_pow = Fr::one();
    for (size_t si = 0; si < sets.size(); si++) {
      for (size_t j = 0; j < sets[si].rots.size(); j++) {
        quotients.emplace_back(n);
        div_in.push_back(fs[si].p);
        div_out.push_back(quotients.back().p);
        div_pts.push_back(point(sets[si].rots[j]));
        weights.push_back(nu_pow * denom_inv[si][j]);
      }
      nu_pow = nu_pow * nu;
    }
    ck(sg_fr_kate_division_batch_dev(div_in.data(), n, div_pts[0].bytes(), (uint32_t)div_in.size(), div_out.data(), main_stream()),
       "kate division batch");
    ck(sg_fr_lincomb_dev(div_out.data(), weights[0].bytes(), (uint32_t)div_out.size(), n, f_all.p, main_stream()), "f lincomb");
  }
  mark("6: f(X) enqueued, commit issued");
  commit_batch({f_all.p}, {0});
  mark("6: W back");
  const Fr mu = tr.squeeze();
  std::vector<int> all_rots = {ROT_LAST, -1, 0, 1};
  std::map<int, Fr> mu_minus;
  for (int r : all_rots) mu_minus[r] = mu - point(r);
  std::vector<Fr> diffs;
  for (auto& set : sets) {
    Fr d = Fr::one();
    for (int r : all_rots)
      if (std::find(set.rots.begin(), set.rots.end(), r) == set.rots.end()) d = d * mu_minus[r];
    diffs.push_back(d);
  }
  const Fr d0_inv = diffs[0].inv();
  Fr z_s0 = Fr::one();
  for (int r : sets[0].rots) z_s0 = z_s0 * mu_minus[r];
  // L(X) = sum_i scale_i (q_i(X) - r_i(mu)) - Z_{S_0}(mu) f(X), with q_i = f_i + r_i:
  //      = sum_i scale_i f_i(X) - Z_{S_0}(mu) f(X) + [ sum_i scale_i (r_i(X) - r_i(mu)) ]      (the bracket: at most four coefficients)
  std::vector<Fr> coeffs;
  std::vector<Fr> low(4, Fr::zero());
  Fr nu_pow = Fr::one();
  for (size_t i = 0; i < sets.size(); i++) {
    const Fr scale = nu_pow * diffs[i] * d0_inv;
    coeffs.push_back(scale);
    Fr r_at_mu = Fr::zero();
    for (size_t t = rs[i].size(); t-- > 0;) r_at_mu = r_at_mu * mu + rs[i][t];
    for (size_t t = 0; t < rs[i].size(); t++) low[t] = low[t] + scale * rs[i][t];
    low[0] = low[0] - scale * r_at_mu;
    nu_pow = nu_pow * nu;
  }
  std::vector<void*> lp;
  for (auto& f : fs) lp.push_back(f.p);
  lp.push_back(f_all.p);
  coeffs.push_back(-z_s0);
  DevCol l_poly(n), w2(n);
  ck(sg_fr_lincomb_low_dev(lp.data(), coeffs[0].bytes(), (uint32_t)lp.size(), n, low[0].bytes(), (uint32_t)low.size(), l_poly.p,
                           main_stream()), "L lincomb");
  // the remainder L(mu) goes to mapped host memory and is looked at once W' is back: the commitment job is issued behind the
  // division without a host wait in between (a non-zero remainder is a bug in this driver, not an input error)
  std::memset(pinned_small_rows() + 4 * MAIL_REMAINDER, 0xff, 32);   // (not a remainder any kernel writes: a stale zero cannot pass)
  ck(sg_fr_kate_division_rem_dev(l_poly.p, n, mu.bytes(), w2.p, pinned_small_dev_row(MAIL_REMAINDER), main_stream()), "final division");
  mark("6: final quotient enqueued, commit issued");
  commit_batch({w2.p}, {0});
  mark("6: W' back");
  {
    Fr rem;
    std::memcpy(rem.l, pinned_small_rows() + 4 * MAIL_REMAINDER, 32);
    if (!rem.is_zero()) throw std::runtime_error("multi-open linearisation does not vanish at mu");
  }
  lap("6_multiopen");
  return tr.proof;
}
// the Keccak / EVM flavour (what tools/create_proof_main.cpp and the bundles use)
inline std::vector<uint8_t> create_proof(const ProvingKey& pk, std::vector<DevCol>& advice, const std::vector<Fr>& instances,
                                         Timings* timings = nullptr) {
  EvmTranscript tr;
  return create_proof_with(pk, advice, instances, tr, Options(), timings);
}

}  // namespace prover
}  // namespace summa
