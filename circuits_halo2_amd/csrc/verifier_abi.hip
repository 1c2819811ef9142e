// sp_verify_proof (include/summa_prover.h): halo2's `verify_proof::<KZGCommitmentScheme<Bn256>, VerifierSHPLONK, _, _,
// SingleStrategy>` for the constraint system of MstInclusionCircuit, as compiled host code -- what `full_verifier` and
// `create_proof_checked` run on every proof [REF zk_prover/src/circuits/utils.rs:110-131, 181-193].  Twin of
// circuits_halo2_amd/verifier.py (same steps, same order; tests compare the two on accepted and rejected proofs).
//
// Host side: transcript replay, Lagrange / instance evaluations, the constraint polynomials at x from the product's own
// expression list (summa::circuit::gates), SHPLONK's scalars.  Group side: every commitment enters the final check
// linearly, so the left-hand side is ONE multi-scalar multiplication of 37 points on the device (sg_msm_g1) followed by
// the two-pairing check (sg_pairing_check):
//     e( sum_i c_i C_i - r G - Z_{S_0}(mu) W + mu W', [1]_2 ) * e( -W', [s]_2 ) == 1
// with the scaling by 1 / Z_{T \ S_0}(mu) of the reference's generated verifier [REF contracts/src/InclusionVerifier.sol:1025-1402].
#include "../../include/summa_prover.h"

#include <map>
#include <tuple>

#include "../../include/summa_circuit.hpp"
#include "host_curve.h"

using namespace summa::prover;
using sg::host::Fq;

namespace {
thread_local char g_sv_err[256] = "";

struct Reject : std::runtime_error {   // the proof (not the call) is at fault: verification answers "no"
  using std::runtime_error::runtime_error;
};

// ---- Fq helpers: canonical big- / little-endian bytes <-> Montgomery words
const Fq FQ_R2{{0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}};
bool fq_from_canonical(const uint64_t c[4], Fq* out) {
  if (Fq::geq_p(c)) return false;
  *out = Fq{{c[0], c[1], c[2], c[3]}} * FQ_R2;
  return true;
}
void fq_to_canonical(const Fq& a, uint64_t out[4]) {
  const Fq c = a * Fq{{1, 0, 0, 0}};
  std::memcpy(out, c.v, 32);
}
bool on_curve(const Fq& x, const Fq& y) {
  const Fq three = Fq::one() + Fq::one() + Fq::one();
  return y.sqr() == x.sqr() * x + three;
}
Fq fq_pow(const Fq& a, const uint64_t e[4]) {
  Fq acc = Fq::one();
  for (int i = 255; i >= 0; i--) {
    acc = acc.sqr();
    if ((e[i >> 6] >> (i & 63)) & 1) acc = acc * a;
  }
  return acc;
}
struct Point {   // affine, Montgomery words, as the C ABI takes points
  uint8_t b[64];
};
Point make_point(const Fq& x, const Fq& y) {
  Point p;
  std::memcpy(p.b, x.v, 32);
  std::memcpy(p.b + 32, y.v, 32);
  return p;
}

// ---- transcript + proof reader of either flavour (`Keccak256Transcript::read_*` / `Blake2bRead::read_*`)
struct Reader {
  const uint8_t* proof;
  size_t len, pos = 0;
  bool evm;
  EvmTranscript evm_tr;
  Blake2bTranscript b2_tr;
  const uint8_t* take(size_t size) {
    if (pos + size > len) throw Reject("proof too short");
    const uint8_t* out = proof + pos;
    pos += size;
    return out;
  }
  void common_scalar(const Fr& v) {
    if (evm) evm_tr.common_scalar(v);
    else b2_tr.common_scalar(v);
  }
  Fr squeeze() { return evm ? evm_tr.squeeze() : b2_tr.squeeze(); }
  Point read_point() {
    Fq x, y;
    if (evm) {   // x || y, canonical big-endian
      const uint8_t* raw = take(64);
      uint64_t c[2][4];
      for (int q = 0; q < 2; q++)
        for (int i = 0; i < 4; i++) {
          uint64_t w = 0;
          for (int j = 0; j < 8; j++) w = (w << 8) | raw[32 * q + 8 * (3 - i) + j];
          c[q][i] = w;
        }
      if (!fq_from_canonical(c[0], &x) || !fq_from_canonical(c[1], &y) || !on_curve(x, y)) throw Reject("commitment not on the curve");
      evm_tr.buf.insert(evm_tr.buf.end(), raw, raw + 64);
      evm_tr.squeezed = false;
    } else {     // compressed: x little-endian, bit 6 of the last byte = parity of y, bit 7 = infinity
      const uint8_t* enc = take(32);
      if (enc[31] & 0x80) throw Reject("point at infinity");
      uint8_t xb[32];
      std::memcpy(xb, enc, 32);
      xb[31] &= 0x3f;
      uint64_t c[4];
      std::memcpy(c, xb, 32);
      if (!fq_from_canonical(c, &x)) throw Reject("x coordinate not reduced");
      const Fq three = Fq::one() + Fq::one() + Fq::one();
      const Fq y2 = x.sqr() * x + three;
      // q = 3 mod 4: sqrt = y2^((q + 1) / 4)
      static constexpr uint64_t E[4] = {0x4f082305b61f3f52ULL, 0x65e05aa45a1c72a3ULL, 0x6e14116da0605617ULL, 0x0c19139cb84c680aULL};
      y = fq_pow(y2, E);
      if (!(y.sqr() == y2)) throw Reject("not on the curve");
      uint64_t yc[4];
      fq_to_canonical(y, yc);
      if ((yc[0] & 1) != (uint64_t)((enc[31] >> 6) & 1)) y = Fq::zero() - y;
      uint8_t msg[65];
      msg[0] = 1;
      std::memcpy(msg + 1, xb, 32);
      fq_to_canonical(y, yc);
      std::memcpy(msg + 33, yc, 32);
      b2_tr.state.update(msg, 65);
    }
    return make_point(x, y);
  }
  Fr read_scalar() {
    const uint8_t* raw = take(32);
    uint64_t c[4];
    if (evm) {
      for (int i = 0; i < 4; i++) {
        uint64_t w = 0;
        for (int j = 0; j < 8; j++) w = (w << 8) | raw[8 * (3 - i) + j];
        c[i] = w;
      }
    } else {
      std::memcpy(c, raw, 32);
    }
    if (Fr::geq_p(c)) throw Reject("scalar not reduced");
    const Fr v = Fr::from_canonical_limbs(c);
    common_scalar(v);
    return v;
  }
};

using EvalKey = std::tuple<int, uint32_t, int>;   // (Kind, index, rotation)

Fr inv_or_reject(const Fr& a) {
  if (a.is_zero()) throw Reject("a challenge landed on the evaluation domain");
  return a.inv();
}

// the constraint polynomials at x, folded with y in the constraint system's order (gates, permutation, lookup), / (x^n - 1)
Fr expected_h_eval(const std::map<EvalKey, Fr>& evals, const Fr& beta, const Fr& gamma, const Fr& y, const Fr& x, const Fr& x_n,
                   const Fr& l_0, const Fr& l_last, const Fr& l_blind, const Fr& instance_eval, uint32_t n_currencies) {
  namespace ci = summa::circuit;
  const Fr one = Fr::one();
  const Fr active = one - l_last - l_blind;
  auto ev = [&](int kind, uint32_t index, int rot) -> const Fr& {
    auto it = evals.find(EvalKey{kind, index, rot});
    if (it == evals.end()) throw std::runtime_error("verifier: evaluation missing from the query list");
    return it->second;
  };
  auto query = [&](uint32_t kind, uint32_t column, int rot) -> Fr {
    if (kind == SG_VS_INSTANCE) return instance_eval;
    return ev(kind == SG_VS_ADVICE ? A_ : F_, column, rot);
  };
  std::map<const ci::Expr*, Fr> memo;   // the chips share subexpressions (s-boxes, selectors)
  std::function<Fr(const ci::E&)> eval = [&](const ci::E& e) -> Fr {
    auto it = memo.find(e.get());
    if (it != memo.end()) return it->second;
    Fr v;
    switch (e->op) {
      case ci::Expr::CONST: v = e->value; break;
      case ci::Expr::QUERY: v = query(e->kind, e->column, e->rotation); break;
      case ci::Expr::ADD: v = eval(e->a) + eval(e->b); break;
      case ci::Expr::SUB: v = eval(e->a) - eval(e->b); break;
      default: v = eval(e->a) * eval(e->b); break;
    }
    memo.emplace(e.get(), v);
    return v;
  };
  std::vector<Fr> terms;
  for (const ci::E& g : ci::gates(n_currencies)) terms.push_back(eval(g));
  // permutation argument: columns (f2, a0, a1, f3, a2, instance) in chunks of CHUNK
  const uint32_t perm_kind[NUM_SIGMA] = {SG_VS_FIXED, SG_VS_ADVICE, SG_VS_ADVICE, SG_VS_FIXED, SG_VS_ADVICE, SG_VS_INSTANCE};
  const uint32_t perm_idx[NUM_SIGMA] = {2, 0, 1, 3, 2, 0};
  const uint32_t chunks = (NUM_SIGMA + CHUNK - 1) / CHUNK, last = chunks - 1;
  auto z = [&](uint32_t j, int rot = 0) -> const Fr& { return ev(Z_, j, rot); };
  terms.push_back(l_0 * (one - z(0)));
  terms.push_back(l_last * (z(last) * z(last) - z(last)));
  for (uint32_t j = 1; j < chunks; j++) terms.push_back(l_0 * (z(j) - z(j - 1, ROT_LAST)));
  const Fr delta = Fr::from_u64(7).pow((uint64_t)1 << 28);
  Fr shift = beta * x;
  for (uint32_t j = 0, col = 0; j < chunks; j++) {
    Fr left = z(j, 1), right = z(j);
    for (uint32_t c = 0; c < CHUNK && col < NUM_SIGMA; c++, col++) {
      const Fr v = query(perm_kind[col], perm_idx[col], 0);
      left = left * (v + beta * ev(SIGMA_, col, 0) + gamma);
      right = right * (v + shift + gamma);
      shift = shift * delta;
    }
    terms.push_back((left - right) * active);
  }
  // the lookup: input f5 (a0 - 2^8 a0_next), table f4
  const Fr inp = ev(F_, 5, 0) * (ev(A_, 0, 0) - ev(A_, 0, 1) * Fr::from_u64(256)), tab = ev(F_, 4, 0);
  const Fr &lz = ev(LZ_, 0, 0), &lz_next = ev(LZ_, 0, 1), &pin = ev(PIN_, 0, 0), &pin_prev = ev(PIN_, 0, -1), &ptab = ev(PTAB_, 0, 0);
  terms.push_back(l_0 * (one - lz));
  terms.push_back(l_last * (lz * lz - lz));
  terms.push_back(active * (lz_next * (pin + beta) * (ptab + gamma) - lz * (inp + beta) * (tab + gamma)));
  terms.push_back(l_0 * (pin - ptab));
  terms.push_back(active * (pin - ptab) * (pin - pin_prev));
  Fr acc = Fr::zero();
  for (const Fr& t : terms) acc = acc * y + t;
  return acc * inv_or_reject(x_n - one);
}

struct VerifyingKeyView {
  uint32_t k, n_currencies;
  const uint8_t* digest_be;
  const uint8_t* fixed_comms;         // NUM_FIXED x 64 B
  const uint8_t* permutation_comms;   // NUM_SIGMA x 64 B
  const uint8_t *g2, *s_g2;           // 128 B each
};

bool verify(const VerifyingKeyView& vk, const uint8_t* proof, size_t len, const std::vector<Fr>& instances, bool evm) {
  const uint32_t k = vk.k;
  const size_t n = (size_t)1 << k;
  if (instances.size() > n - (BLINDING + 1)) return false;      // halo2: Error::InstanceTooLarge
  Reader rd{proof, len, 0, evm, {}, {}};
  rd.common_scalar(Fr::from_be_bytes_reduced(vk.digest_be));
  for (const Fr& v : instances) rd.common_scalar(v);
  std::map<Key, Point> comms;
  for (uint32_t j = 0; j < NUM_ADVICE; j++) comms[Key{A_, j}] = rd.read_point();
  const Fr theta = rd.squeeze();
  (void)theta;   // this circuit's lookup has one input and one table expression: nothing is compressed with theta
  comms[Key{PIN_, 0}] = rd.read_point();
  comms[Key{PTAB_, 0}] = rd.read_point();
  const Fr beta = rd.squeeze(), gamma = rd.squeeze();
  comms[Key{Z_, 0}] = rd.read_point();
  comms[Key{Z_, 1}] = rd.read_point();
  comms[Key{LZ_, 0}] = rd.read_point();
  comms[Key{RANDOM_, 0}] = rd.read_point();
  const Fr y = rd.squeeze();
  std::vector<Point> pieces;
  for (uint32_t j = 0; j < QUOTIENT_PIECES; j++) pieces.push_back(rd.read_point());
  const Fr x = rd.squeeze();
  std::map<EvalKey, Fr> evals;
  for (const Query& q : eval_order()) evals[EvalKey{q.key.kind, q.key.index, q.rot}] = rd.read_scalar();
  const Fr zeta = rd.squeeze(), nu = rd.squeeze();
  const Point w = rd.read_point();
  const Fr mu = rd.squeeze();
  const Point w2 = rd.read_point();
  if (rd.pos != len) return false;

  // Lagrange evaluations: l_0, l_last, l_blind (the five blinding rows), the instance column at x
  uint8_t omega_b[32], n_inv_b[32];
  ck(sg_domain_constant(k, 0, omega_b), "domain constant");
  ck(sg_domain_constant(k, 2, n_inv_b), "domain constant");
  Fr omega, n_inv;
  std::memcpy(omega.l, omega_b, 32);
  std::memcpy(n_inv.l, n_inv_b, 32);
  const Fr one = Fr::one();
  const Fr x_n = x.pow((uint64_t)n);
  const Fr common = (x_n - one) * n_inv;
  auto omega_pow = [&](int i) { return omega.pow((uint64_t)(((long long)i % (long long)n + (long long)n) % (long long)n)); };
  auto li = [&](int i) {
    const Fr wi = omega_pow(i);
    return common * wi * inv_or_reject(x - wi);
  };
  const Fr l_0 = li(0), l_last = li(ROT_LAST);
  Fr l_blind = Fr::zero();
  for (int i = ROT_LAST + 1; i < 0; i++) l_blind = l_blind + li(i);
  Fr instance_eval = Fr::zero();
  {
    Fr wi = one;
    for (size_t i = 0; i < instances.size(); i++) {
      instance_eval = instance_eval + common * wi * inv_or_reject(x - wi) * instances[i];
      wi = wi * omega;
    }
  }
  const Fr h_eval = expected_h_eval(evals, beta, gamma, y, x, x_n, l_0, l_last, l_blind, instance_eval, vk.n_currencies);
  for (uint32_t j = 0; j < NUM_FIXED; j++) std::memcpy(comms[Key{F_, j}].b, vk.fixed_comms + 64 * j, 64);
  for (uint32_t j = 0; j < NUM_SIGMA; j++) std::memcpy(comms[Key{SIGMA_, j}].b, vk.permutation_comms + 64 * j, 64);

  // SHPLONK: per rotation set, the zeta-combination of its polynomials, interpolated through the claimed values and evaluated
  // at mu; sets weighted by nu^i Z_{T \ S_i}(mu) / Z_{T \ S_0}(mu)
  const std::vector<RotationSet> sets = rotation_sets();
  std::map<int, Fr> point, mu_minus;
  for (const RotationSet& s : sets)
    for (int r : s.rots)
      if (!point.count(r)) {
        point[r] = x * omega_pow(r);
        mu_minus[r] = mu - point[r];
      }
  std::vector<Fr> outside;
  for (const RotationSet& s : sets) {
    Fr d = one;
    for (auto& kv : mu_minus)
      if (std::find(s.rots.begin(), s.rots.end(), kv.first) == s.rots.end()) d = d * kv.second;
    outside.push_back(d);
  }
  const Fr norm0 = inv_or_reject(outside[0]);
  Fr z_s0 = one;
  for (int r : sets[0].rots) z_s0 = z_s0 * mu_minus[r];
  std::map<Key, Fr> coeff;   // commitment -> its scalar in the final multi-scalar multiplication
  Fr r_eval = Fr::zero(), nu_pow = one;
  for (size_t si = 0; si < sets.size(); si++) {
    const RotationSet& s = sets[si];
    std::vector<Fr> weights;   // barycentric weights of the set's points, evaluated at mu
    Fr wsum = Fr::zero();
    for (int r : s.rots) {
      Fr den = mu_minus[r];
      for (int r2 : s.rots)
        if (r2 != r) den = den * (point[r] - point[r2]);
      weights.push_back(inv_or_reject(den));
      wsum = wsum + weights.back();
    }
    const Fr total = inv_or_reject(wsum);
    const Fr scale = nu_pow * outside[si] * norm0;
    Fr zeta_pow = one;
    for (const Key& key : s.polys) {
      Fr at_mu = Fr::zero();
      for (size_t t = 0; t < s.rots.size(); t++) {
        const Fr value = key.kind == H_ ? h_eval : evals.at(EvalKey{key.kind, key.index, s.rots[t]});
        at_mu = at_mu + weights[t] * value;
      }
      at_mu = at_mu * total;
      r_eval = r_eval + scale * zeta_pow * at_mu;
      coeff[key] = (coeff.count(key) ? coeff[key] : Fr::zero()) + scale * zeta_pow;
      zeta_pow = zeta_pow * zeta;
    }
    nu_pow = nu_pow * nu;
  }
  std::vector<uint8_t> points, scalars;
  auto push = [&](const Point& p, const Fr& c) {
    points.insert(points.end(), p.b, p.b + 64);
    scalars.insert(scalars.end(), c.bytes(), c.bytes() + 32);
  };
  for (auto& kv : coeff) {
    if (kv.first.kind == H_) {   // h(X) = sum_j x^(n j) h_j(X)
      Fr xp = one;
      for (uint32_t j = 0; j < QUOTIENT_PIECES; j++) {
        push(pieces[j], kv.second * xp);
        xp = xp * x_n;
      }
    } else {
      push(comms.at(kv.first), kv.second);
    }
  }
  const Fq gx = Fq::one(), gy = Fq::one() + Fq::one();
  push(make_point(gx, gy), -r_eval);
  push(w, -z_s0);
  push(w2, mu);
  uint8_t g1[128], g2[256];
  const size_t count = scalars.size() / 32;
  if (sg_msm_g1(scalars.data(), points.data(), count, g1) != SG_OK) throw std::runtime_error(std::string("verifier: msm: ") + sg_last_error());
  {   // -W'
    Fq w2y;
    std::memcpy(w2y.v, w2.b + 32, 32);
    const Fq neg = Fq::zero() - w2y;
    std::memcpy(g1 + 64, w2.b, 32);
    std::memcpy(g1 + 96, neg.v, 32);
  }
  std::memcpy(g2, vk.g2, 128);
  std::memcpy(g2 + 128, vk.s_g2, 128);
  int ok = 0;
  const int rc = sg_pairing_check(g1, g2, 2, &ok);
  if (rc == SG_ERR_INVALID) return false;   // a coordinate not reduced / a point off the curve: the data's fault
  if (rc != SG_OK) throw std::runtime_error(std::string("verifier: pairing: ") + sg_last_error());
  return ok == 1;
}
}  // namespace

extern "C" {

const char* sp_verify_last_error(void) { return g_sv_err; }

int sp_verify_proof(uint32_t k, uint32_t n_currencies, const uint8_t vk_digest_be[32], const uint8_t* fixed_comms,
                    const uint8_t* permutation_comms, const uint8_t g2[128], const uint8_t s_g2[128], const uint8_t* proof,
                    size_t proof_len, const uint8_t* instances, uint32_t n_instances, int transcript, int* accepted) {
  if (!vk_digest_be || !fixed_comms || !permutation_comms || !g2 || !s_g2 || !accepted || (proof_len && !proof) ||
      (n_instances && !instances) || k < 4 || k > 25 || n_currencies == 0 || n_currencies > 64) {
    std::snprintf(g_sv_err, sizeof g_sv_err, "sp_verify_proof: bad argument");
    return SG_ERR_INVALID;
  }
  if (transcript != SP_TRANSCRIPT_EVM && transcript != SP_TRANSCRIPT_BLAKE2B) {
    std::snprintf(g_sv_err, sizeof g_sv_err, "sp_verify_proof: unknown transcript");
    return SG_ERR_INVALID;
  }
  *accepted = 0;
  (void)sg_bind_thread();
  try {
    std::vector<Fr> inst(n_instances);
    for (uint32_t i = 0; i < n_instances; i++) {
      std::memcpy(inst[i].l, instances + 32 * (size_t)i, 32);
      if (Fr::geq_p(inst[i].l)) return SG_OK;          // not a field element: rejected
    }
    const VerifyingKeyView vk{k, n_currencies, vk_digest_be, fixed_comms, permutation_comms, g2, s_g2};
    *accepted = verify(vk, proof, proof_len, inst, transcript == SP_TRANSCRIPT_EVM) ? 1 : 0;
    return SG_OK;
  } catch (const Reject&) {
    return SG_OK;                                         // *accepted stays 0
  } catch (const std::bad_alloc&) {
    std::snprintf(g_sv_err, sizeof g_sv_err, "out of host memory");
    return SG_ERR_NOMEM;
  } catch (const std::exception& e) {
    std::snprintf(g_sv_err, sizeof g_sv_err, "%s", e.what());
    return SG_ERR_HIP;
  }
}

}  // extern "C"
