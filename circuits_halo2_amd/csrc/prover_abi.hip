// C ABI of the compiled-host prover (include/summa_prover.h over include/summa_prover.hpp): key registry, argument
// checks, exception -> status mapping.  Host code only; the kernels are behind the sg_* entry points it calls.
#include "../../include/summa_prover.h"

#include <memory>
#include <mutex>

#include "../../include/summa_prover.hpp"

using namespace summa::prover;

namespace {
thread_local char g_sp_err[512] = "";
int sp_fail(int code, const char* what) {
  std::snprintf(g_sp_err, sizeof g_sp_err, "%s", what);
  return code;
}
std::mutex g_keys_mu;
// never destroyed: keys still registered at process exit must not run their destructors after the HIP runtime and
// the threads' sessions are gone
std::map<uint64_t, std::shared_ptr<ProvingKey>>& g_keys = *new std::map<uint64_t, std::shared_ptr<ProvingKey>>();
uint64_t g_next_key = 1;

Graph copy_graph(const sg_graph& g) {
  Graph out;
  out.constants.assign(g.constants, g.constants + 32 * (size_t)g.n_constants);
  out.rotations.assign(g.rotations, g.rotations + g.n_rotations);
  out.calculations.assign(g.calculations, g.calculations + g.n_calculations);
  out.parts.assign(g.horner_parts, g.horner_parts + g.n_horner_parts);
  return out;
}
// what sp_key_create can see of the constraint-system shape from its arguments (include/summa_prover.h lists the rest):
// every value source of a program names an existing column / constant / earlier intermediate, its rotation is one of
// -1, 0, +1 (the multi-open's rotation sets are fixed), the instance column is read at rotation 0 only
const char* graph_shape_error(const sg_graph& g, uint32_t n_challenges) {
  for (uint32_t r = 0; r < g.n_rotations; r++)
    if (g.rotations[r] < -1 || g.rotations[r] > 1) return "a rotation beyond -1 / 0 / +1";
  auto bad = [&](const sg_value_source& v, uint32_t self) -> const char* {
    switch (v.kind) {
      case SG_VS_CONSTANT: return v.index < g.n_constants ? nullptr : "a constant index out of range";
      case SG_VS_INTERMEDIATE: return v.index < self ? nullptr : "an intermediate that is not defined yet";
      case SG_VS_FIXED: case SG_VS_ADVICE: case SG_VS_INSTANCE: {
        const uint32_t lim = v.kind == SG_VS_FIXED ? NUM_FIXED : v.kind == SG_VS_ADVICE ? NUM_ADVICE : 1u;
        if (v.index >= lim) return "a column index beyond this constraint system's 11 fixed / 3 advice / 1 instance columns";
        if (v.rotation >= g.n_rotations) return "a rotation index out of range";
        if (v.kind == SG_VS_INSTANCE && g.rotations[v.rotation] != 0) return "the instance column read at a rotation";
        return nullptr;
      }
      case SG_VS_CHALLENGE: return v.index < n_challenges ? nullptr : "a challenge index out of range";
      case SG_VS_BETA: case SG_VS_GAMMA: case SG_VS_THETA: case SG_VS_Y: case SG_VS_PREVIOUS_VALUE: return nullptr;
      default: return "an unknown value source";
    }
  };
  for (uint32_t i = 0; i < g.n_calculations; i++) {
    const sg_calculation& c = g.calculations[i];
    if (c.op > SG_OP_STORE) return "an unknown calculation";
    if (const char* e = bad(c.a, i)) return e;
    const bool binary = c.op == SG_OP_ADD || c.op == SG_OP_SUB || c.op == SG_OP_MUL || c.op == SG_OP_HORNER;
    if (binary)
      if (const char* e = bad(c.b, i)) return e;
    if (c.op == SG_OP_HORNER) {
      if ((uint64_t)c.parts_offset + c.parts_len > g.n_horner_parts || (c.parts_len && !g.horner_parts)) return "Horner parts out of range";
      for (uint32_t q = 0; q < c.parts_len; q++)
        if (const char* e = bad(g.horner_parts[c.parts_offset + q], i)) return e;
    }
  }
  return nullptr;
}
template <class F>
int guarded(F&& body) {
  try {
    return body();
  } catch (const WitnessError& e) {
    return sp_fail(SG_ERR_WITNESS, e.what());
  } catch (const std::invalid_argument& e) {
    return sp_fail(SG_ERR_INVALID, e.what());
  } catch (const std::bad_alloc&) {
    return sp_fail(SG_ERR_NOMEM, "out of host memory");
  } catch (const std::exception& e) {
    return sp_fail(SG_ERR_HIP, e.what());
  }
}
}  // namespace

extern "C" {

const char* sp_last_error(void) { return g_sp_err; }

int sp_key_create(uint32_t k, uint64_t srs_handle, const void* const* d_fixed_lagrange, const void* const* d_sigma_lagrange,
                  const uint8_t vk_digest_be[32], const sg_graph* gates, const sg_graph* lookup_input,
                  const uint32_t* gate_challenge_exponents, const uint32_t* gate_challenge_counts, uint32_t n_gate_challenges,
                  void* stream, uint64_t* key_out) {
  if (!d_fixed_lagrange || !d_sigma_lagrange || !vk_digest_be || !gates || !lookup_input || !key_out || k < 4 || k > 25)
    return sp_fail(SG_ERR_INVALID, "sp_key_create: bad argument");
  if (n_gate_challenges > 256 || (n_gate_challenges && (!gate_challenge_exponents || !gate_challenge_counts)))
    return sp_fail(SG_ERR_INVALID, "sp_key_create: bad challenge list");
  for (uint32_t i = 0; i < NUM_FIXED; i++)
    if (!d_fixed_lagrange[i]) return sp_fail(SG_ERR_INVALID, "sp_key_create: null fixed column");
  for (uint32_t i = 0; i < NUM_SIGMA; i++)
    if (!d_sigma_lagrange[i]) return sp_fail(SG_ERR_INVALID, "sp_key_create: null permutation column");
  if ((gates->n_constants && !gates->constants) || (gates->n_calculations && !gates->calculations) ||
      (lookup_input->n_calculations && !lookup_input->calculations))
    return sp_fail(SG_ERR_INVALID, "sp_key_create: malformed program");
  if ((gates->n_rotations && !gates->rotations) || (lookup_input->n_rotations && !lookup_input->rotations) ||
      (lookup_input->n_constants && !lookup_input->constants))
    return sp_fail(SG_ERR_INVALID, "sp_key_create: malformed program");
  for (const sg_graph* g : {gates, lookup_input}) {
    if (const char* why = graph_shape_error(*g, g == gates ? n_gate_challenges : 0u)) {
      std::snprintf(g_sp_err, sizeof g_sp_err, "sp_key_create: the %s program does not fit this prover's constraint-system shape: %s",
                    g == gates ? "gate" : "lookup-input", why);
      return SG_ERR_INVALID;
    }
  }
  return guarded([&]() {
    if (sg_bind_thread() != SG_OK) throw std::runtime_error(sg_last_error());   // this thread's own HIP calls: the library's device
    StreamScope scope(static_cast<hipStream_t>(stream));
    const size_t n = (size_t)1 << k;
    auto pk = std::make_shared<ProvingKey>();
    std::memcpy(pk->vk_digest_be, vk_digest_be, 32);
    pk->gates = copy_graph(*gates);
    pk->lookup_input = copy_graph(*lookup_input);
    pk->gate_challenge_exps.clear();
    for (uint32_t i = 0, at = 0; i < n_gate_challenges; at += gate_challenge_counts[i], i++)
      pk->gate_challenge_exps.emplace_back(gate_challenge_exponents + at, gate_challenge_exponents + at + gate_challenge_counts[i]);
    std::vector<DevCol> fixed, sigma;
    auto clone = [&](const void* src) {
      DevCol c(n);
      hk(hipMemcpyAsync(c.p, src, 32 * n, hipMemcpyDeviceToDevice, main_stream()), "D2D");
      return c;
    };
    for (uint32_t i = 0; i < NUM_FIXED; i++) fixed.push_back(clone(d_fixed_lagrange[i]));
    for (uint32_t i = 0; i < NUM_SIGMA; i++) sigma.push_back(clone(d_sigma_lagrange[i]));
    pk->build(k, srs_handle, std::move(fixed), std::move(sigma));
    std::lock_guard<std::mutex> lk(g_keys_mu);
    *key_out = g_next_key++;
    g_keys[*key_out] = pk;
    return (int)SG_OK;
  });
}

int sp_key_destroy(uint64_t key) {
  std::shared_ptr<ProvingKey> pk;
  {
    std::lock_guard<std::mutex> lk(g_keys_mu);
    auto it = g_keys.find(key);
    if (it == g_keys.end()) return sp_fail(SG_ERR_INVALID, "sp_key_destroy: unknown key");
    pk = it->second;
    g_keys.erase(it);
  }
  return guarded([&]() {
    pk.reset();                 // the columns go to this thread's pool ...
    release_column_pool();      // ... and from there back to the device
    return (int)SG_OK;
  });
}

int sp_create_proof(uint64_t key, void* const* d_advice, const uint8_t* instances, uint32_t n_instances, int transcript,
                    int sanity_checks, void* stream, uint8_t* proof_out, size_t proof_cap, size_t* proof_len) {
  if (!d_advice || !proof_out || !proof_len || (n_instances && !instances))
    return sp_fail(SG_ERR_INVALID, "sp_create_proof: null argument");
  if (transcript != SP_TRANSCRIPT_EVM && transcript != SP_TRANSCRIPT_BLAKE2B)
    return sp_fail(SG_ERR_INVALID, "sp_create_proof: unknown transcript");
  std::shared_ptr<ProvingKey> pk;
  {
    std::lock_guard<std::mutex> lk(g_keys_mu);
    auto it = g_keys.find(key);
    if (it == g_keys.end()) return sp_fail(SG_ERR_INVALID, "sp_create_proof: unknown key");
    pk = it->second;
  }
  return guarded([&]() {
    if (sg_bind_thread() != SG_OK) throw std::runtime_error(sg_last_error());   // this thread's own HIP calls: the library's device
    StreamScope scope(static_cast<hipStream_t>(stream));
    std::vector<DevCol> advice;
    for (uint32_t i = 0; i < NUM_ADVICE; i++) {
      if (!d_advice[i]) throw std::invalid_argument("sp_create_proof: null advice column");
      advice.push_back(DevCol::borrow(d_advice[i], pk->n));
    }
    std::vector<Fr> inst(n_instances);
    if (n_instances) std::memcpy(inst.data(), instances, 32 * (size_t)n_instances);
    Options opt;
    opt.sanity_checks = sanity_checks != 0;
    std::vector<uint8_t> proof;
    if (transcript == SP_TRANSCRIPT_EVM) {
      EvmTranscript tr;
      proof = create_proof_with(*pk, advice, inst, tr, opt);
    } else {
      Blake2bTranscript tr;
      proof = create_proof_with(*pk, advice, inst, tr, opt);
    }
    if (proof.size() > proof_cap) throw std::invalid_argument("sp_create_proof: proof buffer too small");
    std::memcpy(proof_out, proof.data(), proof.size());
    *proof_len = proof.size();
    return (int)SG_OK;
  });
}

}  // extern "C"
