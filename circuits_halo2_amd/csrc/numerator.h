// The quotient numerator of a row in ONE pass (see numerator.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "gates.h"
#include "quotient.h"

namespace sg {
static constexpr uint32_t NUM_MAX_CONSTS = 40, NUM_MAX_INPUT_CONSTS = 8, NUM_MAX_COLS = 24;
struct NumeratorArgs {   // kernel argument (by value: no blob upload, nothing to keep alive)
  fp_words* values;                               // out: cosets * 2^k rows, every row written
  const fp_words* cols[NUM_MAX_COLS];             // fixed ++ advice ++ instance, coset-major, as the gate programs index them
  uint32_t consts[NUM_MAX_CONSTS][8];             // the gate program's constant table (constants ++ challenges ++ beta, gamma, theta, y)
  uint32_t input_consts[NUM_MAX_INPUT_CONSTS][8]; // the lookup input program's
  uint32_t n_consts, n_input_consts;
  QuotPermArgs perm;                              // (values unused)
  QuotLookupArgs look;                            // (values, input unused)
};
// true: the pair (gates, lookup input) is one the fused kernel is instantiated for (the reference circuit at N_CURRENCIES 1 .. 4)
bool numerator_fused_available(const GateProgram& gates, const GateProgram& lookup_input);
// launches the fused kernel; hipErrorInvalidValue when numerator_fused_available() is false or the tables do not fit
hipError_t numerator_fused(const GateProgram& gates, const GateProgram& lookup_input, NumeratorArgs& a, hipStream_t stream);
hipError_t numerator_set_side_prio(uint32_t on);
}  // namespace sg
