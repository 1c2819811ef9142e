// Custom-gate block of halo2's evaluate_h: an interpreter for GraphEvaluator programs (see gates.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/summa_gpu.h"
#include "msm.h"

namespace sg {

struct GateOp {  // one instruction of the compiled program (4 words, read with scalar loads)
  uint32_t w0;   // opcode | kidx << 8 | a_kind << 16 | b_kind << 24   (MULADD: the kidx byte holds the kind of c)
  uint32_t dst;  // LDS slot (low 16 bits); MULADD: the third operand c in the high 16 bits
  uint32_t a, b; // slot / constant index / (LOADCOL: column index, rotation)
};
// G_RED: the same residue below 2p (f29_reduce_small); G_MULADD: a * b + c under one reduction (f29_mul_add)
enum GateOpcode : uint32_t { G_LOADCOL, G_LOADPREV, G_ADD, G_SUB, G_MUL, G_SQR, G_DBL, G_NEG, G_RED, G_MULADD };
enum GateOperandKind : uint32_t { GK_SLOT = 0, GK_CONST = 1 };

struct GateProgram {
  std::vector<GateOp> ops;
  std::vector<uint32_t> const_words;  // 8 words per constant (memory-domain Fr)
  uint32_t n_slots = 0, result_kind = GK_SLOT, result_index = 0;
  uint32_t n_columns = 0;             // fixed ++ advice ++ instance
  std::vector<uint8_t> signature;     // structure bytes this program was lowered from (cache confirmation)
};

// halo2-shaped graph -> compiled program (bound tracking, lazy reductions, slot allocation).
// Returns an empty string on success, a message otherwise.
std::string compile_gates(const sg_graph& g, uint32_t n_fixed, uint32_t n_advice, uint32_t n_instance,
                          const uint8_t* challenges, uint32_t n_challenges, const uint8_t beta[32],
                          const uint8_t gamma[32], const uint8_t theta[32], const uint8_t y[32], GateProgram* out);

// a program the library has straight-line code for (the reference circuit's gate programs and its lookup input): launched with
// columns and constants as kernel arguments -- no blob; returns false (nothing launched) for any other program
bool gates_run_by_value(const GateProgram& p, const void* const* cols, fp_words* d_values, uint32_t k, uint32_t ext_k, hipStream_t stream,
                        uint32_t cosets, hipError_t* err);
// d_blob: [ops][column pointers][constants] as laid out by gates_blob(); values updated in place
size_t gates_blob(const GateProgram& p, const void* const* cols, std::vector<uint8_t>* blob);
// cosets = 0: arrays of 2^ext_k rows in halo2's extended-domain order; cosets = c: coset-major arrays of c * 2^k rows
hipError_t gates_run(const GateProgram& p, const uint8_t* d_blob, fp_words* d_values, uint32_t k, uint32_t ext_k,
                     hipStream_t stream, uint32_t cosets = 0);

}  // namespace sg
