// halo2's `evaluate_h` for one row in ONE pass: the custom gates, the permutation argument and the lookup
// argument of the quotient numerator used to be three kernels (plus one for the lookup's input expression and a memset) that
// each read and wrote `values` and re-read the same advice / fixed / selector columns of the 5 * 2^k coset rows.  The
// counters of round 5 (profiles/r05a_proof_budget.json) show what that costs: the permutation kernel's waves are parked on
// memory half of their cycles, the three kernels move 950 MB where one pass moves 600.  Here a thread keeps its row's running
// value in registers through all three blocks:
//     acc = gates(row)                       straight-line program (gates_device.cuh), previous value = 0
//     acc = permutation terms(acc, row)      quotient_device.cuh
//     acc = lookup terms(acc, input(row))    the input expression is a second straight-line program, evaluated in place
//     values[row] = acc                      the only store
// Same field operations in the same order as the three kernels, so the stored words are identical (tests compare).  Only for
// program pairs known at compile time (the reference circuit's, N_CURRENCIES 1 .. 4); sg_quotient_numerator_cosets_dev
// runs the kernels one after the other for anything else.  Constants and column pointers travel as kernel arguments.
#include "numerator.h"

#include <cstdio>
#include <cstdlib>

#include "gates_device.cuh"
#include "quotient_device.cuh"
#include "side_prio.cuh"

namespace sg {
SG_DEFINE_SIDE_PRIO_SETTER(numerator_set_side_prio)

template <class GATES, class INPUT>
__global__ void __launch_bounds__(256) numerator_fused_kernel(NumeratorArgs a) {
  side_kernel_prio();
  __shared__ uint32_t s_gc[NUM_MAX_CONSTS][9];          // gate constants, 2^261 domain
  __shared__ uint32_t s_ic[NUM_MAX_INPUT_CONSTS][9];    // lookup input constants
  __shared__ uint32_t s_pc[QUOT_PERM_CONSTS][9];
  __shared__ uint32_t s_lc[QUOT_LOOKUP_CONSTS][9];
  const uint32_t tid = threadIdx.x;
  // constants: threads 0 .. 6 the permutation's (one of them a power), 64 .. the gate tables, 128 .. the lookup's
  quot_perm_setup(a.perm, s_pc, (size_t)blockIdx.x * blockDim.x);
  if (tid >= 64 && tid < 64 + a.n_consts) {
    const f29 v = f29_words_to_r261<P>(a.consts[tid - 64]);
#pragma unroll
    for (int q = 0; q < 9; q++) s_gc[tid - 64][q] = v.l[q];
  }
  if (tid >= 128 && tid < 128 + a.n_input_consts) {
    const f29 v = f29_words_to_r261<P>(a.input_consts[tid - 128]);
#pragma unroll
    for (int q = 0; q < 9; q++) s_ic[tid - 128][q] = v.l[q];
  }
  if (tid >= 192 && tid < 196) {
    // (quot_lookup_setup indexes by threadIdx.x < 4: the same conversions from another quarter of the workgroup)
    const uint32_t t = tid - 192;
    f29 v;
    if (t == 0) v = f29_from_words<0>(a.look.beta);
    else if (t == 1) v = f29_from_words<0>(a.look.gamma);
    else if (t == 2) v = f29_words_to_r261<P>(a.look.y);
    else v = f29_const<P>(P::r256);
#pragma unroll
    for (int q = 0; q < 9; q++) s_lc[t][q] = v.l[q];
  }
  __syncthreads();
  const size_t n_ext = (size_t)a.perm.cosets << a.perm.k;
  const size_t row = (size_t)blockIdx.x * blockDim.x + tid;
  if (row >= n_ext) return;
  const size_t mask = ((size_t)1 << a.perm.k) - 1;     // coset-major: a rotation wraps inside its block of 2^k rows
  const GateSrc src{a.cols, nullptr};                  // previous value: zero (a fresh numerator)
  // gates: hat -> memory domain (what gates_fixed_kernel stores and quot_perm_kernel loads back)
  f29 acc = f29_mul<P>(gates_fixed_eval<GATES>(src, &s_gc[0][0], row, mask, 0), f29_const<P>(P::r256));
  acc = quot_perm_terms(a.perm, s_pc, acc, row);
  const f29 input = f29_mul<P>(gates_fixed_eval<INPUT>(src, &s_ic[0][0], row, mask, 0), f29_const<P>(P::r256));
  acc = quot_lookup_terms(a.look, s_lc, acc, input, row);
  f29_store_canonical<P>(a.values + row, acc);
}

template <class F>
static bool for_known_pair(const GateProgram& g, const GateProgram& in, F&& f) {
  if (!is_program<MstLookupInput>(in)) return false;
  if (is_program<MstGatesNc2>(g)) return f(MstGatesNc2{}, 2), true;
  if (is_program<MstGatesNc1>(g)) return f(MstGatesNc1{}, 1), true;
  if (is_program<MstGatesNc3>(g)) return f(MstGatesNc3{}, 3), true;
  if (is_program<MstGatesNc4>(g)) return f(MstGatesNc4{}, 4), true;
  return false;
}
bool numerator_fused_available(const GateProgram& gates, const GateProgram& lookup_input) {
  if (gates.const_words.size() / 8 > NUM_MAX_CONSTS || lookup_input.const_words.size() / 8 > NUM_MAX_INPUT_CONSTS) return false;
  if (gates.n_columns > NUM_MAX_COLS || lookup_input.n_columns != gates.n_columns) return false;
  return for_known_pair(gates, lookup_input, [](auto, int) {});
}
hipError_t numerator_fused(const GateProgram& gates, const GateProgram& lookup_input, NumeratorArgs& a, hipStream_t stream) {
  if (!numerator_fused_available(gates, lookup_input) || a.perm.cosets == 0 || a.perm.ext_k != a.perm.k || a.look.ext_k != a.look.k)
    return hipErrorInvalidValue;
  a.n_consts = (uint32_t)(gates.const_words.size() / 8);
  a.n_input_consts = (uint32_t)(lookup_input.const_words.size() / 8);
  std::memcpy(a.consts, gates.const_words.data(), gates.const_words.size() * sizeof(uint32_t));
  std::memcpy(a.input_consts, lookup_input.const_words.data(), lookup_input.const_words.size() * sizeof(uint32_t));
  const size_t n_ext = (size_t)a.perm.cosets << a.perm.k;
  const unsigned blocks = (unsigned)((n_ext + 255) / 256);
  for_known_pair(gates, lookup_input, [&](auto tag, int nc) {
    using G = decltype(tag);
    if (std::getenv("SG_GATES_DEBUG"))
      std::fprintf(stderr, "gates: ahead-of-time program MstGatesNc%d inside the one-pass numerator, %u blocks\n", nc, blocks);
    numerator_fused_kernel<G, MstLookupInput><<<blocks, 256, 0, stream>>>(a);
  });
  return hipGetLastError();
}
}  // namespace sg
