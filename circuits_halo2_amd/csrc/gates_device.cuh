// Device side of the gate programs, shared by gates.hip (the stand-alone gate kernels) and numerator.hip (the fused
// quotient numerator): kernel arguments, the ahead-of-time program tables and their straight-line evaluation.
#pragma once
#include <cstring>

#include "gates.h"

namespace sg {
typedef Fr29 P;

struct GateArgs {
  fp_words* values;
  const GateOp* ops;
  const fp_words* const* cols;
  const uint32_t* consts;  // 8 words each
  uint32_t n_ops, n_consts, n_slots, result_kind, result_index, k, ext_k;
  // rows of the arrays and the block inside which a rotation wraps: the halo2 layout is ONE block of 2^ext_k rows (a rotation
  // shifts by r * 2^(ext_k - k)); the coset-major layout is `cosets` blocks of 2^k rows (ext_k = k: a rotation shifts by r)
  uint64_t rows, blockmask;
};

// where a straight-line program reads from: column pointers (fixed ++ advice ++ instance, indexed by compile-time constants)
// and the previous values (nullptr: zero)
struct GateSrc {
  const fp_words* const* cols;
  const fp_words* values;
};

// ---- the same, for a program known at compile time (gates_mst_programs.inc: the reference circuit's own gate programs as
// lowered by compile_gates below; tools/gen_gates_programs.py).  Every instruction is a template instantiation: the
// operands are registers (slot[] is indexed by constants only), there is no instruction fetch or decode and no LDS round
// trip per instruction -- what remains of the interpreter's ~50 instructions of overhead per op is the constants' LDS
// reads.  gates_run uses it when the program it is given is word for word one of the tables; SG_GATES_GENERIC=1 keeps
// the interpreter (tests compare the two).
#include "gates_mst_programs.inc"

template <uint32_t KIND, uint32_t IDX, uint32_t NS>
__device__ __forceinline__ f29 fixed_operand(const f29 (&slot)[NS], const uint32_t* s_const) {
  if constexpr (KIND == GK_CONST) {
    f29 r;
#pragma unroll
    for (int q = 0; q < 9; q++) r.l[q] = s_const[IDX * 9 + q];
    return r;
  } else {
    return slot[IDX];
  }
}
template <class PROG, uint32_t PC>
__device__ __forceinline__ void fixed_step(f29 (&slot)[PROG::n_slots], const uint32_t* s_const, const GateSrc& a, size_t row,
                                           size_t mask, uint32_t rot_shift) {
  if constexpr (PC < PROG::n_ops) {
    constexpr GateOp op = PROG::ops[PC];
    constexpr uint32_t code = op.w0 & 0xff, kidx = (op.w0 >> 8) & 0xff, ak = (op.w0 >> 16) & 0xff, bk = op.w0 >> 24;
    constexpr uint32_t dst = op.dst & 0xffff, ci = op.dst >> 16, NS = PROG::n_slots;
    static_assert(dst < NS, "");
    f29 r;
    if constexpr (code == G_LOADCOL) {
      const size_t i = (row & ~mask) | ((row + ((size_t)(int64_t)(int32_t)op.b << rot_shift)) & mask);
      uint32_t w[8];
      fp_words_load(a.cols[op.a] + i, w);
      if constexpr (kidx != 0) r = f29_mul<P>(f29_from_words<0>(w), f29_const<P>(P::r266));
      else r = f29_from_words<5>(w);
    } else if constexpr (code == G_LOADPREV) {
      if (a.values) {
        uint32_t w[8];
        fp_words_load(a.values + row, w);
        r = f29_from_words<5>(w);
      } else {
        r = f29_zero();   // a fresh numerator: the previous value of every row is zero (no memset, no load)
      }
    } else {
      const f29 x = fixed_operand<ak, op.a, NS>(slot, s_const);
      if constexpr (code == G_ADD) r = f29_add(x, fixed_operand<bk, op.b, NS>(slot, s_const));
      else if constexpr (code == G_SUB) r = f29_sub<P, (kidx < 5 ? kidx : 5)>(x, fixed_operand<bk, op.b, NS>(slot, s_const));
      else if constexpr (code == G_MUL) r = f29_mul<P>(x, fixed_operand<bk, op.b, NS>(slot, s_const));
      else if constexpr (code == G_SQR) r = f29_sqr<P>(x);
      else if constexpr (code == G_DBL) r = f29_add(x, x);
      else if constexpr (code == G_NEG) r = f29_sub<P, (kidx < 5 ? kidx : 5)>(f29_zero(), x);
      else if constexpr (code == G_MULADD)
        r = f29_mul_add<P>(x, fixed_operand<bk, op.b, NS>(slot, s_const), fixed_operand<kidx, ci, NS>(slot, s_const));
      else r = f29_reduce_small<P>(x);   // G_RED
    }
    slot[dst] = r;
    fixed_step<PROG, PC + 1>(slot, s_const, a, row, mask, rot_shift);
  }
}

// result of PROG at `row` in the 2^261 (hat) domain, below 2p
template <class PROG>
__device__ __forceinline__ f29 gates_fixed_eval(const GateSrc& src, const uint32_t* s_const, size_t row, size_t mask, uint32_t rot_shift) {
  f29 slot[PROG::n_slots];
#pragma unroll
  for (uint32_t i = 0; i < PROG::n_slots; i++) slot[i] = f29_zero();
  fixed_step<PROG, 0>(slot, s_const, src, row, mask, rot_shift);
  return fixed_operand<PROG::result_kind, PROG::result_index, PROG::n_slots>(slot, s_const);
}
template <class PROG>
static bool is_program(const GateProgram& p) {
  return p.ops.size() == PROG::n_ops && p.n_slots == PROG::n_slots && p.result_kind == PROG::result_kind &&
         p.result_index == PROG::result_index && std::memcmp(p.ops.data(), PROG::ops, sizeof(GateOp) * PROG::n_ops) == 0;
}
}  // namespace sg
