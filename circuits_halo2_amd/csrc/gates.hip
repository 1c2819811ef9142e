// Custom-gate block of the quotient numerator: values[row] <- program(row) for every
// row of the extended coset, where the program is halo2's GraphEvaluator (plonk/evaluation.rs of the
// pinned summa-dev/halo2 fork): a list of calculations over value sources (constants, earlier
// intermediates, fixed / advice / instance columns at a rotation, challenges, beta / gamma / theta / y,
// the previous value of the row).  halo2 runs it per row on the CPU with a Vec of intermediates; the
// circuit-specific part is the program, which the Rust side already holds, not this code.
//
// Host side (compile_gates): the graph is lowered to a straight-line program over LDS slots --
//   * every value lives in the 2^261 (hat) limb form; column words are shifted left by 5 bits on load
//     (which is that form, with bound 32, at no cost), constants are converted once per workgroup;
//   * additions / subtractions are lazy (bound tracking as in the curve code); a reduction (f29_reduce_small: the
//     quotient from the top limb, ~45 instructions, no product) is inserted only where the next product would exceed
//     bound_a * bound_b <= 170;
//   * a product that only an addition reads is fused into it (G_MULADD = f29_mul_add: the addend joins the high columns
//     of the product, one instruction stream entry and one LDS round trip less); Horner is a chain of those;
//   * Store is an alias, loads are emitted at first use;
//   * slots are allocated by liveness (last use), so the LDS footprint is the maximum number of
//     simultaneously live values, not the number of intermediates.
// Device side (gates_kernel): one thread per row, T rows per workgroup, slots in LDS as [slot][limb][row]
// (conflict-free), the instruction stream is uniform (scalar loads, no divergence).  Cost per
// product-type instruction ~260 VALU instructions (206 of them the product): ALU-bound like the rest.
#include "gates.h"
#include "gates_device.cuh"
#include "side_prio.cuh"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>

namespace sg {
SG_DEFINE_SIDE_PRIO_SETTER(gates_set_side_prio)

// ------------------------------------------------------------------ device

template <uint32_t T>
__global__ void __launch_bounds__(T) gates_kernel(GateArgs a) {
  side_kernel_prio();
  extern __shared__ uint32_t lds[];
  uint32_t* s_const = lds;                       // [n_consts][9]
  uint32_t* s_slot = lds + a.n_consts * 9;       // [n_slots][9][T]
  const uint32_t tid = threadIdx.x;
  for (uint32_t c = tid; c < a.n_consts; c += T) {
    f29 v = f29_words_to_r261<P>(a.consts + 8 * c);
#pragma unroll
    for (int q = 0; q < 9; q++) s_const[c * 9 + q] = v.l[q];
  }
  __syncthreads();
  const size_t mask = (size_t)a.blockmask;
  const size_t row = (size_t)blockIdx.x * T + tid;
  if (row >= a.rows) return;
  const uint32_t rot_shift = a.ext_k - a.k;
  auto get = [&](uint32_t kind, uint32_t idx) {
    f29 r;
    if (kind == GK_CONST) {
#pragma unroll
      for (int q = 0; q < 9; q++) r.l[q] = s_const[idx * 9 + q];
    } else {
#pragma unroll
      for (int q = 0; q < 9; q++) r.l[q] = s_slot[(idx * 9 + q) * T + tid];
    }
    return r;
  };
  auto sub_k = [&](uint32_t kidx, const f29& x, const f29& y) {
    switch (kidx) {
      case 0: return f29_sub<P, 0>(x, y);
      case 1: return f29_sub<P, 1>(x, y);
      case 2: return f29_sub<P, 2>(x, y);
      case 3: return f29_sub<P, 3>(x, y);
      case 4: return f29_sub<P, 4>(x, y);
      default: return f29_sub<P, 5>(x, y);
    }
  };
  for (uint32_t pc = 0; pc < a.n_ops; pc++) {
    const GateOp op = a.ops[pc];
    const uint32_t code = op.w0 & 0xff, kidx = (op.w0 >> 8) & 0xff, ak = (op.w0 >> 16) & 0xff, bk = op.w0 >> 24;
    const uint32_t dst = op.dst & 0xffff;
    f29 r;
    switch (code) {
      case G_LOADCOL: {
        const size_t i = (row & ~mask) | ((row + ((size_t)(int64_t)(int32_t)op.b << rot_shift)) & mask);
        uint32_t w[8];
        fp_words_load(a.cols[op.a] + i, w);
        // kidx 0: memory words shifted left by 5 bits ARE the 2^261 form (bound 32), no product;
        // kidx 1: converted with one product (bound 2) -- for values with several consumers
        r = kidx ? f29_mul<P>(f29_from_words<0>(w), f29_const<P>(P::r266)) : f29_from_words<5>(w);
        break;
      }
      case G_LOADPREV: {
        uint32_t w[8];
        fp_words_load(a.values + row, w);
        r = f29_from_words<5>(w);
        break;
      }
      case G_ADD: r = f29_add(get(ak, op.a), get(bk, op.b)); break;
      case G_SUB: r = sub_k(kidx, get(ak, op.a), get(bk, op.b)); break;
      case G_MUL: r = f29_mul<P>(get(ak, op.a), get(bk, op.b)); break;
      case G_SQR: r = f29_sqr<P>(get(ak, op.a)); break;
      case G_DBL: { f29 x = get(ak, op.a); r = f29_add(x, x); break; }
      case G_NEG: r = sub_k(kidx, f29_zero(), get(ak, op.a)); break;
      case G_MULADD: r = f29_mul_add<P>(get(ak, op.a), get(bk, op.b), get(kidx, op.dst >> 16)); break;
      default: r = f29_reduce_small<P>(get(ak, op.a)); break;  // G_RED
    }
#pragma unroll
    for (int q = 0; q < 9; q++) s_slot[(dst * 9 + q) * T + tid] = r.l[q];
  }
  f29 res = get(a.result_kind, a.result_index);
  // hat -> memory domain: x^ * 2^256 * 2^-261 = x~, then canonical
  f29_store_canonical<P>(a.values + row, f29_mul<P>(res, f29_const<P>(P::r256)));
}

template <class PROG>
__global__ void __launch_bounds__(256) gates_fixed_kernel(GateArgs a) {
  side_kernel_prio();
  extern __shared__ uint32_t lds[];
  uint32_t* s_const = lds;                       // [n_consts][9]
  for (uint32_t c = threadIdx.x; c < a.n_consts; c += blockDim.x) {
    f29 v = f29_words_to_r261<P>(a.consts + 8 * c);
#pragma unroll
    for (int q = 0; q < 9; q++) s_const[c * 9 + q] = v.l[q];
  }
  __syncthreads();
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= a.rows) return;
  const GateSrc src{a.cols, a.values};
  const f29 res = gates_fixed_eval<PROG>(src, s_const, row, (size_t)a.blockmask, a.ext_k - a.k);
  f29_store_canonical<P>(a.values + row, f29_mul<P>(res, f29_const<P>(P::r256)));
}

// the same with the column pointers and the constant table as kernel arguments: nothing to upload, nothing to keep alive
static constexpr uint32_t GATES_V_COLS = 24, GATES_V_CONSTS = 40;
struct GateArgsV {
  fp_words* values;
  const fp_words* cols[GATES_V_COLS];
  uint32_t consts[GATES_V_CONSTS][8];
  uint32_t n_consts, k, ext_k;
  uint64_t rows, blockmask;
};
template <class PROG>
__global__ void __launch_bounds__(256) gates_fixed_value_kernel(GateArgsV a) {
  side_kernel_prio();
  __shared__ uint32_t s_const[GATES_V_CONSTS][9];
  if (threadIdx.x < a.n_consts) {
    const f29 v = f29_words_to_r261<P>(a.consts[threadIdx.x]);
#pragma unroll
    for (int q = 0; q < 9; q++) s_const[threadIdx.x][q] = v.l[q];
  }
  __syncthreads();
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= a.rows) return;
  const GateSrc src{a.cols, a.values};
  const f29 res = gates_fixed_eval<PROG>(src, &s_const[0][0], row, (size_t)a.blockmask, a.ext_k - a.k);
  f29_store_canonical<P>(a.values + row, f29_mul<P>(res, f29_const<P>(P::r256)));
}

// ------------------------------------------------------------------ host: compiler
namespace {
struct Val {         // a virtual value of the lowered program
  uint32_t kind;     // GK_SLOT (virtual id in `index`) or GK_CONST
  uint32_t index;
};
struct IrOp {
  uint32_t code, kidx;
  uint32_t dst;      // virtual id
  Val a, b;
  uint32_t col = 0;
  int32_t rot = 0;
  Val c{GK_CONST, 0};   // MULADD: the addend
};
struct Compiler {
  std::vector<IrOp> ir;
  std::vector<uint32_t> bound;     // per virtual id
  std::vector<uint32_t> redirect;  // virtual id -> reduced replacement (or itself)
  uint32_t new_value(uint32_t b) {
    bound.push_back(b);
    redirect.push_back((uint32_t)redirect.size());
    return (uint32_t)bound.size() - 1;
  }
  Val resolve(Val v) {
    if (v.kind == GK_SLOT)
      while (redirect[v.index] != v.index) v.index = redirect[v.index];
    return v;
  }
  uint32_t bnd(const Val& v) const { return v.kind == GK_CONST ? 2u : bound[v.index]; }
  Val reduce(Val v) {  // f29_reduce_small: bound 2
    v = resolve(v);
    if (v.kind == GK_CONST || bound[v.index] <= 2) return v;
    uint32_t d = new_value(2);
    ir.push_back({G_RED, 0, d, v, v});
    redirect[v.index] = d;  // later uses see the reduced copy
    return Val{GK_SLOT, d};
  }
  Val emit2(uint32_t code, uint32_t kidx, Val a, Val b, uint32_t out_bound) {
    uint32_t d = new_value(out_bound);
    ir.push_back({code, kidx, d, a, b});
    return Val{GK_SLOT, d};
  }
  Val add(Val a, Val b) {
    a = resolve(a); b = resolve(b);
    while (bnd(a) + bnd(b) > 80) {
      if (bnd(a) >= bnd(b)) a = reduce(a); else b = reduce(b);
    }
    return emit2(G_ADD, 0, a, b, bnd(a) + bnd(b));
  }
  static uint32_t kidx_for(uint32_t b) {  // smallest K = 2 << kidx >= b
    uint32_t k = 0;
    while ((2u << k) < b) k++;
    return k;
  }
  Val sub(Val a, Val b) {
    a = resolve(a); b = resolve(b);
    if (bnd(b) > 64) b = reduce(b);
    uint32_t k = kidx_for(bnd(b));
    if (bnd(a) + (2u << k) > 120) a = reduce(a);
    return emit2(G_SUB, k, a, b, bnd(a) + (2u << k));
  }
  Val neg(Val a) {
    a = resolve(a);
    if (bnd(a) > 64) a = reduce(a);
    uint32_t k = kidx_for(bnd(a));
    return emit2(G_NEG, k, a, a, 2u << k);
  }
  Val dbl(Val a) {
    a = resolve(a);
    if (bnd(a) > 40) a = reduce(a);
    return emit2(G_DBL, 0, a, a, 2 * bnd(a));
  }
  Val mul(Val a, Val b) {
    a = resolve(a); b = resolve(b);
    while (bnd(a) * bnd(b) > 170) {
      if (bnd(a) >= bnd(b)) a = reduce(a); else b = reduce(b);
    }
    return emit2(G_MUL, 0, a, b, 2);
  }
  Val sqr(Val a) {
    a = resolve(a);
    if (bnd(a) * bnd(a) > 170) a = reduce(a);
    return emit2(G_SQR, 0, a, a, 2);
  }
  Val muladd(Val a, Val b, Val z) {   // a b + z under one reduction: bound 2 + bound(z)
    a = resolve(a); b = resolve(b); z = resolve(z);
    while (bnd(a) * bnd(b) > 170) {
      if (bnd(a) >= bnd(b)) a = reduce(a); else b = reduce(b);
    }
    if (bnd(z) + 2 > 80) z = reduce(z);
    uint32_t d = new_value(bnd(z) + 2);
    IrOp op{G_MULADD, 0, d, a, b};
    op.c = z;
    ir.push_back(op);
    return Val{GK_SLOT, d};
  }
};
}  // namespace

std::string compile_gates(const sg_graph& g, uint32_t n_fixed, uint32_t n_advice, uint32_t n_instance,
                          const uint8_t* challenges, uint32_t n_challenges, const uint8_t beta[32],
                          const uint8_t gamma[32], const uint8_t theta[32], const uint8_t y[32], GateProgram* out) {
  if (g.n_calculations == 0) return "empty program";
  if ((g.n_constants && !g.constants) || (g.n_rotations && !g.rotations) || !g.calculations ||
      (g.n_horner_parts && !g.horner_parts))
    return "null array in graph";
  Compiler c;
  GateProgram prog;
  // constant table: constants ++ challenges ++ beta, gamma, theta, y
  const uint32_t c_chal = g.n_constants, c_beta = c_chal + n_challenges;
  auto push_const = [&](const uint8_t* p) {
    uint32_t w[8];
    std::memcpy(w, p, 32);
    prog.const_words.insert(prog.const_words.end(), w, w + 8);
  };
  for (uint32_t i = 0; i < g.n_constants; i++) push_const(g.constants + 32 * (size_t)i);
  for (uint32_t i = 0; i < n_challenges; i++) push_const(challenges + 32 * (size_t)i);
  push_const(beta); push_const(gamma); push_const(theta); push_const(y);
  // (column, rotation) -> (value, instruction index of its latest use): a loaded value is reused only while the previous use
  // is at most RELOAD_DISTANCE instructions back; beyond that a fresh load (a 32-byte read, L2-resident: the row's cache
  // lines were touched a moment ago) is cheaper than pinning an LDS slot -- the slot count sets the kernel's occupancy.
  size_t RELOAD_DISTANCE = 12;
  if (const char* v = std::getenv("SG_GATES_RELOAD")) RELOAD_DISTANCE = (size_t)std::atoi(v);   // development aid
  // column words (canonical, < p) enter as x~ << 5 = x^ with bound 32 for free; the bound tracker inserts a reduction only
  // where a consumer needs one (a product with a bound-2 value does not), and once a value has been reduced its later
  // uses see the reduced copy (Compiler::redirect), so nothing is gained by converting at the load.
  constexpr uint32_t LOAD_BOUND = 32;
  std::map<std::pair<uint32_t, int32_t>, uint32_t> refs;
  {
    auto note = [&](const sg_value_source& v) {
      if (v.kind < SG_VS_FIXED || v.kind > SG_VS_INSTANCE || v.rotation >= g.n_rotations) return;
      const uint32_t base = v.kind == SG_VS_FIXED ? 0 : v.kind == SG_VS_ADVICE ? n_fixed : n_fixed + n_advice;
      refs[std::make_pair(base + v.index, g.rotations[v.rotation])]++;
    };
    for (uint32_t q = 0; q < g.n_calculations; q++) {
      const sg_calculation& cal = g.calculations[q];
      note(cal.a);
      if (cal.op <= SG_OP_MUL || cal.op == SG_OP_HORNER) note(cal.b);
      if (cal.op == SG_OP_HORNER && (uint64_t)cal.parts_offset + cal.parts_len <= g.n_horner_parts)
        for (uint32_t t = 0; t < cal.parts_len; t++) note(g.horner_parts[cal.parts_offset + t]);
    }
  }
  std::map<std::pair<uint32_t, int32_t>, std::pair<Val, size_t>> loaded;
  Val prev{GK_SLOT, 0xffffffffu};
  std::vector<Val> inter(g.n_calculations);
  std::string err;
  auto source = [&](const sg_value_source& s, uint32_t upto) -> Val {
    switch (s.kind) {
      case SG_VS_CONSTANT:
        if (s.index >= g.n_constants) { err = "constant index out of range"; return Val{GK_CONST, 0}; }
        return Val{GK_CONST, s.index};
      case SG_VS_INTERMEDIATE:
        if (s.index >= upto) { err = "intermediate used before it is defined"; return Val{GK_CONST, 0}; }
        return inter[s.index];
      case SG_VS_FIXED: case SG_VS_ADVICE: case SG_VS_INSTANCE: {
        const uint32_t lim = s.kind == SG_VS_FIXED ? n_fixed : s.kind == SG_VS_ADVICE ? n_advice : n_instance;
        const uint32_t base = s.kind == SG_VS_FIXED ? 0 : s.kind == SG_VS_ADVICE ? n_fixed : n_fixed + n_advice;
        if (s.index >= lim || s.rotation >= g.n_rotations) { err = "column query out of range"; return Val{GK_CONST, 0}; }
        auto key = std::make_pair(base + s.index, g.rotations[s.rotation]);
        auto it = loaded.find(key);
        if (it != loaded.end() && c.ir.size() - it->second.second <= RELOAD_DISTANCE) {
          it->second.second = c.ir.size();
          return it->second.first;
        }
        bool convert = false;
        if (const char* v = std::getenv("SG_GATES_CONVERT")) convert = refs[key] > (uint32_t)std::atoi(v);   // development aid
        uint32_t d = c.new_value(convert ? 2 : LOAD_BOUND);
        IrOp op{G_LOADCOL, convert ? 1u : 0u, d, Val{GK_CONST, 0}, Val{GK_CONST, 0}};
        op.col = key.first; op.rot = key.second;
        c.ir.push_back(op);
        loaded[key] = std::make_pair(Val{GK_SLOT, d}, c.ir.size());
        return Val{GK_SLOT, d};
      }
      case SG_VS_CHALLENGE:
        if (s.index >= n_challenges) { err = "challenge index out of range"; return Val{GK_CONST, 0}; }
        return Val{GK_CONST, c_chal + s.index};
      case SG_VS_BETA: return Val{GK_CONST, c_beta};
      case SG_VS_GAMMA: return Val{GK_CONST, c_beta + 1};
      case SG_VS_THETA: return Val{GK_CONST, c_beta + 2};
      case SG_VS_Y: return Val{GK_CONST, c_beta + 3};
      case SG_VS_PREVIOUS_VALUE:
        if (prev.index == 0xffffffffu) {
          uint32_t d = c.new_value(LOAD_BOUND);
          c.ir.push_back({G_LOADPREV, 0, d, Val{GK_CONST, 0}, Val{GK_CONST, 0}});
          prev = Val{GK_SLOT, d};
        }
        return prev;
      default: err = "unknown value source"; return Val{GK_CONST, 0};
    }
  };
  // Demand-driven emission: an intermediate is lowered when it is first needed, starting from the last
  // calculation.  halo2 ends a program with Horner(PreviousValue, [all gate polynomials], Y); emitted in
  // program order every gate value would stay live until that final fold, emitted on demand each one
  // is folded right after it is computed (and unused calculations disappear).
  if (g.n_calculations > (1u << 16)) return "more than 65536 calculations";
  for (uint32_t q = 0; q < g.n_calculations; q++) {  // validate references up front (the recursion trusts them)
    const sg_calculation& cal = g.calculations[q];
    auto bad = [&](const sg_value_source& s) { return s.kind == SG_VS_INTERMEDIATE && s.index >= q; };
    if (bad(cal.a) || (cal.op <= SG_OP_MUL && bad(cal.b)) || (cal.op == SG_OP_HORNER && bad(cal.b)))
      return "intermediate used before it is defined";
    if (cal.op == SG_OP_HORNER) {
      if ((uint64_t)cal.parts_offset + cal.parts_len > g.n_horner_parts) return "horner parts out of range";
      for (uint32_t t = 0; t < cal.parts_len; t++)
        if (bad(g.horner_parts[cal.parts_offset + t])) return "intermediate used before it is defined";
    }
  }
  {  // the lowering below recurses along dependencies: bound the depth (halo2 graphs are a few hundred deep)
    std::vector<uint32_t> depth(g.n_calculations, 1);
    auto dep = [&](const sg_value_source& s) { return s.kind == SG_VS_INTERMEDIATE ? depth[s.index] : 0u; };
    for (uint32_t q = 0; q < g.n_calculations; q++) {
      const sg_calculation& cal = g.calculations[q];
      uint32_t d = dep(cal.a);
      if (cal.op <= SG_OP_MUL || cal.op == SG_OP_HORNER) d = std::max(d, dep(cal.b));
      if (cal.op == SG_OP_HORNER)
        for (uint32_t t = 0; t < cal.parts_len; t++) d = std::max(d, dep(g.horner_parts[cal.parts_offset + t]));
      depth[q] = d + 1;
      if (depth[q] > 4096) return "dependency chain deeper than 4096 calculations";
    }
  }
  // how often each intermediate is read: a product that is read once, by an addition, is fused into it (G_MULADD)
  std::vector<uint32_t> uses(g.n_calculations, 0);
  {
    auto use = [&](const sg_value_source& s) { if (s.kind == SG_VS_INTERMEDIATE) uses[s.index]++; };
    for (uint32_t q = 0; q < g.n_calculations; q++) {
      const sg_calculation& cal = g.calculations[q];
      use(cal.a);
      if (cal.op <= SG_OP_MUL || cal.op == SG_OP_HORNER) use(cal.b);
      if (cal.op == SG_OP_HORNER)
        for (uint32_t t = 0; t < cal.parts_len; t++) use(g.horner_parts[cal.parts_offset + t]);
    }
    uses[g.n_calculations - 1]++;   // the result
  }
  std::vector<uint8_t> done(g.n_calculations, 0);
  std::function<Val(const sg_value_source&)> need;
  std::function<void(uint32_t)> lower = [&](uint32_t q) {
    if (done[q] || !err.empty()) return;
    done[q] = 1;
    const sg_calculation& cal = g.calculations[q];
    if (cal.op == SG_OP_ADD) {
      // x * y + z: one operand a product (or a square) that nothing else reads and that has not been lowered yet
      auto fusable = [&](const sg_value_source& s) {
        return s.kind == SG_VS_INTERMEDIATE && uses[s.index] == 1 && !done[s.index] &&
               (g.calculations[s.index].op == SG_OP_MUL || g.calculations[s.index].op == SG_OP_SQUARE);
      };
      const bool fa = fusable(cal.a), fb = !fa && fusable(cal.b);
      if (fa || fb) {
        const sg_value_source& prod = fa ? cal.a : cal.b;
        const sg_calculation& m = g.calculations[prod.index];
        done[prod.index] = 1;
        const Val z = need(fa ? cal.b : cal.a);          // the addend first: one live value while the factors are computed
        const Val x = need(m.a);
        const Val y = m.op == SG_OP_SQUARE ? x : need(m.b);
        inter[q] = c.muladd(x, y, z);
        return;
      }
    }
    Val a = need(cal.a);
    switch (cal.op) {
      case SG_OP_ADD: inter[q] = c.add(a, need(cal.b)); break;
      case SG_OP_SUB: inter[q] = c.sub(a, need(cal.b)); break;
      case SG_OP_MUL: inter[q] = c.mul(a, need(cal.b)); break;
      case SG_OP_SQUARE: inter[q] = c.sqr(a); break;
      case SG_OP_DOUBLE: inter[q] = c.dbl(a); break;
      case SG_OP_NEGATE: inter[q] = c.neg(a); break;
      case SG_OP_HORNER: {
        Val f = need(cal.b), acc = a;
        for (uint32_t t = 0; t < cal.parts_len && err.empty(); t++) {
          const Val part = need(g.horner_parts[cal.parts_offset + t]);
          acc = c.muladd(acc, f, part);
        }
        inter[q] = acc;
        break;
      }
      case SG_OP_STORE: inter[q] = a; break;
      default: err = "unknown calculation";
    }
  };
  need = [&](const sg_value_source& s) -> Val {
    if (s.kind == SG_VS_INTERMEDIATE) lower(s.index);
    return source(s, g.n_calculations);
  };
  lower(g.n_calculations - 1);
  if (!err.empty()) return err;
  Val result = c.resolve(inter[g.n_calculations - 1]);
  // liveness: last instruction that reads each virtual value (the result lives to the end)
  const uint32_t nv = (uint32_t)c.bound.size(), END = 0xffffffffu;
  std::vector<uint32_t> last(nv, 0);
  for (uint32_t i = 0; i < c.ir.size(); i++) {
    const IrOp& op = c.ir[i];
    if (op.code == G_LOADCOL || op.code == G_LOADPREV) continue;
    if (op.a.kind == GK_SLOT) last[op.a.index] = i;
    if (op.b.kind == GK_SLOT) last[op.b.index] = i;
    if (op.code == G_MULADD && op.c.kind == GK_SLOT) last[op.c.index] = i;
  }
  if (result.kind == GK_SLOT) last[result.index] = END;
  std::vector<uint32_t> slot(nv, END), free_slots;
  uint32_t n_slots = 0;
  for (uint32_t i = 0; i < c.ir.size(); i++) {
    const IrOp& op = c.ir[i];
    GateOp o{};
    uint32_t a_idx = 0, b_idx = 0, c_idx = 0;
    const bool reads = !(op.code == G_LOADCOL || op.code == G_LOADPREV), three = op.code == G_MULADD;
    if (reads) {
      a_idx = op.a.kind == GK_SLOT ? slot[op.a.index] : op.a.index;
      b_idx = op.b.kind == GK_SLOT ? slot[op.b.index] : op.b.index;
      if (three) c_idx = op.c.kind == GK_SLOT ? slot[op.c.index] : op.c.index;
      // operands dying here free their slots before the destination is chosen (in-place update)
      auto same = [](const Val& u, const Val& v) { return u.kind == GK_SLOT && v.kind == GK_SLOT && u.index == v.index; };
      if (op.a.kind == GK_SLOT && last[op.a.index] == i) free_slots.push_back(slot[op.a.index]);
      if (op.b.kind == GK_SLOT && last[op.b.index] == i && !same(op.a, op.b)) free_slots.push_back(slot[op.b.index]);
      if (three && op.c.kind == GK_SLOT && last[op.c.index] == i && !same(op.a, op.c) && !same(op.b, op.c))
        free_slots.push_back(slot[op.c.index]);
    } else {
      a_idx = op.col;
      b_idx = (uint32_t)op.rot;
    }
    uint32_t d;
    if (last[op.dst] == 0 && !(result.kind == GK_SLOT && result.index == op.dst)) {
      // never read: still needs somewhere to land
      if (free_slots.empty()) free_slots.push_back(n_slots++);
      d = free_slots.back();  // not removed: immediately reusable
    } else {
      if (free_slots.empty()) free_slots.push_back(n_slots++);
      d = free_slots.back();
      free_slots.pop_back();
    }
    slot[op.dst] = d;
    if (std::getenv("SG_GATES_DUMP")) {   // development aid: the lowered program with its slot pressure
      static const char* names[] = {"loadcol", "loadprev", "add", "sub", "mul", "sqr", "dbl", "neg", "red", "muladd"};
      std::fprintf(stderr, "%3u %-8s dst s%-2u", i, names[op.code], d);
      if (!reads) std::fprintf(stderr, " col %u rot %d", op.col, op.rot);
      else std::fprintf(stderr, " %c%u %c%u", op.a.kind == GK_SLOT ? 's' : 'c', a_idx, op.b.kind == GK_SLOT ? 's' : 'c', b_idx);
      if (three) std::fprintf(stderr, " %c%u", op.c.kind == GK_SLOT ? 's' : 'c', c_idx);
      std::fprintf(stderr, "   live %u (last use of dst: %u)\n", n_slots - (uint32_t)free_slots.size(), last[op.dst]);
    }
    o.w0 = op.code | ((three ? op.c.kind : op.kidx) << 8) | (op.a.kind << 16) | (op.b.kind << 24);
    o.dst = d | (c_idx << 16); o.a = a_idx; o.b = b_idx;
    if (d > 0xffff || c_idx > 0xffff) return "program too large (slot or constant index above 65535)";
    prog.ops.push_back(o);
  }
  prog.n_slots = std::max<uint32_t>(1, n_slots);
  prog.result_kind = result.kind;
  prog.result_index = result.kind == GK_SLOT ? slot[result.index] : result.index;
  prog.n_columns = n_fixed + n_advice + n_instance;
  *out = std::move(prog);
  return "";
}

size_t gates_blob(const GateProgram& p, const void* const* cols, std::vector<uint8_t>* blob) {
  const size_t ops_b = p.ops.size() * sizeof(GateOp), cols_b = (size_t)p.n_columns * sizeof(void*),
               const_b = p.const_words.size() * sizeof(uint32_t);
  blob->resize(ops_b + cols_b + const_b + 16);
  std::memcpy(blob->data(), p.ops.data(), ops_b);
  if (cols_b) std::memcpy(blob->data() + ops_b, cols, cols_b);
  std::memcpy(blob->data() + ops_b + cols_b, p.const_words.data(), const_b);
  return blob->size();
}

bool gates_run_by_value(const GateProgram& p, const void* const* cols, fp_words* d_values, uint32_t k, uint32_t ext_k, hipStream_t stream,
                        uint32_t cosets, hipError_t* err) {
  *err = hipSuccess;
  if (std::getenv("SG_GATES_GENERIC") || p.n_columns > GATES_V_COLS || p.const_words.size() / 8 > GATES_V_CONSTS) return false;
  GateArgsV a;
  a.values = d_values;
  for (uint32_t i = 0; i < p.n_columns; i++) a.cols[i] = static_cast<const fp_words*>(cols[i]);
  a.n_consts = (uint32_t)(p.const_words.size() / 8);
  std::memcpy(a.consts, p.const_words.data(), p.const_words.size() * sizeof(uint32_t));
  a.k = k; a.ext_k = cosets ? k : ext_k;
  const size_t n_ext = cosets ? (size_t)cosets << k : (size_t)1 << ext_k;
  a.rows = n_ext;
  a.blockmask = ((uint64_t)1 << a.ext_k) - 1;
  const unsigned blocks = (unsigned)((n_ext + 255) / 256);
  bool done = false;
  auto launch = [&](auto tag, const char* name) {
    using PROG = decltype(tag);
    if (done || !is_program<PROG>(p)) return;
    done = true;
    if (std::getenv("SG_GATES_DEBUG")) std::fprintf(stderr, "gates: ahead-of-time program %s (arguments by value), %u blocks\n", name, blocks);
    gates_fixed_value_kernel<PROG><<<blocks, 256, 0, stream>>>(a);
  };
  launch(MstLookupInput{}, "MstLookupInput");
  launch(MstGatesNc2{}, "MstGatesNc2");
  launch(MstGatesNc1{}, "MstGatesNc1");
  launch(MstGatesNc3{}, "MstGatesNc3");
  launch(MstGatesNc4{}, "MstGatesNc4");
  if (done) *err = hipGetLastError();
  return done;
}

hipError_t gates_run(const GateProgram& p, const uint8_t* d_blob, fp_words* d_values, uint32_t k, uint32_t ext_k,
                     hipStream_t stream, uint32_t cosets) {
  GateArgs a;
  a.values = d_values;
  a.ops = reinterpret_cast<const GateOp*>(d_blob);
  a.cols = reinterpret_cast<const fp_words* const*>(d_blob + p.ops.size() * sizeof(GateOp));
  a.consts = reinterpret_cast<const uint32_t*>(d_blob + p.ops.size() * sizeof(GateOp) + (size_t)p.n_columns * sizeof(void*));
  a.n_ops = (uint32_t)p.ops.size();
  a.n_consts = (uint32_t)(p.const_words.size() / 8);
  a.n_slots = p.n_slots;
  a.result_kind = p.result_kind;
  a.result_index = p.result_index;
  a.k = k; a.ext_k = cosets ? k : ext_k;
  const size_t n_ext = cosets ? (size_t)cosets << k : (size_t)1 << ext_k;
  a.rows = n_ext;
  a.blockmask = ((uint64_t)1 << a.ext_k) - 1;
  if (!std::getenv("SG_GATES_GENERIC")) {   // a program known ahead of time: the straight-line kernel
    const unsigned blocks = (unsigned)((n_ext + 255) / 256);
    const size_t lds = (size_t)a.n_consts * 36;
    // (tables for N_CURRENCIES = 1 .. 4: gates_mst_programs.inc; any other program runs in the interpreter below)
    int which = 0;
    auto launch = [&](auto tag, int nc) {
      using PROG = decltype(tag);
      if (which || !is_program<PROG>(p)) return;
      which = nc;
      gates_fixed_kernel<PROG><<<blocks, 256, lds, stream>>>(a);
    };
    launch(MstGatesNc2{}, 2);
    launch(MstGatesNc1{}, 1);
    launch(MstGatesNc3{}, 3);
    launch(MstGatesNc4{}, 4);
    if (which) {
      if (std::getenv("SG_GATES_DEBUG")) std::fprintf(stderr, "gates: ahead-of-time program MstGatesNc%d, %u blocks\n", which, blocks);
      return hipGetLastError();
    }
  }
  // rows per workgroup from the LDS budget: (constants + slots * T) * 36 B <= 144 KiB
  const size_t budget = 144 * 1024, cbytes = (size_t)a.n_consts * 36;
  // rows per workgroup: what limits the interpreter is waves per SIMD, i.e. LDS per row (the slot count).  Take the shape
  // that keeps the most waves per CU; among shapes within one wave of each other the larger workgroup (measured: for
  // programs of few slots 256 rows beat 64 rows although the latter keeps one more wave)
  uint32_t T = 256, best_waves = 0;
  for (uint32_t t : {256u, 128u, 64u}) {
    const size_t need = cbytes + (size_t)p.n_slots * t * 36;
    if (need > budget) continue;
    const uint32_t waves = std::min<uint32_t>(32, (uint32_t)(160 * 1024 / need) * (t / 64));
    if (waves > best_waves + 1 || best_waves == 0) {
      best_waves = waves;
      T = t;
    }
  }
  if (const char* v = std::getenv("SG_GATES_ROWS")) {   // development aid
    const uint32_t t = (uint32_t)std::atoi(v);
    if (t == 64 || t == 128 || t == 256) T = t;
  }
  const size_t lds = cbytes + (size_t)p.n_slots * T * 36;
  if (lds > budget) return hipErrorInvalidValue;
  if (std::getenv("SG_GATES_DEBUG"))
    std::fprintf(stderr, "gates: %zu ops, %u slots, %u constants -> %u rows per workgroup, %u waves per CU by LDS\n", p.ops.size(),
                 p.n_slots, a.n_consts, T, best_waves);
  const unsigned blocks = (unsigned)((n_ext + T - 1) / T);
  hipError_t e;
  if (T == 256) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gates_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)budget);
    if (e != hipSuccess) return e;
    gates_kernel<256><<<blocks, 256, lds, stream>>>(a);
  } else if (T == 128) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gates_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)budget);
    if (e != hipSuccess) return e;
    gates_kernel<128><<<blocks, 128, lds, stream>>>(a);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gates_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)budget);
    if (e != hipSuccess) return e;
    gates_kernel<64><<<blocks, 64, lds, stream>>>(a);
  }
  return hipGetLastError();
}

}  // namespace sg
