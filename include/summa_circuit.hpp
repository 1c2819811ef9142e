// MstInclusionCircuit<LEVELS, N_CURRENCIES, N_BYTES> as compiled host code: the constraint system's GraphEvaluator programs,
// the reference circuit's floor plan (fixed columns, permutation) and the witness program for the device -- what
// circuits_halo2_amd/mst_inclusion.py holds in Python, for hosts that drive the library from C++ (tools/prove_from_csv.cpp:
// SRS file + CSV + user index -> calldata, no interpreter anywhere).
//
//   gates / lookup input   the 17 + N_CURRENCIES gate polynomials of the circuit [REF zk_prover/src/circuits/merkle_sum_tree.rs:
//                          143-196 (configure), chips/merkle_sum_tree.rs (swap / sum), chips/range/range_check.rs (lookup);
//                          halo2_gadgets' Pow5 chip of width 2] lowered to halo2's GraphEvaluator form; the polynomial list is
//                          the one the generated verifier folds [REF contracts/src/InclusionVerifier.sol:495-1000]
//   FloorPlan              `MstInclusionCircuit::synthesize` [REF circuits/merkle_sum_tree.rs:228-520] replayed over halo2's
//                          SimpleFloorPlanner rule (a region starts at the first row where none of its columns is in use; a
//                          region's constants go to the constants column right after it) and its permutation assembly
//                          (cycles merged smaller into larger in call order)
// Checked byte for byte against the Python twin (tests/test_host_logic.py: tools/circuit_dump), which reproduces the
// reference's verifying key.
#pragma once
#include <array>
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "summa_prover.hpp"

namespace summa {
namespace circuit {

using prover::Fr;
using prover::Graph;

#include "../circuits_halo2_amd/csrc/poseidon_constants.inc"

constexpr uint32_t NUM_ADVICE = 3, NUM_FIXED = 11, NUM_PERM = 6, BLINDING_FACTORS = 5;

struct Poseidon {
  Fr rc[64][2], mds[2][2], mds_inv[2][2];
  Poseidon() {
    for (int r = 0; r < 64; r++)
      for (int j = 0; j < 2; j++) std::memcpy(rc[r][j].l, POSEIDON_RC[r][j], 32);
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 2; j++) std::memcpy(mds[i][j].l, POSEIDON_MDS[i][j], 32);
    const Fr det = (mds[0][0] * mds[1][1] - mds[0][1] * mds[1][0]).inv();
    mds_inv[0][0] = mds[1][1] * det;
    mds_inv[0][1] = -mds[0][1] * det;
    mds_inv[1][0] = -mds[1][0] * det;
    mds_inv[1][1] = mds[0][0] * det;
  }
};
inline const Poseidon& poseidon() {
  static const Poseidon p;
  return p;
}

// ------------------------------------------------------------------ expressions -> GraphEvaluator
// halo2's ValueSource / Calculation numbering (include/summa_gpu.h)
struct GraphBuilder {
  Graph g;
  sg_value_source add_constant(const Fr& v) {
    const uint32_t n = (uint32_t)(g.constants.size() / 32);
    for (uint32_t i = 0; i < n; i++)
      if (!std::memcmp(g.constants.data() + 32 * i, v.l, 32)) return sg_value_source{SG_VS_CONSTANT, i, 0};
    g.constants.insert(g.constants.end(), v.bytes(), v.bytes() + 32);
    return sg_value_source{SG_VS_CONSTANT, n, 0};
  }
  uint32_t add_rotation(int32_t rot) {
    for (size_t i = 0; i < g.rotations.size(); i++)
      if (g.rotations[i] == rot) return (uint32_t)i;
    g.rotations.push_back(rot);
    return (uint32_t)g.rotations.size() - 1;
  }
  sg_value_source query(uint32_t kind, uint32_t column, int32_t rot) { return sg_value_source{kind, column, add_rotation(rot)}; }
  static bool same(const sg_value_source& a, const sg_value_source& b) { return a.kind == b.kind && a.index == b.index && a.rotation == b.rotation; }
  // parts: HORNER only; identical calculations are shared (upstream's add_calculation)
  sg_value_source add_calculation(uint32_t op, sg_value_source a, sg_value_source b = sg_value_source{0, 0, 0},
                                  const std::vector<sg_value_source>* parts = nullptr) {
    for (size_t i = 0; i < g.calculations.size(); i++) {
      const sg_calculation& c = g.calculations[i];
      if (c.op != op || !same(c.a, a) || !same(c.b, b)) continue;
      const bool has = op == SG_OP_HORNER;
      if (!has) return sg_value_source{SG_VS_INTERMEDIATE, (uint32_t)i, 0};
      if (parts && c.parts_len == parts->size()) {
        bool eq = true;
        for (size_t t = 0; t < parts->size(); t++) eq = eq && same(g.parts[c.parts_offset + t], (*parts)[t]);
        if (eq) return sg_value_source{SG_VS_INTERMEDIATE, (uint32_t)i, 0};
      }
    }
    sg_calculation c{};
    c.op = op;
    c.a = a;
    c.b = b;
    if (parts) {
      c.parts_offset = (uint32_t)g.parts.size();
      c.parts_len = (uint32_t)parts->size();
      g.parts.insert(g.parts.end(), parts->begin(), parts->end());
    }
    g.calculations.push_back(c);
    return sg_value_source{SG_VS_INTERMEDIATE, (uint32_t)g.calculations.size() - 1, 0};
  }
};

struct Expr;
using E = std::shared_ptr<const Expr>;
struct Expr {
  enum Op { CONST, QUERY, ADD, SUB, MUL } op;
  Fr value;                    // CONST
  uint32_t kind = 0, column = 0;
  int32_t rotation = 0;        // QUERY
  E a, b;
  sg_value_source lower(GraphBuilder& g) const {
    if (op == CONST) return g.add_constant(value);
    if (op == QUERY) return g.query(kind, column, rotation);
    if (op == MUL && a.get() == b.get()) return g.add_calculation(SG_OP_SQUARE, a->lower(g));
    const sg_value_source va = a->lower(g), vb = b->lower(g);
    if (op == MUL && GraphBuilder::same(va, vb)) return g.add_calculation(SG_OP_SQUARE, va);
    return g.add_calculation(op == ADD ? SG_OP_ADD : op == SUB ? SG_OP_SUB : SG_OP_MUL, va, vb);
  }
};
inline E constant(const Fr& v) {
  auto e = std::make_shared<Expr>();
  e->op = Expr::CONST;
  e->value = v;
  return e;
}
inline E constant(uint64_t v) { return constant(Fr::from_u64(v)); }
inline E query(uint32_t kind, uint32_t column, int32_t rot = 0) {
  auto e = std::make_shared<Expr>();
  e->op = Expr::QUERY;
  e->kind = kind;
  e->column = column;
  e->rotation = rot;
  return e;
}
inline E bin(Expr::Op op, const E& a, const E& b) {
  auto e = std::make_shared<Expr>();
  e->op = op;
  e->a = a;
  e->b = b;
  return e;
}
inline E operator+(const E& a, const E& b) { return bin(Expr::ADD, a, b); }
inline E operator-(const E& a, const E& b) { return bin(Expr::SUB, a, b); }
inline E operator*(const E& a, const E& b) { return bin(Expr::MUL, a, b); }
inline E pow5(const E& v) {
  const E sq = v * v;
  return sq * sq * v;
}

// the gate polynomials in the constraint system's order (17 + one sum gate per currency)
inline std::vector<E> gates(uint32_t n_currencies) {
  const Poseidon& P = poseidon();
  auto a = [](uint32_t c, int32_t r = 0) { return query(SG_VS_ADVICE, c, r); };
  auto f = [](uint32_t c) { return query(SG_VS_FIXED, c, 0); };
  std::vector<E> out;
  auto poseidon_chip = [&](const E& s_full, const E& s_partial) {
    const E sbox[2] = {pow5(a(0) + f(0)), pow5(a(1) + f(1))};
    for (int i = 0; i < 2; i++)   // full round
      out.push_back(s_full * (sbox[0] * constant(P.mds[i][0]) + sbox[1] * constant(P.mds[i][1]) - a(i, 1)));
    out.push_back(s_partial * (sbox[0] - a(2)));   // two partial rounds per row
    const E mid[2] = {a(2), a(1) + f(1)};
    E r_mid[2], nxt[2];
    for (int i = 0; i < 2; i++) r_mid[i] = mid[0] * constant(P.mds[i][0]) + mid[1] * constant(P.mds[i][1]);
    for (int i = 0; i < 2; i++) nxt[i] = a(0, 1) * constant(P.mds_inv[i][0]) + a(1, 1) * constant(P.mds_inv[i][1]);
    out.push_back(s_partial * (pow5(r_mid[0] + f(2)) - nxt[0]));
    out.push_back(s_partial * (r_mid[1] + f(3) - nxt[1]));
  };
  auto simple_selector = [&](uint64_t value) {   // f6 * prod_{v != value} (v - f6)
    E s = f(6);
    for (uint64_t v = 1; v < 5; v++)
      if (v != value) s = s * (constant(v) - f(6));
    return s;
  };
  auto pad_and_add = [&](uint64_t value) {
    const E s = simple_selector(value);
    out.push_back(s * (a(0, -1) + a(0) - a(0, 1)));
    out.push_back(s * (a(1, -1) - a(1, 1)));
  };
  poseidon_chip(f(7), f(8));
  pad_and_add(3);
  poseidon_chip(f(9), f(10));
  pad_and_add(4);
  E s = simple_selector(1);
  out.push_back(s * a(2) * (constant(1) - a(2)));
  out.push_back(s * ((a(1) - a(0)) * a(2) + a(0) - a(0, 1)));
  out.push_back(s * ((a(0) - a(1)) * a(2) + a(1) - a(1, 1)));
  s = simple_selector(2);
  for (uint32_t c = 0; c < n_currencies; c++) out.push_back(s * (a(0) + a(1) - a(2)));
  return out;
}
// evaluate_h's custom-gate block, values <- values * y^Ng + sum_i G_i y^(Ng - 1 - i) over the Ng = 17 + n_currencies polynomials of
// gates(), as the program of mst_inclusion._gate_program (same calculations in the same order; the comment there explains the
// factoring: both Poseidon chips at once, the packed selectors by differences, powers of y as challenges)
struct GateProgramSpec {
  Graph graph;
  std::vector<std::vector<uint32_t>> challenge_exponents;   // challenge i = sum of y^e over group i
};
inline GateProgramSpec gate_program(uint32_t n_currencies) {
  const Poseidon& P = poseidon();
  GraphBuilder g;
  GateProgramSpec out;
  auto& groups = out.challenge_exponents;
  const uint32_t ng = 17 + n_currencies;
  auto e = [&](uint32_t i) { return ng - 1 - i; };
  auto chal = [&](std::vector<uint32_t> exps) {
    for (size_t i = 0; i < groups.size(); i++)
      if (groups[i] == exps) return sg_value_source{SG_VS_CHALLENGE, (uint32_t)i, 0};
    groups.push_back(exps);
    return sg_value_source{SG_VS_CHALLENGE, (uint32_t)groups.size() - 1, 0};
  };
  auto cst = [&](const Fr& v) { return g.add_constant(v); };
  auto a = [&](uint32_t c, int32_t r = 0) { return g.query(SG_VS_ADVICE, c, r); };
  auto f = [&](uint32_t c) { return g.query(SG_VS_FIXED, c, 0); };
  auto add = [&](sg_value_source x, sg_value_source y) { return g.add_calculation(SG_OP_ADD, x, y); };
  auto sub = [&](sg_value_source x, sg_value_source y) { return g.add_calculation(SG_OP_SUB, x, y); };
  auto mul = [&](sg_value_source x, sg_value_source y) { return g.add_calculation(SG_OP_MUL, x, y); };
  auto sqr = [&](sg_value_source x) { return g.add_calculation(SG_OP_SQUARE, x); };
  auto pow5 = [&](sg_value_source v) { return mul(sqr(sqr(v)), v); };
  // NB: C++ leaves the evaluation order of function arguments open; every step below is its own statement so that constants,
  // rotations and calculations are numbered exactly as in the Python twin (which evaluates left to right)
  using V = sg_value_source;
  const V y7 = chal({7});
  const V c_ng = chal({ng});
  V acc = mul(V{SG_VS_PREVIOUS_VALUE, 0, 0}, c_ng);
  // pad-and-add, swap and sum gates
  const V q = f(6);
  const V k1 = cst(Fr::from_u64(1));
  const V u1 = sub(k1, q);
  const V k2 = cst(Fr::from_u64(2));
  const V u2 = sub(k2, q);
  const V k3 = cst(Fr::from_u64(3));
  const V u3 = sub(k3, q);
  const V k4 = cst(Fr::from_u64(4));
  const V u4 = sub(k4, q);
  const V qu1 = mul(q, u1);
  const V lo = mul(qu1, u2);
  const V sel4 = mul(lo, u3);
  const V sel3 = add(sel4, lo);
  const V a0p = a(0, -1);
  const V a0c = a(0);
  const V p0s = add(a0p, a0c);
  const V a0n = a(0, 1);
  const V pad0 = sub(p0s, a0n);
  const V a1p = a(1, -1);
  const V a1n = a(1, 1);
  const V pad1 = sub(a1p, a1n);
  const V c12 = chal({e(12)});
  const V jp0 = mul(pad0, c12);
  const V c13 = chal({e(13)});
  const V jp1 = mul(pad1, c13);
  const V j_pad = add(jp0, jp1);
  const V s3y = mul(sel3, y7);
  const V cpad = add(s3y, sel4);
  const V tpad = mul(j_pad, cpad);
  acc = add(acc, tpad);
  const V qu3 = mul(q, u3);
  const V hi = mul(qu3, u4);
  const V sel2 = mul(hi, u1);
  const V sel1 = add(sel2, hi);
  const V a2c = a(2);
  const V one_m = sub(k1, a2c);
  const V swap_bool = mul(a2c, one_m);
  const V a1c = a(1);
  const V dd = sub(a1c, a0c);
  const V d = mul(dd, a2c);
  const V sl0 = add(d, a0c);
  const V swap_l = sub(sl0, a0n);
  const V sr0 = sub(a1c, d);
  const V swap_r = sub(sr0, a1n);
  const V c14 = chal({e(14)});
  const V i0 = mul(swap_bool, c14);
  const V c15 = chal({e(15)});
  const V i1 = mul(swap_l, c15);
  const V i01 = add(i0, i1);
  const V c16 = chal({e(16)});
  const V i2 = mul(swap_r, c16);
  const V inner = add(i01, i2);
  const V tswap = mul(inner, sel1);
  acc = add(acc, tswap);
  const V t01 = add(a0c, a1c);
  const V total = sub(t01, a2c);
  std::vector<uint32_t> sum_exps;
  for (uint32_t j = 0; j < n_currencies; j++) sum_exps.push_back(e(17 + j));
  const V csum = chal(sum_exps);
  const V ts0 = mul(total, csum);
  const V tsum = mul(ts0, sel2);
  acc = add(acc, tsum);
  // the Poseidon rounds, both chips at once
  const V f1c = f(1);
  const V v1 = add(a1c, f1c);
  const V s1 = pow5(v1);
  const V f0c = f(0);
  const V v0 = add(a0c, f0c);
  const V s0 = pow5(v0);
  auto full = [&](int i, const V& nx) {
    const V m0 = cst(P.mds[i][0]);
    const V x0 = mul(s0, m0);
    const V m1 = cst(P.mds[i][1]);
    const V x1 = mul(s1, m1);
    const V x = add(x0, x1);
    return sub(x, nx);
  };
  const V full0 = full(0, a0n);
  const V full1 = full(1, a1n);
  const V c7 = chal({e(7)});
  const V jf0 = mul(full0, c7);
  const V c8 = chal({e(8)});
  const V jf1 = mul(full1, c8);
  const V j_full = add(jf0, jf1);
  const V f7c = f(7);
  const V f7y = mul(f7c, y7);
  const V f9c = f(9);
  const V cfull = add(f7y, f9c);
  const V tfull = mul(j_full, cfull);
  acc = add(acc, tfull);
  const V e2 = sub(s0, a2c);
  const V c9 = chal({e(9)});
  V j_part = mul(e2, c9);
  V mid[2], nxt[2];
  for (int i = 0; i < 2; i++) {
    const V m0 = cst(P.mds[i][0]);
    const V x0 = mul(a2c, m0);
    const V m1 = cst(P.mds[i][1]);
    const V x1 = mul(v1, m1);
    mid[i] = add(x0, x1);
  }
  for (int i = 0; i < 2; i++) {
    const V m0 = cst(P.mds_inv[i][0]);
    const V x0 = mul(a0n, m0);
    const V m1 = cst(P.mds_inv[i][1]);
    const V x1 = mul(a1n, m1);
    nxt[i] = add(x0, x1);
  }
  const V f3c = f(3);
  const V e4a = add(mid[1], f3c);
  const V e4 = sub(e4a, nxt[1]);
  const V c11 = chal({e(11)});
  const V jp4 = mul(e4, c11);
  j_part = add(j_part, jp4);
  const V f2c = f(2);
  const V e3a = add(mid[0], f2c);
  const V e3b = pow5(e3a);
  const V e3 = sub(e3b, nxt[0]);
  const V c10 = chal({e(10)});
  const V jp3 = mul(e3, c10);
  j_part = add(j_part, jp3);
  const V f8c = f(8);
  const V f8y = mul(f8c, y7);
  const V f10c = f(10);
  const V cpart = add(f8y, f10c);
  const V tpart = mul(j_part, cpart);
  add(acc, tpart);   // the last calculation is the row's new value
  out.graph = g.g;
  return out;
}
inline Graph gate_graph(uint32_t n_currencies) { return gate_program(n_currencies).graph; }
// what that program reads as challenges: challenge i = sum of y^e over group i (ProvingKey::gate_challenge_exps)
inline std::vector<std::vector<uint32_t>> gate_challenge_exponents(uint32_t n_currencies) { return gate_program(n_currencies).challenge_exponents; }
// the lookup's input expression f5 * (a0 - 2^8 a0_next), one value per row
inline Graph lookup_input_graph() {
  GraphBuilder g;
  const E e = query(SG_VS_FIXED, 5, 0) * (query(SG_VS_ADVICE, 0, 0) - query(SG_VS_ADVICE, 0, 1) * constant(256));
  g.add_calculation(SG_OP_STORE, e->lower(g));
  return g.g;
}

// ------------------------------------------------------------------ the reference circuit's floor plan
// column keys of the floor planner: 0-2 advice, 3-13 fixed 0-10, 14 instance, then the selectors
enum Col : int { A0 = 0, A1 = 1, A2 = 2, F0 = 3, INSTANCE = 14, SEL_PAD1 = 15, SEL_PAD2, SEL_FULL1, SEL_FULL2, SEL_PART1, SEL_PART2, SEL_LOOKUP, SEL_SWAP,
                 SEL_SUM, NUM_COLS };
inline int fixed_col(int j) { return F0 + j; }
// permutation columns in the order of their sigma polynomials: f2, a0, a1, f3, a2, instance
inline int perm_index(int col) {
  switch (col) {
    case F0 + 2: return 0;
    case A0: return 1;
    case A1: return 2;
    case F0 + 3: return 3;
    case A2: return 4;
    case INSTANCE: return 5;
    default: throw std::logic_error("column is not under the permutation argument");
  }
}
// witness-program symbols (include/summa_gpu.h: sg_mst_inclusion_witness_dev)
enum SymKind : uint32_t { SYM_USER = 0, SYM_HASH = 1, SYM_BAL = 2, SYM_BIT = 3 };
enum SymMode : uint32_t { MODE_PATH = 0, MODE_SIBLING = 1, MODE_SIBLING_CHILD = 2, MODE_ORDERED_CHILD = 3 };
inline uint32_t sym(uint32_t kind, uint32_t level = 0, uint32_t mode = 0, uint32_t lane = 0) { return kind | level << 4 | mode << 10 | lane << 13; }

struct Cell { int col; uint32_t row; };

struct FloorPlan {
  uint32_t k, levels, nc, n_bytes;
  size_t n;
  std::vector<std::vector<Fr>> fixed;                 // 11 columns of n rows
  std::vector<std::vector<std::pair<uint32_t, uint32_t>>> mapping;   // permutation: (column, row) -> (column, row)
  std::vector<uint32_t> program;                      // items (5 words each) then absorbs (3 words each)
  uint32_t n_items = 0, n_absorbs = 0, rows_used = 0;
  std::vector<uint32_t> instance_symbols;

  FloorPlan(uint32_t k_, uint32_t levels_, uint32_t nc_, uint32_t n_bytes_ = 8)
      : k(k_), levels(levels_), nc(nc_), n_bytes(n_bytes_), n((size_t)1 << k_) {
    fixed.assign(NUM_FIXED, std::vector<Fr>(n, Fr::zero()));
    mapping.resize(NUM_PERM);
    aux_.resize(NUM_PERM);
    sizes_.assign(NUM_PERM, std::vector<uint32_t>(n, 1));
    for (uint32_t c = 0; c < NUM_PERM; c++) {
      mapping[c].resize(n);
      aux_[c].resize(n);
      for (uint32_t r = 0; r < n; r++) mapping[c][r] = aux_[c][r] = {c, r};
    }
    next_free_.assign(NUM_COLS, 0);
    synthesize();
  }

  // sigma columns: the label delta^column * omega^row of the cell each cell maps to
  std::vector<std::vector<Fr>> sigma(const Fr& omega) const {
    const Fr delta = Fr::from_u64(7).pow((uint64_t)1 << 28);
    std::vector<std::vector<Fr>> labels(NUM_PERM, std::vector<Fr>(n)), out(NUM_PERM, std::vector<Fr>(n));
    Fr start = Fr::one();
    for (uint32_t c = 0; c < NUM_PERM; c++) {
      Fr v = start;
      for (size_t r = 0; r < n; r++) {
        labels[c][r] = v;
        v = v * omega;
      }
      start = start * delta;
    }
    for (uint32_t c = 0; c < NUM_PERM; c++)
      for (size_t r = 0; r < n; r++) out[c][r] = labels[mapping[c][r].first][mapping[c][r].second];
    return out;
  }

 private:
  std::vector<uint32_t> next_free_;
  std::vector<std::vector<std::pair<uint32_t, uint32_t>>> aux_;
  std::vector<std::vector<uint32_t>> sizes_;
  struct Item { uint32_t kind, col, row, sym, extra; };
  std::vector<Item> cells_, ranges_, hashes_;
  std::vector<std::array<uint32_t, 3>> absorbs_;

  uint32_t region(std::initializer_list<int> cols, uint32_t rows) {
    uint32_t start = 0;
    for (int c : cols) start = std::max(start, next_free_[c]);
    for (int c : cols) next_free_[c] = start + rows;
    if (start + rows > n - (BLINDING_FACTORS + 1)) throw std::runtime_error("the circuit does not fit 2^k rows");
    return start;
  }
  void copy(Cell l, Cell r) {   // halo2 permutation/keygen.rs: Assembly::copy
    std::pair<uint32_t, uint32_t> left{(uint32_t)perm_index(l.col), l.row}, right{(uint32_t)perm_index(r.col), r.row};
    auto lc = aux_[left.first][left.second], rc = aux_[right.first][right.second];
    if (lc == rc) return;
    if (sizes_[lc.first][lc.second] < sizes_[rc.first][rc.second]) std::swap(lc, rc);
    sizes_[lc.first][lc.second] += sizes_[rc.first][rc.second];
    auto i = rc;
    while (true) {
      aux_[i.first][i.second] = lc;
      i = mapping[i.first][i.second];
      if (i == rc) break;
    }
    std::swap(mapping[left.first][left.second], mapping[right.first][right.second]);
  }
  void constants(std::initializer_list<std::pair<Fr, Cell>> items) {   // into the constants column (fixed 2), right after the region
    for (auto& it : items) {
      const uint32_t row = next_free_[fixed_col(2)]++;
      fixed[2][row] = it.first;
      copy(Cell{fixed_col(2), row}, it.second);
    }
  }
  Cell witness(int column, uint32_t s) {
    const uint32_t row = region({column}, 1);
    cells_.push_back({0, (uint32_t)column, row, s, 0});
    return Cell{column, row};
  }
  // a Poseidon sponge over `inputs` (cells and the symbols they hold): initial state, per word an add-input and a 37-row
  // permute region; returns the digest's cell
  Cell hash(int chip, const std::vector<std::pair<Cell, uint32_t>>& inputs) {
    const Poseidon& P = poseidon();
    const int s_full = chip == 1 ? 7 : 9, s_partial = chip == 1 ? 8 : 10;
    const int sel_pad = chip == 1 ? SEL_PAD1 : SEL_PAD2, sel_full = chip == 1 ? SEL_FULL1 : SEL_FULL2, sel_part = chip == 1 ? SEL_PART1 : SEL_PART2;
    uint32_t st = region({A0, A1}, 1);
    Cell state[2] = {{A0, st}, {A1, st}};
    Fr cap = Fr::zero();
    {   // L * 2^64
      uint64_t limbs[4] = {0, (uint64_t)inputs.size(), 0, 0};
      cap = Fr::from_canonical_limbs(limbs);
    }
    constants({{Fr::zero(), state[0]}, {cap, state[1]}});
    const uint32_t first = (uint32_t)absorbs_.size();
    const uint32_t init_row = st;
    for (auto& in : inputs) {
      st = region({A0, A1, sel_pad}, 3);
      fixed[6][st + 1] = Fr::from_u64(chip == 1 ? 3 : 4);
      copy(Cell{A0, st}, state[0]);
      copy(Cell{A1, st}, state[1]);
      copy(Cell{A0, st + 1}, in.first);
      state[0] = Cell{A0, st + 2};
      state[1] = Cell{A1, st + 2};
      const uint32_t add_row = st;
      st = region({A0, A1, A2, fixed_col(0), fixed_col(1), fixed_col(2), fixed_col(3), sel_full, sel_part}, 37);
      copy(Cell{A0, st}, state[0]);
      copy(Cell{A1, st}, state[1]);
      uint32_t row = st;
      for (int r = 0; r < 4; r++, row++) {
        fixed[s_full][row] = Fr::one();
        fixed[0][row] = P.rc[r][0];
        fixed[1][row] = P.rc[r][1];
      }
      for (int j = 0; j < 28; j++, row++) {
        fixed[s_partial][row] = Fr::one();
        fixed[0][row] = P.rc[4 + 2 * j][0];
        fixed[1][row] = P.rc[4 + 2 * j][1];
        fixed[2][row] = P.rc[5 + 2 * j][0];
        fixed[3][row] = P.rc[5 + 2 * j][1];
      }
      for (int r = 60; r < 64; r++, row++) {
        fixed[s_full][row] = Fr::one();
        fixed[0][row] = P.rc[r][0];
        fixed[1][row] = P.rc[r][1];
      }
      state[0] = Cell{A0, st + 36};
      state[1] = Cell{A1, st + 36};
      absorbs_.push_back({add_row, st, in.second});
    }
    hashes_.push_back({2, 0, init_row, 0, first | (uint32_t)inputs.size() << 20 | (uint32_t)chip << 28});
    return state[0];
  }
  void range_check(Cell cell, uint32_t s) {
    const uint32_t st = region({A0, SEL_LOOKUP}, n_bytes + 1);
    for (uint32_t i = 0; i < n_bytes; i++) fixed[5][st + i] = Fr::one();
    copy(Cell{A0, st}, cell);
    constants({{Fr::zero(), Cell{A0, st + n_bytes}}});
    ranges_.push_back({1, 0, st, s, n_bytes});
  }

  void synthesize() {
    auto path_hash = [](uint32_t level) { return sym(SYM_HASH, level, MODE_PATH); };
    using In = std::pair<Cell, uint32_t>;
    const In user{witness(A0, sym(SYM_USER)), sym(SYM_USER)};
    std::vector<In> cur_bal;
    for (uint32_t c = 0; c < nc; c++) cur_bal.push_back({witness(A1, sym(SYM_BAL, 0, MODE_PATH, c)), sym(SYM_BAL, 0, MODE_PATH, c)});
    std::vector<In> in{user};
    in.insert(in.end(), cur_bal.begin(), cur_bal.end());
    In cur_hash{hash(1, in), path_hash(0)};
    copy(cur_hash.first, Cell{INSTANCE, 0});
    {   // the 8-bit range table
      const uint32_t st = region({fixed_col(4)}, 256);
      for (uint32_t i = 0; i < 256; i++) fixed[4][st + i] = Fr::from_u64(i);
    }
    for (uint32_t level = 0; level < levels; level++) {
      std::vector<In> sib_bal;
      In sib_hash{};
      if (level == 0) {
        const In sib_user{witness(A0, sym(SYM_USER, 0, 1)), sym(SYM_USER, 0, 1)};
        for (uint32_t c = 0; c < nc; c++) sib_bal.push_back({witness(A1, sym(SYM_BAL, 0, MODE_SIBLING, c)), sym(SYM_BAL, 0, MODE_SIBLING, c)});
        std::vector<In> hin{sib_user};
        hin.insert(hin.end(), sib_bal.begin(), sib_bal.end());
        sib_hash = {hash(1, hin), sym(SYM_HASH, 0, MODE_SIBLING)};
        for (uint32_t c = 0; c < nc; c++) {
          range_check(cur_bal[c].first, cur_bal[c].second);
          range_check(sib_bal[c].first, sib_bal[c].second);
        }
      } else {
        for (uint32_t c = 0; c < nc; c++) sib_bal.push_back({witness(A1, sym(SYM_BAL, level, MODE_SIBLING, c)), sym(SYM_BAL, level, MODE_SIBLING, c)});
        const In left{witness(A2, sym(SYM_HASH, level, MODE_SIBLING_CHILD, 0)), sym(SYM_HASH, level, MODE_SIBLING_CHILD, 0)};
        const In right{witness(A2, sym(SYM_HASH, level, MODE_SIBLING_CHILD, 1)), sym(SYM_HASH, level, MODE_SIBLING_CHILD, 1)};
        std::vector<In> hin = sib_bal;
        hin.push_back(left);
        hin.push_back(right);
        sib_hash = {hash(2, hin), sym(SYM_HASH, level, MODE_SIBLING)};
        for (uint32_t c = 0; c < nc; c++) range_check(sib_bal[c].first, sib_bal[c].second);
      }
      const In bit{witness(A0, sym(SYM_BIT, level)), sym(SYM_BIT, level)};
      // swap: 2 rows
      uint32_t st = region({A0, A1, A2, SEL_SWAP}, 2);
      fixed[6][st] = Fr::from_u64(1);
      copy(Cell{A0, st}, cur_hash.first);
      copy(Cell{A1, st}, sib_hash.first);
      copy(Cell{A2, st}, bit.first);
      const In left{Cell{A0, st + 1}, sym(SYM_HASH, level, MODE_ORDERED_CHILD, 0)}, right{Cell{A1, st + 1}, sym(SYM_HASH, level, MODE_ORDERED_CHILD, 1)};
      cells_.push_back({0, A0, st, cur_hash.second, 0});
      cells_.push_back({0, A1, st, sib_hash.second, 0});
      cells_.push_back({0, A2, st, bit.second, 0});
      cells_.push_back({0, A0, st + 1, left.second, 0});
      cells_.push_back({0, A1, st + 1, right.second, 0});
      // sums: 1 row per currency
      std::vector<In> nxt;
      for (uint32_t c = 0; c < nc; c++) {
        st = region({A0, A1, A2, SEL_SUM}, 1);
        fixed[6][st] = Fr::from_u64(2);
        copy(Cell{A0, st}, cur_bal[c].first);
        copy(Cell{A1, st}, sib_bal[c].first);
        const uint32_t total = sym(SYM_BAL, level + 1, MODE_PATH, c);
        cells_.push_back({0, A0, st, cur_bal[c].second, 0});
        cells_.push_back({0, A1, st, sib_bal[c].second, 0});
        cells_.push_back({0, A2, st, total, 0});
        nxt.push_back({Cell{A2, st}, total});
      }
      cur_bal = nxt;
      std::vector<In> hin = cur_bal;
      hin.push_back(left);
      hin.push_back(right);
      cur_hash = {hash(2, hin), path_hash(level + 1)};
    }
    copy(cur_hash.first, Cell{INSTANCE, 1});
    for (uint32_t c = 0; c < nc; c++) copy(cur_bal[c].first, Cell{INSTANCE, 2 + c});
    instance_symbols = {path_hash(0), path_hash(levels)};
    for (uint32_t c = 0; c < nc; c++) instance_symbols.push_back(sym(SYM_BAL, levels, MODE_PATH, c));
    // program: sponges first (whole waves), then the range checks, then the single cells
    for (auto* list : {&hashes_, &ranges_, &cells_})
      for (auto& it : *list) program.insert(program.end(), {it.kind, it.col, it.row, it.sym, it.extra});
    n_items = (uint32_t)(hashes_.size() + ranges_.size() + cells_.size());
    for (auto& ab : absorbs_) program.insert(program.end(), ab.begin(), ab.end());
    n_absorbs = (uint32_t)absorbs_.size();
    for (int c = 0; c < NUM_COLS; c++) rows_used = std::max(rows_used, next_free_[c]);
  }
};

// ------------------------------------------------------------------ verifying-key digest, small host helpers
// halo2's `VerifyingKey::transcript_repr`: Blake2b-512("Halo2-Verify-Key") of len || `{:?}` text of the pinned key, reduced
// wide mod r.  The text spells the constraint system out -- every gate as an expression tree, the query lists in the order
// `MstInclusionConfig::configure` [REF zk_prover/src/circuits/merkle_sum_tree.rs:141-207] made them -- so it is rebuilt by
// replaying configure with halo2's operator rules (a - b = Sum(a, Negated(b)), e * scalar = Scaled).  Twin of
// circuits_halo2_amd/vk_repr.py, which the reference's `vk_digest` (contracts/src/InclusionVerifier.sol:217) pins.
struct PinnedText {
  std::vector<std::pair<uint32_t, int32_t>> queries[3];   // advice, fixed, instance: (column, rotation) in first-use order
  std::vector<std::pair<int, uint32_t>> permutation;      // (kind, column)
  std::vector<std::string> gates;
  enum { ADV = 0, FIX = 1, INST = 2 };
  static const char* kind_name(int kind) { return kind == ADV ? "Advice" : kind == FIX ? "Fixed" : "Instance"; }
  static std::string hex(const Fr& v) {
    uint8_t be[32];
    v.to_be_bytes(be);
    static const char* d = "0123456789abcdef";
    std::string s = "0x";
    for (int i = 0; i < 32; i++) {
      s += d[be[i] >> 4];
      s += d[be[i] & 15];
    }
    return s;
  }
  std::string q(int kind, uint32_t col, int32_t rot = 0) {
    auto& list = queries[kind];
    size_t i = 0;
    while (i < list.size() && list[i] != std::make_pair(col, rot)) i++;
    if (i == list.size()) list.emplace_back(col, rot);
    return std::string(kind_name(kind)) + " { query_index: " + std::to_string(i) + ", column_index: " + std::to_string(col) +
           ", rotation: Rotation(" + std::to_string(rot) + ") }";
  }
  void enable_equality(int kind, uint32_t col) {
    q(kind, col, 0);
    for (auto& c : permutation)
      if (c.first == kind && c.second == col) return;
    permutation.emplace_back(kind, col);
  }
  static std::string neg(const std::string& a) { return "Negated(" + a + ")"; }
  static std::string sum(const std::string& a, const std::string& b) { return "Sum(" + a + ", " + b + ")"; }
  static std::string sub(const std::string& a, const std::string& b) { return sum(a, neg(b)); }
  static std::string mul(const std::string& a, const std::string& b) { return "Product(" + a + ", " + b + ")"; }
  static std::string scaled(const std::string& a, const Fr& c) { return "Scaled(" + a + ", " + hex(c) + ")"; }
  static std::string cst(uint64_t v) { return "Constant(" + hex(Fr::from_u64(v)) + ")"; }
  static std::string pow5(const std::string& v) {
    const std::string v2 = mul(v, v);
    return mul(mul(v2, v2), v);
  }
  static std::string sel(int i) { return "@" + std::to_string(i) + "@"; }   // replaced by compress()

  // halo2_gadgets' Pow5Chip::configure, WIDTH 2 / RATE 1, on state = a0 a1, partial_sbox = a2, rc_a = f0 f1, rc_b = f2 f3
  void pow5_chip(int s_full, int s_partial, int s_pad) {
    const Poseidon& P = poseidon();
    for (uint32_t c = 0; c < 2; c++) enable_equality(ADV, c);
    for (uint32_t c = 2; c < 4; c++) enable_equality(FIX, c);
    for (uint32_t nxt = 0; nxt < 2; nxt++) {   // "full round"
      const std::string state_next = q(ADV, nxt, 1);
      std::string expr;
      for (uint32_t idx = 0; idx < 2; idx++) {
        const std::string term = scaled(pow5(sum(q(ADV, idx), q(FIX, idx))), P.mds[nxt][idx]);
        expr = idx ? sum(expr, term) : term;
      }
      gates.push_back(mul(sel(s_full), sub(expr, state_next)));
    }
    const std::string cur_0 = q(ADV, 0), mid_0 = q(ADV, 2), rc_a0 = q(FIX, 0), rc_b0 = q(FIX, 2);   // "partial rounds"
    auto mid = [&](uint32_t idx) { return sum(scaled(mid_0, P.mds[idx][0]), scaled(sum(q(ADV, 1), q(FIX, 1)), P.mds[idx][1])); };
    auto next = [&](uint32_t idx) { return sum(scaled(q(ADV, 0, 1), P.mds_inv[idx][0]), scaled(q(ADV, 1, 1), P.mds_inv[idx][1])); };
    gates.push_back(mul(sel(s_partial), sub(pow5(sum(cur_0, rc_a0)), mid_0)));
    gates.push_back(mul(sel(s_partial), sub(pow5(sum(mid(0), rc_b0)), next(0))));
    {
      const std::string rc_b1 = q(FIX, 3);
      const std::string m = mid(1);
      gates.push_back(mul(sel(s_partial), sub(sum(m, rc_b1), next(1))));
    }
    const std::string initial_rate = q(ADV, 1, -1), output_rate = q(ADV, 1, 1);                       // "pad-and-add"
    const std::string a_prev = q(ADV, 0, -1), a_cur = q(ADV, 0), a_next = q(ADV, 0, 1);
    gates.push_back(mul(sel(s_pad), sub(sum(a_prev, a_cur), a_next)));
    gates.push_back(mul(sel(s_pad), sub(initial_rate, output_rate)));
  }

  // the text of PinnedConstraintSystem after keygen's selector compression
  std::string build(uint32_t n_currencies) {
    // selectors in creation order: 0 swap, 1 sum, 2 lookup (complex), 3-5 entry hasher, 6-8 middle hasher
    enable_equality(FIX, 2);                                   // enable_constant(fixed[2])
    pow5_chip(3, 4, 5);
    pow5_chip(6, 7, 8);
    for (uint32_t c = 0; c < 3; c++) enable_equality(ADV, c);
    {   // MerkleSumTreeChip::configure [REF chips/merkle_sum_tree.rs:39-95]
      const std::string swap = q(ADV, 2);
      gates.push_back(mul(mul(sel(0), swap), sub(cst(1), swap)));
      const std::string l = q(ADV, 0), r = q(ADV, 1), ln = q(ADV, 0, 1), rn = q(ADV, 1, 1);
      gates.push_back(mul(sel(0), sub(sum(mul(sub(r, l), swap), l), ln)));
      gates.push_back(mul(sel(0), sub(sum(mul(sub(l, r), swap), r), rn)));
      for (uint32_t c = 0; c < n_currencies; c++) gates.push_back(mul(sel(1), sub(sum(q(ADV, 0), q(ADV, 1)), q(ADV, 2))));
    }
    // RangeCheckChip::configure [REF chips/range/range_check.rs:28-56] on advice[0], table f4
    const std::string z_cur = q(ADV, 0), z_next = q(ADV, 0, 1), table = q(FIX, 4);
    std::string lookup_input = mul(sel(2), sub(z_cur, mul(z_next, cst(256))));
    enable_equality(INST, 0);
    // compress_selectors: the complex selector gets f5; swap, sum and the two pad selectors (gate degree <= 3) share f6 as
    // values 1..4; the four degree-6 Poseidon selectors get f7..f10 (the floor plan above writes exactly these columns)
    std::string rep[9];
    rep[2] = q(FIX, 5);
    const std::string q6 = q(FIX, 6);
    const int combo[4] = {0, 1, 5, 8};
    for (int m = 0; m < 4; m++) {
      std::string e = q6;
      for (uint64_t root = 1; root <= 4; root++)
        if (root != (uint64_t)m + 1) e = mul(e, sub(cst(root), q6));
      rep[combo[m]] = e;
    }
    rep[3] = q(FIX, 7);
    rep[4] = q(FIX, 8);
    rep[6] = q(FIX, 9);
    rep[7] = q(FIX, 10);
    auto substitute = [&](std::string& t) {
      for (int i = 0; i < 9; i++) {
        const std::string key = sel(i);
        for (size_t at = t.find(key); at != std::string::npos; at = t.find(key, at + rep[i].size())) t.replace(at, key.size(), rep[i]);
      }
    };
    auto column = [](int kind, uint32_t i) { return "Column { index: " + std::to_string(i) + ", column_type: " + kind_name(kind) + " }"; };
    std::string out = "PinnedConstraintSystem { num_fixed_columns: 11, num_advice_columns: 3, num_instance_columns: 1, num_selectors: 9, gates: [";
    for (size_t i = 0; i < gates.size(); i++) {
      substitute(gates[i]);
      out += (i ? ", " : "") + gates[i];
    }
    out += "]";
    const std::pair<const char*, int> lists[3] = {{"advice_queries", ADV}, {"instance_queries", INST}, {"fixed_queries", FIX}};
    for (auto& l : lists) {
      out += std::string(", ") + l.first + ": [";
      for (size_t i = 0; i < queries[l.second].size(); i++)
        out += std::string(i ? ", " : "") + "(" + column(l.second, queries[l.second][i].first) + ", Rotation(" + std::to_string(queries[l.second][i].second) + "))";
      out += "]";
    }
    out += ", permutation: Argument { columns: [";
    for (size_t i = 0; i < permutation.size(); i++) out += (i ? ", " : "") + column(permutation[i].first, permutation[i].second);
    substitute(lookup_input);
    out += "] }, lookups: [Argument { input_expressions: [" + lookup_input + "], table_expressions: [" + table + "] }]";
    out += ", constants: [" + column(FIX, 2) + "], minimum_degree: None }";
    return out;
  }
};

// comms: 17 commitments as the ABI returns them (64-byte Montgomery affine), fixed first.  Returns 32 bytes big-endian.
inline std::array<uint8_t, 32> verifying_key_digest(uint32_t k, uint32_t n_currencies, const std::vector<std::array<uint8_t, 64>>& comms) {
  auto hex_be = [](const uint8_t mont[32]) {
    uint8_t be[32];
    prover::fq_mont_to_be(mont, be);
    static const char* d = "0123456789abcdef";
    std::string s = "0x";
    for (int i = 0; i < 32; i++) {
      s += d[be[i] >> 4];
      s += d[be[i] & 15];
    }
    return s;
  };
  auto pts = [&](size_t lo, size_t hi) {
    std::string s;
    for (size_t i = lo; i < hi; i++) {
      bool zero = true;
      for (uint8_t b : comms[i]) zero = zero && b == 0;
      s += (i > lo ? ", " : "") + (zero ? std::string("Infinity") : "(" + hex_be(comms[i].data()) + ", " + hex_be(comms[i].data() + 32) + ")");
    }
    return s;
  };
  // generator of the 2^k domain: Fr::ROOT_OF_UNITY (order 2^28) squared down (EvaluationDomain::new)
  static const uint8_t root_2_28[32] = {0x03, 0xdd, 0xb9, 0xf5, 0x16, 0x6d, 0x18, 0xb7, 0x98, 0x86, 0x5e, 0xa9, 0x3d, 0xd3, 0x1f, 0x74,
                                        0x32, 0x15, 0xcf, 0x6d, 0xd3, 0x93, 0x29, 0xc8, 0xd3, 0x4f, 0x1e, 0xd9, 0x60, 0xc3, 0x7c, 0x9c};
  if (k > 28) throw std::invalid_argument("k out of range");
  Fr omega = Fr::from_be_bytes_reduced(root_2_28);
  for (uint32_t i = k; i < 28; i++) omega = omega * omega;
  const std::string r = "PinnedVerificationKey { base_modulus: \"0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47\", "
                        "scalar_modulus: \"0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001\", "
                        "domain: PinnedEvaluationDomain { k: " + std::to_string(k) + ", extended_k: " + std::to_string(k + 3) +
                        ", omega: " + PinnedText::hex(omega) + " }, cs: " + PinnedText().build(n_currencies) +
                        ", fixed_commitments: [" + pts(0, NUM_FIXED) + "], permutation: VerifyingKey { commitments: [" + pts(NUM_FIXED, comms.size()) + "] } }";
  prover::Blake2b h(64, "Halo2-Verify-Key");
  const uint64_t len = r.size();
  h.update(reinterpret_cast<const uint8_t*>(&len), 8);
  h.update(reinterpret_cast<const uint8_t*>(r.data()), r.size());
  uint8_t d[64];
  h.finalize(d, 64);
  Fr lo, hi, r2;
  std::memcpy(lo.l, d, 32);
  std::memcpy(hi.l, d + 32, 32);
  std::memcpy(r2.l, Fr::R2, 32);
  const Fr v = lo * r2 + (hi * r2) * r2;
  std::array<uint8_t, 32> out;
  v.to_be_bytes(out.data());
  return out;
}
// decimal string -> Fr (reduced mod r, like `Fp::from_str_vartime` in zk_prover/src/merkle_sum_tree/utils/operation_helpers.rs:10-12);
// only digits are accepted (the reference's BigUint::parse_bytes rejects signs, blanks and underscores)
inline Fr fr_from_decimal(const std::string& s) {
  if (s.empty()) throw std::invalid_argument("Invalid balance");
  Fr acc = Fr::zero();
  const Fr ten = Fr::from_u64(10);
  for (char ch : s) {
    if (ch < '0' || ch > '9') throw std::invalid_argument("Invalid balance");
    acc = acc * ten + Fr::from_u64((uint64_t)(ch - '0'));
  }
  return acc;
}
// keccak256(username) as a big-endian integer mod r (zk_prover/src/merkle_sum_tree/entry.rs:21)
inline Fr fr_from_username(const std::string& name) {
  const auto h = prover::keccak256(reinterpret_cast<const uint8_t*>(name.data()), name.size());
  return Fr::from_be_bytes_reduced(h.data());
}

}  // namespace circuit
}  // namespace summa
