"""Constraint system of the reference's `MstInclusionCircuit<LEVELS, N_CURRENCIES = 2, N_BYTES = 8>` as input
for the device-side `evaluate_h` (arithmetic.GraphEvaluator / quotient_permutation / quotient_lookup).

Host mirror of what `halo2_proofs::plonk::Evaluator::new(cs)` derives from the circuit's `ConstraintSystem`
[REF zk_prover/src/circuits/merkle_sum_tree.rs:143-196 (configure), chips/poseidon/poseidon_chip.rs (Pow5 chip,
WIDTH 2 / RATE 1), chips/merkle_sum_tree.rs (swap / sum gates), chips/range/range_check.rs (8-bit lookup)];
the resulting polynomial list is the one the generated verifier folds [REF contracts/src/InclusionVerifier.sol:495-1000]
and is pinned against it by tests/test_verifier_cpu.py (same gate values as the restated verifier that accepts the
reference's shipped proof) and, on the GPU, row by row by tests/test_gpu_parity.py.

Column layout (halo2's column numbering after selector compression, as in the verifying key):
  advice a0, a1 (Poseidon state / chip operands), a2 (partial-round s-box, swap bit, sum)
  fixed  f0, f1 = rc_a; f2, f3 = rc_b; f4 = range table; f5 = lookup selector; f6 = compressed simple selector
         (1: swap, 2: sum, 3 / 4: pad-and-add of Poseidon chip 1 / 2); f7, f8 = s_full, s_partial of chip 1; f9, f10 of chip 2
  instance i0;  permutation over (f2, a0, a1, f3, a2, i0) in chunks of 4;  one lookup.
"""
from __future__ import annotations

from . import arithmetic as A
from .arithmetic import ADVICE, FIXED, INSTANCE
from functools import lru_cache

from .poseidon_params import generate as _generate_poseidon


@lru_cache(maxsize=None)
def _poseidon():
    return _generate_poseidon()


R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
NUM_ADVICE, NUM_FIXED, NUM_INSTANCE = 3, 11, 1
BLINDING_FACTORS = 5
DEGREE = 6                      # cs.degree(): extended domain = 2^(k + 3), 5 quotient pieces
PERMUTATION_COLUMNS = [(A.FIXED, 2), (A.ADVICE, 0), (A.ADVICE, 1), (A.FIXED, 3), (A.ADVICE, 2), (A.INSTANCE, 0)]
PERMUTATION_CHUNK = DEGREE - 2


class Expr:
    """Polynomial expression over column queries (halo2's `Expression`), just enough for this circuit."""

    def __init__(self, op, *args):
        self.op, self.args = op, args

    @staticmethod
    def const(v: int):
        return Expr("const", v % R)

    @staticmethod
    def query(kind: int, column: int, rotation: int = 0):
        return Expr("query", kind, column, rotation)

    def _wrap(self, o):
        return o if isinstance(o, Expr) else Expr.const(o)

    def __add__(self, o):
        return Expr("add", self, self._wrap(o))

    def __sub__(self, o):
        return Expr("sub", self, self._wrap(o))

    def __rsub__(self, o):
        return Expr("sub", self._wrap(o), self)

    def __mul__(self, o):
        return Expr("mul", self, self._wrap(o))

    __radd__, __rmul__ = __add__, __mul__

    def pow5(self):
        sq = self * self
        return sq * sq * self

    def degree(self) -> int:
        if self.op == "const":
            return 0
        if self.op == "query":
            return 1
        d = [a.degree() for a in self.args]
        return sum(d) if self.op == "mul" else max(d)

    def evaluate(self, q) -> int:
        """q(kind, column, rotation) -> int; integer evaluation (used by the CPU tests)"""
        if self.op == "const":
            return self.args[0]
        if self.op == "query":
            return q(*self.args)
        a, b = (x.evaluate(q) for x in self.args)
        return (a + b) % R if self.op == "add" else (a - b) % R if self.op == "sub" else a * b % R

    def lower(self, g: A.GraphEvaluator):
        """append to a GraphEvaluator the way upstream's `add_expression` does; returns the value source"""
        if self.op == "const":
            return g.add_constant(((self.args[0] << 256) % R).to_bytes(32, "little"))
        if self.op == "query":
            return g.query(*self.args)
        a, b = self.args
        if self.op == "mul" and a is b:
            return g.add_calculation(A.SQUARE, a.lower(g))
        va, vb = a.lower(g), b.lower(g)
        if self.op == "mul" and va == vb:
            return g.add_calculation(A.SQUARE, va)
        return g.add_calculation({"add": A.ADD, "sub": A.SUB, "mul": A.MUL}[self.op], va, vb)


def gates(n_currencies: int = 2):
    """the gate polynomials in the constraint system's order: 17 + one sum gate per currency (19 for the reference's
    `MstInclusionCircuit<4, 2, 8>`)"""
    _, mds, mds_inv = _poseidon()
    a = lambda c, r=0: Expr.query(A.ADVICE, c, r)
    f = lambda c: Expr.query(A.FIXED, c, 0)
    out = []

    def poseidon_chip(s_full, s_partial):
        sbox = [(a(j) + f(j)).pow5() for j in range(2)]
        for i in range(2):   # full round
            out.append(s_full * (sbox[0] * mds[i][0] + sbox[1] * mds[i][1] - a(i, 1)))
        # two partial rounds per row
        out.append(s_partial * (sbox[0] - a(2)))
        mid = [a(2), a(1) + f(1)]
        r_mid = [mid[0] * mds[i][0] + mid[1] * mds[i][1] for i in range(2)]
        nxt = [a(0, 1) * mds_inv[i][0] + a(1, 1) * mds_inv[i][1] for i in range(2)]
        out.append(s_partial * ((r_mid[0] + f(2)).pow5() - nxt[0]))
        out.append(s_partial * (r_mid[1] + f(3) - nxt[1]))

    def simple_selector(value):
        s = f(6)
        for v in range(1, 5):
            if v != value:
                s = s * (v - f(6))
        return s

    def pad_and_add(value):
        s = simple_selector(value)
        out.append(s * (a(0, -1) + a(0) - a(0, 1)))
        out.append(s * (a(1, -1) - a(1, 1)))

    poseidon_chip(f(7), f(8))
    pad_and_add(3)
    poseidon_chip(f(9), f(10))
    pad_and_add(4)
    s = simple_selector(1)
    out.append(s * a(2) * (1 - a(2)))
    out.append(s * ((a(1) - a(0)) * a(2) + a(0) - a(0, 1)))
    out.append(s * ((a(0) - a(1)) * a(2) + a(1) - a(1, 1)))
    s = simple_selector(2)
    for _ in range(n_currencies):   # one sum gate per currency
        out.append(s * (a(0) + a(1) - a(2)))
    return out


def lookup_expressions():
    """(input, table) of the 8-bit range check: f5 * (a0 - 2^8 * a0_next) in f4"""
    a0, a0n = Expr.query(A.ADVICE, 0, 0), Expr.query(A.ADVICE, 0, 1)
    return Expr.query(A.FIXED, 5, 0) * (a0 - a0n * 256), Expr.query(A.FIXED, 4, 0)


# The gate block of evaluate_h is   values * y^Ng + sum_i G_i y^(Ng - 1 - i)   over the Ng = 17 + n_currencies gate polynomials
# of `gates()`.  halo2's GraphEvaluator spells that as one Horner fold over the G_i; the value is what matters, and this
# circuit's structure allows a much cheaper program for the same value:
#   * the two Poseidon chips sit on the same columns: gates 7..13 are gates 0..6 with another selector, i.e. the same
#     expressions E_k shifted by y^-7.  With J = sum_k E_k y^(Ng - 8 - k) both chips' gates of a selector pair (s, s') are
#     J * (s y^7 + s'): every E_k is evaluated once and multiplied once.
#   * the four simple selectors packed into f6 are q prod_{j != i} (j - q); since (j + 1 - q) = (j - q) + 1,
#     sel3 = sel4 + q u1 u2 and sel1 = sel2 + q u3 u4 (u_j = j - q): 6 products instead of 12.
#   * no fold is in gate order any more, so the needed powers of y come in as challenges (gate_challenge_exponents);
#     every term is added to the running value as soon as it is complete and its operands die.
# 53 products per row instead of 73, and 8 simultaneously live values instead of 13 under the interpreter's allocator
# (csrc/gates.hip; `arithmetic.gates_program_info`): the kernel's occupancy is set by that number (LDS slots per row).
# tests: the value against `gates()` folded with y on the CPU (test_verifier_cpu.py) and row by row on the GPU
# (test_gpu_parity.py), the proofs against the verifier.
@lru_cache(maxsize=None)
def _gate_program(n_currencies: int = 2):
    from .utils import ints_to_fr
    _, mds, mds_inv = _poseidon()
    g = A.GraphEvaluator()
    ng = 17 + n_currencies
    e = lambda i: ng - 1 - i                  # exponent of y on gate i
    groups = []                               # challenge i = sum of y^x over groups[i]

    def chal(exps):
        exps = list(exps)
        if exps not in groups:
            groups.append(exps)
        return (A.CHALLENGE, groups.index(exps), 0)

    const = lambda v: g.add_constant(ints_to_fr([v % R]).tobytes())
    a = lambda c, r=0: g.query(A.ADVICE, c, r)
    f = lambda c: g.query(A.FIXED, c, 0)
    add = lambda x, y: g.add_calculation(A.ADD, x, y)
    sub = lambda x, y: g.add_calculation(A.SUB, x, y)
    mul = lambda x, y: g.add_calculation(A.MUL, x, y)
    sqr = lambda x: g.add_calculation(A.SQUARE, x)

    def pow5(v):
        return mul(sqr(sqr(v)), v)

    y7 = chal([7])
    acc = mul((A.PREVIOUS_VALUE, 0, 0), chal([ng]))
    # -- pad-and-add, swap and sum gates first: their columns die before the Poseidon rounds need the slots
    q = f(6)
    u1, u2, u3, u4 = (sub(const(v), q) for v in (1, 2, 3, 4))
    lo = mul(mul(q, u1), u2)
    sel4 = mul(lo, u3)
    sel3 = add(sel4, lo)
    pad0 = sub(add(a(0, -1), a(0)), a(0, 1))
    pad1 = sub(a(1, -1), a(1, 1))
    j_pad = add(mul(pad0, chal([e(12)])), mul(pad1, chal([e(13)])))
    acc = add(acc, mul(j_pad, add(mul(sel3, y7), sel4)))
    hi = mul(mul(q, u3), u4)
    sel2 = mul(hi, u1)
    sel1 = add(sel2, hi)
    swap_bool = mul(a(2), sub(const(1), a(2)))
    d = mul(sub(a(1), a(0)), a(2))
    swap_l = sub(add(d, a(0)), a(0, 1))
    swap_r = sub(sub(a(1), d), a(1, 1))                   # (a0 - a1) a2 + a1 - a1_next
    inner = add(add(mul(swap_bool, chal([e(14)])), mul(swap_l, chal([e(15)]))), mul(swap_r, chal([e(16)])))
    acc = add(acc, mul(inner, sel1))
    total = sub(add(a(0), a(1)), a(2))
    acc = add(acc, mul(mul(total, chal([e(17 + j) for j in range(n_currencies)])), sel2))
    # -- the Poseidon rounds, both chips at once
    v1 = add(a(1), f(1))
    s1 = pow5(v1)
    s0 = pow5(add(a(0), f(0)))
    full0 = sub(add(mul(s0, const(int(mds[0][0]))), mul(s1, const(int(mds[0][1])))), a(0, 1))
    full1 = sub(add(mul(s0, const(int(mds[1][0]))), mul(s1, const(int(mds[1][1])))), a(1, 1))
    j_full = add(mul(full0, chal([e(7)])), mul(full1, chal([e(8)])))
    acc = add(acc, mul(j_full, add(mul(f(7), y7), f(9))))
    j_part = mul(sub(s0, a(2)), chal([e(9)]))
    mid = [add(mul(a(2), const(int(mds[i][0]))), mul(v1, const(int(mds[i][1])))) for i in range(2)]
    nxt = [add(mul(a(0, 1), const(int(mds_inv[i][0]))), mul(a(1, 1), const(int(mds_inv[i][1])))) for i in range(2)]
    j_part = add(j_part, mul(sub(add(mid[1], f(3)), nxt[1]), chal([e(11)])))
    j_part = add(j_part, mul(sub(pow5(add(mid[0], f(2))), nxt[0]), chal([e(10)])))
    add(acc, mul(j_part, add(mul(f(8), y7), f(10))))      # the last calculation is the row's new value
    return g, tuple(tuple(x) for x in groups)


def gate_graph(n_currencies: int = 2) -> A.GraphEvaluator:
    """the custom-gate part of evaluate_h as a GraphEvaluator program: values <- values * y^Ng + sum_i G_i y^(Ng - 1 - i)"""
    return _gate_program(n_currencies)[0]


def gate_challenge_exponents(n_currencies: int = 2):
    """what the gate program reads as SG_VS_CHALLENGE sources: challenge i = sum of y^e over group i"""
    return [list(x) for x in _gate_program(n_currencies)[1]]


def gate_challenges(y: int, n_currencies: int = 2):
    """the `challenges` array the gate program expects (gate_challenge_exponents evaluated at y)"""
    from .utils import ints_to_fr
    return ints_to_fr([sum(pow(y, x, R) for x in group) % R for group in gate_challenge_exponents(n_currencies)])


@lru_cache(maxsize=None)
def lookup_input_graph() -> A.GraphEvaluator:
    """the lookup's input expression as a program (one value per row)"""
    return expression_graph(lookup_expressions()[0])


def expression_graph(expr: Expr) -> A.GraphEvaluator:
    """a program that stores one expression per row (the compressed lookup input / table columns)"""
    g = A.GraphEvaluator()
    g.add_calculation(A.STORE, expr.lower(g))
    return g


def example_assignment(k: int):
    """A satisfying assignment of this constraint system on 2^k rows (k >= 9) with a floor plan of this repository's
    own: one whole Poseidon permutation on chip 1 (4 + 56 + 4 rounds in 36 rows), single rounds on chip 2, both
    pad-and-add gates, two swaps, a sum, a two-byte range decomposition, copy constraints across advice, fixed and
    instance columns.  Python integers: {fixed [11][n], advice [3][n], instances [4], sigma [6][n], usable_rows, ..}.
    (The reference circuit's own layout is produced by halo2's layouter from zk_prover/src/circuits/merkle_sum_tree.rs
    and is not restated; the constraint system is the same.)"""
    from .prover import DELTA, ROOT_OF_UNITY
    n = 1 << k
    u = n - (BLINDING_FACTORS + 1)
    if n < 512:
        raise ValueError("the 8-bit range table needs 2^9 rows")
    rcs, mds, _ = _poseidon()
    fixed = [[0] * n for _ in range(NUM_FIXED)]
    adv = [[0] * n for _ in range(NUM_ADVICE)]
    pow5 = lambda v: pow(v, 5, R)
    mix = lambda s: [(mds[i][0] * s[0] + mds[i][1] * s[1]) % R for i in range(2)]

    def full_round(row, state, rc, sel):
        fixed[sel][row] = 1
        fixed[0][row], fixed[1][row] = rc
        adv[0][row], adv[1][row] = state
        return mix([pow5((state[j] + rc[j]) % R) for j in range(2)])

    def partial_pair(row, state, rc_a, rc_b, sel):
        fixed[sel][row] = 1
        fixed[0][row], fixed[1][row] = rc_a
        fixed[2][row], fixed[3][row] = rc_b
        adv[0][row], adv[1][row] = state
        adv[2][row] = pow5((state[0] + rc_a[0]) % R)
        mid = mix([adv[2][row], (state[1] + rc_a[1]) % R])
        return mix([pow5((mid[0] + rc_b[0]) % R), (mid[1] + rc_b[1]) % R])

    # chip 1: a whole permutation, rows 0 .. 36
    state = poseidon_in = [0x1234567, (3 << 64) % R]
    row = 0
    for r in range(4):
        state = full_round(row, state, rcs[r], 7)
        row += 1
    for j in range(28):
        state = partial_pair(row, state, rcs[4 + 2 * j], rcs[5 + 2 * j], 8)
        row += 1
    for r in range(60, 64):
        state = full_round(row, state, rcs[r], 7)
        row += 1
    adv[0][row], adv[1][row] = state          # row 36: the output
    poseidon_out = list(state)
    # sum gate (row 55) feeds pad-and-add of chip 1 (rows 37 .. 39) through a copy constraint
    adv[0][55], adv[1][55], adv[2][55] = 1000, 234, 1234
    fixed[6][55] = 2
    adv[0][37], adv[0][38], adv[0][39] = 1234, 66, 1300
    adv[1][37] = adv[1][39] = 77
    fixed[6][38] = 3
    # chip 2: one full round (41 -> 42), one pair of partial rounds (44 -> 45), pad-and-add (46 .. 48)
    adv[0][42], adv[1][42] = full_round(41, [5, 6], rcs[10], 9)
    adv[0][45], adv[1][45] = partial_pair(44, [7, 8], rcs[20], rcs[21], 10)
    adv[0][46], adv[0][47], adv[0][48] = 40, 9, 49
    fixed[6][47] = 4
    # swaps
    adv[0][50], adv[1][50], adv[2][50] = 111, 222, 1
    adv[0][51], adv[1][51] = 222, 111
    fixed[6][50] = 1
    adv[0][52], adv[1][52], adv[2][52] = 333, 444, 0
    adv[0][53], adv[1][53] = 333, 444
    fixed[6][52] = 1
    # range check: 0xABCD = 0xAB * 256 + 0xCD
    adv[0][60], adv[0][61], adv[0][62] = 0xABCD, 0xAB, 0
    fixed[5][60] = fixed[5][61] = 1
    for i in range(256):
        fixed[4][i] = i
    # a constant in a permutation-enabled fixed column, copied into advice
    fixed[2][70] = 5
    adv[1][70] = 5
    instances = [poseidon_out[0], adv[1][51], 556862, 556862]
    inst_col = instances + [0] * (n - len(instances))
    # copy constraints over the permutation columns
    cells = {(ADVICE, j): adv[j] for j in range(NUM_ADVICE)}
    cells.update({(FIXED, 2): fixed[2], (FIXED, 3): fixed[3], (INSTANCE, 0): inst_col})
    groups = [[((ADVICE, 0), 36), ((INSTANCE, 0), 0)], [((ADVICE, 1), 51), ((INSTANCE, 0), 1)],
              [((ADVICE, 2), 55), ((ADVICE, 0), 37)], [((FIXED, 2), 70), ((ADVICE, 1), 70)], [((ADVICE, 0), 53), ((ADVICE, 0), 52)]]
    omega = pow(ROOT_OF_UNITY, 1 << (28 - k), R)
    sigma = []
    for c in range(len(PERMUTATION_COLUMNS)):
        col, v = [], pow(DELTA, c, R)
        for _ in range(n):
            col.append(v)
            v = v * omega % R
        sigma.append(col)
    label = lambda c, i: sigma_identity[c][i]
    sigma_identity = [list(col) for col in sigma]
    for grp in groups:
        if len({cells[col][i] for col, i in grp}) != 1:
            raise AssertionError(grp)
        idx = [(PERMUTATION_COLUMNS.index(col), i) for col, i in grp]
        for (c, i), (c2, i2) in zip(idx, idx[1:] + idx[:1]):
            sigma[c][i] = label(c2, i2)
    return {"fixed": fixed, "advice": adv, "instances": instances, "sigma": sigma, "usable_rows": u,
            "poseidon_in": poseidon_in, "poseidon_out": poseidon_out}


class _Layout:
    """a small region allocator over the constraint system's columns: Poseidon hashes on either chip, swap / sum /
    range-check regions, copy constraints -- enough to assign a Merkle-sum-tree inclusion witness (assign_inclusion)"""

    def __init__(self, k: int):
        self.k, self.n = k, 1 << k
        self.u = self.n - (BLINDING_FACTORS + 1)
        self.fixed = [[0] * self.n for _ in range(NUM_FIXED)]
        self.adv = [[0] * self.n for _ in range(NUM_ADVICE)]
        self.inst = []
        self.row = 0
        self.const_row = self.u - 1          # constants live in f2 from the top of the usable rows downwards
        self.consts = {}
        self.copies = []
        self.rcs, self.mds, _ = _poseidon()
        for i in range(256):
            self.fixed[4][i] = i

    def constant(self, v: int):
        if v not in self.consts:
            self.fixed[2][self.const_row] = v
            self.consts[v] = ((FIXED, 2), self.const_row)
            self.const_row -= 1
        return self.consts[v]

    def copy(self, a, b):
        self.copies.append((a, b))

    def _mix(self, s):
        m = self.mds
        return [(m[i][0] * s[0] + m[i][1] * s[1]) % R for i in range(2)]

    def _full(self, row, state, rc, sel):
        self.fixed[sel][row] = 1
        self.fixed[0][row], self.fixed[1][row] = rc
        self.adv[0][row], self.adv[1][row] = state
        return self._mix([pow((state[j] + rc[j]) % R, 5, R) for j in range(2)])

    def _partial(self, row, state, rc_a, rc_b, sel):
        self.fixed[sel][row] = 1
        self.fixed[0][row], self.fixed[1][row] = rc_a
        self.fixed[2][row], self.fixed[3][row] = rc_b
        self.adv[0][row], self.adv[1][row] = state
        self.adv[2][row] = pow((state[0] + rc_a[0]) % R, 5, R)
        mid = self._mix([self.adv[2][row], (state[1] + rc_a[1]) % R])
        return self._mix([pow((mid[0] + rc_b[0]) % R, 5, R), (mid[1] + rc_b[1]) % R])

    def hash(self, chip: int, inputs):
        """Poseidon(t = 2, rate 1) of `inputs` = [(value, source cell or None)]: initial state [0, L * 2^64] from
        constants, one pad-and-add row + 36 round rows per input; returns (digest, its cell)"""
        s_full, s_partial, pad = (7, 8, 3) if chip == 1 else (9, 10, 4)
        r = self.row
        state = [0, (len(inputs) << 64) % R]
        self.adv[0][r], self.adv[1][r] = state
        self.copy(((ADVICE, 0), r), self.constant(0))
        self.copy(((ADVICE, 1), r), self.constant(state[1]))
        for value, src in inputs:
            self.adv[0][r + 1] = value % R
            self.fixed[6][r + 1] = pad
            if src is not None:
                self.copy(((ADVICE, 0), r + 1), src)
            state = [(state[0] + value) % R, state[1]]
            row = r + 2
            for j in range(4):
                state = self._full(row, state, self.rcs[j], s_full)
                row += 1
            for j in range(28):
                state = self._partial(row, state, self.rcs[4 + 2 * j], self.rcs[5 + 2 * j], s_partial)
                row += 1
            for j in range(60, 64):
                state = self._full(row, state, self.rcs[j], s_full)
                row += 1
            self.adv[0][row], self.adv[1][row] = state
            r = row
        self.row = r + 2
        return state[0], ((ADVICE, 0), r)

    def range_check(self, value: int, n_bytes: int = 8):
        """value < 2^(8 n_bytes) by a running decomposition a0[i+1] = (a0[i] - byte) / 2^8, each byte looked up"""
        t = self.row
        v = value
        for i in range(n_bytes):
            self.adv[0][t + i] = v
            self.fixed[5][t + i] = 1
            v >>= 8
        if v:
            raise ValueError("balance out of range")
        self.adv[0][t + n_bytes] = 0
        self.copy(((ADVICE, 0), t + n_bytes), self.constant(0))
        self.row = t + n_bytes + 2
        return ((ADVICE, 0), t)

    def swap(self, cur, cur_cell, sibling, bit: int):
        t = self.row
        self.adv[0][t], self.adv[1][t], self.adv[2][t] = cur, sibling, bit
        self.fixed[6][t] = 1
        self.copy(((ADVICE, 0), t), cur_cell)
        left, right = (sibling, cur) if bit else (cur, sibling)
        self.adv[0][t + 1], self.adv[1][t + 1] = left, right
        self.row = t + 3
        return (left, ((ADVICE, 0), t + 1)), (right, ((ADVICE, 1), t + 1))

    def add(self, a, a_cell, b, b_cell):
        t = self.row
        self.adv[0][t], self.adv[1][t], self.adv[2][t] = a, b, (a + b) % R
        self.fixed[6][t] = 2
        self.copy(((ADVICE, 0), t), a_cell)
        self.copy(((ADVICE, 1), t), b_cell)
        self.row = t + 2
        return (a + b) % R, ((ADVICE, 2), t)

    def expose(self, value, cell):
        self.copy(cell, ((INSTANCE, 0), len(self.inst)))
        self.inst.append(value)

    def finish(self):
        from .prover import DELTA, ROOT_OF_UNITY
        if self.row > self.const_row:
            raise ValueError("the assignment does not fit 2^k rows")
        n = self.n
        inst_col = self.inst + [0] * (n - len(self.inst))
        cells = {(ADVICE, j): self.adv[j] for j in range(NUM_ADVICE)}
        cells.update({(FIXED, 2): self.fixed[2], (FIXED, 3): self.fixed[3], (INSTANCE, 0): inst_col})
        omega = pow(ROOT_OF_UNITY, 1 << (28 - self.k), R)
        identity = []
        for c in range(len(PERMUTATION_COLUMNS)):
            col, v = [], pow(DELTA, c, R)
            for _ in range(n):
                col.append(v)
                v = v * omega % R
            identity.append(col)
        # union the copy constraints into classes, one cycle per class
        parent = {}

        def find(x):
            while parent.setdefault(x, x) != x:
                parent[x] = parent[parent[x]]
                x = parent[x]
            return x
        for a, b in self.copies:
            if cells[a[0]][a[1]] != cells[b[0]][b[1]]:
                raise AssertionError((a, b))
            parent[find(a)] = find(b)
        classes = {}
        for x in list(parent):
            classes.setdefault(find(x), []).append(x)
        sigma = [list(col) for col in identity]
        for members in classes.values():
            idx = [(PERMUTATION_COLUMNS.index(col), row) for col, row in members]
            for (c, i), (c2, i2) in zip(idx, idx[1:] + idx[:1]):
                sigma[c][i] = identity[c2][i2]
        return {"fixed": self.fixed, "advice": self.adv, "instances": list(self.inst), "sigma": sigma, "usable_rows": self.u,
                "rows_used": self.row, "copies": len(self.copies)}


def assign_inclusion(k: int, username: int, balances, siblings, path_bits):
    """Witness of "this entry is a leaf of the Merkle sum tree with that root" over this constraint system, with the
    public inputs of the reference circuit [REF zk_prover/src/circuits/merkle_sum_tree.rs:40-60, 198-330: leaf hash,
    root hash, root balances]: leaf = H(username, balances..) on the first Poseidon chip, per level a range check of
    the sibling's balances, the swap by the path bit, the per-currency sums and the middle node
    H(sums.., left, right) on the second chip.  siblings: [(hash, [balances])] bottom-up; integers throughout.
    The floor plan is this repository's (see example_assignment); the relations enforced are the circuit's."""
    lay = _Layout(k)
    bal_cells = [lay.range_check(b) for b in balances]
    cur_hash, cur_cell = lay.hash(1, [(username, None)] + list(zip(balances, bal_cells)))
    lay.expose(cur_hash, cur_cell)
    cur_bal = list(zip(balances, bal_cells))
    for (sib_hash, sib_bal), bit in zip(siblings, path_bits):
        sib_cells = [lay.range_check(b) for b in sib_bal]
        (left, left_cell), (right, right_cell) = lay.swap(cur_hash, cur_cell, sib_hash, bit)
        cur_bal = [lay.add(a, a_cell, b, b_cell) for (a, a_cell), b, b_cell in zip(cur_bal, sib_bal, sib_cells)]
        cur_hash, cur_cell = lay.hash(2, cur_bal + [(left, left_cell), (right, right_cell)])
    lay.expose(cur_hash, cur_cell)
    for v, cell in cur_bal:
        lay.expose(v, cell)
    return lay.finish()


# ---------------------------------------------------------------------------------------------------------------------
# The reference circuit's OWN floor plan.  `MstInclusionCircuit::synthesize` [REF zk_prover/src/circuits/merkle_sum_tree.rs:
# 225-519] calls its chips in a fixed order; halo2's SimpleFloorPlanner (upstream: halo2_proofs circuit/floor_planner/
# single_pass.rs) starts every region at the first row where none of the region's columns (advice, fixed, selectors) is in
# use, and assigns a region's constants to the constants column (fixed column 2 here, `enable_constant`, :160) right after
# the region, at that column's next free row.  Region shapes: the reference's own chips [REF chips/merkle_sum_tree.rs:104-
# 226 (swap: 2 rows, sum: 1 row), chips/range/range_check.rs:93-153 (N_BYTES + 1 rows, the final running sum constrained to
# the constant 0), circuits/traits.rs:21-52 (single witness cells; the 256-row table)] and halo2_gadgets' Pow5Chip / sponge
# (upstream, restated from its published source: "initial state" 1 row with two constants, per absorbed word "add input"
# 3 rows with the pad-and-add selector on the middle one and "permute state" 37 rows: state, 4 full rounds, 28 rows of two
# partial rounds, 4 full rounds).  Copy constraints are replayed in call order through halo2's permutation assembly
# (upstream permutation/keygen.rs: cycles merged smaller-into-larger, then the two mapping entries swapped).
# PINNED: with these rules the 11 fixed-column and 6 permutation commitments of the reference's verifying key
# [REF contracts/src/InclusionVerifier.sol:238-271] come out bit for bit (tests/test_verifier_cpu.py).
class _PermutationAssembly:
    def __init__(self, n: int, columns: int):
        self.mapping = [[(i, j) for j in range(n)] for i in range(columns)]
        self.aux = [[(i, j) for j in range(n)] for i in range(columns)]
        self.sizes = [[1] * n for _ in range(columns)]

    def copy(self, left, right):
        left_cycle, right_cycle = self.aux[left[0]][left[1]], self.aux[right[0]][right[1]]
        if left_cycle == right_cycle:
            return
        if self.sizes[left_cycle[0]][left_cycle[1]] < self.sizes[right_cycle[0]][right_cycle[1]]:
            left_cycle, right_cycle = right_cycle, left_cycle
        self.sizes[left_cycle[0]][left_cycle[1]] += self.sizes[right_cycle[0]][right_cycle[1]]
        i = right_cycle
        while True:
            self.aux[i[0]][i[1]] = left_cycle
            i = self.mapping[i[0]][i[1]]
            if i == right_cycle:
                break
        (self.mapping[left[0]][left[1]], self.mapping[right[0]][right[1]]) = (self.mapping[right[0]][right[1]],
                                                                              self.mapping[left[0]][left[1]])


class _ReferenceFloorPlan:
    """cells are (column key, row) with the keys of PERMUTATION_COLUMNS; values are tracked next to the layout"""

    def __init__(self, k: int, lenient: bool = False, instances=None):
        """lenient (the mock prover's view, mock_prover.py): the witness is laid out as given even where it violates a
        copy or range constraint -- halo2's synthesize does not check values either -- and the instance column holds
        `instances` instead of the exposed cells' values"""
        self.k, self.n = k, 1 << k
        self.lenient = lenient
        self.next_free = {}
        self.fixed = [[0] * self.n for _ in range(NUM_FIXED)]
        self.adv = [[0] * self.n for _ in range(NUM_ADVICE)]
        self.instances = dict(enumerate(int(v) % R for v in instances)) if instances is not None else {}
        self.given_instances = instances is not None
        self.asm = _PermutationAssembly(self.n, len(PERMUTATION_COLUMNS))
        self.rcs, self.mds, _ = _poseidon()
        # regions in `assign_region` order with the cells assigned inside them (what halo2's MockProver records:
        # name, assigned columns, first / last assigned row); constants and instance cells lie outside every region
        self.regions = []
        self._cur = None

    # -- floor planner
    def region(self, columns, rows: int, name: str = "") -> int:
        start = max([self.next_free.get(c, 0) for c in columns] + [0])
        for c in columns:
            self.next_free[c] = start + rows
        if start + rows > self.n - (BLINDING_FACTORS + 1):
            raise ValueError("the circuit does not fit 2^k rows")
        self._cur = {"name": name, "columns": set(), "lo": None, "hi": None}
        self.regions.append(self._cur)
        return start

    def _touch(self, column, row: int):
        r = self._cur
        if r is not None:
            r["columns"].add(column)
            r["lo"] = row if r["lo"] is None else min(r["lo"], row)
            r["hi"] = row if r["hi"] is None else max(r["hi"], row)

    def copy(self, left, right):
        lv, rv = self.value(left), self.value(right)
        if lv != rv and not self.lenient:
            raise AssertionError(("copy constraint between unequal cells", left, right))
        self.asm.copy((PERMUTATION_COLUMNS.index(left[0]), left[1]), (PERMUTATION_COLUMNS.index(right[0]), right[1]))

    def value(self, cell):
        (kind, idx), row = cell
        if kind == ADVICE:
            return self.adv[idx][row]
        if kind == FIXED:
            return self.fixed[idx][row]
        return self.instances.get(row, 0)

    def constants(self, items):
        col = (FIXED, 2)
        self._cur = None          # the floor planner assigns a region's constants after it has left the region
        for v, cell in items:
            row = self.next_free.get(col, 0)
            self.next_free[col] = row + 1
            self.fixed[2][row] = v % R
            self.copy((col, row), cell)

    def set(self, cell, v):
        (kind, idx), row = cell
        self.adv[idx][row] = v % R
        self._touch(cell[0], row)
        return cell

    # -- chips
    def witness(self, column: int, v: int, label: str = "value"):
        return self.set(((ADVICE, column), self.region([(ADVICE, column)], 1, "assign " + label)), v)

    def hash(self, chip: int, inputs):
        s_full, s_partial, pad = (7, 8, 3) if chip == 1 else (9, 10, 4)
        a0, a1, a2 = (ADVICE, 0), (ADVICE, 1), (ADVICE, 2)
        st = self.region([a0, a1], 1, "initial state for domain ConstantLength")
        vals = [0, (len(inputs) << 64) % R]
        state = [self.set((a0, st), vals[0]), self.set((a1, st), vals[1])]
        self.constants([(vals[0], state[0]), (vals[1], state[1])])
        mix = lambda s: [(self.mds[i][0] * s[0] + self.mds[i][1] * s[1]) % R for i in range(2)]
        for cell in inputs:
            st = self.region([a0, a1, ("selector", "pad", chip)], 3, "add input for domain ConstantLength")
            self.fixed[6][st + 1] = pad
            self.set((a0, st), vals[0]); self.set((a1, st), vals[1])
            self.copy((a0, st), state[0])
            self.copy((a1, st), state[1])
            self.set((a0, st + 1), self.value(cell))
            self.copy((a0, st + 1), cell)
            vals = [(vals[0] + self.value(cell)) % R, vals[1]]
            state = [self.set((a0, st + 2), vals[0]), self.set((a1, st + 2), vals[1])]
            st = self.region([a0, a1, a2] + [(FIXED, j) for j in range(4)] + [("selector", "full", chip), ("selector", "partial", chip)], 37,
                             "permute state")
            self.set((a0, st), vals[0]); self.set((a1, st), vals[1])
            self.copy((a0, st), state[0])
            self.copy((a1, st), state[1])
            row = st
            for half in (range(0, 4), None, range(60, 64)):
                if half is None:
                    for j in range(28):
                        rc_a, rc_b = self.rcs[4 + 2 * j], self.rcs[5 + 2 * j]
                        self.fixed[s_partial][row] = 1
                        self.fixed[0][row], self.fixed[1][row] = rc_a
                        self.fixed[2][row], self.fixed[3][row] = rc_b
                        for j in range(4):
                            self._touch((FIXED, j), row)
                        sbox = pow((vals[0] + rc_a[0]) % R, 5, R)
                        self.set((a2, row), sbox)
                        mid = mix([sbox, (vals[1] + rc_a[1]) % R])
                        vals = mix([pow((mid[0] + rc_b[0]) % R, 5, R), (mid[1] + rc_b[1]) % R])
                        row += 1
                        self.set((a0, row), vals[0]); self.set((a1, row), vals[1])
                else:
                    for r in half:
                        self.fixed[s_full][row] = 1
                        self.fixed[0][row], self.fixed[1][row] = self.rcs[r]
                        self._touch((FIXED, 0), row); self._touch((FIXED, 1), row)
                        vals = mix([pow((vals[j] + self.rcs[r][j]) % R, 5, R) for j in range(2)])
                        row += 1
                        self.set((a0, row), vals[0]); self.set((a1, row), vals[1])
            state = [(a0, st + 36), (a1, st + 36)]
        return state[0]

    def range_check(self, cell, n_bytes: int):
        a0 = (ADVICE, 0)
        st = self.region([a0, ("selector", "lookup")], n_bytes + 1, "assign value to perform range check")
        v = self.value(cell)
        for i in range(n_bytes):
            self.fixed[5][st + i] = 1
            self.set((a0, st + i), v)
            v >>= 8
        if v and not self.lenient:
            raise ValueError("balance out of range")
        # the last running sum: 0 for a value of n_bytes bytes; what is left of a larger one (the chip decomposes the value's
        # low bytes only, chips/range/utils.rs:12-33, so the running sum does not close and the copy to the constant 0 breaks)
        self.set((a0, st + n_bytes), v)
        self.copy((a0, st), cell)
        self.constants([(0, (a0, st + n_bytes))])

    def swap(self, cur, sibling, bit):
        a0, a1, a2 = (ADVICE, 0), (ADVICE, 1), (ADVICE, 2)
        st = self.region([a0, a1, a2, ("selector", "swap")], 2, "assign nodes hashes per merkle tree level")
        self.fixed[6][st] = 1
        for col, cell in ((a0, cur), (a1, sibling), (a2, bit)):
            self.set((col, st), self.value(cell))
            self.copy((col, st), cell)
        l, r = self.value(cur), self.value(sibling)
        if self.value(bit):       # chips/merkle_sum_tree.rs:150-158: any non-zero bit swaps
            l, r = r, l
        return self.set((a0, st + 1), l), self.set((a1, st + 1), r)

    def add(self, cur, sibling):
        a0, a1, a2 = (ADVICE, 0), (ADVICE, 1), (ADVICE, 2)
        st = self.region([a0, a1, a2, ("selector", "sum")], 1, "sum nodes balances per currency")
        self.fixed[6][st] = 2
        for col, cell in ((a0, cur), (a1, sibling)):
            self.set((col, st), self.value(cell))
            self.copy((col, st), cell)
        return self.set((a2, st), self.value(cur) + self.value(sibling))

    def expose(self, cell, row: int):
        if not self.given_instances:
            self.instances[row] = self.value(cell)
        self.copy(cell, ((INSTANCE, 0), row))

    def finish(self):
        from .prover import DELTA, ROOT_OF_UNITY
        omega = pow(ROOT_OF_UNITY, 1 << (28 - self.k), R)
        labels = []
        for c in range(len(PERMUTATION_COLUMNS)):
            col, v = [], pow(DELTA, c, R)
            for _ in range(self.n):
                col.append(v)
                v = v * omega % R
            labels.append(col)
        sigma = [[labels[m[0]][m[1]] for m in col] for col in self.asm.mapping]
        inst = [self.instances[i] for i in range(len(self.instances))]
        return {"fixed": self.fixed, "advice": self.adv, "instances": inst, "sigma": sigma,
                "usable_rows": self.n - (BLINDING_FACTORS + 1), "rows_used": max(self.next_free.values()),
                "regions": [(r["name"], r["lo"], r["hi"], frozenset(r["columns"])) for r in self.regions],
                "mapping": self.asm.mapping}


def reference_assignment(k: int, username: int, balances, path_bits, sibling_leaf_preimage, sibling_middle_preimages,
                         n_bytes: int = 8, lenient: bool = False, instances=None):
    """`MstInclusionCircuit<LEVELS, N_CURRENCIES, N_BYTES>::synthesize` replayed over the reference's own floor plan
    (see above): fixed columns, permutation and advice assignment exactly as halo2's keygen / prover would hold them.
    Inputs as in the reference's MerkleProof [REF merkle_sum_tree/tree.rs, circuits/merkle_sum_tree.rs:30-60]:
    sibling_leaf_preimage = [username, balances..], sibling_middle_preimages[level - 1] = [balances.., left hash,
    right hash]; integers.  Empty inputs (zeros) give the key-generation view (`init_empty`)."""
    fp = _ReferenceFloorPlan(k, lenient, instances)
    nc = len(balances)
    user = fp.witness(0, username, "entry username")
    cur_bal = [fp.witness(1, b, "entry balance") for b in balances]
    cur_hash = fp.hash(1, [user] + cur_bal)
    fp.expose(cur_hash, 0)
    st = fp.region([(FIXED, 4)], 256, "load range check table of 8 bits")
    for i in range(256):
        fp.fixed[4][st + i] = i
        fp._touch((FIXED, 4), st + i)
    for level, bit_value in enumerate(path_bits):
        if level == 0:
            sib_user = fp.witness(0, sibling_leaf_preimage[0], "sibling leaf node username")
            sib_bal = [fp.witness(1, b, "sibling leaf balance") for b in sibling_leaf_preimage[1:1 + nc]]
            sib_hash = fp.hash(1, [sib_user] + sib_bal)
            for c in range(nc):
                fp.range_check(cur_bal[c], n_bytes)
                fp.range_check(sib_bal[c], n_bytes)
        else:
            pre = sibling_middle_preimages[level - 1]
            sib_bal = [fp.witness(1, b, "sibling node balance") for b in pre[:nc]]
            left = fp.witness(2, pre[nc], "sibling left hash")
            right = fp.witness(2, pre[nc + 1], "sibling right hash")
            sib_hash = fp.hash(2, sib_bal + [left, right])
            for c in range(nc):
                fp.range_check(sib_bal[c], n_bytes)
        bit = fp.witness(0, bit_value, "swap bit")
        left, right = fp.swap(cur_hash, sib_hash, bit)
        cur_bal = [fp.add(cur_bal[c], sib_bal[c]) for c in range(nc)]
        cur_hash = fp.hash(2, cur_bal + [left, right])
    fp.expose(cur_hash, 1)
    for c, cell in enumerate(cur_bal):
        fp.expose(cell, 2 + c)
    return fp.finish()


# ---------------------------------------------------------------------------------------------------------------------
# The same floor plan as a PROGRAM for the device (csrc/witness.hip, C ABI sg_mst_inclusion_witness_dev): where
# reference_assignment above writes values, witness_program records which tree node every cell holds.  In an inclusion
# circuit every witnessed value is a node of the Merkle sum tree or a path bit -- the running hash / balances are the
# path's nodes, the swap outputs are the two children of the path's parent -- so a cell is a "symbol" relative to the
# user's leaf index and the whole assignment of any number of users is one kernel launch.
SYM_USER, SYM_HASH, SYM_BAL, SYM_BIT = 0, 1, 2, 3
MODE_PATH, MODE_SIBLING, MODE_SIBLING_CHILD, MODE_ORDERED_CHILD = 0, 1, 2, 3


def sym(kind: int, level: int = 0, mode: int = 0, lane: int = 0) -> int:
    return kind | level << 4 | mode << 10 | lane << 13


@lru_cache(maxsize=None)
def witness_program(k: int, levels: int, nc: int, n_bytes: int = 8):
    """-> (program uint32 array: n_items x 5 items then n_absorbs x 3 absorbs, n_items, n_absorbs, instance symbols,
    rows used).  Walks `MstInclusionCircuit::synthesize` exactly as reference_assignment does (same regions in the same
    order through the same floor planner, constants column included); checked cell by cell against it."""
    import numpy as np
    fp = _ReferenceFloorPlan(k)               # for its floor planner only
    a0, a1, a2 = (ADVICE, 0), (ADVICE, 1), (ADVICE, 2)
    cells, ranges, hashes, absorbs = [], [], [], []

    def constants(count):
        col = (FIXED, 2)
        fp.next_free[col] = fp.next_free.get(col, 0) + count

    def witness(column, s):
        cells.append((column, fp.region([(ADVICE, column)], 1), s))
        return s

    def hash_(chip, inputs, digest):
        st = fp.region([a0, a1], 1)
        constants(2)
        first = len(absorbs)
        for s in inputs:
            add = fp.region([a0, a1, ("selector", "pad", chip)], 3)
            perm = fp.region([a0, a1, a2] + [(FIXED, j) for j in range(4)] + [("selector", "full", chip), ("selector", "partial", chip)], 37)
            absorbs.append((add, perm, s))
        hashes.append((st, first, len(inputs), chip))
        return digest

    def range_check(s):
        ranges.append((fp.region([a0, ("selector", "lookup")], n_bytes + 1), s))
        constants(1)

    path_hash = lambda level: sym(SYM_HASH, level, MODE_PATH)
    user = witness(0, sym(SYM_USER))
    cur_bal = [witness(1, sym(SYM_BAL, 0, MODE_PATH, c)) for c in range(nc)]
    cur_hash = hash_(1, [user] + cur_bal, path_hash(0))
    fp.region([(FIXED, 4)], 256)
    for level in range(levels):
        if level == 0:
            sib_user = witness(0, sym(SYM_USER, 0, 1))
            sib_bal = [witness(1, sym(SYM_BAL, 0, MODE_SIBLING, c)) for c in range(nc)]
            sib_hash = hash_(1, [sib_user] + sib_bal, sym(SYM_HASH, 0, MODE_SIBLING))
            for c in range(nc):
                range_check(cur_bal[c])
                range_check(sib_bal[c])
        else:
            sib_bal = [witness(1, sym(SYM_BAL, level, MODE_SIBLING, c)) for c in range(nc)]
            left = witness(2, sym(SYM_HASH, level, MODE_SIBLING_CHILD, 0))
            right = witness(2, sym(SYM_HASH, level, MODE_SIBLING_CHILD, 1))
            sib_hash = hash_(2, sib_bal + [left, right], sym(SYM_HASH, level, MODE_SIBLING))
            for c in range(nc):
                range_check(sib_bal[c])
        bit = witness(0, sym(SYM_BIT, level))
        st = fp.region([a0, a1, a2, ("selector", "swap")], 2)
        left, right = sym(SYM_HASH, level, MODE_ORDERED_CHILD, 0), sym(SYM_HASH, level, MODE_ORDERED_CHILD, 1)
        cells.extend([(0, st, cur_hash), (1, st, sib_hash), (2, st, bit), (0, st + 1, left), (1, st + 1, right)])
        nxt = []
        for c in range(nc):
            st = fp.region([a0, a1, a2, ("selector", "sum")], 1)
            total = sym(SYM_BAL, level + 1, MODE_PATH, c)
            cells.extend([(0, st, cur_bal[c]), (1, st, sib_bal[c]), (2, st, total)])
            nxt.append(total)
        cur_bal = nxt
        cur_hash = hash_(2, cur_bal + [left, right], path_hash(level + 1))
    items = [(2, 0, st, 0, first | count << 20 | chip << 28) for st, first, count, chip in hashes]      # sponges first: whole waves
    items += [(1, 0, st, s, n_bytes) for st, s in ranges]
    items += [(0, col, row, s, 0) for col, row, s in cells]
    prog = np.concatenate([np.asarray(items, dtype=np.uint32).reshape(-1), np.asarray(absorbs, dtype=np.uint32).reshape(-1)])
    instances = [path_hash(0), path_hash(levels)] + [sym(SYM_BAL, levels, MODE_PATH, c) for c in range(nc)]
    return prog, len(items), len(absorbs), instances, max(fp.next_free.values())
