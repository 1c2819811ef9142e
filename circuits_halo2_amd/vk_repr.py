"""halo2's `VerifyingKey::transcript_repr` for `MstInclusionCircuit`: the Blake2b digest of the pinned key's Debug text.

Upstream hashes `format!("{:?}", vk.pinned())` [halo2_proofs (summa-dev/halo2 @ 8386d6e, the reference's Cargo.lock pin)
plonk.rs `VerifyingKey::from_parts`: Blake2b-512, personal "Halo2-Verify-Key", over len(s) as u64 LE || s; the digest
is reduced wide into Fr].  That text spells out the whole constraint system -- column counts, every gate polynomial as
an expression TREE (not just its value), the query lists in the order `configure` made them, the permutation and lookup
arguments -- followed by the fixed and permutation commitments.  So reproducing the digest means replaying
`MstInclusionConfig::configure` [REF zk_prover/src/circuits/merkle_sum_tree.rs:141-207] against a constraint system
that records queries and builds trees the way halo2's operators do (`a - b` = Sum(a, Negated(b)), `e * scalar` =
Scaled, selector compression substitutes q * prod (i - q)).

The reference's own golden value pins this: the `vk_digest` constant of its generated verifier
[REF contracts/src/InclusionVerifier.sol, committed as tests/golden/kat.json "vk_digest"] is this digest for
k = 17, <LEVELS, 2 currencies, 8 bytes> (LEVELS does not enter the constraint system).
"""
from __future__ import annotations

import hashlib

from .poseidon_params import generate as _generate_poseidon

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
ROOT_OF_UNITY_2_28 = 0x03ddb9f5166d18b798865ea93dd31f743215cf6dd39329c8d34f1ed960c37c9c


def _hex(v: int) -> str:
    return "0x%064x" % v


# --- expression trees, with halo2's Debug text -----------------------------------------------------------------------
class E:
    __slots__ = ("kind", "a", "b")

    def __init__(self, kind, a=None, b=None):
        self.kind, self.a, self.b = kind, a, b

    def __neg__(self):
        return E("Negated", self)

    def __add__(self, o):
        return E("Sum", self, o)

    def __sub__(self, o):
        return E("Sum", self, -o)

    def __mul__(self, o):
        if isinstance(o, int):
            return E("Scaled", self, o % R)
        return E("Product", self, o)

    def text(self, out: list):
        k = self.kind
        if k == "Constant":
            out.append("Constant(%s)" % _hex(self.a))
        elif k in ("Fixed", "Advice", "Instance"):
            qi, col, rot = self.a
            out.append("%s { query_index: %d, column_index: %d, rotation: Rotation(%d) }" % (k, qi, col, rot))
        elif k == "Negated":
            out.append("Negated(")
            self.a.text(out)
            out.append(")")
        elif k == "Scaled":
            out.append("Scaled(")
            self.a.text(out)
            out.append(", %s)" % _hex(self.b))
        elif k == "Selector":
            raise ValueError("selector left in a pinned expression")
        else:
            out.append(k + "(")
            self.a.text(out)
            out.append(", ")
            self.b.text(out)
            out.append(")")

    def substitute(self, selectors):
        k = self.kind
        if k == "Selector":
            return selectors[self.a]
        if k in ("Constant", "Fixed", "Advice", "Instance"):
            return self
        if k == "Negated":
            return E(k, self.a.substitute(selectors))
        if k == "Scaled":
            return E(k, self.a.substitute(selectors), self.b)
        return E(k, self.a.substitute(selectors), self.b.substitute(selectors))

    def degree(self) -> int:
        k = self.kind
        if k == "Constant":
            return 0
        if k in ("Fixed", "Advice", "Instance", "Selector"):
            return 1
        if k in ("Negated", "Scaled"):
            return self.a.degree()
        d = (self.a.degree(), self.b.degree())
        return sum(d) if k == "Product" else max(d)

    def selectors(self, acc: set):
        if self.kind == "Selector":
            acc.add(self.a)
        elif self.kind in ("Negated", "Scaled"):
            self.a.selectors(acc)
        elif self.kind in ("Sum", "Product"):
            self.a.selectors(acc)
            self.b.selectors(acc)
        return acc


def const(v: int) -> E:
    return E("Constant", v % R)


class ConstraintSystem:
    """the part of halo2's `ConstraintSystem` that the pinned text shows"""

    def __init__(self):
        self.num_fixed = self.num_advice = self.num_instance = 0
        self.simple = []                 # per selector: True = simple
        self.gates = []                  # polynomials, flattened
        self.queries = {"Advice": [], "Fixed": [], "Instance": []}
        self.permutation = []            # (kind, index)
        self.lookups = []                # (inputs, tables)
        self.constants = []

    def advice_column(self):
        self.num_advice += 1
        return ("Advice", self.num_advice - 1)

    def fixed_column(self):
        self.num_fixed += 1
        return ("Fixed", self.num_fixed - 1)

    def instance_column(self):
        self.num_instance += 1
        return ("Instance", self.num_instance - 1)

    def selector(self, simple=True):
        self.simple.append(simple)
        return E("Selector", len(self.simple) - 1)

    def query(self, column, rotation=0) -> E:
        kind, index = column
        q = self.queries[kind]
        if (index, rotation) not in q:
            q.append((index, rotation))
        return E(kind, (q.index((index, rotation)), index, rotation))

    def enable_equality(self, column):
        self.query(column, 0)
        if column not in self.permutation:
            self.permutation.append(column)

    def enable_constant(self, column):
        if column not in self.constants:
            self.constants.append(column)
            self.enable_equality(column)

    def create_gate(self, polys):
        self.gates.extend(polys)

    def degree(self) -> int:
        d = 3 if self.permutation else 1           # permutation argument: required degree 3
        for inputs, tables in self.lookups:        # lookup argument: 2 + max(1, input degree) + max(1, table degree)
            di = max([1] + [e.degree() for e in inputs])
            dt = max([1] + [e.degree() for e in tables])
            d = max(d, 2 + di + dt)
        return max([d] + [g.degree() for g in self.gates])

    def compress_selectors(self, activations):
        """`activations[s]` = set of rows where selector s is enabled; returns the new fixed columns' row values as
        {column index: {row: value}} and substitutes the selectors in gates and lookups"""
        n_sel = len(self.simple)
        degrees = [0] * n_sel
        for g in self.gates:
            sel = g.selectors(set())
            if sel:
                d = g.degree()
                for s in sel:
                    degrees[s] = max(degrees[s], d)
        max_degree = self.degree()
        replacement = [None] * n_sel
        columns = {}

        def allocate():
            col = self.fixed_column()
            return col, self.query(col, 0)

        for s in range(n_sel):               # complex (and unused) selectors: a column each
            if not self.simple[s] or degrees[s] == 0:
                col, q = allocate()
                replacement[s] = q
                columns[col[1]] = {r: 1 for r in activations[s]}
        todo = [s for s in range(n_sel) if replacement[s] is None]
        added = set()
        for i, s in enumerate(todo):
            if s in added:
                continue
            combo, d = [s], degrees[s] - 1      # degree of the gate without its selector
            added.add(s)
            for t in todo[i + 1:]:
                if d + len(combo) == max_degree:
                    break
                if t in added or any(activations[t] & activations[c] for c in combo):
                    continue
                nd = max(d, degrees[t] - 1)
                if nd + len(combo) + 1 > max_degree:
                    continue
                d = nd
                combo.append(t)
                added.add(t)
            col, q = allocate()
            values = {}
            for root0, c in enumerate(combo):
                expr = q
                for root in range(1, len(combo) + 1):
                    if root != root0 + 1:
                        expr = expr * (const(root) - q)
                replacement[c] = expr
                for r in activations[c]:
                    values[r] = root0 + 1
            columns[col[1]] = values
        self.gates = [g.substitute(replacement) for g in self.gates]
        self.lookups = [([e.substitute(replacement) for e in i], [e.substitute(replacement) for e in t])
                        for i, t in self.lookups]
        return columns

    # -- Debug text ---------------------------------------------------------------------------------------------------
    def pinned_text(self, shuffles: bool = False) -> str:
        out = ["PinnedConstraintSystem { num_fixed_columns: %d, num_advice_columns: %d, num_instance_columns: %d, "
               "num_selectors: %d, gates: [" % (self.num_fixed, self.num_advice, self.num_instance, len(self.simple))]
        for j, g in enumerate(self.gates):
            if j:
                out.append(", ")
            g.text(out)
        out.append("]")
        col = lambda kind, i: "Column { index: %d, column_type: %s }" % (i, kind)
        for name, kind in (("advice_queries", "Advice"), ("instance_queries", "Instance"), ("fixed_queries", "Fixed")):
            out.append(", %s: [%s]" % (name, ", ".join("(%s, Rotation(%d))" % (col(kind, i), r)
                                                        for i, r in self.queries[kind])))
        out.append(", permutation: Argument { columns: [%s] }" % ", ".join(col(k, i) for k, i in self.permutation))
        out.append(", lookups: [")
        for j, (inputs, tables) in enumerate(self.lookups):
            if j:
                out.append(", ")
            out.append("Argument { input_expressions: [")
            for n, e in enumerate(inputs):
                if n:
                    out.append(", ")
                e.text(out)
            out.append("], table_expressions: [")
            for n, e in enumerate(tables):
                if n:
                    out.append(", ")
                e.text(out)
            out.append("] }")
        out.append("]")
        if shuffles:
            out.append(", shuffles: []")
        out.append(", constants: [%s], minimum_degree: None }" % ", ".join(col(k, i) for k, i in self.constants))
        return "".join(out)


# --- the circuit's configure ---------------------------------------------------------------------------------------
def _pow5_chip(cs: ConstraintSystem, state, partial_sbox, rc_a, rc_b):
    """halo2_gadgets' `Pow5Chip::configure` for WIDTH 2, RATE 1 (gate and query order)"""
    _, m_reg, m_inv = _generate_poseidon()
    W = 2
    for c in list(state) + list(rc_b):
        cs.enable_equality(c)
    s_full, s_partial, s_pad = cs.selector(), cs.selector(), cs.selector()

    def pow5(v):
        v2 = v * v
        return v2 * v2 * v

    polys = []
    for nxt in range(W):
        state_next = cs.query(state[nxt], 1)
        expr = None
        for idx in range(W):
            term = pow5(cs.query(state[idx], 0) + cs.query(rc_a[idx], 0)) * int(m_reg[nxt][idx])
            expr = term if expr is None else expr + term
        polys.append(s_full * (expr - state_next))
    cs.create_gate(polys)

    cur_0, mid_0 = cs.query(state[0], 0), cs.query(partial_sbox, 0)
    rc_a0, rc_b0 = cs.query(rc_a[0], 0), cs.query(rc_b[0], 0)

    def mid(idx):
        acc = mid_0 * int(m_reg[idx][0])
        for c in range(1, W):
            acc = acc + (cs.query(state[c], 0) + cs.query(rc_a[c], 0)) * int(m_reg[idx][c])
        return acc

    def nxt_(idx):
        acc = None
        for n in range(W):
            t = cs.query(state[n], 1) * int(m_inv[idx][n])
            acc = t if acc is None else acc + t
        return acc

    def linear(idx):
        rcb = cs.query(rc_b[idx], 0)
        return mid(idx) + rcb - nxt_(idx)

    polys = [pow5(cur_0 + rc_a0) - mid_0, pow5(mid(0) + rc_b0) - nxt_(0)] + [linear(i) for i in range(1, W)]
    cs.create_gate([s_partial * p for p in polys])

    RATE = 1
    initial_rate = cs.query(state[RATE], -1)
    output_rate = cs.query(state[RATE], 1)
    polys = []
    for idx in range(RATE):
        polys.append(cs.query(state[idx], -1) + cs.query(state[idx], 0) - cs.query(state[idx], 1))
    polys.append(initial_rate - output_rate)
    cs.create_gate([s_pad * p for p in polys])
    return s_full, s_partial, s_pad


def configure(n_currencies: int = 2, n_bytes: int = 8) -> ConstraintSystem:
    """`MstInclusionConfig::configure` [REF circuits/merkle_sum_tree.rs:141-207], call by call"""
    cs = ConstraintSystem()
    adv = [cs.advice_column() for _ in range(3)]
    fix = [cs.fixed_column() for _ in range(4)]
    range_u8 = cs.fixed_column()
    sel = [cs.selector(), cs.selector()]
    lookup_enable = cs.selector(simple=False)
    cs.enable_constant(fix[2])
    chips = []
    for _ in range(2):       # entry hasher, middle hasher: the same columns
        chips.append(_pow5_chip(cs, adv[0:2], adv[2], fix[0:2], fix[2:4]))
    for c in adv:
        cs.enable_equality(c)
    # MerkleSumTreeChip::configure [REF chips/merkle_sum_tree.rs:39-95]
    a, b, c = adv
    s = sel[0]
    swap = cs.query(c, 0)
    cs.create_gate([s * swap * (const(1) - swap)])
    swap = cs.query(c, 0)
    l_cur, r_cur, l_next, r_next = cs.query(a, 0), cs.query(b, 0), cs.query(a, 1), cs.query(b, 1)
    cs.create_gate([s * ((r_cur - l_cur) * swap + l_cur - l_next), s * ((l_cur - r_cur) * swap + r_cur - r_next)])
    polys = []
    for _ in range(n_currencies):
        polys.append(sel[1] * (cs.query(a, 0) + cs.query(b, 0) - cs.query(c, 0)))
    cs.create_gate(polys)
    # RangeCheckChip::configure [REF chips/range/range_check.rs:28-56] on advice[0]
    z_cur, z_next = cs.query(adv[0], 0), cs.query(adv[0], 1)
    table = cs.query(range_u8, 0)
    cs.lookups.append(([lookup_enable * (z_cur - z_next * const(1 << 8))], [table]))
    inst = cs.instance_column()
    cs.enable_equality(inst)
    cs.selector_names = {"swap": 0, "sum": 1, "lookup": 2, "chips": chips}
    return cs


def _point(p) -> str:
    x, y = p
    return "Infinity" if (x, y) == (0, 0) else "(%s, %s)" % (_hex(x), _hex(y))


def pinned_text(k: int, cs: ConstraintSystem, fixed_comms, permutation_comms, shuffles: bool = False) -> str:
    extended_k = k + (cs.degree() - 1 - 1).bit_length()
    omega = ROOT_OF_UNITY_2_28
    for _ in range(28 - k):
        omega = omega * omega % R
    return ("PinnedVerificationKey { base_modulus: \"%s\", scalar_modulus: \"%s\", "
            "domain: PinnedEvaluationDomain { k: %d, extended_k: %d, omega: %s }, cs: %s, "
            "fixed_commitments: [%s], permutation: VerifyingKey { commitments: [%s] } }"
            % (_hex(Q), _hex(R), k, extended_k, _hex(omega), cs.pinned_text(shuffles),
               ", ".join(_point(p) for p in fixed_comms), ", ".join(_point(p) for p in permutation_comms)))


def transcript_repr_of(text: str) -> int:
    h = hashlib.blake2b(digest_size=64, person=b"Halo2-Verify-Key")
    s = text.encode()
    h.update(len(s).to_bytes(8, "little"))
    h.update(s)
    return int.from_bytes(h.digest(), "little") % R


_CS_TEXT = {}


def constraint_system_text(n_currencies: int = 2) -> tuple:
    """(text of the pinned constraint system, cs.degree()) after selector compression; cached per currency count"""
    if n_currencies not in _CS_TEXT:
        cs = configure(n_currencies)
        # every selector of this circuit is enabled on rows of its own (the regions do not share rows with another
        # selector of the same kind), so the exclusion matrix is empty: one marker row each stands for that
        cs.compress_selectors([{i} for i in range(len(cs.simple))])
        _CS_TEXT[n_currencies] = (cs, cs.pinned_text())
    return _CS_TEXT[n_currencies]


def transcript_repr(k: int, n_currencies: int, fixed_comms, permutation_comms) -> int:
    cs, _ = constraint_system_text(n_currencies)
    return transcript_repr_of(pinned_text(k, cs, fixed_comms, permutation_comms))
