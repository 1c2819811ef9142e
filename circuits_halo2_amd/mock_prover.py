"""`halo2_proofs::dev::MockProver` for `MstInclusionCircuit`: the constraint checker the reference's circuit tests run
[REF zk_prover/src/circuits/tests.rs:25-43, 158-433: `MockProver::run(K, &circuit, instances)`, `assert_satisfied()`,
`verify()` compared with exact `VerifyFailure` lists].  No MSM, no NTT, no device: integers on the host, as upstream.

    prover = MockProver.run(k, circuit, instances)        # circuit: api.MstInclusionCircuit (host-side fields)
    prover.verify()            -> [] or the failures, in upstream's order: gates, lookups, permutation
    prover.assert_satisfied()

What is checked, as upstream's `verify` does it: every gate polynomial on every usable row (gate index and name, the
polynomial's index inside the gate, the region and offset of the failing row, the values of the queried cells); the
lookup's input on every usable row; for every column of the permutation argument, row by row, that the cell equals the
cell its permutation maps it to.  A failure's location is the region -- in `assign_region` order -- whose assigned rows
contain the failing row and whose assigned columns meet the failing columns (`FailureLocation::find`), else "outside
any region" (constants, instance cells).  Failures are plain tuples shaped like upstream's Debug output:

    ("ConstraintNotSatisfied", (gate index, gate name), poly index, ("InRegion", (region index, name), offset), cell_values)
    ("Lookup", lookup index, location)
    ("Permutation", (column kind, column index), location)            location: ("InRegion", ..) | ("OutsideRegion", row)

The circuit enters through mst_inclusion.reference_assignment in its lenient form: the witness is laid out as it is given,
violations and all (synthesize does not check values), and the instance column holds the caller's `instances`.
"""
from __future__ import annotations

from . import mst_inclusion as M
from .arithmetic import ADVICE, FIXED, INSTANCE

R = M.R
KIND_NAMES = {ADVICE: "advice", FIXED: "fixed", INSTANCE: "instance"}


def format_value(v: int) -> str:
    """halo2's dev::util::format_value: 0, 1, -1 plainly, other values as hex without leading zeros"""
    v %= R
    if v == 0:
        return "0"
    if v == 1:
        return "1"
    if v == R - 1:
        return "-1"
    return "0x" + format(v, "x")


def gate_layout(n_currencies: int):
    """(gate index, gate name, polynomial index inside the gate) for every polynomial of mst_inclusion.gates(), in the
    constraint system's order [REF circuits/merkle_sum_tree.rs:141-207: the two Poseidon chips (halo2_gadgets Pow5:
    "full round", "partial rounds", "pad-and-add"), then chips/merkle_sum_tree.rs:50,56,78]"""
    out = []
    g = 0
    for _chip in range(2):
        for name, polys in (("full round", 2), ("partial rounds", 3), ("pad-and-add", 2)):
            out += [(g, name, j) for j in range(polys)]
            g += 1
    out.append((g, "bool constraint", 0))
    out += [(g + 1, "swap constraint", 0), (g + 1, "swap constraint", 1)]
    out += [(g + 2, "sum constraint", j) for j in range(n_currencies)]
    return out


def _queries(expr, acc):
    if expr.op == "query":
        acc.add(tuple(expr.args))
    elif expr.op != "const":
        for a in expr.args:
            _queries(a, acc)
    return acc


# fixed columns that hold selectors after halo2's selector compression: Expression::Selector in the gates, not cell queries
_SELECTOR_COLUMNS = {5, 6, 7, 8, 9, 10}


class MockProver:
    def __init__(self, k: int, assignment: dict, n_currencies: int):
        self.k, self.n = k, 1 << k
        self.asg = assignment
        self.n_currencies = n_currencies
        self.usable = assignment["usable_rows"]
        inst = list(assignment["instances"]) + [0] * self.n
        self.columns = {ADVICE: assignment["advice"], FIXED: assignment["fixed"], INSTANCE: [inst[:self.n]]}

    @classmethod
    def run(cls, k: int, circuit, instances) -> "MockProver":
        """circuit: api.MstInclusionCircuit with its host-side fields; instances: [[values of the instance column]]"""
        if len(instances) != 1:
            raise ValueError("one instance column expected")      # upstream: Error::InvalidInstances
        if len(instances[0]) > (1 << k) - (M.BLINDING_FACTORS + 1):
            raise ValueError("InstanceTooLarge")
        asg = M.reference_assignment(k, circuit.entry[0], list(circuit.entry[1]), circuit.path_indices,
                                     circuit.sibling_leaf_node_hash_preimage,
                                     circuit.sibling_middle_node_hash_preimages[:max(0, circuit.levels - 1)], circuit.n_bytes,
                                     lenient=True, instances=[int(v) % R for v in instances[0]])
        return cls(k, asg, circuit.n_currencies)

    # ---- FailureLocation::find
    def _locate(self, row: int, columns):
        for index, (name, lo, hi, cols) in enumerate(self.asg["regions"]):
            if lo is not None and lo <= row <= hi and not cols.isdisjoint(columns):
                return ("InRegion", (index, name), row - lo)
        return ("OutsideRegion", row)

    def _cell(self, kind: int, column: int, row: int) -> int:
        return self.columns[kind][column][row % self.n]

    def verify(self):
        failures = []
        layout = gate_layout(self.n_currencies)
        polys = M.gates(self.n_currencies)
        assert len(polys) == len(layout)
        # gates: gate by gate, row by row, polynomial by polynomial (upstream's iteration order)
        by_gate = {}
        for (g, name, j), poly in zip(layout, polys):
            by_gate.setdefault((g, name), []).append((j, poly, sorted(_queries(poly, set()))))
        # a row can only fail where one of its selectors is on; skip the others (all their polynomials vanish)
        live_rows = [r for r in range(self.usable) if any(self.asg["fixed"][c][r] for c in (6, 7, 8, 9, 10))]
        for (g, name), plist in sorted(by_gate.items()):
            for row in live_rows:
                q = lambda kind, column, rot, row=row: self._cell(kind, column, row + rot)
                for j, poly, queries in plist:
                    if poly.evaluate(q) == 0:
                        continue
                    cells = [(kind, column, rot) for kind, column, rot in queries
                             if not (kind == FIXED and column in _SELECTOR_COLUMNS)]
                    cells.sort(key=lambda c: ({ADVICE: 0, FIXED: 1, INSTANCE: 2}[c[0]], c[1], c[2]))
                    where = self._locate(row, {(kind, column) for kind, column, _ in cells})
                    values = [((KIND_NAMES[kind], column), rot, format_value(self._cell(kind, column, row + rot)))
                              for kind, column, rot in cells]
                    failures.append(("ConstraintNotSatisfied", (g, name), j, where, values))
        # the lookup: input expression in the table column, on the usable rows
        inp_e, tab_e = M.lookup_expressions()
        table = {tab_e.evaluate(lambda kind, column, rot, row=row: self._cell(kind, column, row + rot)) for row in range(self.usable)}
        for row in range(self.usable):
            if not self.asg["fixed"][5][row]:
                continue
            v = inp_e.evaluate(lambda kind, column, rot, row=row: self._cell(kind, column, row + rot))
            if v not in table:
                failures.append(("Lookup", 0, self._locate(row, {(ADVICE, 0)})))
        # the permutation: column by column (the argument's column order), row by row
        mapping = self.asg["mapping"]
        for ci, (kind, column) in enumerate(M.PERMUTATION_COLUMNS):
            col_map = mapping[ci]
            for row in range(self.n):
                tc, tr = col_map[row]
                if (tc, tr) == (ci, row):
                    continue
                tk, tcol = M.PERMUTATION_COLUMNS[tc]
                if self._cell(kind, column, row) != self._cell(tk, tcol, tr):
                    failures.append(("Permutation", (KIND_NAMES[kind], column), self._locate(row, {(kind, column)})))
        return failures

    def assert_satisfied(self):
        failures = self.verify()
        if failures:
            raise AssertionError("circuit was not satisfied: %r" % (failures[:8],))
