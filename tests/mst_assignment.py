"""A satisfying assignment of the reference circuit's constraint system (circuits_halo2_amd.mst_inclusion) with
this repository's own floor plan: one full Poseidon permutation on chip 1 (4 + 56 + 4 rounds in 36 rows), single
rounds on chip 2, both pad-and-add gates, two swaps, a sum, a two-byte range decomposition, and copy constraints
across advice, fixed and instance columns.  Python integers; TEST INFRASTRUCTURE (uses oracle/)."""
from oracle import pyref as PR
from oracle import summa_verifier as SV

R = PR.R


def build(k: int):
    """-> dict(fixed=[11][n], advice=[3][n], instances=[4], sigma=[6][n], usable_rows, poseidon_in, poseidon_out)"""
    n = 1 << k
    u = n - 6
    assert n >= 512
    rcs, mds, _ = SV.poseidon_generate()
    fixed = [[0] * n for _ in range(11)]
    adv = [[0] * n for _ in range(3)]
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
    adv[0][55], adv[1][55] = 1000, 234
    adv[2][55] = 1234
    fixed[6][55] = 2
    adv[0][37], adv[0][38], adv[0][39] = 1234, 66, 1300
    adv[1][37] = adv[1][39] = 77
    fixed[6][38] = 3
    # chip 2: one full round (41 -> 42), one pair of partial rounds (44 -> 45), pad-and-add (46 .. 48)
    adv[0][42], adv[1][42] = full_round(41, [5, 6], rcs[10], 9)
    adv[0][45], adv[1][45] = partial_pair(44, [7, 8], rcs[20], rcs[21], 10)
    adv[0][46], adv[0][47], adv[0][48] = 40, 9, 49
    adv[1][46] = adv[1][48] = 0
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
    inst_col = instances + [0] * (n - 4)
    # copy constraints over the permutation columns (f2, a0, a1, f3, a2, i0)
    col_index = {("f", 2): 0, ("a", 0): 1, ("a", 1): 2, ("f", 3): 3, ("a", 2): 4, ("i", 0): 5}
    groups = [[(("a", 0), 36), (("i", 0), 0)], [(("a", 1), 51), (("i", 0), 1)], [(("a", 2), 55), (("a", 0), 37)],
              [(("f", 2), 70), (("a", 1), 70)], [(("a", 0), 53), (("a", 0), 52)]]
    values = {("a", j): adv[j] for j in range(3)}
    values.update({("f", 2): fixed[2], ("f", 3): fixed[3], ("i", 0): inst_col})
    omega = PR.omega_for(k)
    label = lambda c, i: pow(PR.DELTA, c, R) * pow(omega, i, R) % R
    sigma = [[label(c, i) for i in range(n)] for c in range(6)]
    for grp in groups:
        assert len({values[col][i] for col, i in grp}) == 1, grp
        cells = [(col_index[col], i) for col, i in grp]
        for (c, i), (c2, i2) in zip(cells, cells[1:] + cells[:1]):
            sigma[c][i] = label(c2, i2)
    return {"fixed": fixed, "advice": adv, "instances": instances, "sigma": sigma, "usable_rows": u,
            "poseidon_in": poseidon_in, "poseidon_out": poseidon_out}


def check_gates(asg, k: int):
    """row-wise: every gate polynomial vanishes on the usable rows, the lookup inputs are table values"""
    n, u = 1 << k, asg["usable_rows"]
    table = set(asg["fixed"][4][:u])
    for row in range(u):
        q = lambda kind, c, rot: (asg["fixed"] if kind == "f" else asg["advice"])[c][(row + rot) % n]
        if any(SV.gate_values(q)):
            return False
        if SV.lookup_input_table(q)[0] not in table:
            return False
    return True
