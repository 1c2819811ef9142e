"""Checks of the example assignment of the reference circuit's constraint system
(circuits_halo2_amd.mst_inclusion.example_assignment) against the restated verifier's gate polynomials.
TEST INFRASTRUCTURE (uses oracle/)."""
from oracle import pyref as PR
from oracle import summa_verifier as SV

R = PR.R


def build(k: int):
    """the product module's example assignment (circuits_halo2_amd.mst_inclusion.example_assignment)"""
    from circuits_halo2_amd import mst_inclusion as M
    return M.example_assignment(k)


def check_gates(asg, k: int, n_currencies: int = 2):
    """row-wise: every gate polynomial vanishes on the usable rows, the lookup inputs are table values"""
    n, u = 1 << k, asg["usable_rows"]
    table = set(asg["fixed"][4][:u])
    for row in range(u):
        q = lambda kind, c, rot: (asg["fixed"] if kind == "f" else asg["advice"])[c][(row + rot) % n]
        if any(SV.gate_values(q, n_currencies)):
            return False
        if SV.lookup_input_table(q)[0] not in table:
            return False
    return True
