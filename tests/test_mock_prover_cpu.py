"""The reference's MockProver circuit tests [REF zk_prover/src/circuits/tests.rs:25-43, 158-433], restated on the
product's constraint system and its replay of the reference's floor plan (circuits_halo2_amd/mst_inclusion.py,
mock_prover.py).  The expected failure lists are the REFERENCE'S OWN: region indices and names, offsets, gate indices
and names, polynomial indices, failing columns, cell values -- values held by the reference's test file.  They pin, on
the CPU and without an SRS: the region order of `synthesize`, the shape of every region (a Poseidon permutation ends at
offset 36, a range check at offset 8), the floor planner's row allocation (the constant behind the first range check
sits in fixed column 2 at row 246), the permutation assembly's cycle structure (WHICH cells of a broken cycle differ
from their successor), the gate order of `configure`, and the leaf hashes.

The witness side comes from the oracle's Merkle sum tree (big integers; pinned on K5 in test_witness_cpu.py): the
product's tree runs on the GPU and is compared with it there (tests/test_gpu_api.py)."""
import csv
import os

import pytest

from conftest import GOLDEN
from oracle import pyref as P

from circuits_halo2_amd import api
from circuits_halo2_amd.mock_prover import MockProver, format_value, gate_layout

N_CURRENCIES, LEVELS, N_BYTES, K = 2, 4, 8, 11
LEAF0 = "0x167505f45c4ef4a0b051c30e881d2e8f881f26f5edb231396198a2cc1712f5ad"      # circuits/tests.rs:341
LEAF1 = "0x2c688f624d2bca741a1c2ad1ad2880721fbfd1613bbc5fe3d2ba66eb672e3aab"      # circuits/tests.rs:346


def _tree(name):
    rows = list(csv.reader(open(os.path.join(GOLDEN, name))))[1:]
    entries = [P.mst_entry(r[0], [int(v) for v in r[1:]]) for r in rows]
    return entries, P.mst_build(entries)


def _circuit(name, user_index):
    """`MstInclusionCircuit::init(merkle_sum_tree.generate_proof(user_index))` from the oracle's tree"""
    entries, (root, levels) = _tree(name)
    depth = len(levels) - 1
    assert depth == LEVELS
    sib0 = user_index ^ 1
    sib_entry = entries[sib0] if sib0 < len(entries) else (0, [0] * N_CURRENCIES)
    middle = []
    for lvl in range(1, depth):
        sib = (user_index >> lvl) ^ 1
        node = levels[lvl][sib]
        middle.append(list(node[1]) + [levels[lvl - 1][2 * sib][0], levels[lvl - 1][2 * sib + 1][0]])
    return api.MstInclusionCircuit(LEVELS, N_CURRENCIES, N_BYTES, (entries[user_index][0], list(entries[user_index][1])),
                                   [(user_index >> l) & 1 for l in range(depth)], [sib_entry[0]] + list(sib_entry[1]), middle,
                                   (root[0], list(root[1])))


def _instances(circuit):
    """`circuit.instances()` with the leaf hash from the oracle (the product's runs Poseidon on the device)"""
    return [[P.mst_leaf(circuit.entry[0], circuit.entry[1]), circuit.root[0]] + list(circuit.root[1])]


def perm(kind, column, location):
    return ("Permutation", (kind, column), location)


def in_region(index, name, offset):
    return ("InRegion", (index, name), offset)


def test_valid_merkle_sum_tree():
    """circuits/tests.rs:25-43: the circuit of every user of entry_16.csv is satisfied"""
    for user_index in range(16):
        circuit = _circuit("entry_16.csv", user_index)
        inst = _instances(circuit)
        assert len(inst[0]) == circuit.num_instances() == 2 + N_CURRENCIES
        prover = MockProver.run(K, circuit, inst)
        assert prover.verify() == [], user_index
        prover.assert_satisfied()
    assert format_value(inst[0][0]) != LEAF0 and format_value(_instances(_circuit("entry_16.csv", 0))[0][0]) == LEAF0


def test_invalid_entry_balance_as_witness():
    """circuits/tests.rs:158-229"""
    circuit = _circuit("entry_16.csv", 0)
    instances = _instances(circuit)
    circuit.entry = (circuit.entry[0], [1000, 1000])
    assert MockProver.run(K, circuit, instances).verify() == [
        perm("advice", 0, in_region(26, "assign nodes hashes per merkle tree level", 0)),
        perm("advice", 0, in_region(121, "permute state", 36)),
        perm("advice", 2, in_region(111, "sum nodes balances per currency", 0)),
        perm("advice", 2, in_region(112, "sum nodes balances per currency", 0)),
        perm("instance", 0, ("OutsideRegion", 0)),
        perm("instance", 0, ("OutsideRegion", 1)),
        perm("instance", 0, ("OutsideRegion", 2)),
        perm("instance", 0, ("OutsideRegion", 3)),
    ]


def test_invalid_leaf_hash_as_instance():
    """circuits/tests.rs:232-266"""
    circuit = _circuit("entry_16.csv", 0)
    instances = _instances(circuit)
    instances[0][0] = 1000
    assert MockProver.run(K, circuit, instances).verify() == [
        perm("advice", 0, in_region(26, "assign nodes hashes per merkle tree level", 0)),
        perm("instance", 0, ("OutsideRegion", 0)),
    ]


def test_balance_not_in_range():
    """circuits/tests.rs:268-299: entry_16_overflow.csv, whose first balance does not fit N_BYTES = 8 bytes"""
    circuit = _circuit("entry_16_overflow.csv", 0)
    assert circuit.entry[1][0] >= 1 << 64
    assert MockProver.run(K, circuit, _instances(circuit)).verify() == [
        perm("fixed", 2, ("OutsideRegion", 246)),
        perm("advice", 0, in_region(21, "assign value to perform range check", 8)),
    ]


def test_non_binary_index():
    """circuits/tests.rs:302-395"""
    circuit = _circuit("entry_16.csv", 0)
    instances = _instances(circuit)
    circuit.path_indices[0] = 2
    region = in_region(26, "assign nodes hashes per merkle tree level", 0)
    assert MockProver.run(K, circuit, instances).verify() == [
        ("ConstraintNotSatisfied", (6, "bool constraint"), 0, region, [(("advice", 2), 0, "0x2")]),
        ("ConstraintNotSatisfied", (7, "swap constraint"), 0, region,
         [(("advice", 0), 0, LEAF0), (("advice", 0), 1, LEAF1), (("advice", 1), 0, LEAF1), (("advice", 2), 0, "0x2")]),
        ("ConstraintNotSatisfied", (7, "swap constraint"), 1, region,
         [(("advice", 0), 0, LEAF0), (("advice", 1), 0, LEAF1), (("advice", 1), 1, LEAF0), (("advice", 2), 0, "0x2")]),
        perm("advice", 0, in_region(121, "permute state", 36)),
        perm("instance", 0, ("OutsideRegion", 1)),
    ]


def test_swapping_index():
    """circuits/tests.rs:398-433"""
    circuit = _circuit("entry_16.csv", 0)
    instances = _instances(circuit)
    circuit.path_indices[0] = 1
    assert MockProver.run(K, circuit, instances).verify() == [
        perm("advice", 0, in_region(121, "permute state", 36)),
        perm("instance", 0, ("OutsideRegion", 1)),
    ]


def test_gate_names_follow_configure():
    """circuits/merkle_sum_tree.rs:141-207: two Pow5 chips, then bool / swap / sum"""
    lay = gate_layout(2)
    assert len(lay) == 19 and lay[14] == (6, "bool constraint", 0) and lay[15:17] == [(7, "swap constraint", 0), (7, "swap constraint", 1)]
    assert lay[17:] == [(8, "sum constraint", 0), (8, "sum constraint", 1)] and lay[0][:2] == (0, "full round")
    assert [format_value(v) for v in (0, 1, P.R - 1, 2, 255)] == ["0", "1", "-1", "0x2", "0xff"]


def test_a_lookup_violation_is_reported():
    """not one of the reference's cases (its overflow shows up as a broken copy): a running sum whose step is not a byte"""
    circuit = _circuit("entry_16.csv", 3)
    prover = MockProver.run(K, circuit, _instances(circuit))
    name, lo, hi, cols = prover.asg["regions"][21]
    assert name == "assign value to perform range check"
    prover.asg["advice"][0][lo + 1] += 1          # z_1 + 1: z_0 - 256 (z_1 + 1) is negative, hence outside the byte table
    failures = prover.verify()
    assert failures[0] == ("Lookup", 0, ("InRegion", (21, name), 0)) and [f[0] for f in failures].count("Lookup") >= 1
    with pytest.raises(AssertionError):
        prover.assert_satisfied()
