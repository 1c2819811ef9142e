"""The reference's Merkle-sum-tree tests [REF zk_prover/src/merkle_sum_tree/tests.rs:12-263], one test per reference test
name, on the product's tree (every Poseidon hash on the device, circuits_halo2_amd/merkle_sum_tree.py) with the
reference's own CSV fixtures (byte copies under tests/golden/) and its own expected values: depths, root balances
(556862 / 556862, 385969 / 459661, 556863 / 556863), zero-entry padding, the root that `update_leaf` must restore.
Every tree is also compared node by node with the oracle's big-integer tree."""
import os
import random

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

N_CURRENCIES, N_BYTES = 2, 8
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
RINV = pow(1 << 256, -1, R)


def _gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from circuits_halo2_amd import ffi
    ffi.check(ffi.lib().sg_init(0))


def _csv(name):
    return os.path.join(GOLDEN, name)


def _ints(buf):
    raw = bytes(buf)
    return [int.from_bytes(raw[i:i + 32], "little") * RINV % R for i in range(0, len(raw), 32)]


def _tree(name, sorted_=False):
    _gpu()
    from circuits_halo2_amd.merkle_sum_tree import MerkleSumTree
    return (MerkleSumTree.from_csv_sorted if sorted_ else MerkleSumTree.from_csv)(_csv(name), N_CURRENCIES, N_BYTES)


def _same_as_oracle(tree):
    """every node of the device-hashed tree equals the oracle's (big integers)"""
    from oracle import pyref as P
    ents = [P.mst_entry(n, b) if n is not None else (0, [0] * N_CURRENCIES) for n, b in tree.entries]
    root, levels = P.mst_build(ents)
    for lvl, nodes in enumerate(levels):
        for i, (h, bal) in enumerate(nodes):
            nh, nb = tree.node(lvl, i)
            assert _ints(nh) == [h] and _ints(nb) == [b % R for b in bal], (lvl, i)
    return root


def test_mst():
    """:12-67"""
    from circuits_halo2_amd.merkle_sum_tree import MerkleSumTree
    merkle_tree = _tree("entry_16.csv")
    root_hash, root_bal = merkle_tree.root()
    assert _ints(root_hash)[0] != 0
    assert _ints(root_bal) == [556862, 556862]
    assert merkle_tree.depth == 4
    proof = merkle_tree.generate_proof(0)
    assert merkle_tree.verify_proof(proof)
    # different root hashes when the entry order changes, the same totals
    merkle_tree_2 = _tree("entry_16_switched_order.csv")
    assert bytes(root_hash) != bytes(merkle_tree_2.root()[0])
    assert bytes(root_bal) == bytes(merkle_tree_2.root()[1])
    for i in range(16):
        assert merkle_tree.verify_proof(merkle_tree.generate_proof(i))
    with pytest.raises(IndexError):
        merkle_tree.generate_proof(16)
    # a proof with a wrong leaf
    proof_invalid_1 = dict(proof)
    proof_invalid_1["entry"] = ("AtwIxZHo", [35479, 35479])
    assert not merkle_tree.verify_proof(proof_invalid_1)
    # a proof with a wrong root hash
    proof_invalid_2 = dict(proof)
    proof_invalid_2["root"] = (np.zeros(32, dtype=np.uint8), proof["root"][1])
    assert not merkle_tree.verify_proof(proof_invalid_2)
    assert _same_as_oracle(merkle_tree)[0] == _ints(root_hash)[0] and _same_as_oracle(merkle_tree_2)


def test_update_mst_leaf():
    """:69-95: entry_16_modified.csv differs from entry_16.csv in its 7th entry; update_leaf restores the root"""
    merkle_tree_1 = _tree("entry_16.csv")
    root_hash_1 = merkle_tree_1.root()[0]
    merkle_tree_2 = _tree("entry_16_modified.csv")
    assert bytes(root_hash_1) != bytes(merkle_tree_2.root()[0])
    new_root = merkle_tree_2.update_leaf("RkLzkDun", [2087, 79731])
    assert bytes(root_hash_1) == bytes(new_root[0])
    assert (merkle_tree_2._h == merkle_tree_1._h).all() and (merkle_tree_2._b == merkle_tree_1._b).all()


def test_update_invalid_mst_leaf():
    """:97-110"""
    merkle_tree = _tree("entry_16.csv", sorted_=True)
    with pytest.raises(KeyError, match="Username not found"):
        merkle_tree.update_leaf("non_existing_user", [11888, 41163])


def test_sorted_mst():
    """:112-131"""
    merkle_tree = _tree("entry_16.csv")
    sorted_merkle_tree = _tree("entry_16.csv", sorted_=True)
    assert bytes(merkle_tree.root()[1]) == bytes(sorted_merkle_tree.root()[1])
    assert bytes(merkle_tree.root()[0]) != bytes(sorted_merkle_tree.root()[0])
    names = [e[0] for e in sorted_merkle_tree.entries]
    assert names == sorted(names, key=lambda s: s.encode())
    _same_as_oracle(sorted_merkle_tree)


def test_big_uint_conversion():
    """:133-149 `big_uint_to_fp` (the tree's balance conversion), and the CSV whose first balance is 2^64
    (csv/entry_16_bigints.csv): the tree takes it as it is -- only the circuit's range check would object"""
    from circuits_halo2_amd.merkle_sum_tree import _to_fr_bytes, parse_csv_to_entries
    fp = _ints(_to_fr_bytes(3))[0]
    assert fp == 3
    big_int_over_64 = 18446744073709551616
    fp_2 = _ints(_to_fr_bytes(big_int_over_64))[0]
    assert fp_2.to_bytes(32, "little") == big_int_over_64.to_bytes(9, "little").ljust(32, b"\0")
    assert (fp_2 - fp) % R == 18446744073709551613
    entries, _ = parse_csv_to_entries(_csv("entry_16_bigints.csv"), N_CURRENCIES)
    assert entries[0] == ("dxGaEAii", [18446744073709551616, 79731])
    tree = _tree("entry_16_bigints.csv")
    assert _ints(tree.root()[1]) == [sum(e[1][c] for e in entries) for c in range(N_CURRENCIES)]
    assert _ints(tree.node(0, 0)[1])[0] == 1 << 64
    _same_as_oracle(tree)
    assert tree.verify_proof(tree.generate_proof(0))


def test_get_middle_node_hash_preimage():
    """:151-180"""
    from circuits_halo2_amd.merkle_sum_tree import MerkleSumTree
    merkle_tree = _tree("entry_16.csv")
    rng = random.Random(11)
    for _ in range(6):
        level = rng.randrange(1, merkle_tree.depth)
        index = rng.randrange(0, 1 << (merkle_tree.depth - level))
        middle_hash, middle_bal = merkle_tree.node(level, index)
        hash_preimage = merkle_tree.get_middle_node_hash_preimage(level, index)
        assert len(hash_preimage) == 32 * (N_CURRENCIES + 2)
        computed_hash, computed_bal = MerkleSumTree.middle_node_from_preimage(hash_preimage, N_CURRENCIES)
        assert bytes(middle_hash) == bytes(computed_hash) and bytes(middle_bal) == bytes(computed_bal)
    root_pre = merkle_tree.get_middle_node_hash_preimage(merkle_tree.depth, 0)
    assert bytes(MerkleSumTree.middle_node_from_preimage(root_pre, N_CURRENCIES)[0]) == bytes(merkle_tree.root()[0])
    with pytest.raises(ValueError, match="Invalid depth"):
        merkle_tree.get_middle_node_hash_preimage(0, 0)
    with pytest.raises(ValueError, match="Invalid depth"):
        merkle_tree.get_middle_node_hash_preimage(5, 0)
    with pytest.raises(IndexError, match="Node not found"):
        merkle_tree.get_middle_node_hash_preimage(1, 8)


def test_get_leaf_node_hash_preimage():
    """:182-200"""
    from circuits_halo2_amd.merkle_sum_tree import MerkleSumTree
    merkle_tree = _tree("entry_16.csv")
    for index in random.Random(5).sample(range(16), 5):
        leaf_hash, _ = merkle_tree.node(0, index)
        hash_preimage = merkle_tree.get_leaf_node_hash_preimage(index)
        assert len(hash_preimage) == 32 * (N_CURRENCIES + 1)
        assert bytes(MerkleSumTree.leaf_node_from_preimage(hash_preimage, N_CURRENCIES)[0]) == bytes(leaf_hash)


def test_tree_with_zero_element_1():
    """:202-232: 13 entries -> 16 leaves"""
    merkle_tree = _tree("entry_13.csv")
    for i in range(13, 16):
        assert merkle_tree.entries[i] == (None, [0, 0])                 # Entry::zero_entry()
        assert _ints(merkle_tree.get_leaf_node_hash_preimage(i)) == [0, 0, 0]
    root_hash, root_bal = merkle_tree.root()
    assert _ints(root_hash)[0] != 0
    assert _ints(root_bal) == [385969, 459661]
    assert merkle_tree.depth == 4
    for i in range(16):
        assert merkle_tree.verify_proof(merkle_tree.generate_proof(i))
    with pytest.raises(IndexError):
        merkle_tree.generate_proof(16)
    _same_as_oracle(merkle_tree)


def test_tree_with_zero_element_2():
    """:234-263: 17 entries -> 32 leaves"""
    merkle_tree = _tree("entry_17.csv")
    for i in range(17, 32):
        assert merkle_tree.entries[i] == (None, [0, 0])
    root_hash, root_bal = merkle_tree.root()
    assert _ints(root_hash)[0] != 0
    assert _ints(root_bal) == [556863, 556863]
    assert merkle_tree.depth == 5
    for i in range(32):
        assert merkle_tree.verify_proof(merkle_tree.generate_proof(i))
    with pytest.raises(IndexError):
        merkle_tree.generate_proof(32)
    _same_as_oracle(merkle_tree)


def test_from_params_nodes_and_entry_leaves():
    """mst.rs:137-159 `from_params`, tree.rs:15 `nodes()`, entry.rs:40-58 `compute_leaf` / `recompute_leaf`: a tree taken
    apart and put together again is the same tree (root, proofs, lookups); ragged parts are refused; an entry's leaf is
    the tree's leaf, and with other balances the leaf `update_leaf` would store"""
    from circuits_halo2_amd.merkle_sum_tree import MerkleSumTree, big_uint_to_fp, fp_to_big_uint, big_intify_username
    tree = _tree("entry_16.csv")
    nodes = tree.nodes()
    assert len(nodes) == tree.depth + 1 == 5 and [len(l) for l in nodes] == [16, 8, 4, 2, 1]
    again = MerkleSumTree.from_params(tree.root(), nodes, tree.depth, tree.entries, tree.cryptocurrencies, tree.is_sorted)
    assert again.n_currencies == N_CURRENCIES and again.cryptocurrencies == tree.cryptocurrencies
    assert bytes(again.root()[0]) == bytes(tree.root()[0]) and bytes(again.root()[1]) == bytes(tree.root()[1])
    assert again.index_of_username("AtwIxZHo") == tree.index_of_username("AtwIxZHo")
    for i in (0, 7, 15):
        assert again.verify_proof(again.generate_proof(i)) and tree.verify_proof(again.generate_proof(i))
    with pytest.raises(ValueError):
        MerkleSumTree.from_params(tree.root(), nodes[:-1], tree.depth, tree.entries, [], False)
    with pytest.raises(ValueError):
        MerkleSumTree.from_params(tree.node(0, 0), nodes, tree.depth, tree.entries, [], False)
    # Entry::compute_leaf == the stored leaf; recompute_leaf == what update_leaf stores
    for i in (0, 3, 15):
        h, b = tree.compute_leaf(i)
        assert bytes(h) == bytes(tree.node(0, i)[0]) and bytes(b) == bytes(tree.node(0, i)[1])
    name, _ = tree.entries[3]
    h2, b2 = tree.recompute_leaf(3, [11, 22])
    assert _ints(b2) == [11, 22] and bytes(h2) != bytes(tree.node(0, 3)[0])
    tree.update_leaf(name, [11, 22])
    assert bytes(tree.node(0, 3)[0]) == bytes(h2)
    # operation_helpers.rs
    assert fp_to_big_uint(big_uint_to_fp(18446744073709551616)) == 18446744073709551616
    assert _ints(tree.get_leaf_node_hash_preimage(3))[0] == big_intify_username(name) % R
