"""The reference's own prover tests, on the GPU back-end, through the mirror of its L3 API (circuits_halo2_amd/api.py):
zk_prover/src/circuits/tests.rs:45-88 (`test_valid_merkle_sum_tree_with_full_prover`), :125-152
(`test_invalid_root_hash_as_instance_with_full_prover`), the production path of backend/src/apis/round.rs:153-174
(`gen_proof_solidity_calldata` under the reference's SRS and verifying key) and `generate_setup_artifacts`' three ways
to obtain parameters (utils.rs:52-72).  Every proof is also checked by the oracle's restated verifier."""
import json
import os

import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

N_CURRENCIES, LEVELS, N_BYTES, K = 2, 4, 8, 11
CSV = os.path.join(GOLDEN, "entry_16.csv")
SRS = os.path.join(GOLDEN, "hermez-raw-11")
H = lambda s: int(s, 16)


def _gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from circuits_halo2_amd import ffi
    ffi.check(ffi.lib().sg_init(0))


def oracle_vk(params, vk):
    """the verifying key in the form the oracle's restated verifier takes"""
    from oracle import pyref as PR
    f2 = lambda b: (PR.fq_from_bytes(b[:32]), PR.fq_from_bytes(b[32:64]))
    g2 = (f2(params.g2[:64]), f2(params.g2[64:]))
    s_g2 = (f2(params.s_g2[:64]), f2(params.s_g2[64:]))
    return {"k": vk.k, "n_currencies": vk.n_currencies, "vk_digest": vk.transcript_repr, "fixed_comms": vk.fixed_comms,
            "permutation_comms": vk.permutation_comms, "g2": g2, "neg_s_g2": (s_g2[0], ((-s_g2[1][0]) % PR.Q, (-s_g2[1][1]) % PR.Q))}


@pytest.fixture(scope="module")
def artifacts():
    """`generate_setup_artifacts(K, None, init_empty())`: unsafe setup + keys, shared by the tests below"""
    _gpu()
    from circuits_halo2_amd.api import MstInclusionCircuit, generate_setup_artifacts
    circuit = MstInclusionCircuit.init_empty(LEVELS, N_CURRENCIES, N_BYTES)
    params, pk, vk = generate_setup_artifacts(K, None, circuit)
    yield params, pk, vk
    params.free()


def test_valid_merkle_sum_tree_with_full_prover(artifacts, kat):
    from circuits_halo2_amd.api import MstInclusionCircuit, full_prover, full_verifier
    from circuits_halo2_amd.merkle_sum_tree import MerkleSumTree
    from oracle import summa_verifier as SV
    params, pk, vk = artifacts
    merkle_sum_tree = MerkleSumTree.from_csv(CSV, N_CURRENCIES, N_BYTES)
    user_index = 0
    merkle_proof = merkle_sum_tree.generate_proof(user_index)
    # only now the circuit is instantiated with the actual inputs
    circuit = MstInclusionCircuit.init(merkle_proof, LEVELS)
    proof = full_prover(params, pk, circuit, circuit.instances())
    assert full_verifier(params, vk, proof, circuit.instances())
    assert len(proof) == 1632                                   # 16 compressed points + 35 scalars (Blake2b flavour)
    # the oracle's restated verifier (its own Blake2b, its own pairing) agrees
    assert SV.verify(proof, circuit.instances()[0], oracle_vk(params, vk), flavour="blake2b")
    # public input #0 is the leaf hash, #1 the root hash, then the root balances -- the reference's expected values (K5)
    inst = circuit.instances()[0]
    k5 = kat["k5"]
    assert len(inst) == circuit.num_instances() == 2 + N_CURRENCIES
    assert inst == [H(k5["leaf0"]), H(k5["root"])] + k5["root_balances"]
    # a second proof of the same statement differs (OsRng blinding) and verifies; a flipped byte does not
    proof2 = full_prover(params, pk, circuit, circuit.instances())
    assert proof2 != proof and full_verifier(params, vk, proof2, circuit.instances())
    for off in (5, 300, 600, 1000, 1600):
        bad = bytearray(proof)
        bad[off] ^= 1
        assert not full_verifier(params, vk, bytes(bad), circuit.instances()), off
        assert not SV.verify(bytes(bad), inst, oracle_vk(params, vk), flavour="blake2b")


def test_invalid_root_hash_as_instance_with_full_prover(artifacts):
    from circuits_halo2_amd.api import MstInclusionCircuit, full_prover, full_verifier
    from circuits_halo2_amd.merkle_sum_tree import MerkleSumTree
    from oracle import summa_verifier as SV
    params, pk, vk = artifacts
    merkle_proof = MerkleSumTree.from_csv(CSV, N_CURRENCIES, N_BYTES).generate_proof(0)
    circuit = MstInclusionCircuit.init(merkle_proof, LEVELS)
    instances = circuit.instances()
    instances[0][1] = 1000                                      # invalid root hash
    proof = full_prover(params, pk, circuit, instances)        # the prover does not fail ...
    assert not full_verifier(params, vk, proof, instances)     # ... the proof is rejected
    assert not SV.verify(proof, instances[0], oracle_vk(params, vk), flavour="blake2b")


def test_every_user_of_the_csv_gets_a_valid_proof(artifacts):
    """tests.rs:25-43 loops over the 16 users with the MockProver; here each gets a real proof, both flavours alternating"""
    from circuits_halo2_amd.api import MstInclusionCircuit, full_prover, full_verifier, gen_proof_solidity_calldata
    from circuits_halo2_amd.merkle_sum_tree import MerkleSumTree
    from circuits_halo2_amd import verifier as V
    params, pk, vk = artifacts
    tree = MerkleSumTree.from_csv(CSV, N_CURRENCIES, N_BYTES)
    for user_index in range(16):
        circuit = MstInclusionCircuit.init(tree.generate_proof(user_index), LEVELS)
        if user_index % 2:
            proof, public_inputs = gen_proof_solidity_calldata(params, pk, circuit)     # re-verified inside
            assert len(proof) == 2144 and public_inputs == circuit.instances()[0]
            assert V.verify_proof(params, vk, proof, public_inputs, "evm")
        else:
            assert full_verifier(params, vk, full_prover(params, pk, circuit, circuit.instances()), circuit.instances())


def test_gen_proof_solidity_calldata_under_the_reference_srs_and_key(kat):
    """backend/src/apis/round.rs:136-174: `Snapshot::new` loads ptau/hermez-raw-11 and generates the keys from the empty
    circuit; `generate_proof_of_inclusion` is `gen_proof_solidity_calldata`.  Key generation must reproduce the verifying
    key baked into the reference's verifier contract, and the calldata proof must be accepted on that key."""
    _gpu()
    from circuits_halo2_amd.api import MstInclusionCircuit, gen_proof_solidity_calldata, generate_setup_artifacts
    from circuits_halo2_amd.merkle_sum_tree import MerkleSumTree
    from oracle import summa_verifier as SV
    want = [(H(a), H(b)) for a, b in kat["fixed_comms"] + kat["permutation_comms"]]
    params, pk, vk = generate_setup_artifacts(K, SRS, MstInclusionCircuit.init_empty(LEVELS, N_CURRENCIES, N_BYTES))
    try:
        assert vk.fixed_comms + vk.permutation_comms == want
        assert pk.vk_digest == H(kat["vk_digest"])     # halo2's transcript_repr, derived (vk_repr), not passed in
        tree = MerkleSumTree.from_csv(CSV, N_CURRENCIES, N_BYTES)
        circuit = MstInclusionCircuit.init(tree.generate_proof(0), LEVELS)
        proof, public_inputs = gen_proof_solidity_calldata(params, pk, circuit)
        tr = json.load(open(os.path.join(GOLDEN, "k6_verifier_trace.json")))["vk"]
        ref_vk = {"k": K, "vk_digest": H(tr["vk_digest"]), "fixed_comms": want[:11], "permutation_comms": want[11:],
                  "g2": ((H(tr["g2_x_2"]), H(tr["g2_x_1"])), (H(tr["g2_y_2"]), H(tr["g2_y_1"]))),
                  "neg_s_g2": ((H(tr["neg_s_g2_x_2"]), H(tr["neg_s_g2_x_1"])), (H(tr["neg_s_g2_y_2"]), H(tr["neg_s_g2_y_1"])))}
        assert SV.verify(proof, public_inputs, ref_vk)          # the contract's constants, not this run's
        assert public_inputs[:2] == [H(kat["k5"]["leaf0"]), H(kat["k5"]["root"])]
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        json.dump({"proof": "0x" + proof.hex(), "public_inputs": [hex(v) for v in public_inputs]},
                  open(os.path.join(out, "api_calldata_entry16_user0.json"), "w"))
    finally:
        params.free()


def test_generate_setup_artifacts_downsizes_larger_params():
    """utils.rs:62-66: a params file with a larger k is downsized (g truncated, g_lagrange recomputed by a G1 FFT on
    the device); proofs under the downsized parameters verify"""
    _gpu()
    from circuits_halo2_amd.api import MstInclusionCircuit, full_prover, full_verifier, generate_setup_artifacts
    from circuits_halo2_amd.merkle_sum_tree import MerkleSumTree
    from oracle import summa_verifier as SV
    levels, k = 2, 10
    params, pk, vk = generate_setup_artifacts(k, SRS, MstInclusionCircuit.init_empty(levels, N_CURRENCIES, N_BYTES))
    try:
        assert params.k == k
        entries = [(f"user{i}", [100 + i, 7 * i]) for i in range(4)]
        tree = MerkleSumTree.from_entries(entries, N_CURRENCIES, N_BYTES)
        circuit = MstInclusionCircuit.init(tree.generate_proof(3), levels)
        proof = full_prover(params, pk, circuit, circuit.instances())
        assert full_verifier(params, vk, proof, circuit.instances())
        assert SV.verify(proof, circuit.instances()[0], oracle_vk(params, vk), flavour="blake2b")
        # a key made for other dimensions is refused before anything is proven
        with pytest.raises(ValueError):
            full_prover(params, pk, MstInclusionCircuit.init_empty(3, N_CURRENCIES, N_BYTES), circuit.instances())
    finally:
        params.free()


def test_prove_from_csv_compiled_program(tmp_path, kat):
    """tools/prove_from_csv (C++ over the library, include/summa_circuit.hpp + summa_prover.hpp; no interpreter): the
    reference's SRS file + its csv/entry_16.csv + user 0 -> calldata.  Key generation inside the program reproduces the
    reference's verifying key, the public inputs are the reference's expected values (K5), and with the contract's vk digest
    the proof is accepted on the reference's key by the oracle's verifier"""
    import subprocess
    from oracle import summa_verifier as SV
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tools", "prove_from_csv")
    if not os.path.exists(exe):
        pytest.skip("tools/prove_from_csv not built (python __graft_entry__.py)")
    out = str(tmp_path / "calldata.json")
    r = subprocess.run([exe, SRS, CSV, "0", str(K), out, kat["vk_digest"] if len(kat["vk_digest"]) == 66 else "0x" + kat["vk_digest"][2:].rjust(64, "0"), "3"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout.strip().splitlines()[-1])
    print(info)
    assert (info["k"], info["levels"], info["n_currencies"], info["users"], info["rows_used"]) == (11, 4, 2, 16, 1489)
    assert info["verified"] is True and info["verify_ms"] > 0          # create_proof_checked: verified by sp_verify_proof inside the program
    cd = json.load(open(out))
    want = [(H(a), H(b)) for a, b in kat["fixed_comms"] + kat["permutation_comms"]]
    assert [(H(a), H(b)) for a, b in cd["commitments"]] == want                  # the reference's verifying key
    inst = [H(v) for v in cd["public_inputs"]]
    assert inst == [H(kat["k5"]["leaf0"]), H(kat["k5"]["root"])] + kat["k5"]["root_balances"]
    tr = json.load(open(os.path.join(GOLDEN, "k6_verifier_trace.json")))["vk"]
    ref_vk = {"k": K, "vk_digest": H(tr["vk_digest"]), "fixed_comms": want[:11], "permutation_comms": want[11:],
              "g2": ((H(tr["g2_x_2"]), H(tr["g2_x_1"])), (H(tr["g2_y_2"]), H(tr["g2_y_1"]))),
              "neg_s_g2": ((H(tr["neg_s_g2_x_2"]), H(tr["neg_s_g2_x_1"])), (H(tr["neg_s_g2_y_2"]), H(tr["neg_s_g2_y_1"])))}
    proof = bytes.fromhex(cd["proof"][2:])
    assert len(proof) == 2144 and SV.verify(proof, inst, ref_vk)
    # another user, no digest passed in: the program derives halo2's transcript_repr itself (summa_circuit.hpp) and
    # arrives at the contract's constant; a key for more levels than the tree has is refused
    r = subprocess.run([exe, SRS, CSV, "11", str(K), out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    cd2 = json.load(open(out))
    vk2 = dict(ref_vk, vk_digest=H(cd2["vk_digest"]))
    assert H(cd2["vk_digest"]) == H(tr["vk_digest"]) and SV.verify(bytes.fromhex(cd2["proof"][2:]), [H(v) for v in cd2["public_inputs"]], vk2)
    assert subprocess.run([exe, SRS, CSV, "16", str(K), out], capture_output=True, text=True).returncode == 1     # user index out of bounds
    assert subprocess.run([exe, SRS, CSV, "0", "12", out], capture_output=True, text=True).returncode == 1        # k is too large for the given params


def test_params_read_rejects_points_off_the_curve(tmp_path):
    """ParamsKZG::read with SerdeFormat::RawBytes rejects points that are not on the curve; here the check runs on the
    device (sg_srs_check) when generate_setup_artifacts loads a file"""
    _gpu()
    from circuits_halo2_amd.api import MstInclusionCircuit, generate_setup_artifacts
    from circuits_halo2_amd.params import ParamsKZG
    raw = bytearray(open(SRS, "rb").read())
    good = ParamsKZG.read(bytes(raw))
    good.check()
    good.free()
    for offset in (4 + 64 * 100 + 3, 4 + 64 * 2048 + 64 * 777 + 40):       # one byte of g[100].x, one of g_lagrange[777].y
        bad = bytearray(raw)
        bad[offset] ^= 1
        path = tmp_path / "hermez-raw-11"
        path.write_bytes(bytes(bad))
        with pytest.raises(ValueError, match="Failed to read params"):
            generate_setup_artifacts(K, str(path), MstInclusionCircuit.init_empty(LEVELS, N_CURRENCIES, N_BYTES))


# --- the reference's MockProver cases [REF zk_prover/src/circuits/tests.rs:158-433] on the real prover: the same tampered
# circuits, in the reference's floor plan, go through `full_prover` -- which, like upstream's create_proof, lays out and proves
# whatever witness it is given -- and the verifier must reject the proof.  Beside each: halo2's constraint checker restated
# (circuits_halo2_amd/mock_prover.py) on the product's own device-hashed tree reports exactly the failures the reference's
# test expects (tests/test_mock_prover_cpu.py checks the same lists on the oracle's tree).
def _tampered_case(artifacts, csv_name, tamper, expected_failures):
    from circuits_halo2_amd.api import MstInclusionCircuit, full_prover, full_verifier
    from circuits_halo2_amd.merkle_sum_tree import MerkleSumTree
    from circuits_halo2_amd.mock_prover import MockProver
    from oracle import summa_verifier as SV
    params, pk, vk = artifacts
    tree = MerkleSumTree.from_csv(os.path.join(GOLDEN, csv_name), N_CURRENCIES, N_BYTES)
    circuit = MstInclusionCircuit.init(tree.generate_proof(0), LEVELS)
    instances = circuit.instances()
    tamper(circuit, instances)
    assert MockProver.run(K, circuit, instances).verify() == expected_failures
    proof = full_prover(params, pk, circuit, instances)                     # "prover should not fail" (utils.rs:102)
    assert len(proof) == 1632
    assert not full_verifier(params, vk, proof, instances)
    assert not SV.verify(proof, instances[0], oracle_vk(params, vk), flavour="blake2b")
    # the untampered circuit of the same user, same keys: accepted (the rejection above is the witness's doing)
    if csv_name == "entry_16.csv":
        good = MstInclusionCircuit.init(tree.generate_proof(0), LEVELS)
        assert MockProver.run(K, good, good.instances()).verify() == []
        assert full_verifier(params, vk, full_prover(params, pk, good, good.instances()), good.instances())


_SWAP = ("InRegion", (26, "assign nodes hashes per merkle tree level"), 0)
_ROOT_HASH = ("InRegion", (121, "permute state"), 36)
_LEAF0 = "0x167505f45c4ef4a0b051c30e881d2e8f881f26f5edb231396198a2cc1712f5ad"
_LEAF1 = "0x2c688f624d2bca741a1c2ad1ad2880721fbfd1613bbc5fe3d2ba66eb672e3aab"


def test_invalid_entry_balance_as_witness(artifacts):
    """:158-229"""
    def tamper(circuit, instances):
        circuit.entry = (circuit.entry[0], [1000, 1000])
    _tampered_case(artifacts, "entry_16.csv", tamper, [
        ("Permutation", ("advice", 0), _SWAP), ("Permutation", ("advice", 0), _ROOT_HASH),
        ("Permutation", ("advice", 2), ("InRegion", (111, "sum nodes balances per currency"), 0)),
        ("Permutation", ("advice", 2), ("InRegion", (112, "sum nodes balances per currency"), 0)),
    ] + [("Permutation", ("instance", 0), ("OutsideRegion", row)) for row in range(4)])


def test_invalid_leaf_hash_as_instance(artifacts):
    """:232-266"""
    def tamper(circuit, instances):
        instances[0][0] = 1000
    _tampered_case(artifacts, "entry_16.csv", tamper, [("Permutation", ("advice", 0), _SWAP),
                                                       ("Permutation", ("instance", 0), ("OutsideRegion", 0))])


def test_balance_not_in_range(artifacts):
    """:268-299: csv/entry_16_overflow.csv; the tree is built (mst.rs does not range-check), the circuit's range check fails"""
    _tampered_case(artifacts, "entry_16_overflow.csv", lambda circuit, instances: None, [
        ("Permutation", ("fixed", 2), ("OutsideRegion", 246)),
        ("Permutation", ("advice", 0), ("InRegion", (21, "assign value to perform range check"), 8))])


def test_non_binary_index(artifacts):
    """:302-395"""
    def tamper(circuit, instances):
        circuit.path_indices[0] = 2
    _tampered_case(artifacts, "entry_16.csv", tamper, [
        ("ConstraintNotSatisfied", (6, "bool constraint"), 0, _SWAP, [(("advice", 2), 0, "0x2")]),
        ("ConstraintNotSatisfied", (7, "swap constraint"), 0, _SWAP,
         [(("advice", 0), 0, _LEAF0), (("advice", 0), 1, _LEAF1), (("advice", 1), 0, _LEAF1), (("advice", 2), 0, "0x2")]),
        ("ConstraintNotSatisfied", (7, "swap constraint"), 1, _SWAP,
         [(("advice", 0), 0, _LEAF0), (("advice", 1), 0, _LEAF1), (("advice", 1), 1, _LEAF0), (("advice", 2), 0, "0x2")]),
        ("Permutation", ("advice", 0), _ROOT_HASH), ("Permutation", ("instance", 0), ("OutsideRegion", 1))])


def test_swapping_index(artifacts):
    """:398-433"""
    def tamper(circuit, instances):
        circuit.path_indices[0] = 1
    _tampered_case(artifacts, "entry_16.csv", tamper, [("Permutation", ("advice", 0), _ROOT_HASH),
                                                       ("Permutation", ("instance", 0), ("OutsideRegion", 1))])


def test_compiled_verifier_equals_its_python_twin(artifacts):
    """sp_verify_proof (csrc/verifier_abi.hip, what `full_verifier` / `create_proof_checked` run) against verifier.py's
    Python twin and the oracle's restated verifier: the reference's shipped proof K6 under the reference's key, GPU-made
    proofs of both flavours, and the same rejections -- flipped bytes all over the proof, a changed public input, wrong
    length, wrong flavour, an unreduced scalar, a point off the curve, another key"""
    from circuits_halo2_amd import api, params as PM, verifier as V
    from circuits_halo2_amd.merkle_sum_tree import MerkleSumTree
    R = V.R
    cases = []
    # K6: the reference's own proof, key and SRS
    tr = json.load(open(os.path.join(GOLDEN, "k6_verifier_trace.json")))["vk"]
    comm = [(H(a), H(b)) for a, b in tr["commitments"]]
    vk6 = api.VerifyingKey(11, 2, comm[:11], comm[11:], H(tr["vk_digest"]))
    cd = json.load(open(os.path.join(GOLDEN, "k6_inclusion_proof_solidity_calldata.json")))
    p6 = PM.ParamsKZG.read(open(SRS, "rb"))
    cases.append((p6, vk6, bytes.fromhex(cd["proof"][2:]), [H(x) for x in cd["public_inputs"]], "evm"))
    # proofs made here, both flavours
    params, pk, vk = artifacts
    circuit = api.MstInclusionCircuit.init(MerkleSumTree.from_csv(CSV, N_CURRENCIES, N_BYTES).generate_proof(3), LEVELS)
    inst = circuit.instances()[0]
    cases.append((params, vk, api._create_proof(params, pk, circuit, [inst], "evm"), inst, "evm"))
    cases.append((params, vk, api.full_prover(params, pk, circuit, [inst]), inst, "blake2b"))
    both = lambda *a: (V.verify_proof(*a, driver="native"), V.verify_proof(*a, driver="python"))
    for prm, key, proof, pub, flavour in cases:
        assert both(prm, key, proof, pub, flavour) == (True, True)
        step = max(1, len(proof) // 41)
        for off in list(range(0, len(proof), step)) + [len(proof) - 1]:
            bad = bytearray(proof)
            bad[off] ^= 0x04
            assert both(prm, key, bytes(bad), pub, flavour) == (False, False), (flavour, off)
        assert both(prm, key, proof, [pub[0], pub[1], pub[2] + 1, pub[3]], flavour) == (False, False)
        assert both(prm, key, proof, pub[:3], flavour) == (False, False)
        assert both(prm, key, proof[:-1], pub, flavour) == (False, False)
        assert both(prm, key, proof + b"\0", pub, flavour) == (False, False)
        assert both(prm, key, proof, pub, "blake2b" if flavour == "evm" else "evm") == (False, False)
        assert both(prm, key, proof, [R] + pub[1:], flavour) == (False, False)
        other = type(key)(key.k, key.n_currencies, key.fixed_comms, key.permutation_comms, (key.transcript_repr + 1) % R)
        assert both(prm, other, proof, pub, flavour) == (False, False)
        swapped = type(key)(key.k, key.n_currencies, key.fixed_comms[1:] + key.fixed_comms[:1], key.permutation_comms, key.transcript_repr)
        assert both(prm, swapped, proof, pub, flavour) == (False, False)
    # an unreduced evaluation (evm: scalar 0 of the proof set to r) and a first commitment off the curve
    prm, key, proof, pub, flavour = cases[0]
    bad = bytearray(proof)
    bad[0x380:0x3a0] = R.to_bytes(32, "big")
    assert both(prm, key, bytes(bad), pub, flavour) == (False, False)
    bad = bytearray(proof)
    bad[32:64] = (int.from_bytes(proof[32:64], "big") ^ 1).to_bytes(32, "big")
    assert both(prm, key, bytes(bad), pub, flavour) == (False, False)
    p6.free()


@pytest.mark.parametrize("nc", [1, 3, 4])
def test_other_currency_counts_prove_and_verify(nc, capfd):
    """`MstInclusionCircuit<LEVELS, N_CURRENCIES, N_BYTES>` for N_CURRENCIES other than the reference tests' 2 (its bench runs
    1: benches/full_solvency_flow.rs:15; one sum gate and one Poseidon input per currency, so the gate program, the floor
    plan, the verifying key's digest and the witness kernel all depend on it): key generation, a proof from the host
    witness and one from the device witness, both flavours, accepted by the product's verifier and the oracle's; a proof
    under another currency count's key is rejected; the gate block ran as the ahead-of-time kernel of that count"""
    _gpu()
    import torch
    from circuits_halo2_amd import api, arithmetic as A
    from circuits_halo2_amd.merkle_sum_tree import DeviceMerkleSumTree
    from circuits_halo2_amd.mock_prover import MockProver
    from circuits_halo2_amd.utils import random_fr_canonical
    from oracle import summa_verifier as SV
    levels, k = 5, 12
    size = 1 << levels
    bal = random_fr_canonical(500 + nc, size * nc).reshape(-1, 32).copy()
    bal[:, 4:] = 0
    tree = DeviceMerkleSumTree(A.fr_random(bytes(range(32)), 40 + nc, size), A.fr_to_montgomery(torch.from_numpy(bal.reshape(-1)).cuda()), levels, nc)
    params, pk, vk = api.generate_setup_artifacts(k, None, api.MstInclusionCircuit.init_empty(levels, nc))
    try:
        ovk = oracle_vk(params, vk)
        host = api.MstInclusionCircuit.init(tree.generate_proof(9), levels)
        assert host.n_currencies == nc and len(host.instances()[0]) == 2 + nc
        assert MockProver.run(k, host, host.instances()).verify() == []
        os.environ["SG_GATES_DEBUG"] = "1"
        try:
            proof, inst = api.gen_proof_solidity_calldata(params, pk, host)
        finally:
            os.environ.pop("SG_GATES_DEBUG", None)
        assert f"ahead-of-time program MstGatesNc{nc}" in capfd.readouterr().err
        assert SV.verify(proof, inst, ovk) and inst == tree.public_inputs(9)
        dev = api.MstInclusionCircuit.init_from_tree(tree, 9)
        blake = api.full_prover(params, pk, dev, dev.instances())
        assert api.full_verifier(params, vk, blake, dev.instances()) and SV.verify(blake, inst, ovk, flavour="blake2b")
        bad = list(inst)
        bad[-1] = (bad[-1] + 1) % SV.R
        assert not api.full_verifier(params, vk, blake, [bad])
        other = api.VerifyingKey(vk.k, 2, vk.fixed_comms, vk.permutation_comms, vk.transcript_repr)   # the gate list of two currencies
        from circuits_halo2_amd import verifier as V
        assert not V.verify_proof(params, other, proof, inst, "evm") or nc == 2
    finally:
        params.free()
