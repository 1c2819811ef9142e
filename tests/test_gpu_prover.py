"""create_proof on the GPU (circuits_halo2_amd.prover) for the reference circuit's constraint system, checked by the
restated verifier (oracle/summa_verifier.py) -- the same code that accepts the reference's own shipped proof and
matches the reference verifier's intermediate values (tests/test_verifier_cpu.py).  Every MSM, NTT, grand product,
evaluate_h pass, evaluation and division of the proof ran on the device through the C ABI: one wrong bit anywhere
and the pairing check fails."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def make_setup(k):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    import circuits_halo2_amd as sg
    from circuits_halo2_amd import ffi, prover
    from circuits_halo2_amd.utils import ints_to_fr
    import mst_assignment as MA
    from oracle import pyref as PR
    ffi.check(ffi.lib().sg_init(0))
    asg = MA.build(k)
    if k <= 10:
        assert MA.check_gates(asg, k)      # row by row with Python integers (the floor plan does not depend on k)
    tau = ints_to_fr([0x1D0C0FFEE1234567890ABCDEF])
    params = sg.ParamsKZG.setup(k, tau)
    dev = lambda ints: torch.from_numpy(ints_to_fr(ints)).cuda()
    pk = prover.ProvingKey(params, k, [dev(c) for c in asg["fixed"]], [dev(c) for c in asg["sigma"]])
    f2 = lambda b: (PR.fq_from_bytes(b[:32]), PR.fq_from_bytes(b[32:64]))
    g2 = (f2(params.g2[:64]), f2(params.g2[64:]))
    s_g2 = (f2(params.s_g2[:64]), f2(params.s_g2[64:]))
    vk = {"k": k, "vk_digest": pk.vk_digest, "fixed_comms": pk.fixed_comms, "permutation_comms": pk.permutation_comms, "g2": g2,
          "neg_s_g2": (s_g2[0], ((-s_g2[1][0]) % PR.Q, (-s_g2[1][1]) % PR.Q))}
    return {"k": k, "asg": asg, "params": params, "pk": pk, "vk": vk, "dev": dev, "prover": prover}


@pytest.fixture(scope="module")
def setup():
    s = make_setup(9)
    yield s
    s["params"].free()


def seeded_rng(seed):
    """a deterministic 32-byte ChaCha20 key for the prover's blinding values"""
    return seed.to_bytes(4, "little") * 8


def test_gpu_proof_is_accepted_by_the_restated_verifier(setup):
    from oracle import summa_verifier as SV
    s = setup
    advice = [s["dev"](c) for c in s["asg"]["advice"]]
    proof = s["prover"].create_proof(s["params"], s["pk"], advice, s["asg"]["instances"], seeded_rng(1))
    assert len(proof) == 2144                       # the reference's proof length (InclusionVerifier.sol:274)
    assert SV.verify(proof, s["asg"]["instances"], s["vk"])
    # a second proof of the same statement differs (fresh blinding) and verifies too
    proof2 = s["prover"].create_proof(s["params"], s["pk"], advice, s["asg"]["instances"], seeded_rng(1000))
    assert proof2 != proof and SV.verify(proof2, s["asg"]["instances"], s["vk"])
    # the proof is bound to its public inputs and to every byte
    wrong = list(s["asg"]["instances"])
    wrong[2] += 1
    assert not SV.verify(proof, wrong, s["vk"])
    for off in (0x10, 0x150, 0x390, 0x700, 0x7f0, 0x850):
        p = bytearray(proof)
        p[off] ^= 1
        assert not SV.verify(bytes(p), s["asg"]["instances"], s["vk"]), hex(off)


def test_quotient_on_cosets_gives_the_same_proof_as_on_the_extended_domain(setup):
    """the quotient h is unique, so evaluating its numerator on 5 cosets (the product's default) or on halo2's whole
    extended domain must lead to the same pieces: with the same blinding stream the two proofs are equal byte for byte
    (both transcript flavours), and so is a proof of a witness whose gates fail (h is then not a polynomial: garbage,
    but the same garbage is not required -- only the valid case is compared)"""
    from oracle import summa_verifier as SV
    s = setup
    P = s["prover"]
    assert s["pk"].quotient_domain == "cosets"
    dev = s["dev"]
    pk_ext = P.ProvingKey(s["params"], s["k"], [dev(c) for c in s["asg"]["fixed"]], [dev(c) for c in s["asg"]["sigma"]],
                          quotient_domain="extended")
    assert pk_ext.vk_digest == s["pk"].vk_digest and pk_ext.fixed_ext[0].numel() == 8 * s["pk"].fixed_ext[0].numel() // 5
    advice = [dev(c) for c in s["asg"]["advice"]]
    inst = s["asg"]["instances"]
    for seed in (31, 32):
        a = P.create_proof(s["params"], s["pk"], advice, inst, seeded_rng(seed))
        b = P.create_proof(s["params"], pk_ext, advice, inst, seeded_rng(seed))
        assert a == b and SV.verify(a, inst, s["vk"])
    a = P.create_proof(s["params"], s["pk"], advice, inst, seeded_rng(33), transcript=P.Blake2bWrite())
    b = P.create_proof(s["params"], pk_ext, advice, inst, seeded_rng(33), transcript=P.Blake2bWrite())
    assert a == b and SV.verify(a, inst, s["vk"], flavour="blake2b")
    with pytest.raises(ValueError):
        P.ProvingKey(s["params"], s["k"], [dev(c) for c in s["asg"]["fixed"]], [dev(c) for c in s["asg"]["sigma"]], quotient_domain="x")


@pytest.mark.parametrize("what", ["gate", "lookup", "copy"])
def test_gpu_prover_unsatisfied_assignments(setup, what):
    """a violated gate yields a proof the verifier rejects (the quotient is not a polynomial); violated lookup /
    copy constraints are detected while the grand products are built"""
    from oracle import summa_verifier as SV
    s = setup
    adv = [list(c) for c in s["asg"]["advice"]]
    if what == "gate":
        adv[2][55] += 1            # sum gate: a2 != a0 + a1 (and its copy a0[37] follows, so only the gate breaks)
        adv[0][37] += 1
        adv[0][39] += 1
    elif what == "lookup":
        adv[0][60] += 1 << 16      # the low "byte" is no longer below 2^8
    else:
        adv[1][70] = 6             # copy of the fixed constant 5
    advice = [s["dev"](c) for c in adv]
    if what == "gate":
        proof = s["prover"].create_proof(s["params"], s["pk"], advice, s["asg"]["instances"], seeded_rng(7))
        assert not SV.verify(proof, s["asg"]["instances"], s["vk"])
    else:
        with pytest.raises(ValueError):
            s["prover"].create_proof(s["params"], s["pk"], advice, s["asg"]["instances"], seeded_rng(7))
    # malformed calls are refused before anything is enqueued
    for bad in (lambda: s["prover"].create_proof(s["params"], s["pk"], advice[:2], s["asg"]["instances"]),
                lambda: s["prover"].create_proof(s["params"], s["pk"], [advice[0][:64]] * 3, s["asg"]["instances"]),
                lambda: s["prover"].create_proof(s["params"], s["pk"], advice, [1 << 255]),
                lambda: s["prover"].create_proof(s["params"], s["pk"], advice, s["asg"]["instances"], b"short")):
        with pytest.raises(ValueError):
            bad()


def test_gpu_proof_under_the_reference_srs_and_contract_constants():
    """k = 11 with the reference's own SRS file (backend/ptau/hermez-raw-11, the reference circuit's size): the proof
    verifies with the G2 constants of the reference's verifier contract (tests/golden/k6_verifier_trace.json)"""
    import json
    import torch
    import circuits_halo2_amd as sg
    from circuits_halo2_amd import ffi, prover
    from circuits_halo2_amd import mst_inclusion as M
    from circuits_halo2_amd.utils import ints_to_fr
    from oracle import summa_verifier as SV
    ffi.check(ffi.lib().sg_init(0))
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    params = sg.ParamsKZG.read(open(os.path.join(gold, "hermez-raw-11"), "rb"))
    try:
        k = params.k
        assert k == 11
        asg = M.example_assignment(k)
        dev = lambda ints: torch.from_numpy(ints_to_fr(ints)).cuda()
        pk = prover.ProvingKey(params, k, [dev(c) for c in asg["fixed"]], [dev(c) for c in asg["sigma"]])
        proof = prover.create_proof(params, pk, [dev(c) for c in asg["advice"]], asg["instances"], seeded_rng(11))
        v = json.load(open(os.path.join(gold, "k6_verifier_trace.json")))["vk"]
        H = lambda s: int(s, 16)
        vk = {"k": k, "vk_digest": pk.vk_digest, "fixed_comms": pk.fixed_comms, "permutation_comms": pk.permutation_comms,
              "g2": ((H(v["g2_x_2"]), H(v["g2_x_1"])), (H(v["g2_y_2"]), H(v["g2_y_1"]))),
              "neg_s_g2": ((H(v["neg_s_g2_x_2"]), H(v["neg_s_g2_x_1"])), (H(v["neg_s_g2_y_2"]), H(v["neg_s_g2_y_1"])))}
        assert SV.verify(proof, asg["instances"], vk)
        # the range table's commitment is the reference's own fixed_comms[4] (K2): same column, same SRS
        assert pk.fixed_comms[4] == (H(v["commitments"][4][0]), H(v["commitments"][4][1]))
    finally:
        params.free()


def test_gpu_proof_at_k17(gpu_time_budget=None):
    """the configuration the reference benchmarks (k = 17: 2^17 rows, 2^20 extended rows): one proof, verified"""
    import time
    from oracle import summa_verifier as SV
    s = make_setup(17)
    try:
        advice = [s["dev"](c) for c in s["asg"]["advice"]]
        s["params"].precompute()
        s["prover"].create_proof(s["params"], s["pk"], advice, s["asg"]["instances"], seeded_rng(5))   # warm-up
        t = time.perf_counter()
        proof = s["prover"].create_proof(s["params"], s["pk"], advice, s["asg"]["instances"], seeded_rng(6))
        print(f"create_proof k=17: {(time.perf_counter() - t) * 1e3:.1f} ms")
        assert SV.verify(proof, s["asg"]["instances"], s["vk"])
        p = bytearray(proof)
        p[0x400] ^= 1
        assert not SV.verify(bytes(p), s["asg"]["instances"], s["vk"])
    finally:
        s["params"].free()


@pytest.mark.parametrize("k", [9, 17])
def test_cpp_prover_proof_is_accepted(k, tmp_path):
    """include/summa_prover.hpp (the compiled-host create_proof over the C ABI), run as its own process on a bundle
    exported from the Python proving key: the proof it writes is accepted by the restated verifier"""
    import json
    import subprocess
    from oracle import summa_verifier as SV
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tools", "create_proof_cpp")
    if not os.path.exists(exe):
        pytest.skip("tools/create_proof_cpp not built (python __graft_entry__.py)")
    s = make_setup(k)
    try:
        advice = [s["dev"](c) for c in s["asg"]["advice"]]
        bundle, out = str(tmp_path / "bundle.bin"), str(tmp_path / "proof.bin")
        s["prover"].export_bundle(bundle, s["params"], s["pk"], advice, s["asg"]["instances"])
    finally:
        s["params"].free()
    r = subprocess.run([exe, bundle, out, "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout.strip().splitlines()[-1])
    print(info)
    proof = open(out, "rb").read()
    assert len(proof) == 2144 == info["proof_bytes"]
    assert SV.verify(proof, s["asg"]["instances"], s["vk"])
    p = bytearray(proof)
    p[0x3a0] ^= 1
    assert not SV.verify(bytes(p), s["asg"]["instances"], s["vk"])


def test_fr_random_matches_the_chacha20_twin():
    """sg_fr_random_dev: bit-exact with the oracle's RFC 8439 ChaCha20 twin (pinned by the RFC's test vector in
    tests/test_verifier_cpu.py), for several keys / stream ids / lengths; rejection sampling included (~24 % redraws)"""
    import torch
    from circuits_halo2_amd import arithmetic as A, ffi
    from oracle import pyref as PR
    ffi.check(ffi.lib().sg_init(0))
    for key, stream, n in ((bytes(range(32)), 1, 700), (bytes(32), 0, 1), (bytes([255] * 32), (1 << 40) + 5, 333), (b"summa" * 6 + b"ab", 2, 0)):
        got = A.fr_random(key, stream, n).cpu().numpy().tobytes()
        want = b"".join(v.to_bytes(32, "little") for v in PR.chacha_field_elements(key, stream, n))
        assert got == want
    # seeded proofs are reproducible, unseeded ones are not
    s = make_setup(9)
    try:
        advice = [s["dev"](c) for c in s["asg"]["advice"]]
        mk = lambda seed: s["prover"].create_proof(s["params"], s["pk"], advice, s["asg"]["instances"], seed)
        assert mk(seeded_rng(3)) == mk(seeded_rng(3)) != mk(seeded_rng(4))
        assert mk(None) != mk(None)
    finally:
        s["params"].free()


@pytest.mark.parametrize("rows,top,seed", [(500, 256, 1), (4090, 65536, 2), (131066, 256, 3), (7, 3, 4)])
def test_lookup_permute_on_device(rows, top, seed):
    """sg_lookup_permute_small_dev == the host permutation (prover.permute_expression_pair, itself checked on the CPU
    against the defining properties): A' sorted, S' a rearrangement with A'[i] == S'[i] or A'[i] == A'[i-1]"""
    import torch
    from circuits_halo2_amd import arithmetic as A, ffi
    from circuits_halo2_amd.prover import permute_expression_pair
    ffi.check(ffi.lib().sg_init(0))
    rng = np.random.default_rng(seed)
    table = np.zeros((rows, 4), dtype=np.uint64)
    m = min(rows, top)
    table[:m, 0] = rng.permutation(top)[:m] if top <= rows * 4 else rng.integers(0, top, m)
    inp = table[rng.integers(0, m, rows)]
    if rows > 100:
        inp[rng.integers(0, rows, rows // 2)] = table[0]          # long runs of one value
    dev = lambda limbs: A.fr_to_montgomery(torch.from_numpy(limbs.view(np.uint8).reshape(-1).copy()).cuda())
    want_a, want_s = permute_expression_pair(inp, table)
    got = A.lookup_permute_small(dev(inp), dev(table), rows)
    canon = lambda t: A.fr_from_montgomery(t).cpu().numpy().view(np.uint64).reshape(-1, 4)
    assert (canon(got[0]) == want_a).all() and (canon(got[1]) == want_s).all()
    # an input value outside the table; a table outside the range this entry point handles
    bad = inp.copy()
    bad[rows // 2, 0] = int(table[:, 0].max()) + 1
    if bad[rows // 2, 0] < 65536:
        with pytest.raises(ValueError):
            A.lookup_permute_small(dev(bad), dev(table), rows)
    big = table.copy()
    big[0, 0] = 70000
    assert A.lookup_permute_small(dev(inp), dev(big), rows) is None
    big[0] = [5, 1, 0, 0]
    assert A.lookup_permute_small(dev(inp), dev(big), rows) is None


def test_inclusion_proof_of_the_reference_csv_and_srs(tmp_path):
    """BASELINE configs[0] end to end on the device: the reference's csv/entry_16.csv (N_CURRENCIES = 2), user 0,
    k = 11, the reference's SRS file.  Merkle sum tree on the GPU -> inclusion witness over the reference circuit's
    constraint system (mst_inclusion.assign_inclusion) -> create_proof (Python driver and the C++ one) -> the
    restated verifier with the contract's pairing constants accepts, and the public inputs are the reference's own
    expected values (K5: zk_prover/src/circuits/tests.rs:341,346 -- leaf hash, root hash, root balances)."""
    import json
    import subprocess
    import torch
    import circuits_halo2_amd as sg
    from circuits_halo2_amd import ffi, prover
    from circuits_halo2_amd import mst_inclusion as M
    from circuits_halo2_amd.utils import ints_to_fr
    from oracle import pyref as PR
    from oracle import summa_verifier as SV
    import mst_assignment as MA
    ffi.check(ffi.lib().sg_init(0))
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    from circuits_halo2_amd.merkle_sum_tree import MerkleSumTree
    tree = MerkleSumTree.from_csv(os.path.join(gold, "entry_16.csv"), 2)
    mp = tree.generate_proof(0)
    assert tree.verify_proof(mp)
    toi = lambda b: PR.fr_from_bytes(bytes(b))
    ints = lambda b: [toi(b[i:i + 32]) for i in range(0, len(b), 32)]
    name, bal = mp["entry"]
    username = int.from_bytes(PR.keccak256(name.encode()), "big") % PR.R
    siblings = [(toi(h), ints(b)) for h, b in mp["siblings"]]
    asg = M.assign_inclusion(11, username, [int(x) for x in bal], siblings, mp["path_indices"])
    kat = json.load(open(os.path.join(gold, "kat.json")))["k5"]
    assert asg["instances"] == [int(kat["leaf0"], 16), int(kat["root"], 16)] + kat["root_balances"]
    assert MA.check_gates(asg, 11)
    params = sg.ParamsKZG.read(open(os.path.join(gold, "hermez-raw-11"), "rb"))
    try:
        dev = lambda v: torch.from_numpy(ints_to_fr(v)).cuda()
        pk = prover.ProvingKey(params, 11, [dev(c) for c in asg["fixed"]], [dev(c) for c in asg["sigma"]])
        advice = [dev(c) for c in asg["advice"]]
        proof = prover.create_proof(params, pk, advice, asg["instances"])
        v = json.load(open(os.path.join(gold, "k6_verifier_trace.json")))["vk"]
        H = lambda s: int(s, 16)
        vk = {"k": 11, "vk_digest": pk.vk_digest, "fixed_comms": pk.fixed_comms, "permutation_comms": pk.permutation_comms,
              "g2": ((H(v["g2_x_2"]), H(v["g2_x_1"])), (H(v["g2_y_2"]), H(v["g2_y_1"]))),
              "neg_s_g2": ((H(v["neg_s_g2_x_2"]), H(v["neg_s_g2_x_1"])), (H(v["neg_s_g2_y_2"]), H(v["neg_s_g2_y_1"])))}
        assert SV.verify(proof, asg["instances"], vk)
        wrong = list(asg["instances"])
        wrong[3] += 1                                   # a different root balance
        assert not SV.verify(proof, wrong, vk)
        # a witness for a balance that is not the leaf's fails at the copy / gate level
        forged = [list(c) for c in asg["advice"]]
        forged[0][0] += 1                               # the first range-checked balance cell
        with pytest.raises(ValueError):
            prover.create_proof(params, pk, [dev(c) for c in forged], asg["instances"])
        exe = os.path.join(os.path.dirname(gold.rstrip("/")), "..", "tools", "create_proof_cpp")
        if os.path.exists(exe):
            bundle, out = str(tmp_path / "b.bin"), str(tmp_path / "p.bin")
            prover.export_bundle(bundle, params, pk, advice, asg["instances"])
            r = subprocess.run([exe, bundle, out, "3"], capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr
            print(r.stdout.strip().splitlines()[-1])
            assert SV.verify(open(out, "rb").read(), asg["instances"], vk)
    finally:
        params.free()


@pytest.mark.parametrize("nc", [1, 2])
def test_inclusion_proof_levels20_k13(nc):
    """the reference bench's shape (LEVELS = 20, k = 13; N_CURRENCIES = 1 is its exact configuration,
    zk_prover/benches/full_solvency_flow.rs:13-16): a 2^20-user Merkle sum tree on the device, the inclusion
    witness of one user in the reference circuit's own floor plan (7041 rows), proof, verification; the public inputs
    are the device tree's leaf and root"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import circuits_halo2_amd as sg
    from circuits_halo2_amd import ffi
    from circuits_halo2_amd.utils import ints_to_fr
    from oracle import pyref as PR
    from oracle import summa_verifier as SV
    import full_flow
    import mst_assignment as MA
    ffi.check(ffi.lib().sg_init(0))
    k = 13
    asg = full_flow.build(20, k, nc=nc)
    assert MA.check_gates(asg, k, nc)
    params = sg.ParamsKZG.setup(k, ints_to_fr([0xABCDEF0123456789]))
    try:
        pk, advice, proof = full_flow.keygen_and_prove(asg, k, params, reps=1, nc=nc)
        f2 = lambda b: (PR.fq_from_bytes(b[:32]), PR.fq_from_bytes(b[32:64]))
        s_g2 = (f2(params.s_g2[:64]), f2(params.s_g2[64:]))
        vk = {"k": k, "n_currencies": nc, "vk_digest": pk.vk_digest, "fixed_comms": pk.fixed_comms,
              "permutation_comms": pk.permutation_comms, "g2": PR.G2_GENERATOR,
              "neg_s_g2": (s_g2[0], ((-s_g2[1][0]) % PR.Q, (-s_g2[1][1]) % PR.Q))}
        assert SV.verify(proof, asg["instances"], vk)
        assert not SV.verify(proof, asg["instances"][:1] + [asg["instances"][1] ^ 1] + asg["instances"][2:], vk)
    finally:
        params.free()


def test_reference_floor_plan_proof_under_the_reference_verifying_key():
    """The reference circuit's OWN layout (mst_inclusion.reference_assignment: halo2's floor planner and permutation
    assembly replayed over the reference's synthesize), csv/entry_16.csv user 0, the reference's SRS: key generation on
    the GPU reproduces the reference's verifying key -- all 11 fixed and 6 permutation commitments of
    contracts/src/InclusionVerifier.sol:238-271 --, and a proof made here with the contract's vk digest is accepted by
    the restated verifier running on the REFERENCE'S verifying key (tests/golden/k6_verifier_trace.json), with the
    reference's expected public inputs (K5).  The proof is also written to gpurun_out/ so that the reference's
    contract itself can be run on it (tests/test_verifier_cpu.py, where the reference checkout exists)."""
    import json
    import torch
    import circuits_halo2_amd as sg
    from circuits_halo2_amd import ffi, prover
    from circuits_halo2_amd import mst_inclusion as M
    from circuits_halo2_amd.merkle_sum_tree import MerkleSumTree
    from circuits_halo2_amd.utils import ints_to_fr
    from oracle import pyref as PR
    from oracle import summa_verifier as SV
    ffi.check(ffi.lib().sg_init(0))
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    tree = MerkleSumTree.from_csv(os.path.join(gold, "entry_16.csv"), 2)
    mp = tree.generate_proof(0)
    toi = lambda b: PR.fr_from_bytes(bytes(b))
    ints = lambda b: [toi(b[i:i + 32]) for i in range(0, len(b), 32)]
    name, bal = mp["entry"]
    username = int.from_bytes(PR.keccak256(name.encode()), "big") % PR.R
    asg = M.reference_assignment(11, username, [int(x) for x in bal], mp["path_indices"], ints(mp["sibling_leaf_node_hash_preimage"]),
                                 [ints(p) for p in mp["sibling_middle_node_hash_preimages"]])
    trace = json.load(open(os.path.join(gold, "k6_verifier_trace.json")))["vk"]
    H = lambda s: int(s, 16)
    ref_comms = [(H(a), H(b)) for a, b in trace["commitments"]]
    params = sg.ParamsKZG.read(open(os.path.join(gold, "hermez-raw-11"), "rb"))
    try:
        dev = lambda v: torch.from_numpy(ints_to_fr(v)).cuda()
        pk = prover.ProvingKey(params, 11, [dev(c) for c in asg["fixed"]], [dev(c) for c in asg["sigma"]])
        assert pk.fixed_comms == ref_comms[:11]              # the reference's verifying key, from this key generation
        assert pk.permutation_comms == ref_comms[11:]
        pk.vk_digest = H(trace["vk_digest"])                  # the transcript starts from the contract's digest
        proof = prover.create_proof(params, pk, [dev(c) for c in asg["advice"]], asg["instances"])
        vk = {"k": 11, "vk_digest": H(trace["vk_digest"]), "fixed_comms": ref_comms[:11], "permutation_comms": ref_comms[11:],
              "g2": ((H(trace["g2_x_2"]), H(trace["g2_x_1"])), (H(trace["g2_y_2"]), H(trace["g2_y_1"]))),
              "neg_s_g2": ((H(trace["neg_s_g2_x_2"]), H(trace["neg_s_g2_x_1"])), (H(trace["neg_s_g2_y_2"]), H(trace["neg_s_g2_y_1"])))}
        assert SV.verify(proof, asg["instances"], vk)
        kat = json.load(open(os.path.join(gold, "kat.json")))["k5"]
        assert asg["instances"] == [int(kat["leaf0"], 16), int(kat["root"], 16)] + kat["root_balances"]
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        json.dump({"proof": "0x" + proof.hex(), "public_inputs": [hex(v) for v in asg["instances"]]},
                  open(os.path.join(out, "gpu_proof_entry16_user0.json"), "w"))
        # the compiled-host prover on the same key and witness
        exe = os.path.join(os.path.dirname(out), "tools", "create_proof_cpp")
        if os.path.exists(exe):
            import subprocess
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                prover.export_bundle(os.path.join(td, "b.bin"), params, pk, [dev(c) for c in asg["advice"]], asg["instances"])
                r = subprocess.run([exe, os.path.join(td, "b.bin"), os.path.join(td, "p.bin"), "3"], capture_output=True, text=True, timeout=300)
                assert r.returncode == 0, r.stderr
                cpp_proof = open(os.path.join(td, "p.bin"), "rb").read()
            assert SV.verify(cpp_proof, asg["instances"], vk)
            json.dump({"proof": "0x" + cpp_proof.hex(), "public_inputs": [hex(v) for v in asg["instances"]]},
                      open(os.path.join(out, "gpu_proof_entry16_user0_cpp.json"), "w"))
    finally:
        params.free()


def test_native_driver_matches_the_python_driver(setup):
    """the library's compiled-host prover behind the C ABI of include/summa_prover.h (sp_create_proof) against
    circuits_halo2_amd.prover.create_proof on the same key and witness: both transcript flavours verify with the
    oracle's restated verifier, errors map to the same exceptions"""
    from circuits_halo2_amd import ffi
    from oracle import summa_verifier as SV
    s = setup
    P = s["prover"]
    advice = [s["dev"](c) for c in s["asg"]["advice"]]
    inst = s["asg"]["instances"]
    before = [a.clone() for a in advice]
    for flavour, size in (("evm", 2144), ("blake2b", 1632)):
        proof = P.create_proof_native(s["params"], s["pk"], advice, inst, flavour)
        assert len(proof) == size and SV.verify(proof, inst, s["vk"], flavour=flavour)
        assert P.create_proof_native(s["params"], s["pk"], advice, inst, flavour) != proof       # fresh blinding
        bad = bytearray(proof)
        bad[size // 2] ^= 1
        assert not SV.verify(bytes(bad), inst, s["vk"], flavour=flavour)
    assert all((a == b).all() for a, b in zip(advice, before))                  # the caller's columns are untouched ...
    P.create_proof_native(s["params"], s["pk"], advice, inst, "evm", in_place=True)
    assert not all((a == b).all() for a, b in zip(advice, before))              # ... unless it hands them over
    advice = before
    # the Python driver's Blake2b flavour on the same key
    blake = P.create_proof(s["params"], s["pk"], advice, inst, seeded_rng(21), transcript=P.Blake2bWrite())
    assert len(blake) == 1632 and SV.verify(blake, inst, s["vk"], flavour="blake2b")
    assert not SV.verify(blake, inst, s["vk"], flavour="evm")
    # witness errors: SG_ERR_WITNESS -> ValueError, as the Python driver raises
    adv = [list(c) for c in s["asg"]["advice"]]
    adv[0][60] += 1 << 16
    with pytest.raises(ValueError, match="not in the table"):
        P.create_proof_native(s["params"], s["pk"], [s["dev"](c) for c in adv], inst)
    adv = [list(c) for c in s["asg"]["advice"]]
    adv[1][70] = 6
    with pytest.raises(ValueError, match="permutation"):
        P.create_proof_native(s["params"], s["pk"], [s["dev"](c) for c in adv], inst)
    unchecked = P.create_proof_native(s["params"], s["pk"], [s["dev"](c) for c in adv], inst, sanity_checks=False)
    assert len(unchecked) == 2144 and not SV.verify(unchecked, inst, s["vk"])       # upstream's default build: a proof that fails
    # advice words that are not canonical field elements (>= r) are refused by both drivers when the checks are on
    raw = [a.clone() for a in advice]
    raw[2][32 * 9 + 31] = 0xff
    with pytest.raises(ValueError, match="canonical"):
        P.create_proof_native(s["params"], s["pk"], raw, inst)
    with pytest.raises(ValueError, match="canonical"):
        P.create_proof(s["params"], s["pk"], raw, inst, seeded_rng(22))
    # malformed calls
    with pytest.raises(ValueError):
        P.create_proof_native(s["params"], s["pk"], advice[:2], inst)
    with pytest.raises(ValueError):
        P.create_proof_native(s["params"], s["pk"], advice, [1 << 255])
    import ctypes as C
    L = ffi.prover_lib()
    assert L.sp_key_destroy(C.c_uint64(987654)) == -1 and b"unknown key" in L.sp_last_error()
    size = C.c_size_t(0)
    out = np.zeros(2144, dtype=np.uint8)
    ptrs = (C.c_void_p * 3)(*[a.data_ptr() for a in advice])
    key = P.native_key(s["pk"], s["params"])
    from circuits_halo2_amd.utils import ints_to_fr
    inst_b = ints_to_fr(inst)
    args = (C.c_uint64(key), ptrs, ffi.ptr(inst_b), C.c_uint32(len(inst)))
    assert L.sp_create_proof(*args, 7, 1, None, ffi.ptr(out), C.c_size_t(2144), C.byref(size)) == -1                  # unknown transcript
    assert L.sp_create_proof(*args, 0, 1, None, ffi.ptr(out), C.c_size_t(100), C.byref(size)) == -1                   # proof buffer too small
    assert L.sp_create_proof(C.c_uint64(key), ptrs, None, C.c_uint32(0), 0, 1, None, ffi.ptr(out), C.c_size_t(2144), C.byref(size)) == -6   # no instances: the copy constraints to the instance column fail
    assert L.sp_create_proof(*args, 0, 1, None, ffi.ptr(out), C.c_size_t(2144), C.byref(size)) == 0 and size.value == 2144


def test_native_proofs_from_several_threads(setup):
    """four host threads, each on its own stream, proving concurrently through sp_create_proof (per-thread prover
    sessions over per-call library lanes): every proof verifies"""
    import threading
    import torch
    from oracle import summa_verifier as SV
    s = setup
    P = s["prover"]
    advice = [s["dev"](c) for c in s["asg"]["advice"]]
    inst = s["asg"]["instances"]
    P.create_proof_native(s["params"], s["pk"], advice, inst)
    torch.cuda.synchronize()
    proofs, errors = [], []

    def worker(flavour):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for _ in range(5):
                    proofs.append((flavour, P.create_proof_native(s["params"], s["pk"], advice, inst, flavour)))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
    threads = [threading.Thread(target=worker, args=(f,)) for f in ("evm", "blake2b", "evm", "blake2b")]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(proofs) == 20 and len({p for _, p in proofs}) == 20
    assert all(SV.verify(p, inst, s["vk"], flavour=f) for f, p in proofs)
