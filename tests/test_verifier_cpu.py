"""The restated verifier (oracle/summa_verifier.py) against the reference's own artefacts: its shipped proof K6
(tests/golden/k6_inclusion_proof_solidity_calldata.json, a data file of the reference:
zk_prover/examples/inclusion_proof_solidity_calldata.json) and the values its generated verifier computes on that
proof (tests/golden/k6_verifier_trace.json, produced by oracle/yul_verifier_run.py from
contracts/src/InclusionVerifier.sol).  Also: the product-side constraint system (circuits_halo2_amd/mst_inclusion.py)
yields the same gate values as the restatement."""
import json
import os

import pytest

from oracle import pairing as PA
from oracle import pyref as PR
from oracle import summa_verifier as SV

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
H = lambda s: int(s, 16)


def load_k6():
    tr = json.load(open(os.path.join(GOLD, "k6_verifier_trace.json")))
    v = tr["vk"]
    comm = [(H(a), H(b)) for a, b in v["commitments"]]
    vk = {"vk_digest": H(v["vk_digest"]), "fixed_comms": comm[:11], "permutation_comms": comm[11:],
          "g2": ((H(v["g2_x_2"]), H(v["g2_x_1"])), (H(v["g2_y_2"]), H(v["g2_y_1"]))),
          "neg_s_g2": ((H(v["neg_s_g2_x_2"]), H(v["neg_s_g2_x_1"])), (H(v["neg_s_g2_y_2"]), H(v["neg_s_g2_y_1"])))}
    cd = json.load(open(os.path.join(GOLD, "k6_inclusion_proof_solidity_calldata.json")))
    return bytes.fromhex(cd["proof"][2:]), [H(x) for x in cd["public_inputs"]], vk, tr


def test_pairing_bilinear_and_nondegenerate():
    g2 = PR.G2_GENERATOR
    assert PA.g2_is_on_curve(g2)
    a, b = 0x1234567, 0x89ABCDEF01
    pa, qb = PR.g1_mul(PR.G1_GEN, a), PR.g2_mul(g2, b)
    pab = PR.g1_mul(PR.G1_GEN, a * b % PR.R)
    assert PA.pairing_check([(pa, qb), (PR.g1_neg(pab), g2)])          # e(aG, bH) = e(abG, H)
    assert not PA.pairing_check([(pa, qb), (PR.g1_neg(pa), g2)])
    assert PA.pairing_check([(None, g2), (pa, None)])                  # identities pair to 1
    assert PA.pairing(g2, PR.G1_GEN) != PA.F12_ONE


def test_vk_constants_match_the_committed_kat():
    """the trace's verifying key equals the constants transcribed earlier from the same contract (kat.json) and
    -[s]_2 is a point of G2"""
    _, _, vk, tr = load_k6()
    kat = json.load(open(os.path.join(GOLD, "kat.json")))
    assert vk["vk_digest"] == H(kat["vk_digest"])
    assert vk["fixed_comms"] == [(H(a), H(b)) for a, b in kat["fixed_comms"]]
    assert vk["permutation_comms"] == [(H(a), H(b)) for a, b in kat["permutation_comms"]]
    assert vk["g2"] == PR.G2_GENERATOR
    assert PA.g2_is_on_curve(vk["neg_s_g2"])


def test_k6_proof_verifies_and_every_intermediate_matches_the_reference():
    proof, inst, vk, tr = load_k6()
    assert tr["trace"]["result"] == 1
    got = {}
    assert SV.verify(proof, inst, vk, got)
    for name, want in tr["trace"].items():
        if name != "result":
            assert got[name] == H(want), name


@pytest.mark.parametrize("where", ["commitment", "evaluation", "opening", "instance", "quotient"])
def test_k6_tampered_proof_is_rejected(where):
    proof, inst, vk, _ = load_k6()
    p = bytearray(proof)
    if where == "commitment":      # a different valid curve point as a_0: the generator
        p[0:64] = (1).to_bytes(32, "big") + (2).to_bytes(32, "big")
    elif where == "evaluation":    # f_6 evaluation + 1
        off = 0x380 + 32 * SV.EVAL_ORDER.index(("f", 6, 0))
        p[off:off + 32] = ((int.from_bytes(p[off:off + 32], "big") + 1) % SV.R).to_bytes(32, "big")
    elif where == "opening":       # W' replaced by W
        p[0x820:0x860] = p[0x7e0:0x820]
    elif where == "quotient":      # swap two quotient pieces
        p[0x240:0x280], p[0x280:0x2c0] = p[0x280:0x2c0], p[0x240:0x280]
    else:
        inst = [inst[0], inst[1], inst[2] + 1, inst[3]]
    assert not SV.verify(bytes(p), inst, vk)
    # malformed inputs are rejected, not raised
    assert not SV.verify(bytes(p[:-1]), inst, vk)
    q = bytearray(proof)
    q[63] ^= 1                     # off the curve
    assert not SV.verify(bytes(q), inst, vk)


def test_transcript_restates_the_evm_rule():
    """challenge = keccak(buffer) mod r, the hash becomes the buffer; a second squeeze hashes hash || 0x01"""
    t = SV.EvmTranscript(5)
    t.absorb_scalar(7)
    h = PR.keccak256((5).to_bytes(32, "big") + (7).to_bytes(32, "big"))
    assert t.squeeze() == int.from_bytes(h, "big") % SV.R
    h2 = PR.keccak256(h + b"\x01")
    assert t.squeeze_again() == int.from_bytes(h2, "big") % SV.R
    t.absorb_point((1, 2))
    assert t.squeeze() == int.from_bytes(PR.keccak256(h2 + (1).to_bytes(32, "big") + (2).to_bytes(32, "big")), "big") % SV.R


def test_poseidon_matrices_in_the_gates_are_the_contracts():
    """the MDS / MDS^-1 entries the restated gates multiply by are the literals of the generated verifier
    (first entries; contracts/src/InclusionVerifier.sol:508, 579)"""
    _, mds, mds_inv = SV.poseidon_generate()
    assert mds[0][0] == 0x066f6f85d6f68a85ec10345351a23a3aaf07f38af8c952a7bceca70bd2af7ad5
    assert mds_inv[0][0] == 0x13abec390ada7f4370819ab1c7846f210554569d9b29d1ea8dbebd0fa8c53e66


def test_product_constraint_system_equals_the_restatement():
    """circuits_halo2_amd.mst_inclusion (what the GPU evaluates) and oracle.summa_verifier.gate_values agree on
    random column values, gate by gate, as do the lookup expressions; degrees fit the extended domain"""
    from circuits_halo2_amd import arithmetic as A
    from circuits_halo2_amd import mst_inclusion as M
    vals = {}
    rnd = iter(PR.random_fr(0xC0FFEE, 64))
    kinds = {A.ADVICE: "a", A.FIXED: "f"}

    def q_oracle(kind, c, rot):
        if (kind, c, rot) not in vals:
            vals[(kind, c, rot)] = next(rnd)
        return vals[(kind, c, rot)]
    want = SV.gate_values(q_oracle)
    got = [e.evaluate(lambda kind, c, rot: q_oracle(kinds[kind], c, rot)) for e in M.gates()]
    assert got == want and len(got) == 19
    inp, tab = M.lookup_expressions()
    q = lambda kind, c, rot: q_oracle(kinds[kind], c, rot)
    assert (inp.evaluate(q), tab.evaluate(q)) == SV.lookup_input_table(q_oracle)
    assert max(e.degree() for e in M.gates()) == M.DEGREE
    assert [("f" if k == A.FIXED else "a" if k == A.ADVICE else "i", c) for k, c in M.PERMUTATION_COLUMNS] == SV.PERMUTATION_COLUMNS
    assert sorted(M.gate_graph().rotations) == [-1, 0, 1]


@pytest.mark.parametrize("nc", [1, 2, 3])
def test_gate_program_value_is_the_plain_fold_of_the_gate_polynomials(nc):
    """`mst_inclusion.gate_graph` is a factored program (both Poseidon chips at once, packed selectors by differences, powers of
    y as challenges); its VALUE must be halo2's  prev * y^Ng + sum_i G_i y^(Ng - 1 - i)  over `gates()`, the polynomials the
    restated verifier folds.  Run by the oracle's graph interpreter on every row of a 2^4-row domain with random columns."""
    from circuits_halo2_amd import arithmetic as A
    from circuits_halo2_amd import mst_inclusion as M
    from oracle import oracle as O
    from oracle import pyref as PR
    k, n = 4, 16
    g = M.gate_graph(nc)
    groups = M.gate_challenge_exponents(nc)
    assert len(groups) == 12 and [17 + nc] in groups and [7] in groups and sorted(g.rotations) == [-1, 0, 1]
    n_ops, n_slots = A.gates_program_info(g, M.NUM_FIXED, M.NUM_ADVICE, 1, len(groups))
    assert n_slots <= 8 and n_ops < 140            # what the kernel's occupancy hangs on (csrc/gates.hip)
    fixed = [O.random_fr(6100 + i, n) for i in range(M.NUM_FIXED)]
    advice = [O.random_fr(6200 + i, n) for i in range(M.NUM_ADVICE)]
    beta, gamma, theta, y = (O.random_fr(6300 + i, 1) for i in range(4))
    prev = O.random_fr(6310, n)
    yi = PR.fr_from_bytes(bytes(y))
    chal = M.gate_challenges(yi, nc)
    got = O.quotient_gates(prev, g.as_dict(), fixed, advice, [], chal, beta, gamma, theta, y, k, k)
    cell = lambda arr, row: PR.fr_from_bytes(bytes(arr[32 * row:32 * row + 32]))
    for row in range(n):
        q = lambda kind, c, rot: cell((fixed if kind == A.FIXED else advice)[c], (row + rot) % n)
        acc = cell(prev, row)
        for e in M.gates(nc):
            acc = (acc * yi + e.evaluate(q)) % PR.R
        assert cell(got, row) == acc, row


@pytest.mark.skipif(not os.path.exists("/root/reference/contracts/src/InclusionVerifier.sol"),
                    reason="needs the reference checkout (authoring container only)")
def test_trace_fixture_is_what_the_reference_verifier_computes():
    """re-run the reference's generated verifier (read as text, interpreted by oracle/yul_verifier_run.py) on its
    shipped proof: it accepts, and the committed fixture is its output"""
    from oracle import yul_verifier_run as Y
    out = Y.run()
    assert out["trace"]["result"] == 1
    assert out == json.load(open(os.path.join(GOLD, "k6_verifier_trace.json")))
    cd = json.load(open(Y.CALLDATA))
    assert cd == json.load(open(os.path.join(GOLD, "k6_inclusion_proof_solidity_calldata.json")))


@pytest.mark.parametrize("small", [True, False])
def test_permute_expression_pair_host_logic(small):
    """the prover's lookup permutation (halo2's permute_expression_pair): A' sorted, every row has A'[i] == S'[i]
    or A'[i] == A'[i-1], both are permutations of their inputs; a value outside the table is an error"""
    import numpy as np
    from circuits_halo2_amd.prover import permute_expression_pair
    rng = np.random.default_rng(3)
    n = 777
    table = np.zeros((n, 4), dtype=np.uint64)
    table[:256, 0] = np.arange(256)
    if not small:
        table[:, 2] = 9                      # multi-limb values take the general path
    inp = table[rng.integers(0, 256, n)]
    a, s = permute_expression_pair(inp, table)
    assert all((a[i] == s[i]).all() or (a[i] == a[i - 1]).all() for i in range(n))
    srt = lambda x: x[np.lexsort((x[:, 0], x[:, 1], x[:, 2], x[:, 3]))]
    assert (srt(s) == srt(table)).all() and (a == srt(inp)).all()
    bad = inp.copy()
    bad[5, 0] = 999
    with pytest.raises(ValueError):
        permute_expression_pair(bad, table)


def test_contract_pairing_constants_are_the_srs_files():
    """the verifier contract's G2 constants are the last 256 bytes of the reference's SRS file
    (backend/ptau/hermez-raw-11: g2, s_g2): generator, and -[s]_2 = the negated s_g2"""
    _, _, vk, _ = load_k6()
    raw = open(os.path.join(GOLD, "hermez-raw-11"), "rb").read()
    tail = raw[4 + 128 * 2048:]
    f2 = lambda b: (PR.fq_from_bytes(b[:32]), PR.fq_from_bytes(b[32:64]))
    g2, s_g2 = (f2(tail[:64]), f2(tail[64:128])), (f2(tail[128:192]), f2(tail[192:256]))
    assert vk["g2"] == g2
    assert vk["neg_s_g2"] == (s_g2[0], ((-s_g2[1][0]) % PR.Q, (-s_g2[1][1]) % PR.Q))


def test_chacha20_block_matches_rfc8439():
    """the twin of sg_fr_random_dev's generator: RFC 8439 section 2.3.2 test vector; field elements are < r"""
    key, nonce = bytes(range(32)), bytes.fromhex("000000090000004a00000000")
    assert PR.chacha20_block(key, 1, nonce).hex() == (
        "10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e"
        "d2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e")
    vals = PR.chacha_field_elements(key, 7, 64)
    assert all(v < PR.R for v in vals) and len(set(vals)) == 64
    assert vals != PR.chacha_field_elements(key, 8, 64)


@pytest.mark.parametrize("index", [0, 5, 15])
def test_inclusion_assignment_reproduces_the_reference_public_inputs(index):
    """mst_inclusion.assign_inclusion on the reference's csv/entry_16.csv (tree from the oracle's Poseidon): the
    witness satisfies every gate and lookup of the restated constraint system, its copy constraints hold, and the
    public inputs are the leaf hash, the root hash and the root balances -- for user 0 the reference's own expected
    values (K5: zk_prover/src/circuits/tests.rs:341,346)"""
    import csv
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import mst_assignment as MA
    from circuits_halo2_amd import mst_inclusion as M
    rows = list(csv.reader(open(os.path.join(GOLD, "entry_16.csv"))))[1:]
    entries = [PR.mst_entry(r[0], [int(r[1]), int(r[2])]) for r in rows]
    root, levels = PR.mst_build(entries)
    siblings, bits, i = [], [], index
    for level in range(4):
        siblings.append(levels[level][i ^ 1])
        bits.append(i & 1)
        i >>= 1
    asg = M.assign_inclusion(11, entries[index][0], entries[index][1], siblings, bits)
    assert MA.check_gates(asg, 11)
    assert asg["instances"] == [levels[0][index][0], root[0]] + root[1]
    if index == 0:
        k5 = json.load(open(os.path.join(GOLD, "kat.json")))["k5"]
        assert asg["instances"] == [int(k5["leaf0"], 16), int(k5["root"], 16)] + k5["root_balances"]
    assert asg["rows_used"] < asg["usable_rows"] and asg["copies"] > 50
    with pytest.raises(ValueError):
        M.assign_inclusion(11, entries[index][0], [1 << 64, 5], siblings, bits)     # a balance beyond N_BYTES = 8


def _entry16_reference_inputs(index=0):
    import csv
    rows = list(csv.reader(open(os.path.join(GOLD, "entry_16.csv"))))[1:]
    entries = [PR.mst_entry(r[0], [int(r[1]), int(r[2])]) for r in rows]
    root, levels = PR.mst_build(entries)
    bits, pre_mid, i = [], [], index
    for level in range(4):
        bits.append(i & 1)
        s = i ^ 1
        if level:
            pre_mid.append(levels[level][s][1] + [levels[level - 1][2 * s][0], levels[level - 1][2 * s + 1][0]])
        i >>= 1
    sib = entries[index ^ 1]
    return entries[index][0], entries[index][1], bits, [sib[0]] + sib[1], pre_mid, root, levels


@pytest.mark.parametrize("witness", ["entry_16.csv user 0", "empty (keygen view)"])
def test_reference_verifying_key_is_reproduced(witness):
    """KEYGEN LEVEL, bit-exact: replaying the reference circuit's synthesize over halo2's floor planner and permutation
    assembly (circuits_halo2_amd.mst_inclusion.reference_assignment) yields fixed and permutation columns whose KZG
    commitments under the reference's SRS are the 11 fixed_comms and 6 permutation_comms of the reference's verifying
    key (contracts/src/InclusionVerifier.sol:238-271; SURVEY.md "pinned bit-exactly but needs a layout restatement").
    The key does not depend on the witness; the real witness also satisfies every gate and gives the K5 public inputs."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import mst_assignment as MA
    from circuits_halo2_amd import mst_inclusion as M
    from oracle import oracle as O
    if witness.startswith("entry"):
        user, bal, bits, pre_leaf, pre_mid, root, levels = _entry16_reference_inputs(0)
    else:
        user, bal, bits, pre_leaf, pre_mid = 0, [0, 0], [0] * 4, [0, 0, 0], [[0, 0, 0, 0]] * 3
    asg = M.reference_assignment(11, user, bal, bits, pre_leaf, pre_mid)
    assert asg["rows_used"] == 1489
    kat = json.load(open(os.path.join(GOLD, "kat.json")))
    gl = np.frombuffer(PR.parse_srs(open(os.path.join(GOLD, "hermez-raw-11"), "rb").read())["g_lagrange"], dtype=np.uint8).copy()

    def commit(col):
        sc = np.frombuffer(b"".join(PR.fr_to_bytes(v) for v in col), dtype=np.uint8).copy()
        return PR.g1_from_bytes(bytes(O.best_multiexp(sc, gl, O.ncpu())))
    for j in range(11):
        assert commit(asg["fixed"][j]) == (H(kat["fixed_comms"][j][0]), H(kat["fixed_comms"][j][1])), f"fixed_comms[{j}]"
    for j in range(6):
        assert commit(asg["sigma"][j]) == (H(kat["permutation_comms"][j][0]), H(kat["permutation_comms"][j][1])), f"permutation_comms[{j}]"
    assert MA.check_gates(asg, 11)
    if witness.startswith("entry"):
        assert asg["instances"] == [H(kat["k5"]["leaf0"]), H(kat["k5"]["root"])] + kat["k5"]["root_balances"]


@pytest.mark.parametrize("fixture", ["gpu_proof_entry16_user0.json", "gpu_proof_entry16_user0_cpp.json"])
def test_gpu_made_proof_under_the_reference_key(fixture):
    """tests/golden/gpu_proof_entry16_user0.json was produced on an MI355X by tests/test_gpu_prover.py::
    test_reference_floor_plan_proof_under_the_reference_verifying_key (reference layout, reference SRS, entry_16.csv
    user 0).  The restated verifier accepts it on the REFERENCE'S verifying key, and -- where the reference checkout
    exists -- so does the reference's own verifier contract (contracts/src/InclusionVerifier.sol::verifyProof, run by
    oracle/yul_verifier_run.py), which rejects it after a flipped byte."""
    proof_k6, _, vk, _ = load_k6()
    d = json.load(open(os.path.join(GOLD, fixture)))      # Python driver / C++ driver (include/summa_prover.hpp)
    proof, inst = bytes.fromhex(d["proof"][2:]), [H(x) for x in d["public_inputs"]]
    kat = json.load(open(os.path.join(GOLD, "kat.json")))["k5"]
    assert inst == [H(kat["leaf0"]), H(kat["root"])] + kat["root_balances"] and proof != proof_k6
    assert SV.verify(proof, inst, vk)
    if os.path.exists("/root/reference/contracts/src/InclusionVerifier.sol"):
        from oracle import yul_verifier_run as Y
        assert Y.run(proof=proof, instances=inst)["trace"]["result"] == 1
        bad = bytearray(proof)
        bad[0x500] ^= 1
        assert Y.run(proof=bytes(bad), instances=inst)["trace"]["result"] == 0
        assert Y.run(proof=proof, instances=inst[:3] + [inst[3] + 1])["trace"]["result"] == 0
