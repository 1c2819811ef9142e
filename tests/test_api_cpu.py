"""CPU checks of the reference's L3 API mirror (circuits_halo2_amd/api.py), of the two transcript flavours, and of the
product-side verifier (circuits_halo2_amd/verifier.py + the library's host-side pairing), pinned on the reference's own
artefacts: its shipped proof K6 is accepted by the PRODUCT verifier (with the oracle's MSM injected for the one
multi-scalar multiplication that needs the GPU), RFC 7693's vector pins the oracle's Blake2b, and the oracle's Blake2b /
Keccak transcripts agree with the product's writers challenge by challenge."""
import ctypes as C
import hashlib
import json
import os
import random

import numpy as np
import pytest

from conftest import GOLDEN, fr_np

from oracle import pairing as PA
from oracle import pyref as PR
from oracle import summa_verifier as SV

H = lambda s: int(s, 16)


# ------------------------------------------------------------------ Blake2b and the transcripts
def test_oracle_blake2b_rfc7693_vector_and_hashlib():
    want = ("ba80a53f981c4d0d6a2797b69f12f6e94c212f14685ac4b74b12bb6fdbffa2d1"
            "7d87c5392aab792dc252d5de4533cc9518d38aa8dbf1925ab92386edd4009923")
    assert PR.blake2b(b"abc").hex() == want                                 # RFC 7693 appendix A
    assert PR.blake2b(b"", 64).hex().startswith("786a02f742015903c6c6fd852552d272")
    rng = random.Random(5)
    for n in (0, 1, 63, 64, 127, 128, 129, 255, 256, 257, 1000):
        d = bytes(rng.randrange(256) for _ in range(n))
        assert PR.blake2b(d, 64, b"Halo2-Transcript") == hashlib.blake2b(d, digest_size=64, person=b"Halo2-Transcript").digest()
        assert PR.blake2b(d, 32, b"p", b"s", b"key") == hashlib.blake2b(d, digest_size=32, person=b"p", salt=b"s", key=b"key").digest()


def _random_points(rng, count):
    return [PR.g1_mul(PR.G1_GEN, rng.randrange(1, PR.R)) for _ in range(count)]


@pytest.mark.parametrize("flavour", ["evm", "blake2b"])
def test_product_transcript_writers_match_the_oracle_readers(flavour):
    """same absorb / squeeze sequence (the shape of a proof: digest, instances, point groups, consecutive challenges)
    through the product's writer and the oracle's restated reader: equal challenges; the proof stream parses back"""
    from circuits_halo2_amd import prover as P
    rng = random.Random(11)
    digest = rng.randrange(PR.R)
    w = (P.EvmTranscriptWriter if flavour == "evm" else P.Blake2bWrite)()
    w.common_scalar(digest)
    o = (SV.EvmTranscript if flavour == "evm" else SV.Blake2bTranscript)(digest)
    pts, scalars = _random_points(rng, 5), [rng.randrange(PR.R) for _ in range(4)]
    for v in scalars[:2]:
        w.common_scalar(v); o.absorb_scalar(v)
    for p in pts[:3]:
        w.write_point(p); o.absorb_point(p)
    assert w.squeeze_challenge() == o.squeeze()
    for p in pts[3:]:
        w.write_point(p); o.absorb_point(p)
    assert w.squeeze_challenge() == o.squeeze()
    assert w.squeeze_challenge() == o.squeeze_again()            # beta, gamma: two challenges without new input
    for v in scalars[2:]:
        w.write_scalar(v); o.absorb_scalar(v)
    c1, c2 = w.squeeze_challenge(), w.squeeze_challenge()
    assert (c1, c2) == (o.squeeze(), o.squeeze_again()) and c1 != c2
    proof = w.finalize()
    if flavour == "evm":
        assert len(proof) == 64 * 5 + 32 * 2
        assert proof[:32] == pts[0][0].to_bytes(32, "big") and proof[-32:] == scalars[3].to_bytes(32, "big")
    else:
        assert len(proof) == 32 * 5 + 32 * 2
        assert [SV.decompress_g1(proof[32 * i:32 * i + 32]) for i in range(5)] == pts
        assert proof[-32:] == scalars[3].to_bytes(32, "little")
    with pytest.raises(ValueError):
        w.write_point(None)                                       # "cannot write points at infinity to the transcript"


def test_blake2b_challenge_is_the_wide_reduction_of_the_stream_digest():
    from circuits_halo2_amd import prover as P
    w = P.Blake2bWrite()
    w.common_scalar(7)
    p = PR.g1_mul(PR.G1_GEN, 12345)
    w.write_point(p)
    stream = b"\x02" + (7).to_bytes(32, "little") + b"\x01" + p[0].to_bytes(32, "little") + p[1].to_bytes(32, "little") + b"\x00"
    want = int.from_bytes(hashlib.blake2b(stream, digest_size=64, person=b"Halo2-Transcript").digest(), "little") % PR.R
    assert w.squeeze_challenge() == want


def test_point_compression_round_trip_and_rejections():
    from circuits_halo2_amd import prover as P, verifier as V
    rng = random.Random(3)
    for p in _random_points(rng, 20) + [PR.G1_GEN]:
        enc = P.compress_g1(p)
        assert len(enc) == 32 and enc[31] & 0x80 == 0 and (enc[31] >> 6) & 1 == p[1] & 1
        assert V.decompress_g1(enc) == p == SV.decompress_g1(enc)
        flipped = bytearray(enc); flipped[31] ^= 0x40
        assert V.decompress_g1(bytes(flipped)) == (p[0], PR.Q - p[1])
    assert P.compress_g1(None) == bytes(31) + b"\x80"
    off = next(x for x in range(1, 50) if pow(x ** 3 + 3, (PR.Q - 1) // 2, PR.Q) != 1)           # no y with y^2 = x^3 + 3
    for bad in (bytes(31) + b"\x80", (PR.Q).to_bytes(32, "little"), off.to_bytes(32, "little")):   # infinity, x >= q, off the curve
        for fn in (V.decompress_g1, SV.decompress_g1):
            with pytest.raises(ValueError):
                fn(bad)


# ------------------------------------------------------------------ pairing (host code of the library)
def _pairing_check(pairs):
    from circuits_halo2_amd import ffi
    g1 = np.frombuffer(b"".join(PR.g1_to_bytes(p) for p, _ in pairs), dtype=np.uint8).copy()
    g2 = np.frombuffer(b"".join(PR.g2_to_bytes(q) for _, q in pairs), dtype=np.uint8).copy()
    ok = C.c_int(-1)
    rc = ffi.lib().sg_pairing_check(ffi.ptr(g1), ffi.ptr(g2), C.c_size_t(len(pairs)), C.byref(ok))
    return rc, ok.value


def test_library_pairing_check_agrees_with_the_oracle():
    g2 = PR.G2_GENERATOR
    rng = random.Random(9)
    for _ in range(3):
        a, b = rng.randrange(1, PR.R), rng.randrange(1, PR.R)
        pa, qb = PR.g1_mul(PR.G1_GEN, a), PR.g2_mul(g2, b)
        pab = PR.g1_mul(PR.G1_GEN, a * b % PR.R)
        good, bad = [(pa, qb), (PR.g1_neg(pab), g2)], [(pa, qb), (PR.g1_neg(pa), g2)]
        assert _pairing_check(good) == (0, 1) and _pairing_check(bad) == (0, 0)
    assert PA.pairing_check(good) and not PA.pairing_check(bad)              # the oracle on the last case
    # three pairs: e(aG, H) e(bG, H) e(-(a+b)G, H) = 1
    pb, pc = PR.g1_mul(PR.G1_GEN, b), PR.g1_neg(PR.g1_mul(PR.G1_GEN, (a + b) % PR.R))
    assert _pairing_check([(pa, g2), (pb, g2), (pc, g2)]) == (0, 1)
    assert _pairing_check([(None, g2), (pa, None)]) == (0, 1)               # identities pair to 1
    assert _pairing_check([(pa, g2)]) == (0, 0)
    assert _pairing_check([]) == (0, 1)
    # points off the curve are refused, not paired
    rc, _ = _pairing_check([((1, 3), g2)])
    assert rc < 0
    rc, _ = _pairing_check([(pa, ((1, 2), (3, 4)))])
    assert rc < 0


# ------------------------------------------------------------------ the product verifier on the reference's shipped proof
def _k6():
    from circuits_halo2_amd import api, params as PM
    tr = json.load(open(os.path.join(GOLDEN, "k6_verifier_trace.json")))["vk"]
    comm = [(H(a), H(b)) for a, b in tr["commitments"]]
    vk = api.VerifyingKey(11, 2, comm[:11], comm[11:], H(tr["vk_digest"]))
    cd = json.load(open(os.path.join(GOLDEN, "k6_inclusion_proof_solidity_calldata.json")))
    params = PM.ParamsKZG.read(open(os.path.join(GOLDEN, "hermez-raw-11"), "rb"))
    return params, vk, bytes.fromhex(cd["proof"][2:]), [H(x) for x in cd["public_inputs"]]


@pytest.fixture()
def oracle_msm(monkeypatch):
    """the verifier's one multi-scalar multiplication runs on the GPU in the product; on the CPU box the oracle's MSM
    stands in for it (the pairing, transcript and scalar side under test are the product's own)"""
    from circuits_halo2_amd import verifier as V
    from oracle import oracle as O
    monkeypatch.setattr(V, "best_multiexp", lambda s, b: O.best_multiexp(np.ascontiguousarray(s), np.ascontiguousarray(b), 2))
    monkeypatch.setattr(V, "DRIVER", "python")     # the twin of the compiled verifier (which runs its MSM on the GPU: tests/test_gpu_api.py)
    return V


def test_product_verifier_accepts_the_reference_shipped_proof(oracle_msm):
    V = oracle_msm
    params, vk, proof, inst = _k6()
    assert V.verify_proof(params, vk, proof, inst, "evm")
    # the SRS file's g2 / s_g2 are what the contract pairs against: same verdict as the restated verifier
    for off in (0x10, 0x150, 0x390, 0x700, 0x7f0, 0x850):
        p = bytearray(proof)
        p[off] ^= 1
        assert not V.verify_proof(params, vk, bytes(p), inst, "evm"), hex(off)
    assert not V.verify_proof(params, vk, proof, [inst[0], inst[1], inst[2] + 1, inst[3]], "evm")
    assert not V.verify_proof(params, vk, proof[:-1], inst, "evm")
    assert not V.verify_proof(params, vk, proof + b"\0", inst, "evm")
    assert not V.verify_proof(params, vk, proof, inst, "blake2b")            # wrong flavour: does not even parse
    assert not V.verify_proof(params, vk, proof, [PR.R] + inst[1:], "evm")   # unreduced public input
    wrong_key = type(vk)(vk.k, vk.n_currencies, vk.fixed_comms, vk.permutation_comms, vk.transcript_repr + 1)
    assert not V.verify_proof(params, wrong_key, proof, inst, "evm")


# ------------------------------------------------------------------ API host logic
def test_generate_setup_artifacts_k_too_large():
    from circuits_halo2_amd import api
    circuit = api.MstInclusionCircuit.init_empty(4, 2, 8)
    with pytest.raises(ValueError, match="k is too large for the given params"):       # utils.rs:58-60
        api.generate_setup_artifacts(12, os.path.join(GOLDEN, "hermez-raw-11"), circuit)
    with pytest.raises(OSError):                                                        # "couldn't load params"
        api.generate_setup_artifacts(11, os.path.join(GOLDEN, "no-such-file"), circuit)


def test_mst_inclusion_circuit_init_shapes():
    from circuits_halo2_amd import api
    c = api.MstInclusionCircuit.init_empty(4, 2, 8)
    assert c.num_instances() == 4 and len(c.path_indices) == 4 and len(c.sibling_middle_node_hash_preimages) == 4
    assert c.entry == (0, [0, 0]) and c.root == (0, [0, 0])


def test_empty_and_real_circuits_share_fixed_columns_and_permutation(kat):
    """keygen from `init_empty` and proving from `init` must agree on everything that enters the keys"""
    from circuits_halo2_amd import api
    empty = api.MstInclusionCircuit.init_empty(4, 2, 8).synthesize(11)
    rng = random.Random(1)
    real = api.MstInclusionCircuit(4, 2, 8, (rng.randrange(PR.R), [5, 7]), [1, 0, 1, 1], [rng.randrange(PR.R), 11, 13],
                                   [[rng.randrange(1 << 40), rng.randrange(1 << 40), rng.randrange(PR.R), rng.randrange(PR.R)] for _ in range(3)],
                                   (0, [0, 0])).synthesize(11)
    assert empty["sigma"] == real["sigma"]
    # fixed columns are equal too (round constants, table, selectors, the constants 0 and L * 2^64)
    assert empty["fixed"] == real["fixed"]
    assert len(real["instances"]) == 4


def test_vk_transcript_repr_is_the_digest_baked_into_the_reference_verifier(kat):
    """halo2's `VerifyingKey::transcript_repr` (Blake2b of the pinned key's Debug text) rebuilt from a replay of
    `MstInclusionConfig::configure`: for k = 11 and the reference's commitments it must be the `vk_digest` constant of
    contracts/src/InclusionVerifier.sol:217 (kat.json)."""
    from circuits_halo2_amd import prover as P, vk_repr as V
    fixed = [(H(a), H(b)) for a, b in kat["fixed_comms"]]
    perm = [(H(a), H(b)) for a, b in kat["permutation_comms"]]
    assert P.verifying_key_digest(H(kat["k"]), 2, fixed, perm) == H(kat["vk_digest"])
    cs, text = V.constraint_system_text(2)
    # the query lists are the evaluation order of the reference's verifier (InclusionVerifier.sol: a_0, a_1, a_0 next,
    # a_1 next, a_2, a_1 prev, a_0 prev; f_2, f_3, f_0, f_1, f_4 .. f_10)
    assert cs.queries["Advice"] == [(0, 0), (1, 0), (0, 1), (1, 1), (2, 0), (1, -1), (0, -1)]
    assert [c for c, _ in cs.queries["Fixed"]] == [2, 3, 0, 1, 4, 5, 6, 7, 8, 9, 10]
    assert cs.degree() == 6 and len(cs.gates) == 19 and cs.num_fixed == 11
    assert text.count("Scaled(") == 2 * 20 and "Selector" not in text   # per chip: 4 (full rounds) + 5 * 2 + 2 (s-box of the mixed state, written out five times) + 4
    # any change of the text changes the digest: one more currency = one more sum gate
    assert P.verifying_key_digest(H(kat["k"]), 3, fixed, perm) != H(kat["vk_digest"])
    assert len(V.constraint_system_text(3)[0].gates) == 20


def test_selector_compression_replay_gives_the_floor_plans_selector_columns():
    """`compress_selectors` replayed on the activations of the real floor plan assigns the same fixed columns and
    values as the floor plan's f5 .. f10 (lookup selector alone; swap / sum / pad / pad combined as 1 .. 4; the four
    degree-6 Poseidon selectors alone)"""
    from circuits_halo2_amd import api, vk_repr as V
    fixed = api.MstInclusionCircuit.init_empty(4, 2, 8).synthesize(11)["fixed"]
    rows = lambda col, value: {r for r, v in enumerate(fixed[col]) if v == value}
    # selector order of configure: swap, sum, lookup (complex), chip 1 full / partial / pad, chip 2 full / partial / pad
    act = [rows(6, 1), rows(6, 2), rows(5, 1), rows(7, 1), rows(8, 1), rows(6, 3), rows(9, 1), rows(10, 1), rows(6, 4)]
    assert all(act)
    cs = V.configure(2)
    cols = cs.compress_selectors(act)
    assert sorted(cols) == [5, 6, 7, 8, 9, 10]
    for c, values in cols.items():
        assert values == {r: v for r, v in enumerate(fixed[c]) if v}
    assert cs.pinned_text() == V.constraint_system_text(2)[1]


def test_init_asserts_the_reference_lengths():
    from circuits_halo2_amd import api
    z32 = np.zeros(32, dtype=np.uint8)
    mp = {"entry": ("alice", [1, 2]), "path_indices": [0, 1, 0], "root": (z32, np.zeros(64, dtype=np.uint8)),
          "sibling_leaf_node_hash_preimage": np.zeros(96, dtype=np.uint8),
          "sibling_middle_node_hash_preimages": [np.zeros(128, dtype=np.uint8)] * 2}
    c = api.MstInclusionCircuit.init(mp, 3)
    assert c.shape() == (3, 2, 8) and c.entry[0] == int.from_bytes(PR.keccak256(b"alice"), "big") % PR.R
    with pytest.raises(AssertionError):
        api.MstInclusionCircuit.init(mp, 4)                         # assert_eq!(path_indices.len(), LEVELS)
    mp2 = dict(mp, sibling_middle_node_hash_preimages=[np.zeros(128, dtype=np.uint8)] * 3)
    with pytest.raises(AssertionError):
        api.MstInclusionCircuit.init(mp2, 3)                        # assert_eq!(preimages.len(), LEVELS - 1)


def test_field_element_to_solidity_calldata():
    from circuits_halo2_amd import api
    assert api.field_element_to_solidity_calldata(556862) == 556862
    assert api.field_element_to_solidity_calldata(PR.R + 5) == 5


# ------------------------------------------------------------------ the floor plan as a device program
def _tree_value(symbol, idx, users, node_hash, node_bal, nc):
    """value of a witness-program symbol for leaf `idx` (integers): users[i], node_hash[level][i], node_bal[level][i][c]"""
    kind, level, mode, lane = symbol & 15, (symbol >> 4) & 63, (symbol >> 10) & 7, (symbol >> 13) & 127
    at = idx >> level
    if kind == 0:
        return users[idx ^ (mode & 1)]
    if kind == 3:
        return at & 1
    node, lv = at, level
    if mode == 1:
        node = at ^ 1
    elif mode == 2:
        node, lv = 2 * (at ^ 1) + (lane & 1), level - 1
    elif mode == 3:
        node = (at & ~1) + (lane & 1)
    return node_hash[lv][node] if kind == 1 else node_bal[lv][node][lane]


@pytest.mark.parametrize("levels,nc,k,idx", [(4, 2, 11, 5), (3, 1, 10, 6), (5, 3, 12, 17)])
def test_witness_program_covers_the_reference_assignment_cell_by_cell(levels, nc, k, idx):
    """mst_inclusion.witness_program (what the device kernel executes) against mst_inclusion.reference_assignment (the
    replay of the reference's synthesize with integers, itself pinned by the reproduced verifying key): every cell the
    program writes holds the value the assignment has there, the sponge regions start from the sponge's initial state,
    absorb the right words and end in the tree's own nodes -- and the cells the program covers are ALL the non-zero
    cells of the assignment"""
    from circuits_halo2_amd import mst_inclusion as M
    rng = random.Random(levels * 100 + nc)
    size = 1 << levels
    users = [rng.randrange(PR.R) for _ in range(size)]
    bals = [[rng.randrange(1 << 40) for _ in range(nc)] for _ in range(size)]
    node_hash = [[PR.poseidon_hash([users[i]] + bals[i]) for i in range(size)]]
    node_bal = [bals]
    for level in range(1, levels + 1):
        hs, bs = [], []
        for p in range(size >> level):
            b = [(node_bal[-1][2 * p][c] + node_bal[-1][2 * p + 1][c]) % PR.R for c in range(nc)]
            bs.append(b)
            hs.append(PR.poseidon_hash(b + [node_hash[-1][2 * p], node_hash[-1][2 * p + 1]]))
        node_hash.append(hs)
        node_bal.append(bs)
    bits = [(idx >> l) & 1 for l in range(levels)]
    sib = idx ^ 1
    pre_leaf = [users[sib]] + bals[sib]
    pre_mid = []
    for level in range(1, levels):
        s = (idx >> level) ^ 1
        pre_mid.append(node_bal[level][s] + [node_hash[level - 1][2 * s], node_hash[level - 1][2 * s + 1]])
    asg = M.reference_assignment(k, users[idx], bals[idx], bits, pre_leaf, pre_mid)
    adv = asg["advice"]
    prog, n_items, n_absorbs, inst_syms, rows_used = M.witness_program(k, levels, nc, 8)
    assert rows_used == asg["rows_used"]
    items = prog[:5 * n_items].reshape(-1, 5).tolist()
    absorbs = prog[5 * n_items:].reshape(-1, 3).tolist()
    assert len(absorbs) == n_absorbs
    val = lambda s: _tree_value(s, idx, users, node_hash, node_bal, nc) % PR.R
    assert [val(s) for s in inst_syms] == asg["instances"] == [node_hash[0][idx], node_hash[levels][0]] + node_bal[levels][0]
    covered = set()
    seen_hash = False
    for kind, col, row, s, extra in items:
        if kind == 0:
            assert not seen_hash or True
            assert adv[col][row] == val(s), (col, row)
            covered.add((col, row))
        elif kind == 1:
            for b in range(extra):
                assert adv[0][row + b] == val(s) >> (8 * b)
                covered.add((0, row + b))
            assert adv[0][row + extra] == 0
        else:
            first, count = extra & 0xFFFFF, (extra >> 20) & 255
            assert (adv[0][row], adv[1][row]) == (0, (count << 64) % PR.R)
            covered.add((1, row))
            state0 = 0
            for add_row, perm_row, s in absorbs[first:first + count]:
                assert adv[0][add_row] == state0 and adv[0][add_row + 1] == val(s)
                assert adv[0][add_row + 2] == (state0 + val(s)) % PR.R == adv[0][perm_row]
                for r in range(3):
                    covered.update({(0, add_row + r), (1, add_row + r)})
                for r in range(37):
                    covered.update({(0, perm_row + r), (1, perm_row + r)})
                for r in range(4, 32):
                    covered.add((2, perm_row + r))
                state0 = adv[0][perm_row + 36]
            # the digest is a node of the tree: the leaf, a sibling, or the next node of the path
            assert state0 in {h for lv in node_hash for h in lv}
    nonzero = {(c, r) for c in range(3) for r in range(1 << k) if adv[c][r]}
    assert nonzero <= covered
    assert [it[0] for it in items] == sorted((it[0] for it in items), reverse=True)      # sponges first


def test_library_keccak_and_the_cross_checked_final_exponentiation():
    """host utilities of the library: sg_keccak256 against the Python twin and the oracle's Keccak (padding edge cases
    around the 136-byte rate); sg_pairing_check_slow = the same verdicts with the final exponentiation's addition chain
    compared against the plain exponentiation by (q^12 - 1) / r"""
    from circuits_halo2_amd import merkle_sum_tree as T
    rng = random.Random(17)
    for n in (0, 1, 31, 135, 136, 137, 271, 272, 273, 1000, 4321):
        d = bytes(rng.randrange(256) for _ in range(n))
        assert T.keccak256(d) == T.keccak256_python(d) == PR.keccak256(d), n
    assert T.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    from circuits_halo2_amd import ffi
    g2 = PR.G2_GENERATOR
    a, b = rng.randrange(1, PR.R), rng.randrange(1, PR.R)
    pa, qb, pab = PR.g1_mul(PR.G1_GEN, a), PR.g2_mul(g2, b), PR.g1_mul(PR.G1_GEN, a * b % PR.R)
    for pairs, want in (([(pa, qb), (PR.g1_neg(pab), g2)], 1), ([(pa, qb), (PR.g1_neg(pa), g2)], 0)):
        g1b = np.frombuffer(b"".join(PR.g1_to_bytes(p) for p, _ in pairs), dtype=np.uint8).copy()
        g2b = np.frombuffer(b"".join(PR.g2_to_bytes(q) for _, q in pairs), dtype=np.uint8).copy()
        ok = C.c_int(-1)
        assert ffi.lib().sg_pairing_check_slow(ffi.ptr(g1b), ffi.ptr(g2b), C.c_size_t(2), C.byref(ok)) == 0 and ok.value == want


def test_operation_helpers_round_trip():
    """utils/operation_helpers.rs:5-17: `big_uint_to_fp` / `fp_to_big_uint` are inverse on [0, r), reduce beyond it as
    `Fp::from_str_vartime` does, and write halo2curves' memory form (Montgomery, little-endian)"""
    from circuits_halo2_amd.merkle_sum_tree import big_uint_to_fp, fp_to_big_uint
    for v in (0, 1, 3, 18446744073709551616, PR.R - 1):
        b = big_uint_to_fp(v)
        assert len(b) == 32 and int.from_bytes(b, "little") == (v << 256) % PR.R and fp_to_big_uint(b) == v
    assert fp_to_big_uint(big_uint_to_fp(PR.R + 5)) == 5
    with pytest.raises(ValueError):
        big_uint_to_fp(-1)
    with pytest.raises(ValueError):
        fp_to_big_uint((PR.R).to_bytes(32, "little"))
