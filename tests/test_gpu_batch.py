"""Multi-GPU / batch paths on the real device (one rank: the GPU box has one GPU): the point-sharded MSM's exchange and
combination steps with the HIP MSM (no injected oracle), the setup export / import that the broadcast carries, and the
proof-level batch driver with several proofs in flight -- BASELINE configs[4] at a size a test can afford."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from circuits_halo2_amd import ffi
    ffi.check(ffi.lib().sg_init(0))


def test_sharded_msm_hip_path_with_an_rccl_group_of_one():
    """`sharded_msm` with the product's own MSM (HIP) and a real RCCL process group (world 1 here): per-shard partials,
    `all_gather_into_tensor` on the device, `combine_partials` through sg_g1_sum_affine -- compared with the oracle"""
    _gpu()
    import torch
    import torch.distributed as dist
    from circuits_halo2_amd.distributed import combine_partials, shard_bounds, sharded_msm
    from circuits_halo2_amd import best_multiexp
    from oracle import oracle as O
    n = 5000
    sc, bases = O.random_fr(52, n), O.fixed_base_mul(O.random_fr(53, n), 4)
    want = O.best_multiexp(sc, bases, 4)
    d_sc, d_b = torch.from_numpy(sc).cuda(), torch.from_numpy(bases).cuda()
    # the combination step on partials of 1, 2, 3, 8 shards, each reduced by the HIP MSM
    for world in (1, 2, 3, 8):
        parts = []
        for r in range(world):
            lo, hi = shard_bounds(n, r, world)
            parts.append(best_multiexp(d_sc[32 * lo:32 * hi].contiguous(), d_b[64 * lo:64 * hi].contiguous()))
        assert (combine_partials(np.concatenate(parts)) == want).all(), world
    # identity partials (empty shards) and P + (-P)
    zero = np.zeros(64, dtype=np.uint8)
    assert (combine_partials(np.concatenate([zero, want, zero])) == want).all()
    neg = want.copy()
    q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    neg[32:] = np.frombuffer(((q - int.from_bytes(bytes(want[32:]), "little")) % q).to_bytes(32, "little"), dtype=np.uint8)
    assert (combine_partials(np.concatenate([want, neg])) == zero).all()
    # the collective path itself
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        assert (sharded_msm(d_sc, d_b) == want).all()
        assert (sharded_msm(d_sc[:0], d_b[:0]) == zero).all()
    finally:
        dist.destroy_process_group()


def test_setup_export_import_round_trip_and_batch_of_proofs():
    """what `broadcast_setup` carries rebuilds an equal proving key on the receiving side; `prove_batch` with two and
    three proofs in flight proves every user of the reference's CSV, each proof checked by the oracle's verifier"""
    _gpu()
    from circuits_halo2_amd import api, batch as B
    from circuits_halo2_amd.merkle_sum_tree import MerkleSumTree
    from oracle import summa_verifier as SV
    from test_gpu_api import oracle_vk
    levels, nc, k = 4, 2, 11
    params, pk, vk = B.setup_on_all_ranks(k, os.path.join(GOLDEN, "hermez-raw-11"), levels, nc)
    try:
        params2, pk2, vk2 = B.import_setup(B.export_setup(params, pk))
        assert vk2.fixed_comms == vk.fixed_comms and vk2.permutation_comms == vk.permutation_comms
        assert vk2.transcript_repr == vk.transcript_repr and pk2.circuit_shape == pk.circuit_shape
        tree = MerkleSumTree.from_csv(os.path.join(GOLDEN, "entry_16.csv"), nc)
        ovk = oracle_vk(params, vk)
        for in_flight, flavour, (p_, k_) in ((2, "evm", (params2, pk2)), (3, "blake2b", (params, pk))):
            res = B.prove_batch(tree, list(range(16)), p_, k_, levels, flavour=flavour, in_flight=in_flight)
            assert not res.errors and sorted(res.proofs) == list(range(16))
            for user, (proof, inst) in res.proofs.items():
                assert inst[1:] == [int.from_bytes(bytes(tree.root()[0]), "little") * pow(1 << 256, -1, SV.R) % SV.R, 556862, 556862]
                assert SV.verify(proof, inst, ovk, flavour=flavour), (user, flavour)
            print(f"in_flight={in_flight} {flavour}: {res.proofs_per_s():.1f} proofs/s (k = 11)")
        params2.free()
    finally:
        params.free()


def test_device_snapshot_tree_and_its_merkle_proofs():
    """DeviceMerkleSumTree (nodes resident in HBM) hands out the same Merkle proofs as the host-mirrored tree, and a
    proof made from one is accepted"""
    _gpu()
    import torch
    from circuits_halo2_amd import api
    from circuits_halo2_amd.merkle_sum_tree import DeviceMerkleSumTree, MerkleSumTree
    from circuits_halo2_amd.utils import ints_to_fr
    from oracle import pyref as PR
    nc, depth = 2, 5
    entries = [(f"user{i}", [1000 + 3 * i, 17 * i]) for i in range(1 << depth)]
    host_tree = MerkleSumTree.from_entries(entries, nc)
    users = ints_to_fr([int.from_bytes(PR.keccak256(name.encode()), "big") for name, _ in entries])
    bals = ints_to_fr([b for _, bal in entries for b in bal])
    dev_tree = DeviceMerkleSumTree(torch.from_numpy(users).cuda(), torch.from_numpy(bals).cuda(), depth, nc)
    assert bytes(dev_tree.root()[0]) == bytes(host_tree.root()[0]) and bytes(dev_tree.root()[1]) == bytes(host_tree.root()[1])
    for index in (0, 1, 13, 31):
        a, b = dev_tree.generate_proof(index), host_tree.generate_proof(index)
        assert a["path_indices"] == b["path_indices"] and a["entry"][1] == b["entry"][1]
        assert a["entry"][0] == int.from_bytes(PR.keccak256(b["entry"][0].encode()), "big") % PR.R
        assert bytes(a["sibling_leaf_node_hash_preimage"]) == bytes(b["sibling_leaf_node_hash_preimage"])
        assert [bytes(x) for x in a["sibling_middle_node_hash_preimages"]] == [bytes(x) for x in b["sibling_middle_node_hash_preimages"]]
        ca, cb = api.MstInclusionCircuit.init(a, depth), api.MstInclusionCircuit.init(b, depth)
        assert ca.instances() == cb.instances() and ca.entry == cb.entry
    params, pk, vk = api.generate_setup_artifacts(12, None, api.MstInclusionCircuit.init_empty(depth, nc))
    try:
        circuit = api.MstInclusionCircuit.init(dev_tree.generate_proof(13), depth)
        proof = api.full_prover(params, pk, circuit, circuit.instances())
        assert api.full_verifier(params, vk, proof, circuit.instances())
    finally:
        params.free()


@pytest.mark.parametrize("levels,nc,k,users", [(4, 2, 11, [0, 5, 15]), (20, 1, 13, [123457]), (20, 2, 17, [5, (1 << 20) - 1, 777777])])
def test_device_witness_equals_the_host_assignment(levels, nc, k, users):
    """sg_mst_inclusion_witness_dev (Circuit::synthesize as one kernel launch over the device-resident tree) writes
    bit for bit the advice columns of mst_inclusion.reference_assignment (the integer replay of the reference's
    synthesize, pinned by the reproduced verifying key), several users per launch; public inputs included"""
    _gpu()
    import types
    import torch
    from circuits_halo2_amd import api, arithmetic as A
    from circuits_halo2_amd.merkle_sum_tree import DeviceMerkleSumTree
    from circuits_halo2_amd.utils import ints_to_fr, random_fr_canonical
    size = 1 << levels
    d_users = A.fr_random(bytes(range(32)), 7, size)
    bal = random_fr_canonical(91, size * nc).reshape(-1, 32).copy()
    bal[:, 5:] = 0
    d_bals = A.fr_to_montgomery(torch.from_numpy(bal.reshape(-1)).cuda())
    tree = DeviceMerkleSumTree(d_users, d_bals, levels, nc)
    key = types.SimpleNamespace(k=k, n=1 << k, circuit_shape=(levels, nc, 8))
    adv = api.synthesize_on_device(key, tree, users).cpu().numpy()
    for u, idx in enumerate(users):
        host = api.MstInclusionCircuit.init(tree.generate_proof(idx), levels)
        asg = host.synthesize(k)
        for c in range(3):
            want = ints_to_fr(asg["advice"][c])
            assert (adv[u, c] == want).all(), (idx, c, int(np.nonzero(adv[u, c] != want)[0][0]) // 32)
        dev_circuit = api.MstInclusionCircuit.init_from_tree(tree, idx)
        assert dev_circuit.instances() == host.instances() == [asg["instances"]]
        assert dev_circuit.path_indices == host.path_indices


def test_proofs_from_the_device_witness_verify():
    """the production path of the batch driver: snapshot on the device -> witness kernel -> create_proof -> verified,
    both flavours, and the same through prove_batch"""
    _gpu()
    import torch
    from circuits_halo2_amd import api, arithmetic as A, batch as B
    from circuits_halo2_amd.merkle_sum_tree import DeviceMerkleSumTree
    from circuits_halo2_amd.utils import random_fr_canonical
    from oracle import summa_verifier as SV
    from test_gpu_api import oracle_vk
    levels, nc, k = 6, 2, 12
    size = 1 << levels
    bal = random_fr_canonical(92, size * nc).reshape(-1, 32).copy()
    bal[:, 4:] = 0
    tree = DeviceMerkleSumTree(A.fr_random(bytes(range(32)), 9, size), A.fr_to_montgomery(torch.from_numpy(bal.reshape(-1)).cuda()), levels, nc)
    params, pk, vk = api.generate_setup_artifacts(k, None, api.MstInclusionCircuit.init_empty(levels, nc))
    try:
        ovk = oracle_vk(params, vk)
        circuit = api.MstInclusionCircuit.init_from_tree(tree, 37)
        proof, inst = api.gen_proof_solidity_calldata(params, pk, circuit)
        assert SV.verify(proof, inst, ovk) and inst == tree.public_inputs(37)
        blake = api.full_prover(params, pk, circuit, circuit.instances())
        assert api.full_verifier(params, vk, blake, circuit.instances()) and SV.verify(blake, inst, ovk, flavour="blake2b")
        res = B.prove_batch(tree, list(range(0, 64, 5)), params, pk, levels, in_flight=3)
        assert not res.errors and sorted(res.proofs) == list(range(0, 64, 5))
        assert all(SV.verify(p, i, ovk) for p, i in res.proofs.values())
        with pytest.raises(IndexError):
            api.MstInclusionCircuit.init_from_tree(tree, size)
    finally:
        params.free()


def _run_bench(extra, timeout=1100):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra, capture_output=True, text=True, timeout=timeout, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def _check_two_rank_line(line, backend_word):
    # what the collectives library saw, and the blocking point-sharded MSM beside the amortised exchange (round 5)
    assert line["config"]["ranks_seen_by_process_group"] == 2 and line["config"]["per_call_sharded_msm_ms"] > 0
    assert line["config"]["rccl_ranks_seen"] == (2 if backend_word == "nccl" else None)
    assert line["sharded_msm_per_call"]["calls"] >= 3 and line["sharded_msm_per_call"]["ms_per_msm"] >= line["ms_per_step"] * 0.5
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["sharding"].startswith("point-sharded") and backend_word in line["config"]["backend"]
    b = line["batch_k17"]
    assert b["n_gpus"] == 2 and b["verified_sample"] is True and b["proofs_total"] == 6 and b["errors"] == 0
    assert all(r["proofs"] == 6 and r["errors"] == 0 for r in b["repeats"])          # dealt 3 + 3, both ranks counted
    assert all(v["errors"] == 0 for v in b["pre_sweep_by_in_flight"].values())
    st = line["msm_strong_scaling_2^23"]
    assert st["n_gpus"] == 2 and st["scaling"] == "strong" and st["points_per_gpu"] == 1 << 22
    assert st["every_rank_partial_equals_inner_product_times_G"] is True
    return st["result_x"]


def test_bench_starts_its_own_ranks_two_rank_rehearsal():
    """`python bench.py --gpus 2` launched bare: the parent starts the two ranks itself (torch.distributed.run child, before
    it touches the GPU), rank 0 prints the one JSON line with n_gpus = 2.  On this one-GPU box the ranks rehearse with the
    gloo backend and share the GPU (`--backend gloo`); everything else -- point-sharded MSM with the all_gather of partials
    (weak: 2^20 per rank; strong: one 2^23-point MSM cut in two), setup broadcast, the batch's users dealt to the ranks and
    kept in flight, max-over-ranks timing -- is the N > 1 code path.  The strong-scaling MSM's result must be the point the
    same extra finds on ONE rank (same 2^23 inputs at every N)."""
    two = _run_bench(["--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--batch-proofs", "6", "--batch-repeats", "1",
                      "--no-cpu", "--log-n", "20"])
    x2 = _check_two_rank_line(two, "gloo")
    one = _run_bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--batch-proofs", "0", "--no-cpu", "--log-n", "20", "--strong-only"])
    assert one["msm_strong_scaling_2^23"]["result_x"] == x2 and one["msm_strong_scaling_2^23"]["points_per_gpu"] == 1 << 23


def test_four_rank_rehearsal_on_one_gpu():
    """the bare launch with FOUR ranks sharing the one GPU (gloo; the GPU boxes of this pool allow six processes on a card, so the
    eight of a whole node are rehearsed on the CPU: tests/test_distributed_cpu.py, test_bench_failsafe_cpu.py): the users of the
    batch dealt over four ranks, the setup broadcast to three receivers, max-over-ranks timing, the watchdog's per-step
    allowance under four-fold contention for the device, and still ONE JSON line; the strong-scaling MSM cut in four gives the
    point it gives cut in two or not at all"""
    line = _run_bench(["--gpus", "4", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--batch-proofs", "8", "--batch-repeats", "1",
                       "--batch-in-flight", "2", "--no-cpu", "--log-n", "20"])
    assert line["n_gpus"] == 4 and line["value"] > 0 and line["scaling"] == "weak"
    assert line["config"]["ranks_seen_by_process_group"] == 4 and line["config"]["rccl_ranks_seen"] is None
    assert line["config"]["per_call_sharded_msm_ms"] > 0
    b = line["batch_k17"]
    assert b["n_gpus"] == 4 and b["proofs_total"] == 8 and b["errors"] == 0 and b["verified_sample"] is True
    assert all(r["proofs"] == 8 and r["errors"] == 0 for r in b["repeats"])          # dealt 2 + 2 + 2 + 2, every rank counted
    st = line["msm_strong_scaling_2^23"]
    assert st["n_gpus"] == 4 and st["points_per_gpu"] == 1 << 21 and st["every_rank_partial_equals_inner_product_times_G"] is True
    one = _run_bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--batch-proofs", "0", "--no-cpu", "--log-n", "20", "--strong-only"])
    assert one["msm_strong_scaling_2^23"]["result_x"] == st["result_x"]


def test_a_rank_killed_mid_batch_costs_an_error_line_not_the_run():
    """the two-rank gloo rehearsal again, and rank 1 is killed (SIGKILL) one second into the headline batch
    (SUMMA_BENCH_FAULT): the launcher ends the surviving rank, whose watchdog prints the error line with what had been
    measured, and the parent of the bare launch returns non-zero with that ONE JSON line -- within two minutes, not after
    the process group's (or the driver's) ten."""
    import json
    import subprocess
    import sys
    import time
    _gpu()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["SUMMA_BENCH_FAULT"] = "1:batch:1.0"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1",
                        "--batch-proofs", "512", "--batch-repeats", "1", "--batch-in-flight", "8", "--no-cpu", "--log-n", "20"],
                       capture_output=True, text=True, timeout=400, env=env, cwd=root)
    took = time.time() - t0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode != 0 and len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    line = json.loads(lines[0])
    assert line["value"] is None and line["error"] and line["n_gpus"] == 2
    # the headline steps had been measured before the batch began: the partial results say so
    part = line.get("partial") or (line.get("rank0_line") or {}).get("partial") or {}
    assert part.get("value", 0) > 0 and part.get("n_gpus") == 2, line
    assert took < 240, took          # set-up of two ranks + one second of batch + the launcher's clean-up; never the timeouts


def test_two_rank_nccl():
    """the same bare launch with the nccl (= RCCL) backend, one GPU per rank: the first box with two GPUs exercises the
    collectives over xGMI -- the all_gather of partials, the device-to-device setup broadcast, the timing all_reduces --
    without anyone editing code.  One-GPU boxes skip it (the gloo rehearsal above covers the logic there)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL between ranks)")
    line = _run_bench(["--gpus", "2", "--backend", "nccl", "--steps", "3", "--warmup", "1", "--batch-proofs", "6", "--batch-repeats", "1",
                       "--no-cpu", "--log-n", "20"])
    _check_two_rank_line(line, "nccl")


def test_batch_of_1024_users_at_k17():
    """BASELINE configs[4] at its stated size on one GPU: inclusion proofs for 1024 users of a 2^20-user snapshot,
    MstInclusionCircuit<20,2,8> at k = 17, through the batch driver (what the reference's backend serves one call at a
    time, backend/src/apis/round.rs:132-174).  Every proof is verified by the product's verifier inside
    gen_proof_solidity_calldata (create_proof_checked, utils.rs:162-196); here additionally: none failed, every user
    got a proof of 2144 bytes whose public inputs are the tree's leaf / root / root balances for THAT user, all leaf
    hashes differ, and the oracle's verifier accepts a sample."""
    _gpu()
    import sys
    import time
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from bench import oracle_vk, snapshot_tree
    from circuits_halo2_amd import batch as B
    from oracle import summa_verifier as SV
    levels, nc, k = 20, 2, 17
    params, pk, vk = B.setup_on_all_ranks(k, None, levels, nc)
    try:
        tree = snapshot_tree(levels, nc)
        users = [(7919 * i + 13) % (1 << levels) for i in range(1024)]
        assert len(set(users)) == 1024
        B.prove_batch(tree, users[:24], params, pk, levels, in_flight=12)        # warm: lanes, sessions
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = B.prove_batch(tree, users, params, pk, levels, flavour="evm", in_flight=12)   # twelve in flight: commitment jobs fused
        dt = time.perf_counter() - t0
        assert not res.errors and sorted(res.proofs) == sorted(users)
        root_inputs = None
        leaves = set()
        for u, (proof, inst) in res.proofs.items():
            assert len(proof) == 2144 and len(inst) == 2 + nc
            assert inst == tree.public_inputs(u)
            root_inputs = root_inputs or inst[1:]
            assert inst[1:] == root_inputs
            leaves.add(inst[0])
        assert len(leaves) == 1024
        ovk = oracle_vk(params, vk)
        for u in users[::256]:
            assert SV.verify(res.proofs[u][0], res.proofs[u][1], ovk), u
        # a proof does not verify for another user's public inputs
        a, b = users[0], users[1]
        from circuits_halo2_amd import verifier as V
        assert V.verify_proof(params, vk, res.proofs[a][0], res.proofs[a][1], "evm")
        assert not V.verify_proof(params, vk, res.proofs[a][0], res.proofs[b][1], "evm")
        print(f"1024 proofs at k = 17 in {dt:.2f} s = {1024 / dt:.1f} proofs/s (12 in flight, every proof re-verified)")
    finally:
        params.free()


def test_commit_combiner_fuses_the_jobs_of_proofs_in_flight():
    """sg_commit_combine_begin / _end: proofs in flight on eight threads hand their commitment jobs to the combiner; the
    proofs are what they would be alone (each accepted by the product's verifier and the oracle's), requests outnumber
    the fused jobs, and two keys of different sizes proving at the same time are never fused with each other (their jobs
    differ in SRS and length) -- both batches come out right"""
    _gpu()
    import ctypes as C
    import threading
    import torch
    from circuits_halo2_amd import api, arithmetic as A, batch as B, ffi
    from circuits_halo2_amd.merkle_sum_tree import DeviceMerkleSumTree
    from circuits_halo2_amd.utils import random_fr_canonical
    from oracle import summa_verifier as SV
    from test_gpu_api import oracle_vk

    def stats():
        a, b = C.c_uint64(0), C.c_uint64(0)
        ffi.check(ffi.lib().sg_commit_combine_stats(C.byref(a), C.byref(b)))
        return a.value, b.value

    def snapshot(levels, nc, seed):
        size = 1 << levels
        bal = random_fr_canonical(seed, size * nc).reshape(-1, 32).copy()
        bal[:, 4:] = 0
        return DeviceMerkleSumTree(A.fr_random(bytes(range(32)), seed, size), A.fr_to_montgomery(torch.from_numpy(bal.reshape(-1)).cuda()), levels, nc)

    setups = []
    try:
        for levels, k, seed in ((6, 12, 31), (5, 13, 32)):
            params, pk, vk = api.generate_setup_artifacts(k, None, api.MstInclusionCircuit.init_empty(levels, 2))
            setups.append((levels, params, pk, vk, snapshot(levels, 2, seed)))
        j0, r0 = stats()
        levels, params, pk, vk, tree = setups[0]
        users = list(range(0, 64, 2))
        res = B.prove_batch(tree, users, params, pk, levels, in_flight=8, combine=True)
        j1, r1 = stats()
        assert not res.errors and sorted(res.proofs) == users
        assert r1 - r0 == 5 * len(users) and j1 - j0 < r1 - r0            # five commitment jobs per proof, fused
        ovk = oracle_vk(params, vk)
        assert all(SV.verify(p, i, ovk) for p, i in list(res.proofs.values())[::5])
        # two keys at once, from two driver threads with four workers each
        out = {}

        def drive(idx):
            lv, pr, key, _, tr = setups[idx]
            out[idx] = B.prove_batch(tr, list(range(20)), pr, key, lv, in_flight=4, combine=True)
        threads = [threading.Thread(target=drive, args=(i,)) for i in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for idx in range(2):
            lv, pr, key, vkey, tr = setups[idx]
            assert not out[idx].errors and sorted(out[idx].proofs) == list(range(20))
            ovk = oracle_vk(pr, vkey)
            assert all(SV.verify(p, i, ovk) for p, i in list(out[idx].proofs.values())[::4])
    finally:
        for s_ in setups:
            s_[1].free()


def _small_snapshot(levels, nc, seed):
    import torch
    from circuits_halo2_amd import arithmetic as A
    from circuits_halo2_amd.merkle_sum_tree import DeviceMerkleSumTree
    from circuits_halo2_amd.utils import random_fr_canonical
    size = 1 << levels
    bal = random_fr_canonical(seed, size * nc).reshape(-1, 32).copy()
    bal[:, 4:] = 0
    return DeviceMerkleSumTree(A.fr_random(bytes(range(32)), seed, size), A.fr_to_montgomery(torch.from_numpy(bal.reshape(-1)).cuda()), levels, nc)


def test_a_failed_fused_job_does_not_fail_its_members():
    """the commit combiner fuses the commitment jobs of up to sixteen proofs; a fused job that fails as a whole (injected:
    `debug.fail_next_fused_job` makes the next one report SG_ERR_NOMEM without running) must cost nobody a proof: every
    member is re-run as a job of its own -- all sixteen proofs come out and verify (batch.py: "one bad witness must not
    lose the batch"), and the statistics show that members were isolated"""
    _gpu()
    import ctypes as C
    from circuits_halo2_amd import api, batch as B, ffi
    from oracle import summa_verifier as SV
    from test_gpu_api import oracle_vk
    levels, k = 6, 12
    params, pk, vk = api.generate_setup_artifacts(k, None, api.MstInclusionCircuit.init_empty(levels, 2))
    try:
        tree = _small_snapshot(levels, 2, 41)
        users = list(range(16))
        B.prove_batch(tree, users, params, pk, levels, in_flight=16, combine=True)          # warm: sessions, lanes
        j0, r0 = C.c_uint64(0), C.c_uint64(0)
        ffi.check(ffi.lib().sg_commit_combine_stats(C.byref(j0), C.byref(r0)))
        ffi.check(ffi.lib().sg_set_param(b"debug.fail_next_fused_job", 1))
        res = B.prove_batch(tree, users, params, pk, levels, in_flight=16, combine=True)
        ffi.check(ffi.lib().sg_set_param(b"debug.fail_next_fused_job", 0))
        assert not res.errors, res.errors
        assert sorted(res.proofs) == users
        ovk = oracle_vk(params, vk)
        assert all(SV.verify(p, i, ovk) for p, i in res.proofs.values())
        j1, r1 = C.c_uint64(0), C.c_uint64(0)
        ffi.check(ffi.lib().sg_commit_combine_stats(C.byref(j1), C.byref(r1)))
        assert r1.value - r0.value == 5 * len(users) and j1.value - j0.value < r1.value - r0.value    # fused as usual around the failure
    finally:
        params.free()


def test_batches_in_a_long_lived_process_do_not_grow():
    """the reference's backend is a long-lived server that proves for one snapshot after another
    (backend/src/apis/round.rs:132-174): three batches of 256 proofs under each of two keys of different size, and
    afterwards -- keys freed, sg_collect_retired() -- the device has the memory it had before (within 64 MiB) and the
    process the threads it had after the first batch (worker pools and library sessions are reused, not re-created)"""
    _gpu()
    import threading
    import torch
    from circuits_halo2_amd import api, batch as B, ffi

    def free_mib():
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        return torch.cuda.mem_get_info()[0] / 2 ** 20

    def os_threads():
        for ln in open("/proc/self/status"):
            if ln.startswith("Threads:"):
                return int(ln.split()[1])
        return -1

    def one_round(levels, k, seed, batches):
        params, pk, vk = api.generate_setup_artifacts(k, None, api.MstInclusionCircuit.init_empty(levels, 2))
        try:
            tree = _small_snapshot(levels, 2, seed)
            users = [(37 * i + 5) % (1 << levels) for i in range(256)]
            for _ in range(batches):
                res = B.prove_batch(tree, users, params, pk, levels, in_flight=8)
                assert not res.errors and len(res.proofs) == len(set(users))
            del tree
        finally:
            params.free()
        del pk, vk

    # warm-up rounds first: worker threads and their sessions, the plans of both sizes, and the work spaces of the library's
    # lanes -- a fused job runs on whichever lane is free, so it takes a few batches until every lane has met the largest job
    history = []
    for rnd in range(6):
        one_round(6, 12, 60 + rnd, 1)
        one_round(7, 13, 70 + rnd, 1)
        ffi.check(ffi.lib().sg_collect_retired())
        history.append((round(free_mib()), threading.active_count(), os_threads()))
    print("free MiB / python threads / OS threads after each round:", history)
    (base_mem, base_py, base_os), (mem, py, osn) = history[2], history[5]
    assert abs(mem - base_mem) <= 64, history           # three batches under each key later: what the device had
    assert py == base_py and osn <= base_os + 2, history
