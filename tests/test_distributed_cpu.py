"""world_size-2 and -8 gloo tests of the point-sharded MSM driver (circuits_halo2_amd/distributed.py).
The collective logic (shard -> partial -> all_gather -> sum of partials) is the product's;
the per-shard MSM is injected, and on this GPU-less box the injected callable is the CPU
oracle, so the test checks the N > 1 exchange path end to end without a GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["REPO_ROOT"])
import torch.distributed as dist
from oracle import oracle as O
from circuits_halo2_amd.distributed import sharded_msm, shard_bounds

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n = 3000
sc = O.random_fr(42, n)
bases = O.fixed_base_mul(O.random_fr(43, n), 2)
lo, hi = shard_bounds(n, rank, world)
msm = lambda s, b: O.best_multiexp(np.ascontiguousarray(s), np.ascontiguousarray(b), 2)
got = sharded_msm(sc[32 * lo:32 * hi], bases[64 * lo:64 * hi], msm=msm)
want = O.best_multiexp(sc, bases, 2)
assert (got == want).all(), "sharded MSM differs from the single-process result"
# an empty shard contributes the identity
got2 = sharded_msm(sc[:32 * n] if rank == 0 else sc[:0], bases[:64 * n] if rank == 0 else bases[:0], msm=msm)
assert (got2 == want).all()
# the exchange of several steps in ONE collective gives what one collective per step gives
from circuits_halo2_amd.distributed import exchange_partials, exchange_partials_many
parts = [msm(sc[32 * (lo + j):32 * hi], bases[64 * (lo + j):64 * hi]) for j in range(3)]
many = exchange_partials_many(parts, msm=msm)
assert len(many) == 3 and all((many[j] == exchange_partials(parts[j], msm=msm)).all() for j in range(3))
assert (many[0] == want).all() and exchange_partials_many([], msm=msm) == []
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


@pytest.mark.parametrize("world", [2, 8])
def test_point_sharded_msm_gloo(tmp_path, world):
    """world 8 = the size the driver's scaling run uses (one rank per GPU of a node): shard bounds, the all_gather of eight partials
    and the host sum of eight points, rehearsed without the hardware"""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, REPO_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29533 + 10 * world), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("ok") == world


BATCH_WORKER = r'''
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.environ["REPO_ROOT"])
import torch.distributed as dist
from circuits_halo2_amd import batch as B

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
# --- the setup artifacts exist once: rank 0 holds them, the broadcast hands every rank the same bytes
rng = np.random.default_rng(7)
k = 5
setup = None
if rank == 0:
    setup = {"k": k, "shape": (4, 2, 8), "g": rng.integers(0, 256, 64 << k, dtype=np.uint8), "g_lagrange": rng.integers(0, 256, 64 << k, dtype=np.uint8),
             "g2": bytes(range(128)), "s_g2": bytes(range(128, 256)), "fixed": [rng.integers(0, 256, 32 << k, dtype=np.uint8) for _ in range(11)],
             "sigma": [rng.integers(0, 256, 32 << k, dtype=np.uint8) for _ in range(6)], "vk_digest": 0x1234567890ABCDEF << 100}
got = B.broadcast_setup(setup, 0)
ref = np.random.default_rng(7)                      # every rank re-derives what rank 0 drew
assert got["k"] == k and got["shape"] == (4, 2, 8) and got["vk_digest"] == 0x1234567890ABCDEF << 100
assert (got["g"] == ref.integers(0, 256, 64 << k, dtype=np.uint8)).all() and (got["g_lagrange"] == ref.integers(0, 256, 64 << k, dtype=np.uint8)).all()
assert bytes(got["g2"]) == bytes(range(128)) and bytes(got["s_g2"]) == bytes(range(128, 256))
for col in got["fixed"] + got["sigma"]:
    assert (col == ref.integers(0, 256, 32 << k, dtype=np.uint8)).all()
assert len(got["fixed"]) == 11 and len(got["sigma"]) == 6
# --- a source with nothing valid to send says so in the header: EVERY rank raises, none waits in the next broadcast
bad = dict(setup, g=setup["g"][:-1]) if rank == 0 else None
try:
    B.broadcast_setup(bad, 0)
    raise SystemExit("a short buffer on the source must fail the broadcast on every rank")
except RuntimeError as ex:
    assert "nothing valid to send" in str(ex)
    assert (ex.__cause__ is not None and "g has" in str(ex.__cause__)) == (rank == 0)
again = B.broadcast_setup(setup, 0)                  # and the group is still usable
assert again["k"] == k and (again["g"] == got["g"]).all()
# --- users are dealt round-robin; every user is proven exactly once across the group; several proofs in flight
users = list(range(3, 40))
mine = B.deal(users)
assert mine == users[rank::world] and all(B.owner_of(users.index(u), world) == rank for u in mine)
peak, live, lock = [0], [0], threading.Lock()
def prove(circuit):
    with lock:
        live[0] += 1
        peak[0] = max(peak[0], live[0])
    time.sleep(0.02)
    with lock:
        live[0] -= 1
    if circuit == 17:
        raise ValueError("lookup input value not in the table")
    return (b"proof-%d" % circuit, [circuit, rank])
res = B.prove_batch(None, users, None, None, levels=4, in_flight=3, prove=prove, make_circuit=lambda i: i)
assert sorted(list(res.proofs) + list(res.errors)) == mine
assert (17 in res.errors) == (17 in mine) and all(res.proofs[u] == (b"proof-%d" % u, [u, rank]) for u in res.proofs)
assert peak[0] == 3, peak                                 # three proofs were in flight on this rank
assert res.proofs_per_s() > 0
serial = B.prove_batch(None, users, None, None, levels=4, in_flight=1, prove=prove, make_circuit=lambda i: i)
assert serial.seconds > res.seconds * 1.5                  # the in-flight overlap is real
merged = B.gather_proofs(res, 0)
if rank == 0:
    assert sorted(merged) == [u for u in users if u != 17]
    assert all(merged[u][1][1] == B.owner_of(users.index(u), world) for u in merged)
else:
    assert merged is None
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


@pytest.mark.parametrize("world", [2, 8])
def test_proof_batch_scheduler_gloo(tmp_path, world):
    """bookkeeping of the proof-level batch driver (circuits_halo2_amd/batch.py) with two and with eight ranks: the setup broadcast,
    the round-robin deal, proofs in flight per rank, error isolation, the gather -- with stand-in provers (the real
    prover needs the GPU; tests/test_gpu_batch.py runs it)"""
    script = tmp_path / "batch_worker.py"
    script.write_text(BATCH_WORKER)
    env = dict(os.environ, REPO_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29534 + 10 * world), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("ok") == world
