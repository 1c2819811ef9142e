"""world_size-2 gloo test of the point-sharded MSM driver (circuits_halo2_amd/distributed.py).
The collective logic (shard -> partial -> all_gather -> sum of partials) is the product's;
the per-shard MSM is injected, and on this GPU-less box the injected callable is the CPU
oracle, so the test checks the N > 1 exchange path end to end without a GPU."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["REPO_ROOT"])
import torch.distributed as dist
from oracle import oracle as O
from circuits_halo2_amd.distributed import sharded_msm, shard_bounds, assign_ops

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
assert sorted(sum((assign_ops(35, r, world) for r in range(world)), [])) == list(range(35))
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_point_sharded_msm_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, REPO_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("ok") == 2
