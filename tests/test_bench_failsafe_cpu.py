"""bench.py owes the driver ONE JSON line whatever happens: the watchdog of a rank (`FailSafe`:
wall-clock limit, per-step allowances, SIGTERM from the launcher through the wake-up descriptor while the main thread
is blocked) and the parent of a bare `--gpus N` launch (`run_ranks_and_relay`: failed ranks, no line, a timeout).  CPU
only: the ranks here never reach a GPU -- on this box they fail at once, which is one of the cases."""
import json
import os
import signal
import subprocess
import sys
import time

from conftest import ROOT

FAILSAFE_WORKER = r'''
import os, sys, time, types
sys.path.insert(0, os.environ["REPO_ROOT"])
import bench
args = types.SimpleNamespace(gpus=1, steps=20, warmup=5)
mode = sys.argv[1]
fs = bench.FailSafe(int(os.environ.get("FS_RANK", "0")), args, 60.0 if mode != "limit" else 0.4)
fs.arm()
fs.partial["value"] = 123.0
print("armed", flush=True)
if mode == "step":
    fs.beat("a collective nobody else joins", 0.3)
if mode == "extra":
    fs.base_line = {"metric": "msm_points_per_sec", "value": 7.5e8, "n_gpus": 1, "roofline": {"frac": 0.0118}}
    fs.beat("an extra that hangs", 0.3)
if mode == "ok":
    assert fs.emit({"metric": "msm_points_per_sec", "value": 1.0})
    assert not fs.emit({"second": True})          # one line only
    sys.exit(0)
t0 = time.time()
while time.time() - t0 < 30:                      # "blocked": the watchdog has to end this
    time.sleep(0.05)
print("NOT REACHED", flush=True)
'''


def _run_worker(tmp_path, mode, rank="0", kill_after=None):
    script = tmp_path / "fs_worker.py"
    script.write_text(FAILSAFE_WORKER)
    env = dict(os.environ, REPO_ROOT=ROOT, FS_RANK=rank)
    p = subprocess.Popen([sys.executable, str(script), mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if kill_after is not None:
        assert p.stdout.readline().strip() == "armed"
        time.sleep(kill_after)
        p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=60)
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    return p.returncode, lines, out, err


def test_wall_clock_limit_prints_the_error_line(tmp_path):
    rc, lines, out, _ = _run_worker(tmp_path, "limit")
    assert rc == 124 and len(lines) == 1 and "NOT REACHED" not in out
    line = json.loads(lines[0])
    assert line["value"] is None and "wall-clock limit" in line["error"] and line["partial"]["value"] == 123.0
    assert line["metric"] == "msm_points_per_sec" and line["n_gpus"] == 1 and line["steps"] == 20


def test_a_step_that_overstays_its_allowance(tmp_path):
    rc, lines, _, _ = _run_worker(tmp_path, "step")
    assert rc == 124 and len(lines) == 1
    assert "a collective nobody else joins" in json.loads(lines[0])["error"]


def test_an_extra_that_hangs_costs_the_extras_not_the_headline(tmp_path):
    """single rank, timed steps done (FailSafe.base_line is set), then an extra overstays: the run's line is the headline,
    whole, with `incomplete` naming what is missing -- exit code 0, no `error` key"""
    rc, lines, out, _ = _run_worker(tmp_path, "extra")
    assert rc == 0 and len(lines) == 1 and "NOT REACHED" not in out
    line = json.loads(lines[0])
    assert line["value"] == 7.5e8 and "error" not in line and "an extra that hangs" in line["incomplete"]
    assert line["roofline"]["frac"] == 0.0118 and line["partial_extras"]["value"] == 123.0


def test_sigterm_from_the_launcher_while_blocked(tmp_path):
    t0 = time.time()
    rc, lines, _, _ = _run_worker(tmp_path, "sigterm", kill_after=0.3)
    assert time.time() - t0 < 20
    assert rc == 143 and len(lines) == 1 and "signal" in json.loads(lines[0])["error"]
    # a rank other than 0 leaves without a line (rank 0 owns stdout's JSON)
    rc, lines, _, err = _run_worker(tmp_path, "sigterm", rank="1", kill_after=0.3)
    assert rc == 143 and lines == [] and "rank 1" in err


def test_the_one_line_is_printed_once(tmp_path):
    rc, lines, _, _ = _run_worker(tmp_path, "ok")
    assert rc == 0 and len(lines) == 1 and json.loads(lines[0])["value"] == 1.0


def test_parent_of_a_bare_multi_rank_launch_reports_failed_ranks():
    """`python bench.py --gpus 2` where the ranks cannot run (no GPU on this box: every rank exits at once): the parent
    returns promptly, non-zero, with one JSON line carrying `error`"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
                        "--no-extras", "--no-cpu", "--wall-limit", "60"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert time.time() - t0 < 120
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode != 0 and len(lines) == 1, (r.returncode, r.stdout[-800:], r.stderr[-800:])
    line = json.loads(lines[0])
    assert line["value"] is None and line["error"] and line["n_gpus"] == 2


EIGHT_RANK_WORKER = r'''
import os, sys, time, types, datetime
sys.path.insert(0, os.environ["REPO_ROOT"])
import torch.distributed as dist
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
args = types.SimpleNamespace(gpus=world, steps=20, warmup=5)
fs = bench._FS = bench.FailSafe(rank, args, 60.0)
fs.arm()
dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=90))
# the timing collectives of the bench at the size of a whole node: max over ranks, sums over ranks
assert bench._all_max(0.25 + rank, world, "cpu") == 0.25 + (world - 1)
assert bench._all_sum([1.0, rank, 2.0 * rank], world, "cpu") == [float(world), world * (world - 1) / 2, float(world * (world - 1))]
fs.partial["value"] = 6.0e9
if sys.argv[1] == "straggler" and rank == 5:
    time.sleep(30)                                  # never joins the next collective
bench._beat("a step's collective", 1.5)             # per-step allowance, as bench.timed() sets one around every collective
dist.barrier()
if sys.argv[1] == "ok":
    if rank == 0:
        fs.emit({"metric": "msm_points_per_sec", "value": 6.0e9, "n_gpus": world})
    dist.barrier()
    dist.destroy_process_group()
'''


def _run_eight(tmp_path, mode, port):
    script = tmp_path / "eight.py"
    script.write_text(EIGHT_RANK_WORKER)
    env = dict(os.environ, REPO_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script), mode], env=env, capture_output=True, text=True, timeout=300)
    return r, [ln for ln in r.stdout.splitlines() if ln.startswith("{")], time.time() - t0


def test_eight_ranks_one_line(tmp_path):
    """the bench's timing collectives and its one-JSON-line contract with eight ranks (gloo on the CPU: the size the driver's
    scaling run uses, rehearsed without the hardware)"""
    r, lines, _ = _run_eight(tmp_path, "ok", 29611)
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    assert json.loads(lines[0])["n_gpus"] == 8


def test_eight_ranks_a_straggler_costs_an_error_line_within_the_step_allowance(tmp_path):
    """rank 5 of 8 never reaches a step's collective: the other ranks' watchdogs end them after the step's allowance (not after the
    process group's timeout), rank 0 prints the ONE error line with what had been measured, the launcher returns non-zero"""
    r, lines, took = _run_eight(tmp_path, "straggler", 29621)
    assert r.returncode != 0 and len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    line = json.loads(lines[0])
    assert line["value"] is None and "a step's collective" in line["error"] and line["partial"]["value"] == 6.0e9 and line["n_gpus"] == 8
    assert took < 60, took
