"""One process per GPU over RCCL: the cfg5 driver (training/trajopt_batch.py) and bench.py under ``torch.distributed.run`` with one
rank (a single GPU is what the test box has; the process group, the barrier and the all_gather / all_reduce run exactly as with 8)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(args, port, timeout=900):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(port)] + args
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_trajopt_batch_folding_one_rank(tmp_path):
    r = _launch(["-m", "thinshelllab_amd.training.trajopt_batch", "--env", "folding", "--scenes", "2", "--iter", "2", "--tot_step", "4", "--out", str(tmp_path)], 29641)
    assert r.returncode == 0, r.stderr[-3000:]
    hist = np.load(tmp_path / "plot_data.npy")
    assert hist.shape == (2, 2) and np.isfinite(hist).all()
    t0, t1 = np.load(tmp_path / "traj_scene0.npy"), np.load(tmp_path / "traj_scene1.npy")
    assert t0.shape == t1.shape == (4, 1, 6) and np.abs(t0 - t1).max() > 0          # different seeds per scene
    assert np.load(tmp_path / "best_gripper_grad.npy").shape == (4, 1, 6)


def test_bench_under_launcher_reports_world():
    r = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--grid", "48", "--no-cpu-baseline"], 29643)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["solves_unconverged"] == 0
    # a mismatch between --gpus and the launcher's world size is an error, not a silently different run
    r2 = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--grid", "48", "--no-cpu-baseline"], 29645)
    assert r2.returncode != 0
