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


def test_trajopt_batch_grouped_scenes_of_a_rank_match_ungrouped(tmp_path):
    """more scenes than GPUs (BASELINE configs[4] on fewer GPUs): with --group 1 the scenes of a rank step in lock step as ONE scene group
    (merged sparse factorisations, thinshelllab_amd/scene_group.py); rewards and optimised trajectories of two iterations must equal the
    round-robin run bit for bit"""
    outs = []
    for k, grp in enumerate(("0", "1")):
        out = tmp_path / f"g{grp}"
        r = _launch(["-m", "thinshelllab_amd.training.trajopt_batch", "--env", "balancing", "--scenes", "3", "--iter", "2", "--tot_step", "4", "--cloth_N", "48", "--group", grp,
                     "--out", str(out)], 29651 + 2 * k)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append((np.load(out / "plot_data.npy"), [np.load(out / f"traj_scene{s}.npy") for s in range(3)]))
    assert outs[0][0].shape == (3, 2) and np.isfinite(outs[0][0]).all()
    assert np.array_equal(outs[0][0], outs[1][0]), (outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert np.array_equal(a, b)


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


def test_trajopt_batch_cfg5_at_stated_size_one_rank(tmp_path):
    """BASELINE configs[4] at the size SURVEY section 8d states -- 8 scenes of the cfg3 class (folding, 200 x 100 cloth, 40,000 triangles
    each, trajectory seeds default_rng(1000 + scene)) -- on the one GPU the test box has: the 8 scenes run round-robin on rank 0,
    one optimisation iteration of 10 tape steps each (forward rollout, loss seed, reverse sweep, RCCL gather, Adam)."""
    r = _launch(["-m", "thinshelllab_amd.training.trajopt_batch", "--env", "folding", "--scenes", "8", "--iter", "1", "--tot_step", "10", "--cloth_N", "200",
                 "--out", str(tmp_path)], 29647, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    hist = np.load(tmp_path / "plot_data.npy")
    assert hist.shape == (8, 1) and np.isfinite(hist).all(), hist      # a scene with an unconverged adjoint solve would report NaN
    trajs = [np.load(tmp_path / f"traj_scene{s}.npy") for s in range(8)]
    assert all(t.shape == (10, 1, 6) for t in trajs)
    assert min(np.abs(trajs[a] - trajs[b]).max() for a in range(8) for b in range(a)) > 0   # eight different trajectories
    assert np.abs(np.load(tmp_path / "best_gripper_grad.npy")).max() > 0


def test_bench_cfg3_workload_line():
    """bench.py --workload cfg3: the N = 1 point of the cfg5 curve (one 200 x 100 folding scene per GPU)"""
    r = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "cfg3", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], 29649)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["triangles"] == 40000 and d["value"] > 0 and d["config"]["solves_unconverged"] == 0
    assert d["config"]["workload"].startswith("cfg3")
