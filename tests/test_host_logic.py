"""CPU tests of the host side: reference-surface objects (fields, mesh tables, poses, scenes, agent, optimiser,
readers), the C-ABI library (loads, exports every declared symbol, fails loudly without a GPU) and the
multi-process batch helpers (gloo, world_size 2)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    if not os.path.exists(g.LIB):
        g.build()
    hdr = open(os.path.join(ROOT, "include", "tsl_hip.h")).read()
    declared = sorted(set(re.findall(r"^(?:int|void|const char\*)\s+(tsl_[a-z_0-9]+)\(", hdr, flags=re.M)))
    from thinshelllab_amd import _lib
    assert sorted(_lib.EXPORTS) == declared
    L = ctypes.CDLL(g.LIB)
    for sym in declared:
        assert hasattr(L, sym), sym
    L.tsl_version.restype = ctypes.c_char_p
    assert b"gfx950" in L.tsl_version()
    # the binding declares the argument types of every entry point that takes arguments (a raw int pointer would otherwise be
    # truncated to 32 bits)
    B = _lib.load()
    for sym in declared:
        if sym in ("tsl_version", "tsl_last_error"):
            continue
        assert getattr(B, sym).argtypes is not None, sym


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_engine_call_without_gpu_fails_loudly():
    from thinshelllab_amd._lib import TslError, TslLibraryError
    from thinshelllab_amd.task_scene.Scene_drape import Scene
    s = Scene(N=6)
    s.init_all()
    with pytest.raises((TslLibraryError, TslError)):
        s.time_step(None, 1)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "thinshelllab_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "pyoracle" not in txt and "tslo_" not in txt and "from oracle" not in txt, os.path.join(dp, f)


@pytest.mark.parametrize("N,M", [(15, 3), (7, 10), (16, 16)])
def test_cloth_tables_match_oracle(oracle, N, M):
    from thinshelllab_amd.engine.model_fold_offset import Cloth
    c = Cloth(N, 5e-3, 0.1, 0, 40.0, 0, False, M)
    c.init(0.1, 0.2, 0.3)
    o = oracle.OracleScene()
    ci = o.add_cloth(N, M, 0.1)
    o.cloth_init(ci, 0.1, 0.2, 0.3)
    o.finalize()
    assert np.array_equal(c.f2v.to_numpy(), o.arr("cloth0.f2v", (-1, 3)))
    assert np.array_equal(c.counter_face.to_numpy(), o.arr("cloth0.counter_face", (-1, 3)))
    assert np.array_equal(c.counter_point.to_numpy(), o.arr("cloth0.counter_point", (-1, 3)))
    assert np.abs(c.pos.to_numpy() - o.arr("cloth0.pos", (-1, 3))).max() == 0
    assert np.allclose(c.V.to_numpy(), o.arr("cloth0.V")) and np.allclose(c.l_i.to_numpy(), o.arr("cloth0.l_i", (-1, 3)))


def test_fold_pose_and_initial_ref_angles_match_oracle(oracle):
    from thinshelllab_amd.engine.model_fold_offset import Cloth
    c = Cloth(15, 5e-3, 0.1, 0, 40.0, 0, False, 3)
    c.k_angle[None] = 0.5
    c.init_fold(-0.07, -0.01, 0.0004, 2)
    o = oracle.OracleScene()
    ci = o.add_cloth(15, 3, 0.1)
    o.set_scalar("cloth0.k_angle", 0.5)
    o.cloth_init(ci, -0.07, -0.01, 0.0004, fold=True, curv=2)
    o.finalize()
    assert np.abs(c.pos.to_numpy() - o.arr("cloth0.pos", (-1, 3))).max() < 1e-16
    ra = o.arr("cloth0.ref_angle", (-1, 3))
    assert np.count_nonzero(ra) == 12 and np.abs(c.ref_angle.to_numpy() - ra).max() < 1e-12


@pytest.mark.parametrize("name,nv,nf,ntet", [("folding", 502, 610, 1685), ("lifting", 1209, 1242, 4415), ("balancing", 1332, 1176, 5755)])
def test_scene_sizes_and_init_match_reference_and_oracle(oracle, name, nv, nf, ntet):
    """sizes of BASELINE.md section 1; initialisation (host numpy) cross-checked against the oracle's own init code"""
    import importlib
    from helpers import oracle_from_scene
    Scene = importlib.import_module(f"thinshelllab_amd.task_scene.Scene_{name}").Scene
    s = Scene(cloth_size=0.1 if name == "folding" else 0.06)
    s.init_all()
    assert (s.tot_NV, s.tot_NF, sum(e.n_cells for e in s.elastics)) == (nv, nf, ntet)
    assert s.elastics[1].frozen_cnt == 49 and s.elastics[1].surf_point == 53
    o = oracle_from_scene(oracle, s, check_init=True)
    assert o.tot_NV == nv
    # gripper state
    assert np.array_equal(s.gripper.bound_idx.to_numpy(), o.arr("gripper.bound_idx"))
    fx = s.gripper.F_x_upper.to_numpy() if s.gripper.paired else s.gripper.F_x.to_numpy()
    assert np.abs(fx.reshape(-1, 3) - o.arr("gripper.F_x", (-1, 3))).max() < 1e-15


def test_gripper_step_matches_oracle(oracle):
    from helpers import oracle_from_scene
    from thinshelllab_amd.task_scene.Scene_balancing import Scene
    s = Scene(cloth_size=0.06)
    s.init_all()
    o = oracle_from_scene(oracle, s, check_init=False)
    rng = np.random.default_rng(0)
    for _ in range(3):
        dp = rng.normal(0, 1e-4, (2, 3)); dr = rng.normal(0, 1e-2, (2, 3))
        s.action(1, dp, dr); o.action(dp, dr)
    assert np.abs(s.gripper.pos.to_numpy() - o.arr("gripper.pos", (-1, 3))).max() < 1e-15
    assert np.abs(s.gripper.rot.to_numpy() - o.arr("gripper.rot", (-1, 4))).max() < 1e-15
    assert np.array_equal(s.gripper.rotmat.to_numpy().reshape(-1), o.arr("gripper.rotmat"))  # f32 storage on both sides
    assert np.abs(s.pos.to_numpy() - o.pos).max() < 1e-15
    g = rng.normal(size=s.tot_NV * 3)
    s.gripper.gather_grad(g, s)
    o.arr("tmp_z_frozen")[:] = g
    # oracle gather_grad is exercised through the adjoint tests on the GPU box; here compare against the formula
    b = s.gripper.bound_idx.to_numpy().astype(int)
    R = s.gripper.rotmat.to_numpy().astype(np.float64)
    for j in range(2):
        acc = np.zeros(3); ang = np.zeros(3)
        for e, fx in ((s.elastics[2 * j + 1], s.gripper.F_x_upper.to_numpy()), (s.elastics[2 * j + 2], s.gripper.F_x_lower.to_numpy())):
            gj = g.reshape(-1, 3)[e.offset + b]
            acc += gj.sum(0); ang += np.cross((R[j] @ fx[j, b].T).T, gj).sum(0)
        assert np.allclose(s.gripper.d_pos.to_numpy()[j], np.clip(acc / (2 * len(b)), -10, 10))
        assert np.allclose(s.gripper.d_angle.to_numpy()[j], np.clip(ang / (2 * len(b)), -10, 10))


def test_field_surface():
    from thinshelllab_amd.engine.field import Field, ScalarField
    seen = []
    k = ScalarField(100.0, lambda f: seen.append(f.value))
    k[None] = 400.0
    assert k[None] == 400.0 and seen == [400.0]
    f = Field(torch.zeros((4, 3), dtype=torch.float64))
    f[1] = [1.0, 2.0, 3.0]
    assert f[1][2] == 3.0 and f.to_numpy().shape == (4, 3)
    f.fill(2.0)
    g = Field(torch.zeros((4, 3), dtype=torch.float64)); g.copy_from(f)
    assert g.to_torch().sum().item() == 24.0
    g.from_numpy(np.ones((4, 3)))
    assert g[3][0] == 1.0


def test_agent_and_optimizer_follow_reference_formulas():
    from thinshelllab_amd.agent.traj_opt_single import agent_trajopt
    from thinshelllab_amd.optimizer.optim import Adam_single
    a = agent_trajopt(5, 1, max_moving_dist=0.001)
    a.traj.t[1:, 0, 2] = torch.arange(1, 5, dtype=torch.float64) * -0.002
    a.fix_action(0.015)
    assert torch.allclose(a.traj.t[:, 0, 2], torch.tensor([0, -0.001, -0.002, -0.003, -0.004], dtype=torch.float64), atol=1e-7)  # weight = d / (dist + 1e-8)
    a.get_action(2)
    assert abs(a.delta_pos.t[0, 2].item() + 0.001) < 1e-7
    assert abs(a.calculate_dist(2, 0.015, 0) - 0.001) < 1e-7
    opt = Adam_single((5, 1, 6), 1e-3, 0.9, 0.9999, 1e-8)
    p = torch.zeros((5, 1, 6), dtype=torch.float64)
    g = torch.full((5, 1, 6), 2.0, dtype=torch.float64)
    for it in range(10):
        opt.step(p, g)
    # bias-corrected Adam with sqrt(v + eps): constant gradient -> step = lr * g / sqrt(g^2 + eps)
    assert torch.allclose(p, torch.full_like(p, -10 * 1e-3 * 2.0 / np.sqrt(4.0 + 1e-8)), rtol=1e-9)
    assert abs(opt.lr - 0.9e-3) < 1e-15  # lr *= 0.9 after every 10th iteration (optim.py:74-75)


def test_readfile_and_ply(tmp_path):
    from thinshelllab_amd.engine import readfile
    from thinshelllab_amd.engine.model_fold_offset import Cloth
    n, v = readfile.read_node()
    assert n == 276 and len(v[0]) == 3
    n, t = readfile.read_ele("../data/ball.ele")
    assert n == 295 and len(t[0]) == 4
    c = Cloth(3, 5e-3, 0.03, 0, 40.0, 0, False, 2)
    c.init(0, 0, 0)
    p = tmp_path / "c.ply"
    readfile.save_cloth_mesh(c, str(p))
    assert p.read_text().startswith("ply")


_WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from thinshelllab_amd.batch import Batch
b = Batch(backend="gloo", device=torch.device("cpu"))
assert b.world == 2
ids = b.scene_ids(5)
assert ids == ([0, 2, 4] if b.rank == 0 else [1, 3])
b.barrier()
assert b.max_over_ranks(1.0 + b.rank) == 2.0
assert b.sum_over_ranks(10.0) == 20.0
rs, gs = b.gather_results(100.0 + b.rank, torch.full((4, 1, 6), float(b.rank), dtype=torch.float64))
assert rs == [100.0, 101.0] and gs[1].sum().item() == 24.0 and gs[0].sum().item() == 0.0
calls = []
f = b.share_population(5, lambda k: (calls.append(k), 10.0 * k + 1.0)[1])
assert f == [1.0, 11.0, 21.0, 31.0, 41.0] and calls == ids
b.close()
print("ok", b.rank)
'''


def test_batch_helpers_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=240)
        assert p.returncode == 0, err[-2000:]
        assert "ok" in out


_BATCH_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from thinshelllab_amd.batch import Batch
from thinshelllab_amd.training.trajopt_batch import run_batch

class Quad:
    """stand-in for a scene rollout: reward -|traj - target_s|^2 with its gradient; plain gradient ascent"""
    def __init__(self, s):
        self.s = s
        self.traj = torch.zeros((4, 1, 6), dtype=torch.float64)
        self.target = torch.full((4, 1, 6), 0.1 * (s + 1), dtype=torch.float64)
    def rollout(self):
        d = self.traj - self.target
        return -float((d * d).sum()), 2 * d
    def step(self, g):
        self.traj -= 0.25 * g

b = Batch(backend="gloo", device=torch.device("cpu"))
out = sys.argv[2]
hist, best = run_batch(b, 3, 6, Quad, out_dir=out, log=lambda *a: None)     # 3 scenes on 2 ranks: rank 0 runs scenes 0 and 2, rank 1 scene 1
assert sorted(hist) == [0, 1, 2] and all(len(v) == 6 for v in hist.values()), hist
for s in range(3):
    r0 = -24 * (0.1 * (s + 1)) ** 2
    assert abs(hist[s][0] - r0) < 1e-12 and hist[s][-1] > r0 * 1e-3, (s, hist[s])   # every rank holds the rewards of the whole batch
    assert all(hist[s][k + 1] > hist[s][k] for k in range(5))
assert best[1] == 0 and best[2] == 5, best                                        # the smallest target converges fastest
# a rollout that raises on ONE rank (an unconverged adjoint solve) must not stall the gather of the others: NaN for that scene and
# iteration, no update, everybody continues
class Flaky(Quad):
    n = 0
    def rollout(self):
        Flaky.n += 1
        if self.s == 1 and Flaky.n == 2:
            raise RuntimeError("transfer_grad: linear solve not converged")
        return super().rollout()
hist2, _ = run_batch(b, 3, 4, Flaky, out_dir=None, log=lambda *a: None)
assert all(len(v) == 4 for v in hist2.values())
assert np.isnan(hist2[1][1]) and np.isfinite([hist2[1][0], hist2[1][2], hist2[1][3]]).all(), hist2[1]
assert np.isfinite(hist2[0]).all() and np.isfinite(hist2[2]).all()
assert hist2[1][2] > hist2[1][0]
# the scenes of a rank rolled out TOGETHER (rollout_many: a scene group on the GPU): rank 0 holds scenes 0 and 2, rank 1 scene 1 -- same history as one
# after the other; a grouped rollout that raises costs its rank's scenes one iteration, nobody stalls
calls = []
def many(ps):
    calls.append(len(ps))
    if len(calls) == 2:
        raise RuntimeError("tsl_group_step failed")
    return [p.rollout() for p in ps]
hist3, _ = run_batch(b, 3, 4, Quad, out_dir=None, log=lambda *a: None, rollout_many=many)
if b.rank == 0:
    assert calls == [2, 2, 2, 2], calls
    assert np.isnan(hist3[0][1]) and np.isnan(hist3[2][1]) and np.isfinite(hist3[1]).all(), hist3
else:
    assert calls == [], calls      # one scene on this rank: its own rollout
assert abs(hist3[0][0] - hist[0][0]) < 1e-15 and abs(hist3[1][3] - hist[1][3]) < 1e-15
b.barrier()
if b.rank == 0:
    assert np.load(os.path.join(out, "plot_data.npy")).shape == (3, 6)
    assert os.path.exists(os.path.join(out, "traj_scene0.npy")) and os.path.exists(os.path.join(out, "traj_scene2.npy"))
b.close()
print("ok", b.rank)
'''


def test_trajopt_batch_gather_world_size_2_gloo(tmp_path):
    """cfg5 driver (training/trajopt_batch.py): scene -> rank placement, the per-iteration all_gather of (reward, gripper_grad)
    and the outputs, with a stand-in rollout (the engine needs a GPU), two gloo ranks on the CPU"""
    script = tmp_path / "worker.py"
    script.write_text(_BATCH_WORKER)
    out = tmp_path / "out"
    out.mkdir()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(out)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        o, err = p.communicate(timeout=240)
        assert p.returncode == 0, err[-2000:]
        assert "ok" in o


def test_cmaes_restatement_converges():
    """optimizer/cmaes.py (stand-in for the un-vendored `cma` of run_cmaes_all.py): ask / tell / result / stop surface, convergence on
    the sphere and on Rosenbrock, candidate count checks."""
    from thinshelllab_amd.optimizer.cmaes import CMAEvolutionStrategy
    es = CMAEvolutionStrategy(10 * [5.0], 1.0, {"popsize": 8, "seed": 3})
    X = es.ask()
    assert len(X) == 8 and X[0].shape == (10,)
    with pytest.raises(ValueError):
        es.tell(X[:3], [0.0, 1.0, 2.0])
    for _ in range(150):
        X = es.ask(); es.tell(X, [float(np.sum((x - 2.0) ** 2)) for x in X])
    assert es.result.fbest < 1e-6 and np.abs(es.result.xbest - 2.0).max() < 1e-2
    rosen = lambda x: float(np.sum(100 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2))
    es = CMAEvolutionStrategy(6 * [0.0], 0.5, {"popsize": 12, "seed": 1, "maxiter": 500})
    while not es.stop():
        X = es.ask(); es.tell(X, [rosen(x) for x in X])
    assert es.result.fbest < 1e-8 and es.result.evaluations == 12 * es.result.iterations


def test_cmaes_candidate_decoding():
    """run_cmaes_all.decode (reference run_cmaes_all.py:98-114): a constant candidate of 5 is the zero trajectory; an offset on one
    component gives a linear ramp of slope (x - 5) / sub_steps / scaling, clipped by fix_action."""
    from types import SimpleNamespace
    from thinshelllab_amd.agent.traj_opt_single import agent_trajopt
    from thinshelllab_amd.training.run_cmaes_all import decode
    T, abs_step, cnt = 12, 4, 2
    sub = T // abs_step
    scaling = 5.0 / (sub * 0.0003); scaling_angle = 5.0 / (sub * 0.01)
    args = SimpleNamespace(abs_step=abs_step, env="folding")
    agent = agent_trajopt(T, cnt, max_moving_dist=0.002)
    decode(agent, np.full(abs_step * 6 * cnt, 5.0), args, cnt, sub, scaling, scaling_angle)
    assert float(agent.traj.t.abs().max()) == 0.0
    x = np.full(abs_step * 6 * cnt, 5.0); x[1 * 6 * cnt + 1 * 6 + 2] = 6.0   # segment 1, gripper 1, z
    decode(agent, x, args, cnt, sub, scaling, scaling_angle)
    tr = agent.traj.to_numpy()
    step = 1.0 / sub / scaling
    assert np.allclose(tr[sub - 1], 0) and np.allclose(tr[2 * sub - 1, 1, 2], sub * step) and np.allclose(tr[-1, 1, 2], sub * step)
    assert np.allclose(np.delete(tr.reshape(T, -1), 1 * 6 + 2, axis=1), 0)


def test_rl_env_box_fallback():
    from thinshelllab_amd.training.RL_env import Box
    b = Box(low=-0.001, high=0.001, shape=(12,), dtype=np.float32)
    a = b.sample(np.random.default_rng(0)) if not hasattr(b, "np_random") else b.sample()
    assert a.shape == (12,) and b.contains(a)


_BENCH_WORKER = r'''
import json, os, sys, time
sys.path.insert(0, sys.argv[1])
import torch
import bench
from thinshelllab_amd.batch import Batch

b = Batch(backend="gloo", device=torch.device("cpu"))
T, K, W = 1000, 4, 1
calls = []
def rollout():                      # stand-in for run_rollout: rank 1 is the slow one
    calls.append(1)
    time.sleep(0.20 if b.rank == 1 else 0.05)
    return {"newton": 7 + b.rank}
t_wall0 = time.perf_counter()
stats, elapsed = bench.timed_region(b, rollout)
t_wall = time.perf_counter() - t_wall0
assert calls == [1] and stats["newton"] == 7 + b.rank
assert 0.20 <= elapsed <= t_wall + 1e-3, (elapsed, t_wall)          # the MAX over the ranks: the fast rank reports the slow rank's time too
out = bench.headline(T, K, W, b.world, elapsed)
assert out["n_gpus"] == 2 and out["steps"] == K and out["warmup"] == W and out["scaling"] == "weak" and out["higher_is_better"] is True
assert abs(out["value"] - T * K * 2 / elapsed) < 1e-9 * out["value"] and abs(out["ms_per_step"] - elapsed / K * 1e3) < 1e-9
assert out["metric"] == json.load(open(os.path.join(sys.argv[1], "BASELINE.json")))["metric"]
e_all = b.sum_over_ranks(elapsed)
assert abs(e_all - 2 * elapsed) < 1e-12                              # every rank holds the same elapsed
bench.emit(out, b.rank)
b.close()
'''


def test_bench_rank_logic_world_size_2_gloo(tmp_path):
    """bench.py's own rank logic (timed_region / headline / emit) with a stand-in rollout on two gloo ranks: value = T K world / max-over-ranks seconds,
    n_gpus = world, and ONLY rank 0 prints the JSON line -- so that the first 8-GPU run is a measurement, not a debugging session"""
    script = tmp_path / "worker.py"
    script.write_text(_BENCH_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, err = p.communicate(timeout=240)
        assert p.returncode == 0, err[-2000:]
        outs.append(o)
    lines0 = [ln for ln in outs[0].splitlines() if ln.startswith("{")]
    assert len(lines0) == 1 and not [ln for ln in outs[1].splitlines() if ln.startswith("{")]
    import json
    d = json.loads(lines0[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["unit"] == "element-steps/s"
