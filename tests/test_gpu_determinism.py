"""Bit-reproducible rollouts (round 4): every sum of the step and of its adjoint has a fixed order -- element gradients and Hessian blocks
through staging records and gathers, ordered contact lists, energies from per-workgroup partials, the multifrontal extend-add and the
solve sweeps as parent-side gathers -- so two runs of the same rollout give the SAME BITS in the tape, the gradients and the solver
statistics.  (The reference itself is not reproducible: Taichi's atomic adds and its atomic-append constraint list.)  The golden
vectors under tests/golden/det_*.npz were generated on an MI355X by tests/golden/gen_golden_gpu.py and are compared bit for bit."""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rollout(name, T, grid=None, direct=1, params=()):
    """one fresh scene, T - 1 driven steps, the reverse sweep.  Which kernels invert the pivot blocks depends on the neighbourhood (the
    dataflow token of the device, direct_host.hpp) -- and no longer matters: every path gives the same bits (tests/test_gpu_direct.py)."""
    import gc
    from thinshelllab_amd.engine.analytic_grad_single import Grad
    from thinshelllab_amd.engine.geometry import projection_query
    if name == "balancing":
        from thinshelllab_amd.task_scene.Scene_balancing import Scene
        N = grid or 48
        s = Scene(cloth_size=0.12 * N / 224 if N > 100 else 0.06, cloth_N=N, cloth_M=N)
    else:
        from thinshelllab_amd.task_scene.Scene_folding import Scene
        N = grid or 60
        s = Scene(cloth_size=0.1, cloth_N=N, cloth_M=N // 2)
    s.init_all()
    s.mu_cloth_elastic[None] = 5.0
    s.prev_pos.copy_from(s.pos)
    ctx = s._ensure_ctx()
    ctx.set_param("direct", direct)
    for key, v in params:
        ctx.set_param(key, v)
    n_part = s.gripper.n_part
    g = Grad(s, T, n_part); g.init_mass(s)
    g.copy_pos(s, 0)
    stats = []
    for f in range(1, T):
        dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3))
        if name == "balancing":
            dpos[:, 2] = [1e-4, -1e-4][:n_part]
        else:
            dpos[:, 2] = -2e-4
        s.action(f, dpos, drot)
        st = s.time_step(projection_query, f)
        g.copy_pos(s, f)
        stats.append((st["nc"], st["newton_iters"], st["ls_evals"], st["cg_iters"], st["unconverged"], st["energy"]))
    if name == "balancing":
        g.get_loss_balance(s)
    else:
        g.get_loss_fold(s, 1.0, -1.0, rows=s.fold_rows())
    for k in range(T - 1, 0, -1):
        g.transfer_grad(k, s, projection_query)
    out = dict(pos_buffer=g.pos_buffer.to_numpy().copy(), pos_grad=g.pos_grad.to_numpy().copy(), gripper_grad=g.gripper_grad.to_numpy().copy(),
               angleref_grad=g.angleref_grad.to_numpy().copy(), stats=np.array(stats, dtype=np.float64))
    rollout.last_flow_launches = ctx.direct_counters()["flow_launches"]
    del g, s, ctx
    gc.collect()
    return out


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name", ["balancing", "folding"])
def test_two_runs_give_the_same_bits(name):
    a = rollout(name, 5)
    b = rollout(name, 5)
    assert np.abs(a["gripper_grad"]).max() > 0 and a["stats"][:, 0].max() > 0
    for k in a:
        assert np.array_equal(a[k], b[k]), f"{name}: {k} differs between two runs (max |d| = {np.abs(a[k] - b[k]).max():.3e})"


def _sysid_sweep(name):
    """system identification (analytic_grad_system.Grad): 3 driven steps, reverse sweep with the parameter gradients"""
    import gc
    gc.collect()
    from thinshelllab_amd.engine.analytic_grad_system import Grad
    from thinshelllab_amd.engine.geometry import projection_query
    if name == "balancing":
        from thinshelllab_amd.task_scene.Scene_balancing import Scene
        s = Scene(cloth_size=0.06, cloth_N=48, cloth_M=48)
    else:
        from thinshelllab_amd.task_scene.Scene_folding import Scene
        s = Scene(cloth_size=0.1, cloth_N=60, cloth_M=30)
    s.init_all(); s.mu_cloth_elastic[None] = 5.0; s.prev_pos.copy_from(s.pos)
    s._ensure_ctx().set_param("direct", 1)
    T, n_part = 4, s.gripper.n_part
    g = Grad(s, T, n_part); g.init_mass(s)
    g.count_mu_lam_grad = True
    g.copy_pos(s, 0)
    dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3))
    dpos[:, 2] = -1e-4 if name != "balancing" else 5e-5
    drot[:, 1] = 2e-3
    for f in range(1, T):
        s.action(f, dpos, drot)
        s.time_step(projection_query, f)
        g.copy_pos(s, f)
    g.get_loss_slide(s)
    for k in range(T - 1, 0, -1):
        g.transfer_grad(k, s, projection_query)
    force = s.gather_force() if hasattr(s, "gather_force") else None
    out = dict(pos_grad=g.pos_grad.to_numpy().copy(), params=np.array([g.grad_kb.value, g.grad_mu.value, g.grad_lam.value]))
    if force is not None:
        out["force"] = np.asarray(force, dtype=np.float64).copy()
    del g, s
    gc.collect()
    return out


@pytest.mark.parametrize("name", ["balancing", "folding"])
def test_system_identification_sweep_gives_the_same_bits(name):
    """tsl_param_grad / tsl_elastic_force: staged element contributions + ticketed dot products (no f64 atomics) -- two runs, the same bits"""
    a = _sysid_sweep(name); b = _sysid_sweep(name)
    assert np.abs(a["params"][:2]).min() > 0
    for k in a:
        assert np.array_equal(a[k], b[k]), f"{name}: {k} differs between two runs: {a[k] if a[k].size < 8 else ''} {b[k] if b[k].size < 8 else ''}"


def test_rollout_bits_do_not_depend_on_the_inversion_path():
    """the same rollout with the pivot blocks inverted by the persistent dataflow launches (default, if this process holds the device's
    token), by one launch per block step ("direct_flow" 0), and without the LDS kernel either: the same bits in tape and gradients (round 5:
    every path forms the same products in the same order)"""
    a = rollout("balancing", 4, grid=96)
    fl = rollout.last_flow_launches
    b = rollout("balancing", 4, grid=96, params=(("direct_flow", 0),))
    assert rollout.last_flow_launches == 0
    c = rollout("balancing", 4, grid=96, params=(("direct_flow", 0), ("direct_small_rounds", 1)))
    assert fl > 0, "the default rollout did not run a dataflow launch (token held by a context that is still alive?)"
    for k in a:
        assert np.array_equal(a[k], b[k]) and np.array_equal(a[k], c[k]), f"{k}: max |d| = {np.abs(a[k] - b[k]).max():.3e} / {np.abs(a[k] - c[k]).max():.3e}"


def test_cfg4_steps_are_reproducible_at_full_size():
    a = rollout("balancing", 4, grid=224)
    b = rollout("balancing", 4, grid=224)
    for k in a:
        assert np.array_equal(a[k], b[k]), f"cfg4: {k} differs between two runs (max |d| = {np.abs(a[k] - b[k]).max():.3e})"


@pytest.mark.parametrize("name,grid,T", [("balancing", 224, 11), ("folding", 200, 11)])
def test_golden_rollout_bits(name, grid, T):
    """cfg4 / cfg3 at their stated sizes, 10 driven steps + the reverse sweep against the committed vectors: the digests of the whole tape
    and gradients, a sample of 512 vertices per tape step, and gripper_grad, bit for bit"""
    path = os.path.join(GOLD, f"det_{name}_{grid}.npz")
    if not os.path.exists(path):
        pytest.skip("golden vector not generated yet (tests/golden/gen_golden_gpu.py)")
    gold = np.load(path)
    r = rollout(name, T, grid=grid)
    sel = gold["sample_idx"]
    assert np.array_equal(r["gripper_grad"], gold["gripper_grad"])
    assert np.array_equal(r["pos_buffer"][:, sel], gold["pos_sample"])
    assert np.array_equal(r["pos_grad"][:, sel], gold["pos_grad_sample"])
    assert np.array_equal(r["stats"], gold["stats"])
    assert digest(r["pos_buffer"]) == str(gold["sha_pos_buffer"]) and digest(r["pos_grad"]) == str(gold["sha_pos_grad"])
