"""Scene groups (tsl_group_*, thinshelllab_amd/scene_group.py): several scenes of one GPU stepped in lock step, the sparse direct solves of all
members as ONE factorisation and ONE first application of the merged plan.  The statement under test: a member's rollout -- tape, solver
statistics, gradients of the reverse sweep that follows -- is BIT-IDENTICAL to the rollout of the same scene stepped alone
(training/trajopt_*.py of the reference rolls out independent copies of one scene; nothing may couple them)."""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _make(name, grid, amp):
    if name == "balancing":
        from thinshelllab_amd.task_scene.Scene_balancing import Scene
        s = Scene(cloth_size=0.12 * grid / 224 if grid > 100 else 0.06, cloth_N=grid, cloth_M=grid)
    else:
        from thinshelllab_amd.task_scene.Scene_folding import Scene
        s = Scene(cloth_size=0.1, cloth_N=grid, cloth_M=grid // 2)
    s.init_all()
    s.mu_cloth_elastic[None] = 5.0
    s.prev_pos.copy_from(s.pos)
    s._ensure_ctx().set_param("direct", 1)
    s._test_name, s._test_amp = name, amp
    return s


def _drive(s, f):
    n_part = s.gripper.n_part
    dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3))
    if s._test_name == "balancing":
        dpos[:, 2] = np.array([1e-4, -1e-4][:n_part]) * s._test_amp
    else:
        dpos[:, 2] = -2e-4 * s._test_amp
    s.action(f, dpos, drot)


def _rollout(specs, T, grouped, group_adjoint=False):
    """the scenes of `specs` for T - 1 driven steps and the reverse sweep, stepped together (grouped) or one after the other; group_adjoint: the
    reverse sweep in lock step as well (SceneGroup.transfer_grad), otherwise on every member's own path"""
    from thinshelllab_amd.engine.analytic_grad_single import Grad
    from thinshelllab_amd.engine.geometry import projection_query
    from thinshelllab_amd.scene_group import SceneGroup
    scenes = [_make(*sp) for sp in specs]
    grads = []
    for s in scenes:
        g = Grad(s, T, s.gripper.n_part); g.init_mass(s); g.copy_pos(s, 0)
        grads.append(g)
    stats = [[] for _ in scenes]
    info = None
    if grouped:
        G = SceneGroup(scenes)
        for f in range(1, T):
            for s in scenes:
                _drive(s, f)
            sts = G.time_step(projection_query, f)
            for i, (s, g) in enumerate(zip(scenes, grads)):
                g.copy_pos(s, f)
                stats[i].append(sts[i])
        info = G.info()
    else:
        for i, (s, g) in enumerate(zip(scenes, grads)):
            for f in range(1, T):
                _drive(s, f)
                stats[i].append(s.time_step(projection_query, f))
                g.copy_pos(s, f)
    out = []
    adj = [[] for _ in scenes]
    akeys = ("flag", "iters", "restarts", "rel_residual", "method")
    for s, g in zip(scenes, grads):
        if s._test_name == "balancing":
            g.get_loss_balance(s)
        else:
            g.get_loss_fold(s, 1.0, -1.0, rows=s.fold_rows())
    if grouped and group_adjoint:
        for k in range(T - 1, 0, -1):
            G.transfer_grad(k, grads, projection_query)
            for i, g in enumerate(grads):
                adj[i].append([g.last_stats[a] for a in akeys])
        info = G.info()
    else:
        for i, (s, g) in enumerate(zip(scenes, grads)):
            for k in range(T - 1, 0, -1):
                g.transfer_grad(k, s, projection_query)
                adj[i].append([g.last_stats[a] for a in akeys])
    for i, (s, g) in enumerate(zip(scenes, grads)):
        keys = ("nc", "newton_iters", "ls_evals", "cg_iters", "unconverged", "energy", "last_delta", "last_alpha", "max_rel_residual")
        out.append(dict(pos_buffer=g.pos_buffer.to_numpy().copy(), pos_grad=g.pos_grad.to_numpy().copy(), gripper_grad=g.gripper_grad.to_numpy().copy(),
                        angleref_grad=g.angleref_grad.to_numpy().copy(), end_pos=s.pos.to_numpy().copy(),
                        stats=np.array([[st[k] for k in keys] for st in stats[i]], dtype=np.float64), adjoint_stats=np.array(adj[i], dtype=np.float64)))
    if grouped:
        G.close()
    del scenes, grads
    gc.collect()
    return out, info


@pytest.mark.parametrize("specs", [
    [("balancing", 48, 1.0), ("balancing", 48, 1.3)],                                   # one topology, two trajectories
    [("balancing", 48, 1.0), ("folding", 60, 1.0), ("balancing", 64, 0.7)],             # different topologies, tree depths and contact sets
])
@pytest.mark.parametrize("group_adjoint", [False, True])
def test_group_members_match_their_single_scene_rollouts_bit_for_bit(specs, group_adjoint):
    T = 4
    single, _ = _rollout(specs, T, grouped=False)
    group, info = _rollout(specs, T, grouped=True, group_adjoint=group_adjoint)
    assert info["merged_factorizations"] > 0 and info["merged_applications"] == info["merged_factorizations"], info
    for i, (a, b) in enumerate(zip(single, group)):
        assert a["stats"][:, 0].max() > 0, "no contact in the rollout"
        for k in a:
            assert np.array_equal(a[k], b[k]), f"scene {i} {specs[i]}: {k} differs (max |d| = {np.abs(a[k] - b[k]).max():.3e})"


def test_group_of_one_and_regrouping():
    """a group of a single scene is that scene; members can be regrouped after a group is closed (their buffers come back)"""
    from thinshelllab_amd.engine.geometry import projection_query
    from thinshelllab_amd.scene_group import SceneGroup
    a, b = _make("balancing", 48, 1.0), _make("balancing", 48, 1.0)
    G = SceneGroup([a])
    for f in range(1, 3):
        _drive(a, f); _drive(b, f)
        G.time_step(projection_query, f)
        b.time_step(projection_query, f)
    assert np.array_equal(a.pos.to_numpy(), b.pos.to_numpy())
    G.close()
    G2 = SceneGroup([a, b])
    for f in range(3, 5):
        _drive(a, f); _drive(b, f)
        sts = G2.time_step(projection_query, f)
    assert np.array_equal(a.pos.to_numpy(), b.pos.to_numpy()) and sts[0]["newton_iters"] == sts[1]["newton_iters"]
    G2.close()
    _drive(a, 5); a.time_step(projection_query, 5)     # ... and step alone again
    assert np.isfinite(a.pos.to_numpy()).all()


def test_group_steps_after_a_reverse_sweep_use_cached_plans_correctly():
    """the reverse sweep revisits the constraint sets of the forward rollout, so the members' plans come back from their plan caches; the next
    group step merges such plans (round 5: a cached plan came back without its host block lists and the merge produced NaN factors, hidden by the
    members' own fallback).  Forward, reverse sweep, forward again -- grouped against one after the other, bit for bit, and no member may leave
    the merged solve more than a few times"""
    from thinshelllab_amd.engine.analytic_grad_single import Grad
    from thinshelllab_amd.engine.geometry import projection_query
    from thinshelllab_amd.scene_group import SceneGroup
    res = {}
    for grouped in (False, True):
        scenes = [_make("balancing", 48, 1.0), _make("balancing", 48, 1.3)]
        T = 4
        grads = []
        for s in scenes:
            g = Grad(s, T, s.gripper.n_part); g.init_mass(s); g.copy_pos(s, 0); grads.append(g)
        G = SceneGroup(scenes) if grouped else None

        def forward(f0, f1):
            for f in range(f0, f1):
                for s in scenes:
                    _drive(s, f)
                if grouped:
                    G.time_step(projection_query, f)
                else:
                    for s in scenes:
                        s.time_step(projection_query, f)
                for s, g in zip(scenes, grads):
                    if f < T:
                        g.copy_pos(s, f)
        forward(1, T)
        end = [s.pos.to_numpy().copy() for s in scenes]
        for s, g in zip(scenes, grads):
            g.get_loss_balance(s)
            for k in range(T - 1, 0, -1):
                g.transfer_grad(k, s, projection_query)
        for s, x in zip(scenes, end):      # back to the end of the rollout, then on
            s.pos.from_numpy(x); s.prev_pos.from_numpy(x)
        forward(T, T + 2)
        res[grouped] = [s.pos.to_numpy().copy() for s in scenes] + [g.pos_grad.to_numpy().copy() for g in grads]
        if grouped:
            info = G.info()
            assert info["member_solves_on_own_path"] <= 6, info
            G.close()
        del scenes, grads
        gc.collect()
    for a, b in zip(res[False], res[True]):
        assert np.isfinite(a).all() and np.array_equal(a, b), np.abs(a - b).max()


def test_group_at_cfg4_size_bit_for_bit():
    """BASELINE configs[4]'s scene at its stated size (224 x 224 cloth, 100,352 triangles, ball + four pads) twice with different drives: two driven steps
    and the reverse sweep as a scene group (forward and reverse in lock step) against one scene after the other -- same tapes, gradients and solver
    statistics bit for bit; the merged solves went through the dataflow inversions (the group holds the device's token)"""
    specs = [("balancing", 224, 1.0), ("balancing", 224, 1.3)]
    T = 3
    single, _ = _rollout(specs, T, grouped=False)
    group, info = _rollout(specs, T, grouped=True, group_adjoint=True)
    assert info["merged_factorizations"] >= 2 * (T - 1) and info["member_solves_on_own_path"] <= 8, info
    for i, (a, b) in enumerate(zip(single, group)):
        assert a["stats"][:, 0].max() > 50, "too few contacts for the cfg4 scene"
        for k in a:
            assert np.array_equal(a[k], b[k]), f"scene {i}: {k} differs (max |d| = {np.abs(a[k] - b[k]).max():.3e})"


def test_group_dataflow_abort_refactorises_the_merged_plan():
    """a dataflow launch of the MERGED factorisation that loses a flag leaves garbage factors for every member, and the members' own abort
    branch looks at counters the merged launch never touched.  "ds_dbg" 21 (read from member 0 at every merged solve) forces the group's abort branch:
    the merged factorisation must run once more on the launch-per-block-step path, the step must still equal the single-scene step bit for bit, and the
    group must stay off the dataflow path afterwards"""
    from thinshelllab_amd.engine.geometry import projection_query
    from thinshelllab_amd.scene_group import SceneGroup
    a, b, ref = _make("balancing", 96, 1.0), _make("balancing", 96, 1.3), _make("balancing", 96, 1.0)
    G = SceneGroup([a, b])
    _drive(a, 1); _drive(b, 1); _drive(ref, 1)
    G.time_step(projection_query, 1)
    ref.time_step(projection_query, 1)
    i0 = G.info()
    assert i0["merged_flow_launches"] > 0 and i0["merged_flow_aborts"] == 0, i0
    assert np.array_equal(a.pos.to_numpy(), ref.pos.to_numpy())
    a._ensure_ctx().set_param("ds_dbg", 21)
    _drive(a, 2); _drive(b, 2); _drive(ref, 2)
    sts = G.time_step(projection_query, 2)
    a._ensure_ctx().set_param("ds_dbg", 0)
    ref.time_step(projection_query, 2)
    i1 = G.info()
    assert i1["merged_flow_aborts"] == 1, i1
    assert sts[0]["unconverged"] == 0 and sts[1]["unconverged"] == 0
    assert np.array_equal(a.pos.to_numpy(), ref.pos.to_numpy()), np.abs(a.pos.to_numpy() - ref.pos.to_numpy()).max()
    _drive(a, 3); _drive(b, 3)
    G.time_step(projection_query, 3)
    assert G.info()["merged_flow_launches"] == i1["merged_flow_launches"]     # the group stays on the block-step path
    G.close()
