"""Parity at BASELINE.json's full sizes through size-independent properties (the oracle needs minutes per step there):
residual of the linear solves against the exported operator (scipy), linearity of the solve, energy decrease and Newton
convergence of the steps, sign / definiteness bookkeeping of the adjoint systems.  cfg2 (71x71 drape), cfg3 (200x100 folding),
cfg4 (224x224 cloth on ball + 4 pads), SURVEY.md section 8d."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
# The T = 50 rollout is bit-reproducible since round 4 (deterministic assembly, tests/test_gpu_determinism.py): it ends with NO flagged
# solve (0 of 2456 forward, 0 of 50 adjoint).  Round 3 had widened these bounds to 20 to absorb run-to-run chaos; they
# are back to a small margin for code changes that move the rollout, not for noise.
FLAGGED_STEPS_MAX = 2
ADJ_FLAGGED_MAX = 2


def _free_mask(s):
    return torch.as_tensor(s.frozen.to_numpy().reshape(-1) == 0, device=s.device)


def _residual(ctx, b, x):
    H = ctx.operator_csr()
    bn = b.cpu().numpy()
    return np.linalg.norm(bn - H @ x.cpu().numpy()) / np.linalg.norm(bn), H


def test_cfg2_drape_71_forward_steps():
    """SURVEY 8d cfg2 at its full length: 50 forward steps of the pinned 71 x 71 drape"""
    from thinshelllab_amd.task_scene.Scene_drape import Scene
    s = Scene(cloth_size=0.1 / 15 * 71, N=71, M=71, Kb=100.0, k_angle=3.14, perturb=1e-4)
    s.init_all()
    E_prev = None
    for f in range(1, 51):
        E0 = s.compute_energy()
        st = s.time_step(None, f)
        if f <= 5:  # the first steps converge below the reference's stop rule; later the falling sheet runs into the cap of 50 like the reference
            assert st["newton_iters"] < 50 and st["last_delta"] < 1e-7, st
        assert st["fallback"] == 0 and st["unconverged"] == 0 and st["max_rel_residual"] < 1e-9, (f, st)
        assert np.isfinite(s.pos.to_numpy()).all()
    # linear solve of the last state: residual against the exported operator, and linearity
    s.compute_residual_and_Hessian(spd=True)
    ctx = s._ctx
    b1 = s.F.to_torch().clone()
    rng = np.random.default_rng(0)
    b2 = torch.as_tensor(rng.normal(size=b1.shape), device=b1.device) * _free_mask(s)
    x1, st1 = ctx.solve(b1); x2, _ = ctx.solve(b2); x3, _ = ctx.solve(2.5 * b1 - b2)
    r1, H = _residual(ctx, b1, x1)
    assert st1["flag"] == 0 and r1 < 1e-9
    assert np.abs((2.5 * x1 - x2 - x3).cpu().numpy()).max() <= 1e-7 * np.abs(x3.cpu().numpy()).max()
    assert abs(H - H.T).max() <= 1e-6 * abs(H).max()                           # symmetric up to the area-Hessian quirk


def test_cfg3_folding_200x100_step_and_adjoint():
    from thinshelllab_amd.task_scene.Scene_folding import Scene
    from thinshelllab_amd.engine.analytic_grad_single import Grad
    from thinshelllab_amd.engine.geometry import projection_query
    s = Scene(cloth_size=0.1, cloth_N=200, cloth_M=100)
    s.cloths[0].Kb[None] = 400.0
    s.init_all()
    s.mu_cloth_elastic[None] = 5.0
    s.prev_pos.copy_from(s.pos)
    assert s.cloths[0].NF == 40000
    T = 51   # SURVEY 8d's T = 50: ten steps of -z, then +x
    n_part = s.gripper.n_part
    g = Grad(s, T, n_part); g.init_mass(s)
    g.copy_pos(s, 0)
    for f in range(1, T):
        dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3))
        if f <= 10:
            dpos[:, 2] = -2e-4      # pad moves -z 2e-4 m per step for ten steps,
        else:
            dpos[:, 0] = 2e-4       # then +x 2e-4 m per step
        s.action(f, dpos, drot)
        st = s.time_step(projection_query, f)
        g.copy_pos(s, f)
        assert np.isfinite(s.pos.to_numpy()).all()
        assert st["nc"] >= 0 and st["newton_iters"] >= 1
        assert st["unconverged"] == 0 and st["max_rel_residual"] < 1e-8, (f, st)
    g.get_loss_fold(s, 1.0, -1.0)
    for st_ in range(T - 1, 0, -1):
        g.transfer_grad(st_, s, projection_query)   # raises on an unconverged solve
        ls = g.last_stats
        assert ls["flag"] in (0, 1) and (ls["rel_residual"] < 1e-8 or ls["backward_error"] < 1e-13), (st_, ls)
    assert np.isfinite(g.pos_grad.to_numpy()).all() and np.abs(g.gripper_grad.to_numpy()).max() > 0
    # the adjoint operator of the last processed step: independent residual check of a solve with the un-projected Hessian
    s.compute_Hessian(spd=False)
    ctx = s._ctx
    b = torch.as_tensor(np.random.default_rng(1).normal(size=3 * s.tot_NV), device=s.device) * _free_mask(s)
    x, stx = ctx.solve(b)
    r, _ = _residual(ctx, b, x)
    assert stx["flag"] in (0, 1) and r < 1e-8, (stx, r)


def test_cfg4_balancing_224_rollout_and_adjoint():
    """cfg4 at the bench's drive for 14 steps (contacts build up to ~200) and the complete reverse sweep: every linear solve of the
    forward Newton loops and of the adjoint steps converged (the reference's spsolve is exact every time), and one adjoint solution
    is checked independently against the exported un-projected operator."""
    import scipy.sparse.linalg as spl
    from thinshelllab_amd.engine.analytic_grad_single import Grad
    from thinshelllab_amd.engine.geometry import projection_query
    from thinshelllab_amd.task_scene.Scene_balancing import Scene
    s = Scene(cloth_size=0.12, cloth_N=224, cloth_M=224)
    s.init_all()
    s.mu_cloth_elastic[None] = 5.0
    s.prev_pos.copy_from(s.pos)
    T = 15
    n_part = s.gripper.n_part
    g = Grad(s, T, n_part); g.init_mass(s)
    g.copy_pos(s, 0)
    dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3)); dpos[:, 2] = [1e-4, -1e-4][:n_part]
    nc_max = 0
    for f in range(1, T):
        s.action(f, dpos, drot)
        st = s.time_step(projection_query, f)
        g.copy_pos(s, f)
        nc_max = max(nc_max, st["nc"])
        assert np.isfinite(s.pos.to_numpy()).all()
        assert st["unconverged"] == 0 and st["fallback"] == 0, (f, st)
        assert st["max_rel_residual"] < 1e-8 and st["energy"] == st["energy"], (f, st)
    assert nc_max > 100
    g.get_loss_balance(s)
    worst = 0.0
    for st_ in range(T - 1, 0, -1):
        g.transfer_grad(st_, s, projection_query)   # raises on an unconverged solve
        ls = g.last_stats
        assert ls["flag"] == 0 and ls["method"] in (0, 4) and ls["iters"] <= 12 + 64, (st_, ls)   # (a re-probe of the iterative hierarchy adds <= 60)
        # near-singular un-projected operators (|H| |x| / |b| up to 1e9 late in the sweep): the relative residual a backward-stable
        # direct solve -- the reference's spsolve -- can reach is eps |H| |x| / |b|; those solves are accepted on their normwise
        # backward error (attained = 1), everything else on rel_residual <= cg_tol
        assert ls["rel_residual"] < 1e-8 or (ls["attained"] == 1 and ls["backward_error"] < 1e-14), (st_, ls)
        worst = max(worst, ls["rel_residual"])
    assert np.isfinite(g.pos_grad.to_numpy()).all() and np.abs(g.gripper_grad.to_numpy()).max() > 0
    # independent check: the un-projected operator of the current state, a random right-hand side on the free dofs, scipy's sparse LU
    s.compute_Hessian(spd=False)
    ctx = s._ctx
    b = torch.as_tensor(np.random.default_rng(2).normal(size=3 * s.tot_NV), device=s.device) * _free_mask(s)
    x, stx = ctx.solve(b)
    r, H = _residual(ctx, b, x)
    assert stx["flag"] == 0 and r < 1e-9, (stx, r)
    xs = spl.splu(H.tocsc()).solve(b.cpu().numpy())
    assert np.linalg.norm(x.cpu().numpy() - xs) <= 1e-5 * np.linalg.norm(xs)


def test_cfg4_balancing_224_T50_rollout():
    """SURVEY 8d cfg4 as specified: T = 50, zero trajectory for 10 steps, then +-z 1e-4 m per step, forward rollout + reverse sweep.
    The refined sheet (0.58 kg) and the ball hang on the pads, which are crushed flat in the second half of the rollout (det F of
    their elements through zero, exp_crush.py): the tactile material is evaluated from cofactors there (finite), but contact blocks
    on collapsed surface triangles reach 1e14..1e20 in a few Newton iterates -- operators on which scipy's pivoted SuperLU itself
    stops at residuals of 1e-7..1e-3 (profiles/r02c_cfg4_T50_degenerate_systems.txt).  The engine has to report those solves
    (unconverged > 0, never silently), keep the state finite, and converge everything else.  The rollout itself is reproducible since
    round 4 (no f64 atomics on the step); which steps are affected still moves with every change of the engine's rounding (the states
    are chaotic), so the test bounds their number instead of fixing it."""
    from thinshelllab_amd.engine.analytic_grad_single import Grad
    from thinshelllab_amd.engine.geometry import projection_query
    from thinshelllab_amd.task_scene.Scene_balancing import Scene
    s = Scene(cloth_size=0.12, cloth_N=224, cloth_M=224)
    s.init_all()
    s.mu_cloth_elastic[None] = 5.0
    s.prev_pos.copy_from(s.pos)
    T = 51
    n_part = s.gripper.n_part
    g = Grad(s, T, n_part); g.init_mass(s)
    g.allow_unconverged = True
    g.copy_pos(s, 0)
    zero = np.zeros((n_part, 3)); dpos = zero.copy(); dpos[:, 2] = [1e-4, -1e-4][:n_part]
    flagged_steps, solves, flagged = 0, 0, 0
    for f in range(1, T):
        s.action(f, zero if f <= 10 else dpos, zero)
        st = s.time_step(projection_query, f)
        g.copy_pos(s, f)
        assert np.isfinite(s.pos.to_numpy()).all(), f
        assert 0 <= st["unconverged"] <= st["newton_iters"], (f, st)
        if st["unconverged"] == 0:
            assert st["max_rel_residual"] < 1e-8, (f, st)
        else:
            flagged_steps += 1
        solves += st["newton_iters"]; flagged += st["unconverged"]
        if f <= 10:   # the idle phase settles below the reference's stop rule after the first steps
            assert st["unconverged"] == 0, (f, st)
    assert flagged_steps <= FLAGGED_STEPS_MAX and flagged <= 0.05 * solves, (flagged_steps, flagged, solves)
    g.get_loss_balance(s)
    adj_flagged = 0
    for st_ in range(T - 1, 0, -1):
        g.transfer_grad(st_, s, projection_query)
        ls = g.last_stats
        assert ls["flag"] in (0, 1, 3), ls
        if ls["flag"] == 3:
            adj_flagged += 1
        else:
            assert ls["rel_residual"] < 1e-8 or (ls["attained"] == 1 and ls["backward_error"] < 1e-12), (st_, ls)
    print(f"T = 50 rollout: {flagged_steps} flagged steps, {flagged} of {solves} forward solves flagged, {adj_flagged} adjoint solves flagged")
    assert adj_flagged <= ADJ_FLAGGED_MAX, adj_flagged
    assert np.isfinite(g.pos_grad.to_numpy()).all() and np.isfinite(g.gripper_grad.to_numpy()).all()


def test_cfg4_balancing_224_contact_solve():
    """the iterative hierarchy (multigrid PCG), which stays the solver of coarse scenes and the fallback of the direct path"""
    from thinshelllab_amd.task_scene.Scene_balancing import Scene
    from thinshelllab_amd.engine.geometry import projection_query
    s = Scene(cloth_size=0.12, cloth_N=224, cloth_M=224)
    s.init_all()
    s.mu_cloth_elastic[None] = 5.0
    s.prev_pos.copy_from(s.pos)
    s._ensure_ctx().set_param("direct", 0)
    assert s.cloths[0].NF == 100352
    n_part = s.gripper.n_part
    dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3)); dpos[:, 2] = [1e-4, -1e-4][:n_part]
    nc = 0
    for f in range(1, 3):
        s.action(f, dpos, drot)
        E0 = s.compute_energy()
        st = s.time_step(projection_query, f)
        nc = st["nc"]
        assert np.isfinite(s.pos.to_numpy()).all()
    assert nc > 0
    projection_query(s)
    s.compute_residual_and_Hessian(spd=True)
    ctx = s._ctx
    b = s.F.to_torch().clone()
    x, stx = ctx.solve(b)
    r, H = _residual(ctx, b, x)
    assert r < 1e-7, (stx, r)
    # the step direction is a descent direction of the incremental potential: E(x - a p) < E(x) for a small a
    Ecur = s.compute_energy()
    pos0 = s.pos.to_torch().clone()
    s.pos.t.copy_((pos0.reshape(-1) - 1e-3 * x).reshape(pos0.shape))
    Enew = s.compute_energy()
    s.pos.t.copy_(pos0)
    assert Enew < Ecur
