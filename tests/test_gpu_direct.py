"""Sparse direct preconditioner (multifrontal LU on the GPU, thinshelllab_amd/csrc/k_direct.hpp) through the C ABI:
tsl_solve with "direct" = 1 against scipy's sparse LU of the exported operator -- the reference solves every system with a direct
solver (sparse_solver.py:85-105), so this is the parity statement for the solve itself."""
import numpy as np
import pytest
import torch

from helpers import rel_err

pytestmark = pytest.mark.gpu


def _drape(N, M, amp, seed=0):
    from thinshelllab_amd.task_scene.Scene_drape import Scene
    s = Scene(cloth_size=0.1 / 15 * N, N=N, M=M, Kb=100.0, k_angle=3.14)
    s.init_all()
    rng = np.random.default_rng(seed)
    x = s.pos.to_numpy()
    x += rng.normal(0, amp, x.shape)
    s.pos.from_numpy(x)
    s.prev_pos.from_numpy(x)
    return s


@pytest.mark.parametrize("N,M,leaf", [(12, 12, 8), (48, 32, 32), (33, 70, 16)])
def test_direct_solve_matches_sparse_lu(N, M, leaf):
    import scipy.sparse.linalg as spl
    s = _drape(N, M, 5e-5)
    ctx = s._ensure_ctx()
    ctx.set_param("direct", 1); ctx.set_param("direct_leaf", leaf)
    s.compute_residual_and_Hessian(spd=True)
    b = s.F.to_torch().clone()
    x, st = ctx.solve(b)
    H = ctx.operator_csr().tocsc()
    xs = spl.splu(H).solve(b.cpu().numpy())
    assert st["flag"] == 0 and st["method"] == 4 and st["iters"] <= 3, st
    assert st["rel_residual"] <= 1e-10
    assert rel_err(x.cpu().numpy(), xs) < 1e-9
    # same answer as the iterative hierarchy
    ctx.set_param("direct", 0)
    x2, st2 = ctx.solve(b)
    assert st2["method"] == 0 and rel_err(x2.cpu().numpy(), xs) < 1e-7


def test_direct_solve_indefinite_operator():
    """un-projected Hessian of a strongly perturbed cloth (adjoint systems): indefinite, no pivoting inside the factorisation"""
    import scipy.sparse.linalg as spl
    s = _drape(40, 40, 2e-4, seed=4)
    ctx = s._ensure_ctx()
    ctx.set_param("direct", 1)
    s.compute_Hessian(spd=False)
    H = ctx.operator_csr().tocsc()
    w = spl.eigsh(0.5 * (H + H.T), k=4, which="SA", return_eigenvectors=False)
    assert w.min() < 0
    b = torch.as_tensor(np.random.default_rng(3).normal(size=s.tot_NV * 3), device=s.device)
    b = b * torch.as_tensor(s.frozen.to_numpy().reshape(-1) == 0, device=s.device)
    x, st = ctx.solve(b)
    assert st["flag"] == 0 and st["method"] == 4, st
    r = b.cpu().numpy() - H @ x.cpu().numpy()
    assert np.linalg.norm(r) <= 1e-9 * np.linalg.norm(b.cpu().numpy())
    xs = spl.splu(H).solve(b.cpu().numpy())
    assert rel_err(x.cpu().numpy(), xs) < 1e-6


def test_direct_solve_with_contacts_and_bodies():
    """balancing scene (cloth on ball + 4 pads) at 48 x 48 after two driven steps: bodies are dense supernodes, the contact
    constraints extend the elimination tree; forward (projected) and adjoint (un-projected) operators"""
    import scipy.sparse.linalg as spl
    from thinshelllab_amd.engine.geometry import projection_query
    from thinshelllab_amd.task_scene.Scene_balancing import Scene
    s = Scene(cloth_size=0.06, cloth_N=48, cloth_M=48)
    s.init_all()
    s.mu_cloth_elastic[None] = 5.0
    s.prev_pos.copy_from(s.pos)
    ctx = s._ensure_ctx()
    ctx.set_param("direct", 1)
    n_part = s.gripper.n_part
    dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3)); dpos[:, 2] = [1e-4, -1e-4][:n_part]
    for f in range(1, 3):
        s.action(f, dpos, drot)
        st = s.time_step(projection_query, f)
        assert st["unconverged"] == 0 and st["factorizations"] == st["solves"], st
        assert st["max_rel_residual"] < 1e-8, st
    assert st["nc"] > 0
    projection_query(s)
    free = torch.as_tensor(s.frozen.to_numpy().reshape(-1) == 0, device=s.device)
    for spd in (True, False):
        if spd:
            s.compute_residual_and_Hessian(spd=True)
            b = s.F.to_torch().clone()
        else:
            s.compute_Hessian(spd=False)
            b = torch.as_tensor(np.random.default_rng(5).normal(size=s.tot_NV * 3), device=s.device) * free
        x, stx = ctx.solve(b)
        H = ctx.operator_csr().tocsc()
        xs = spl.splu(H).solve(b.cpu().numpy())
        assert stx["flag"] == 0 and stx["method"] == 4 and stx["iters"] <= 4, stx
        assert rel_err(x.cpu().numpy(), xs) < 1e-7, (spd, stx)
