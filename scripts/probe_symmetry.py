"""Probe (not part of the product): is the solve operator symmetric?  |H - H^T| relative to |H|, forward (SPD-projected) and adjoint (un-projected)
operator, on balancing (ball + pads, friction) and folding (cloth on a table-less fold: cloth-elastic contact) scenes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from thinshelllab_amd.engine.geometry import projection_query
from thinshelllab_amd.engine.analytic_grad_single import Grad


def asym(A):
    D = (A - A.T).tocoo()
    return (np.abs(D.data).max() if D.nnz else 0.0), np.abs(A.data).max()


def run(name, grid):
    if name == "balancing":
        from thinshelllab_amd.task_scene.Scene_balancing import Scene
        s = Scene(cloth_size=0.12 * grid / 224 if grid > 100 else 0.06, cloth_N=grid, cloth_M=grid)
    else:
        from thinshelllab_amd.task_scene.Scene_folding import Scene
        s = Scene(cloth_size=0.1, cloth_N=grid, cloth_M=grid // 2)
    s.init_all(); s.mu_cloth_elastic[None] = 5.0; s.prev_pos.copy_from(s.pos)
    ctx = s._ensure_ctx(); ctx.set_param("direct", 1)
    n_part = s.gripper.n_part
    T = 5
    g = Grad(s, T, n_part); g.init_mass(s); g.copy_pos(s, 0)
    for f in range(1, T):
        dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3))
        if name == "balancing":
            dpos[:, 2] = np.array([1e-4, -1e-4][:n_part])
        else:
            dpos[:, 2] = -2e-4
        s.action(f, dpos, drot)
        st = s.time_step(projection_query, f)
        g.copy_pos(s, f)
    a, m = asym(ctx.operator_csr())
    print(f"{name} {grid}: nc {st['nc']}  forward operator (last Newton iteration): max |H - H^T| = {a:.3e}, max |H| = {m:.3e}, ratio {a / m:.2e}", flush=True)
    if name == "balancing":
        g.get_loss_balance(s)
    else:
        g.get_loss_fold(s, 1.0, -1.0, rows=s.fold_rows())
    g.transfer_grad(T - 1, s, projection_query)
    a, m = asym(ctx.operator_csr())
    print(f"{name} {grid}: adjoint operator (un-projected): max |H - H^T| = {a:.3e}, max |H| = {m:.3e}, ratio {a / m:.2e}", flush=True)


run("balancing", 48)
run("folding", 60)
run("balancing", 96)
