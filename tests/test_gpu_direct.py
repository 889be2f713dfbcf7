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


def test_direct_dataflow_chains_agree_with_block_step_launches():
    """"direct_flow": the Gauss-Jordan chains of the batches that are alone on their tree level as ONE persistent launch each
    (k_ds_gj_flow: a super-tile of 2 x 2 tiles per workgroup in registers, block steps ordered by point-to-point flags) against one
    launch per 32 pivots (k_ds_gj_step) and the LDS kernel (k_ds_inv_small), on a plan whose upper levels have several fronts per
    batch (160 x 96 drape, leaves of 16 vertices).  Every path forms the same products in the same order and inverts its pivot tiles
    with the same routine: the solutions must be EQUAL BIT FOR BIT (round 5; round 4's paths differed in the last digits, which made
    the bits of a rollout depend on which path a neighbour context left free).  Against scipy's LU for the value itself; the class
    replay must show that the dataflow path is the one that ran."""
    import scipy.sparse.linalg as spl
    s = _drape(160, 96, 5e-5, seed=5)
    ctx = s._ensure_ctx()
    ctx.set_param("direct", 1); ctx.set_param("direct_leaf", 16)
    s.compute_residual_and_Hessian(spd=True)
    b = s.F.to_torch().clone()
    xs = spl.splu(ctx.operator_csr().tocsc()).solve(b.cpu().numpy())
    sols = {}
    for flow in (0, 1, 3):
        ctx.set_param("direct_flow", flow)
        s.compute_residual_and_Hessian(spd=True)      # fresh factors on the selected path
        for rep in range(3):                          # the flags carry the launch epoch: repeated factorisations reuse every slot
            x, st = ctx.solve(b.clone())
            assert st["flag"] == 0 and st["method"] == 4 and st["iters"] <= 3, (flow, st)
            assert rel_err(x.cpu().numpy(), xs) < 1e-9
            if rep:
                assert np.array_equal(x.cpu().numpy(), sols[flow]), (flow, rep)
            sols[flow] = x.cpu().numpy()
            s.compute_residual_and_Hessian(spd=True)
        launches = ctx.bench_direct(5, 2)["launches"]
        x, st = ctx.solve(b.clone())                  # (the replays invalidate the factors)
        assert (launches > 0) == (flow > 0), (flow, launches)
    assert np.array_equal(sols[1], sols[0]) and np.array_equal(sols[3], sols[0]), (np.abs(sols[1] - sols[0]).max(), np.abs(sols[3] - sols[0]).max())
    # ... and with the LDS kernel switched off as well (every batch on the block-step launches)
    ctx.set_param("direct_flow", 0); ctx.set_param("direct_small_rounds", 1)
    s.compute_residual_and_Hessian(spd=True)
    x0, st = ctx.solve(b.clone())
    assert st["flag"] == 0 and np.array_equal(x0.cpu().numpy(), sols[0])


@pytest.mark.parametrize("N,M,leaf", [(160, 96, 16), (96, 96, 64)])
def test_direct_lookahead_keeps_the_bits(N, M, leaf):
    """"direct_lookahead" (round 6): on the tree levels of one batch each the leading lead x lead block of every Schur complement -- all that the parent's pivot
    block receives -- is formed first (with the columns of G it reads), the parent's F11 is gathered and inverted on the engine stream while the other Schur tiles
    and the gather of F12 / F21 run on a low-priority side stream; bit 1 of the switch: the upward sweep of the solve's first application runs on that stream next to
    the chains of the last three levels (default 103: the leaf level next to the third chain from the top, two levels next to the one below the root, the others next to the root's).  Every tile, every panel entry and every sweep row is the same sum in the same order
    whichever launch forms it: solutions EQUAL BIT FOR BIT across the settings, over repeated factorisations (events and panels are reused), and right against scipy's LU."""
    import scipy.sparse.linalg as spl
    s = _drape(N, M, 5e-5, seed=9)
    ctx = s._ensure_ctx()
    ctx.set_param("direct", 1); ctx.set_param("direct_leaf", leaf)
    s.compute_residual_and_Hessian(spd=True)
    b = s.F.to_torch().clone()
    xs = spl.splu(ctx.operator_csr().tocsc()).solve(b.cpu().numpy())
    sols = {}
    for la in (0, 103, 1, 3, 11, 0, 103, 67):
        ctx.set_param("direct_lookahead", la)
        for rep in range(3):
            s.compute_residual_and_Hessian(spd=True)      # fresh factors
            x, st = ctx.solve(b.clone())
            assert st["flag"] == 0 and st["method"] == 4 and st["iters"] <= 3, (la, st)
            assert rel_err(x.cpu().numpy(), xs) < 1e-9
            if la in sols:
                assert np.array_equal(x.cpu().numpy(), sols[la]), (la, rep)
            sols[la] = x.cpu().numpy()
    for la in (1, 3, 11, 67, 103):
        assert np.array_equal(sols[0], sols[la]), (la, np.abs(sols[0] - sols[la]).max())
    # ... and with the chains on the launch-per-block-step path (a context without the device's dataflow token): the look-ahead and the eager sweep run all the same
    ctx.set_param("direct_flow", 0); ctx.set_param("direct_lookahead", 103)
    s.compute_residual_and_Hessian(spd=True)
    x, st = ctx.solve(b.clone())
    assert st["flag"] == 0 and np.array_equal(x.cpu().numpy(), sols[0])


@pytest.mark.parametrize("N,M,leaf", [(96, 96, 64), (70, 33, 16)])
def test_direct_dataflow_odd_tile_counts(N, M, leaf):
    """fronts whose tile count is odd (the last super-tile of k_ds_gj_flow is partial) and batches of many small fronts: dataflow launches
    against the block-step launches, bit for bit, and against scipy's LU"""
    import scipy.sparse.linalg as spl
    s = _drape(N, M, 5e-5, seed=21)
    ctx = s._ensure_ctx()
    ctx.set_param("direct", 1); ctx.set_param("direct_leaf", leaf)
    s.compute_residual_and_Hessian(spd=True)
    b = s.F.to_torch().clone()
    xs = spl.splu(ctx.operator_csr().tocsc()).solve(b.cpu().numpy())
    sols = {}
    for flow in (3, 0):
        ctx.set_param("direct_flow", flow)
        s.compute_residual_and_Hessian(spd=True)
        x, st = ctx.solve(b.clone())
        assert st["flag"] == 0 and rel_err(x.cpu().numpy(), xs) < 1e-9, (flow, st)
        sols[flow] = x.cpu().numpy()
    k = ctx.direct_counters()
    assert k["flow_launches"] > 0 and k["flow_aborts"] == 0, k
    assert np.array_equal(sols[3], sols[0])


def test_direct_dataflow_abort_refactorises_on_block_step_path():
    """A dataflow launch that loses a flag (not resident as a whole: another process on the device) leaves garbage factors; the solve
    must throw them away, refactorise on the launch-per-block-step path and still return the right answer (the
    refactorisation used to return early on `numeric_valid`).  "ds_dbg" 21 forces the abort branch after a converged solve."""
    import scipy.sparse.linalg as spl
    s = _drape(160, 96, 5e-5, seed=8)
    ctx = s._ensure_ctx()
    ctx.set_param("direct", 1); ctx.set_param("direct_leaf", 16)
    s.compute_residual_and_Hessian(spd=True)
    b = s.F.to_torch().clone()
    xs = spl.splu(ctx.operator_csr().tocsc()).solve(b.cpu().numpy())
    x, st = ctx.solve(b.clone())
    c0 = ctx.direct_counters(); f0 = ctx.direct_info()["factorizations"]
    assert c0["flow_launches"] > 0 and c0["flow_aborts"] == 0
    s.compute_residual_and_Hessian(spd=True)
    ctx.set_param("ds_dbg", 21)
    x, st = ctx.solve(b.clone())
    ctx.set_param("ds_dbg", 0)
    c1 = ctx.direct_counters(); f1 = ctx.direct_info()["factorizations"]
    assert c1["flow_aborts"] == 1 and f1 - f0 == 2, (c1, f0, f1)          # the aborted factorisation + the one that replaced it
    assert st["flag"] == 0 and rel_err(x.cpu().numpy(), xs) < 1e-9, st
    s.compute_residual_and_Hessian(spd=True)
    x, st = ctx.solve(b.clone())                                            # the context stays off the dataflow path
    assert ctx.direct_counters()["flow_launches"] == c1["flow_launches"] and st["flag"] == 0


def test_first_lost_dataflow_launch_costs_the_lookahead_only():
    """Graceful degradation (round 6): the first dataflow launch a context loses while it runs the look-ahead takes the LOOK-AHEAD away -- side-stream work arriving
    while a persistent launch is still being dispatched is the one cause known inside a process -- and keeps the dataflow path; the next loss takes the dataflow path.
    "ds_dbg" 22 forces the abort branch after a converged solve; the answers stay right throughout."""
    import scipy.sparse.linalg as spl
    s = _drape(160, 96, 5e-5, seed=8)
    ctx = s._ensure_ctx()
    ctx.set_param("direct", 1); ctx.set_param("direct_leaf", 16)
    s.compute_residual_and_Hessian(spd=True)
    b = s.F.to_torch().clone()
    xs = spl.splu(ctx.operator_csr().tocsc()).solve(b.cpu().numpy())
    x, st = ctx.solve(b.clone())
    n0 = ctx.direct_counters()["flow_launches"]
    assert n0 > 0
    for k in (1, 2):
        s.compute_residual_and_Hessian(spd=True)
        ctx.set_param("ds_dbg", 22)
        x, st = ctx.solve(b.clone())
        ctx.set_param("ds_dbg", 0)
        assert ctx.direct_counters()["flow_aborts"] == k and st["flag"] == 0 and rel_err(x.cpu().numpy(), xs) < 1e-9, (k, st)
        n1 = ctx.direct_counters()["flow_launches"]
        s.compute_residual_and_Hessian(spd=True)
        x, st = ctx.solve(b.clone())
        n2 = ctx.direct_counters()["flow_launches"]
        assert st["flag"] == 0 and rel_err(x.cpu().numpy(), xs) < 1e-9
        assert (n2 > n1) == (k == 1), (k, n1, n2)      # after the first loss the chains still run as dataflow launches, after the second they do not


def test_direct_wide_sweep_kernel_agrees():
    """"direct_gemv_wide_below": the sweep launches of the upper levels with four narrow workgroups per chunk (k_ds_gemv_wide) against the
    16-row kernel everywhere and against the narrow kernel everywhere"""
    s = _drape(96, 64, 5e-5, seed=7)
    ctx = s._ensure_ctx()
    ctx.set_param("direct", 1); ctx.set_param("direct_leaf", 16)
    s.compute_residual_and_Hessian(spd=True)
    b = s.F.to_torch().clone()
    sols = []
    for below in (0, 300, 1 << 30):
        ctx.set_param("direct_gemv_wide_below", below)
        x, st = ctx.solve(b.clone())
        assert st["flag"] == 0 and st["method"] == 4 and st["iters"] <= 3, (below, st)
        sols.append(x.cpu().numpy())
    assert rel_err(sols[1], sols[0]) < 1e-10 and rel_err(sols[2], sols[0]) < 1e-10


def test_direct_small_tile_g_kernel_agrees():
    """"direct_g32_below": G = W F12 in 32 x 32 tiles (k_ds_gemm_g32, the upper tree levels by default) against 64 x 64 tiles everywhere"""
    s = _drape(96, 64, 5e-5, seed=6)
    ctx = s._ensure_ctx()
    ctx.set_param("direct", 1); ctx.set_param("direct_leaf", 16)
    s.compute_residual_and_Hessian(spd=True)
    b = s.F.to_torch().clone()
    sols = []
    for g32 in (0, 1 << 30):
        ctx.set_param("direct_g32_below", g32)
        s.compute_residual_and_Hessian(spd=True)
        x, st = ctx.solve(b.clone())
        assert st["flag"] == 0 and st["method"] == 4 and st["iters"] <= 3, (g32, st)
        sols.append(x.cpu().numpy())
    assert rel_err(sols[1], sols[0]) < 1e-10


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


def test_unconverged_solve_fails_loudly():
    """the reference's spsolve is exact every time (sparse_solver.py:85-105): a solve that ends without convergence must not be
    consumed silently.  Direct path off and 5 iterations allowed: tsl_solve reports flag 3, tsl_step counts it, transfer_grad raises."""
    from thinshelllab_amd._lib import TslError
    from thinshelllab_amd.engine.analytic_grad_single import Grad
    s = _drape(24, 24, 1e-4)
    ctx = s._ensure_ctx()
    ctx.set_param("direct", 0); ctx.set_param("cg_maxit", 5); ctx.set_param("gmres_m", 5)
    s.compute_residual_and_Hessian(spd=True)
    x, st = ctx.solve(s.F.to_torch().clone())
    assert st["flag"] == 3 and st["rel_residual"] > 1e-8, st
    g = Grad(s, 3, 0); g.init_mass(s)
    g.copy_pos(s, 0)
    st1 = s.time_step(None, 1)
    assert st1["unconverged"] > 0 and st1["unconverged"] + st1["fallback"] <= st1["solves"], st1
    g.copy_pos(s, 1)
    g.pos_grad.t[1, :, 2] = 1.0
    with pytest.raises(TslError, match="not converged"):
        g.transfer_grad(1, s, None)
    # with the solver's normal budget the same sweep goes through
    ctx.set_param("cg_maxit", 200000); ctx.set_param("gmres_m", 300)
    g.transfer_grad(1, s, None)
    assert g.last_stats["flag"] in (0, 1)


def test_constraint_overflow_is_an_error():
    """constraints beyond max_n_constraints would be dropped in atomic-append order (non-deterministic physics): an error instead"""
    from thinshelllab_amd._lib import TslError
    from thinshelllab_amd.engine.geometry import projection_query
    from thinshelllab_amd.task_scene.Scene_balancing import Scene
    s = Scene(cloth_size=0.06, cloth_N=32, cloth_M=32)
    s.max_n_constraints = 3
    s.init_all()
    s.prev_pos.copy_from(s.pos)
    n_part = s.gripper.n_part
    dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3)); dpos[:, 2] = [1e-4, -1e-4][:n_part]
    with pytest.raises(TslError, match="max_n_constraints"):
        for f in range(1, 4):
            s.action(f, dpos, drot)
            s.time_step(projection_query, f)


def test_oversized_broad_phase_bucket_is_an_error():
    """the broad phase ranks the triangles of a hash bucket against each other (quadratic in the bucket): a body outside grid_extent -- every
    centroid clamped into the boundary cells -- must be reported, not ranked for seconds"""
    from thinshelllab_amd._lib import TslError
    from thinshelllab_amd.engine.geometry import projection_query
    from thinshelllab_amd.task_scene.Scene_balancing import Scene
    s = Scene(cloth_size=0.06, cloth_N=96, cloth_M=96)     # 18,432 cloth triangles
    s.init_all()
    s.prev_pos.copy_from(s.pos)
    s._ensure_ctx().set_param("grid_extent", 0.004)        # two cells per axis: the whole cloth falls into four of them
    with pytest.raises(TslError, match="broad-phase cell"):
        s.time_step(projection_query, 1)


def test_elastic_parameters_reach_the_engine_after_context_creation():
    """Elastic.mu / lam / alpha are 0-d fields like Cloth.Kb: a write after the first engine call must change the engine's material"""
    from thinshelllab_amd.task_scene.Scene_balancing import Scene
    s = Scene(cloth_size=0.06)
    s.init_all()
    rng = np.random.default_rng(0)
    x = s.pos.to_numpy(); x += rng.normal(0, 2e-5, x.shape); s.pos.from_numpy(x)
    E0 = s.compute_energy()
    ball = s.elastics[0]; pad = s.elastics[1]
    ball.mu[None] = 2.0 * ball.mu[None]
    E1 = s.compute_energy()
    pad.lam[None] = 3.0 * pad.lam[None]
    E2 = s.compute_energy()
    assert abs(E1 - E0) > 1e-9 * abs(E0) and abs(E2 - E1) > 1e-9 * abs(E1), (E0, E1, E2)
    with pytest.raises(Exception):
        s._ctx.set_param("elastic99.mu", 1.0)
    s._ctx.set_param("cloth0.Kb", 123.0)     # index parsed up to the dot
    with pytest.raises(Exception):
        s._ctx.set_param("cloth10.Kb", 1.0)


def _token_worker(q_in, q_out):
    """child process of test_dataflow_token_across_processes: a context of its own on the same device"""
    import numpy as np
    import scipy.sparse.linalg as spl
    from helpers import rel_err
    s = _drape(96, 64, 5e-5, seed=12)
    c = s._ensure_ctx()
    c.set_param("direct", 1); c.set_param("direct_leaf", 16)
    s.compute_residual_and_Hessian(spd=True)
    b = s.F.to_torch().clone()
    xs = spl.splu(c.operator_csr().tocsc()).solve(b.cpu().numpy())
    worst = 0.0
    n = 0
    q_out.put("ready")
    q_in.get()
    import time
    t0 = time.time()
    while time.time() - t0 < 4.0:
        s.compute_residual_and_Hessian(spd=True)
        x, st = c.solve(b.clone())
        worst = max(worst, rel_err(x.cpu().numpy(), xs) if st["flag"] == 0 else 1.0)
        n += 1
    q_out.put((n, worst, c.direct_counters()))


def test_two_contexts_share_the_device_through_the_dataflow_token():
    """Several scenes per GPU (bench.py --scenes-per-gpu, trajopt_batch with more scenes than GPUs): the persistent dataflow launch needs every
    one of its workgroups resident, so ONE context per device holds the dataflow token (an advisory lock per PCI bus id, direct_host.hpp)
    and launches it; the others run the same block steps as one launch each.  Both contexts solve concurrently from two host threads: no
    lost flag, right answers, and -- every path giving the same bits -- the second context's solution equals the one the first computes
    for the same operator.  The token passes on when its holder is destroyed."""
    import gc
    import threading
    gc.collect()
    import scipy.sparse.linalg as spl
    s1 = _drape(160, 96, 5e-5, seed=11)
    c1 = s1._ensure_ctx()
    c1.set_param("direct", 1); c1.set_param("direct_leaf", 16)
    s1.compute_residual_and_Hessian(spd=True)
    b1 = s1.F.to_torch().clone()
    x, st = c1.solve(b1.clone())
    n0 = c1.direct_counters()["flow_launches"]
    assert n0 > 0 and st["flag"] == 0, (n0, st, c1.direct_counters())    # (a context of an earlier test that is still alive would hold the token)
    xs1 = spl.splu(c1.operator_csr().tocsc()).solve(b1.cpu().numpy())
    x1_ref = x.cpu().numpy()
    s2 = _drape(160, 96, 5e-5, seed=11)                                   # the same operator in a second context
    c2 = s2._ensure_ctx()
    c2.set_param("direct", 1); c2.set_param("direct_leaf", 16)
    s2.compute_residual_and_Hessian(spd=True)
    b2 = s2.F.to_torch().clone()
    out = {}

    def work(name, s, c, b):
        res = []
        for _ in range(12):
            s.compute_residual_and_Hessian(spd=True)      # fresh factors every time
            x, st = c.solve(b.clone())
            res.append((st["flag"], x.cpu().numpy()))
        out[name] = res
    th = [threading.Thread(target=work, args=("a", s1, c1, b1)), threading.Thread(target=work, args=("b", s2, c2, b2))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for name in "ab":
        for f, x in out[name]:
            assert f == 0 and rel_err(x, xs1) < 1e-9
            assert np.array_equal(x, x1_ref), (name, np.abs(x - x1_ref).max())
    k1, k2 = c1.direct_counters(), c2.direct_counters()
    assert k1["flow_aborts"] == 0 and k2["flow_aborts"] == 0, (k1, k2)
    assert k1["flow_launches"] > n0 and k2["flow_launches"] == 0, (n0, k1, k2)     # the holder kept launching, the neighbour never did
    del c1, s1, th, out
    gc.collect()
    s3 = _drape(96, 64, 5e-5, seed=13)                                    # a context created after the holder is gone gets the token
    c3 = s3._ensure_ctx()
    c3.set_param("direct", 1); c3.set_param("direct_leaf", 16)
    s3.compute_residual_and_Hessian(spd=True)
    x, st = c3.solve(s3.F.to_torch().clone())
    assert st["flag"] == 0 and c3.direct_counters()["flow_launches"] > 0, (st, c3.direct_counters())


def test_dataflow_token_across_processes():
    """Two PROCESSES on one device (two ranks per GPU): the token is a lock on a file named after the device's PCI bus id, so the second
    process finds it taken, stays on the block-step launches, and neither runs into a lost flag or a multi-second stall."""
    import gc
    import multiprocessing as mp
    import time
    gc.collect()
    s1 = _drape(160, 96, 5e-5, seed=11)
    c1 = s1._ensure_ctx()
    c1.set_param("direct", 1); c1.set_param("direct_leaf", 16)
    s1.compute_residual_and_Hessian(spd=True)
    b1 = s1.F.to_torch().clone()
    x, st = c1.solve(b1.clone())
    assert c1.direct_counters()["flow_launches"] > 0 and st["flag"] == 0          # this process holds the token
    mpc = mp.get_context("spawn")
    q_in, q_out = mpc.Queue(), mpc.Queue()
    p = mpc.Process(target=_token_worker, args=(q_in, q_out))
    p.start()
    assert q_out.get(timeout=300) == "ready"
    q_in.put("go")
    t0 = time.time(); n = 0; slowest = 0.0
    while time.time() - t0 < 4.0:
        t1 = time.time()
        s1.compute_residual_and_Hessian(spd=True)
        x, st = c1.solve(b1.clone())
        assert st["flag"] == 0
        slowest = max(slowest, time.time() - t1); n += 1
    n2, worst2, k2 = q_out.get(timeout=300)
    p.join(60)
    k1 = c1.direct_counters()
    assert k1["flow_aborts"] == 0 and k2["flow_aborts"] == 0 and k2["flow_launches"] == 0, (k1, k2)
    assert worst2 < 1e-9 and n2 > 3 and n > 3 and slowest < 1.0, (n, n2, worst2, slowest)
