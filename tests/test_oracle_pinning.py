"""CPU checks that pin the oracle (the reference ships no tests / golden vectors for this path and cannot be
imported here): finite differences, numpy eigh, scipy spsolve and the mesh facts of
SURVEY.md App. A/C.  Also the reference's only self-check input (engine/linalg.py:155-171) and its only data
fixture (data/balance_state) are exercised."""
import os
import random

import numpy as np
import pytest

from helpers import rel_err

DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "thinshelllab_amd", "data")
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cloth(po, N, M, fold=False, frozen_row=True, gravity=(0, 0, -9.8), Kb=100.0, k_angle=3.14):
    o = po.OracleScene(gravity=gravity, newton_cap=50)
    ci = o.add_cloth(N, M, 0.1 / 15 * N)
    o.set_scalar("cloth0.Kb", Kb); o.set_scalar("cloth0.k_angle", k_angle)
    if fold:
        o.cloth_init(ci, -0.07, -0.01, 0.0004, fold=True, curv=2)
    else:
        o.cloth_init(ci, 0, 0, 0)
    o.finalize()
    if frozen_row:
        o.frozen.reshape(-1, 3)[N * (M + 1):] = 1
    return o


# ------------------------------------------------------------------------------------------------ mesh facts
@pytest.mark.parametrize("N,M,hinges,wrong", [(15, 3, 117, 59), (15, 15, 645, 329)])
def test_hinge_tables(oracle, N, M, hinges, wrong):
    """all hinges with counter_face > f are correct; 59/270 (329/1350) table entries are wrong."""
    o = _cloth(oracle, N, M)
    f2v = o.arr("cloth0.f2v", (-1, 3)); cf = o.arr("cloth0.counter_face", (-1, 3)); cp = o.arr("cloth0.counter_point", (-1, 3))
    NF = len(f2v)
    edge_faces = {}
    for f in range(NF):
        for l in range(3):
            e = tuple(sorted((f2v[f][(l + 1) % 3], f2v[f][(l + 2) % 3])))
            edge_faces.setdefault(e, []).append((f, l))
    n_h = 0; n_wrong = 0
    for f in range(NF):
        for l in range(3):
            e = tuple(sorted((f2v[f][(l + 1) % 3], f2v[f][(l + 2) % 3])))
            others = [x for x in edge_faces[e] if x[0] != f]
            true_nb = others[0] if others else (-1, 0)
            ok = cf[f][l] == true_nb[0] and (true_nb[0] == -1 or cp[f][l] == true_nb[1])
            n_wrong += (not ok)
            if cf[f][l] > f:
                n_h += 1
                assert ok, (f, l)
    assert n_h == hinges == sum(1 for v in edge_faces.values() if len(v) == 2)
    assert n_wrong == wrong


def test_tactile_and_ball_mesh_facts(oracle):
    nodes = oracle.read_node(os.path.join(DATA, "tactile.node")); tets = oracle.read_ele(os.path.join(DATA, "tactile.ele")); faces = oracle.read_face(os.path.join(DATA, "tactile.face"))
    assert (len(nodes), len(tets), len(faces)) == (276, 1365, 200)
    o = oracle.OracleScene()
    o.add_cloth(2, 2, 0.01)
    o.L.tslo_cloth_init_mesh(o.h, 0)
    ei = o.add_tactile(1.0, nodes, tets, faces)
    o.elastic_init(ei, 0, 0, 0, False)
    o.finalize()
    assert o.int("elastic0.frozen_cnt") == 49 and o.int("elastic0.surf_point") == 53
    W = o.arr("elastic0.F_W")
    assert abs(W.sum() - 6.4306e-6) < 1e-10          # rest volume
    x = nodes
    Ds = np.stack([x[tets[:, k]] - x[tets[:, 3]] for k in range(3)], axis=2)
    assert (np.linalg.det(Ds) > 0).all()             # all tets positively oriented
    assert len(np.unique(faces)) == 102              # surface vertices
    bn = oracle.read_node(os.path.join(DATA, "ball.node")); bt = oracle.read_ele(os.path.join(DATA, "ball.ele")); bf = oracle.read_face(os.path.join(DATA, "ball.face"))
    assert (len(bn), len(bt), len(bf)) == (100, 295, 166)


# ------------------------------------------------------------------------------------------------ SPD projections
def test_spd_projector_reference_selfcheck_matrix(oracle):
    """engine/linalg.py:155-171: random.seed(1), 9x9 symmetric from random.random(); the projected matrix must have the
    clamped spectrum of the input (what the reference's __main__ prints for eyeballing)."""
    random.seed(1)
    n = 9
    A = np.zeros((n, n))
    for i in range(n):
        for j in range(n):
            A[i, j] = random.random()
    for i in range(n):
        for j in range(i):
            A[i, j] = A[j, i]
    P, sweeps = oracle.spd_project(A, 9)   # K = n as in the self-check
    w0 = np.linalg.eigvalsh(A); w1 = np.linalg.eigvalsh(P)
    # K = 9 sweeps are NOT enough for this 9x9 (the reference's own self-check is only eyeballed): percent-level agreement
    assert np.abs(np.sort(np.maximum(w0, 0)) - np.sort(w1)).max() < 2e-2
    assert sweeps == 9
    P20, _ = oracle.spd_project(A, 20)
    V = np.linalg.eigh(A)
    E = (V[1] * np.maximum(V[0], 0)) @ V[1].T
    assert np.abs(P20 - E).max() < 2e-5
    assert np.abs(oracle.spd_project_jacobi(A) - E).max() < 1e-13


@pytest.mark.parametrize("n,K,scale,tol", [(3, 10, 3e5, 1e-10), (9, 20, 1e3, 1e-8)])
def test_spd_projector_vs_eigh(oracle, n, K, scale, tol):
    """literal Householder + shifted-QR projector == exact eigen-clamp whenever its sweeps converged before the cap K
    (D=9, K=20 needs ~16.5 sweeps on average and occasionally hits the cap, leaving an unconverged result);
    the converged Jacobi variant (what the GPU runs) is exact always."""
    rng = np.random.default_rng(n)
    worst = 0
    capped = 0
    trials = 400
    for _ in range(trials):
        A = rng.normal(size=(n, n)) * scale
        A = A + A.T
        P, sweeps = oracle.spd_project(A, K)
        w, V = np.linalg.eigh(A)
        E = (V * np.maximum(w, 0)) @ V.T
        if sweeps < K:
            worst = max(worst, np.abs(P - E).max() / np.abs(A).max())
        else:
            capped += 1
        assert np.abs(oracle.spd_project_jacobi(A) - E).max() / np.abs(A).max() < 1e-13
    assert worst < tol
    assert capped <= 0.03 * trials


def test_spd_project_2d(oracle):
    rng = np.random.default_rng(0)
    for _ in range(200):
        A = rng.normal(size=(2, 2)); A = A + A.T
        w, V = np.linalg.eigh(A)
        E = (V * np.maximum(w, 0)) @ V.T
        assert np.abs(oracle.spd_project_2d(A) - E).max() < 1e-13


# ------------------------------------------------------------------------------------------------ finite differences
def _fd_grad(o, x0, idx, eps=1e-7):
    g = np.zeros(len(idx))
    for n, k in enumerate(idx):
        xp = x0.copy().reshape(-1); xp[k] += eps
        xm = x0.copy().reshape(-1); xm[k] -= eps
        o.pos[:] = xp.reshape(-1, 3); o.push_down_all(); ep = o.compute_energy()
        o.pos[:] = xm.reshape(-1, 3); o.push_down_all(); em = o.compute_energy()
        g[n] = (ep - em) / (2 * eps)
    o.pos[:] = x0; o.push_down_all()
    return g


@pytest.mark.parametrize("fold", [False, True])
def test_cloth_gradient_is_derivative_of_energy(oracle, fold):
    o = _cloth(oracle, 15, 3, fold=fold, Kb=400.0, k_angle=0.5)
    rng = np.random.default_rng(0)
    o.pos[:] += rng.normal(0, 2e-4, o.pos.shape)
    o.vel[:] = rng.normal(0, 1e-2, o.pos.shape); o.prev_pos[:] = o.pos - rng.normal(0, 1e-4, o.pos.shape)
    o.push_down_all()
    x0 = o.pos.copy()
    o.newton_step_init(); o.compute_energy(); o.compute_residual_and_Hessian(False)
    F = o.arr("F").copy()
    free = np.nonzero(o.frozen == 0)[0]
    idx = rng.choice(free, 40, replace=False)
    g = _fd_grad(o, x0, idx)
    assert np.abs(F[idx] - g).max() / np.abs(g).max() < 1e-7


def _full_scene(oracle):
    """folding-like scene: cloth + frozen box + tactile pad, contacts active"""
    nodes = oracle.read_node(os.path.join(DATA, "tactile.node")); tets = oracle.read_ele(os.path.join(DATA, "tactile.ele")); faces = oracle.read_face(os.path.join(DATA, "tactile.face"))
    o = oracle.OracleScene(k_contact=1e4, eps_contact=4e-4, gravity=(0, 0, 0), newton_cap=50, effector_cnt=2, mu_cloth_elastic=5.0)
    ci = o.add_cloth(15, 3, 0.1)
    o.set_scalar("cloth0.Kb", 400.0); o.set_scalar("cloth0.k_angle", 0.5)
    o.cloth_init(ci, -0.07, -0.01, 0.0004, fold=True, curv=2)
    b = o.add_box(0.07, 9, 9, 2)
    o.elastic_init(b, -0.035, -0.035, -0.00875)
    t = o.add_tactile(0.5, nodes, tets, faces)
    r = 0.1 / 15 * 3 / 3.1415
    x = -0.07 + 9 / 16 * 0.1 - r * 0.86 + 0.005
    o.elastic_init(t, x, 0.0, 2 * r + 0.0079 - 0.0003, True)   # pressed 0.3 mm into the fold
    o.finalize()
    nv_c, nv_b = o.int("cloth0.NV"), o.int("elastic0.n_verts")
    for (bi, vs, ve) in ((0, nv_c, nv_c + nv_b), (1, 0, nv_c), (0, nv_c + nv_b, o.tot_NV), (2, 0, nv_c)):
        o.add_pair(bi, vs, ve)
    fr = o.frozen.reshape(-1, 3)
    fr[nv_c:nv_c + nv_b] = 1
    fr[15 * 4:16 * 4] = 1
    return o


def test_full_scene_gradient_is_derivative_of_energy(oracle):
    o = _full_scene(oracle)
    rng = np.random.default_rng(1)
    x = o.pos; x[:64, 2] += rng.normal(0, 5e-5, 64)
    o.prev_pos[:] = o.pos; o.push_down_all()
    o.calc_vn(); o.projection_query(); o.contact_analysis()
    assert o.nc > 5
    xp = o.pos + rng.normal(0, 1e-5, o.pos.shape) * (o.frozen.reshape(-1, 3) == 0)
    o.pos[:] = xp; o.push_down_all()
    x0 = o.pos.copy()
    o.newton_step_init(); o.compute_energy(); o.compute_residual_and_Hessian(False)
    F = o.arr("F").copy()
    cidx = o.arr("const_idx", (-1, 4))[: o.nc]
    touched = np.unique(np.concatenate([3 * cidx.ravel() + k for k in range(3)]))
    free = np.nonzero(o.frozen == 0)[0]
    cand = np.intersect1d(touched, free)
    idx = np.concatenate([rng.choice(cand, min(30, len(cand)), replace=False), rng.choice(free, 30, replace=False)])
    g = _fd_grad(o, x0, idx, eps=1e-8)
    assert np.abs(F[idx] - g).max() / np.abs(g).max() < 2e-5
    assert o.stats()["missing"] == 0


def test_parameter_derivative_fields_are_derivatives_of_the_force(oracle):
    """System identification (analytic_grad_system.py / compute_deri): d_kb, d_kl, d_ka, d_mu are the derivatives of the
    elastic force (= -gradient) with respect to the stiffness parameters; checked by central differences of the
    oracle's own gradient.  The tactile derivative is exact for its energy (alpha = 1 + mu/lam moves with mu)."""
    o = _full_scene(oracle)
    rng = np.random.default_rng(4)
    o.pos[:] += rng.normal(0, 1e-5, o.pos.shape) * (o.frozen.reshape(-1, 3) == 0)
    o.prev_pos[:] = o.pos; o.push_down_all()
    nv_c, nv_b = o.int("cloth0.NV"), o.int("elastic0.n_verts")
    pad = slice(nv_c + nv_b, o.tot_NV)

    def grad():
        o.newton_step_init(); o.compute_energy(); o.compute_residual_and_Hessian(False)
        return o.arr("F").copy().reshape(-1, 3)

    grad()                     # normals / prepare_bending of the current pose (the reference calls init_folding first)
    o.get_paramters_grad()
    d = {k: o.arr(k, (-1, 3)).copy() for k in ("d_kb", "d_kl", "d_ka", "d_mu")}
    free = o.frozen.reshape(-1, 3) == 0
    for key, name, val in (("d_kb", "cloth0.Kb", 400.0), ("d_kl", "cloth0.Kl", 1000.0), ("d_ka", "cloth0.Ka", 1000.0)):
        h = 1e-3 * val
        o.set_scalar(name, val + h); gp = grad()
        o.set_scalar(name, val - h); gm = grad()
        o.set_scalar(name, val)
        fd = -(gp - gm) / (2 * h)
        m = free[:nv_c]
        assert np.abs(d[key][:nv_c][m]).max() > 0
        assert np.abs(fd[:nv_c][m] - d[key][:nv_c][m]).max() <= 1e-7 * np.abs(d[key][:nv_c]).max(), key
    mu0 = o.double("elastic1.mu")
    h = 1e-4 * mu0
    o.set_scalar("elastic1.mu", mu0 + h); gp = grad()
    o.set_scalar("elastic1.mu", mu0 - h); gm = grad()
    o.set_scalar("elastic1.mu", mu0)
    fd = -(gp - gm) / (2 * h)
    m = free[pad]
    assert np.abs(d["d_mu"][pad][m]).max() > 0
    assert np.abs(fd[pad][m] - d["d_mu"][pad][m]).max() <= 1e-6 * np.abs(d["d_mu"][pad]).max()


def test_hessian_matches_fd_where_the_reference_is_exact(oracle):
    """FEM (both materials) and contact blocks are exact second derivatives; the cloth terms are not
    (edge off-diagonal sign, area factor 2, bending second-order term) and are excluded by zeroing their stiffness."""
    o = _full_scene(oracle)
    for k in ("Kl", "Ka", "Kb"):
        o.set_scalar(f"cloth0.{k}", 0.0)
    rng = np.random.default_rng(3)
    o.pos[:64, 2] += rng.normal(0, 5e-5, 64)
    o.prev_pos[:] = o.pos; o.push_down_all()
    o.calc_vn(); o.projection_query(); o.contact_analysis()
    o.pos[:] = o.pos + rng.normal(0, 1e-5, o.pos.shape) * (o.frozen.reshape(-1, 3) == 0)
    o.push_down_all()
    x0 = o.pos.copy()

    def grad(x):
        o.pos[:] = x; o.push_down_all(); o.newton_step_init(); o.compute_energy(); o.compute_residual_and_Hessian(False)
        return o.arr("F").copy()
    grad(x0)
    H = o.H_csr()
    free = np.nonzero(o.frozen == 0)[0]
    cols = rng.choice(free, 25, replace=False)
    eps = 1e-8
    for k in cols:
        xp = x0.copy().reshape(-1); xp[k] += eps
        xm = x0.copy().reshape(-1); xm[k] -= eps
        col = (grad(xp.reshape(-1, 3)) - grad(xm.reshape(-1, 3))) / (2 * eps)
        hk = np.asarray(H[:, k].todense()).ravel()
        assert np.abs(col[free] - hk[free]).max() <= 2e-4 * max(np.abs(hk).max(), 1.0), k


# ------------------------------------------------------------------------------------------------ linear solve
def test_solver_matches_spsolve(oracle):
    import scipy.sparse.linalg as spl
    o = _cloth(oracle, 15, 3, fold=True, Kb=400.0, k_angle=0.5, gravity=(0, 0, 0))
    rng = np.random.default_rng(0)
    o.pos[:] += rng.normal(0, 3e-5, o.pos.shape) * (o.frozen.reshape(-1, 3) == 0); o.prev_pos[:] = o.pos; o.push_down_all()
    o.set_solver(1e-12)
    for spd in (True, False):
        o.newton_step_init(); o.compute_energy(); o.compute_residual_and_Hessian(spd)
        b = rng.normal(size=o.tot_NV * 3)
        x, flag = o.solve(b)
        xs = spl.spsolve(o.H_csr().tocsc(), b)
        assert flag in (0, 1, 2)
        assert rel_err(x, xs) < 1e-7, (spd, flag)


def test_solver_direct_hook_is_the_same_solve(oracle):
    """set_direct(1) makes scipy's SuperLU the oracle's solver (the reference calls spsolve, sparse_solver.py:85-105): same
    solution as the iterative stages, and a whole time step lands on the same state"""
    o = _cloth(oracle, 15, 3, fold=True, Kb=400.0, k_angle=0.5, gravity=(0, 0, 0))
    rng = np.random.default_rng(0)
    o.pos[:] += rng.normal(0, 3e-5, o.pos.shape) * (o.frozen.reshape(-1, 3) == 0); o.prev_pos[:] = o.pos; o.push_down_all()
    o.set_solver(1e-12)
    x0 = o.pos.copy()
    for spd in (True, False):
        o.newton_step_init(); o.compute_energy(); o.compute_residual_and_Hessian(spd)
        b = rng.normal(size=o.tot_NV * 3)
        o.set_direct(0); xi, fi = o.solve(b)
        o.set_direct(1); xd, fd = o.solve(b)
        assert fi in (0, 1, 2) and fd == 4
        assert rel_err(xd, xi) < 1e-7, spd
    o.set_direct(1); o.time_step(); x_direct = o.pos.copy()
    o.pos[:] = x0; o.prev_pos[:] = x0; o.vel[:] = 0; o.push_down_all(); o.clear_proj()
    o.set_direct(0); o.time_step()
    assert np.abs(o.pos - x_direct).max() < 1e-9


def test_newton_step_converges_and_decreases_energy(oracle):
    o = _cloth(oracle, 15, 3, fold=True, Kb=400.0, k_angle=0.5, gravity=(0, 0, 0))
    o.set_scalar("plastic", 1)
    o.set_solver(1e-10)
    o.timestep_init(); o.calc_vn(); o.projection_query(); o.contact_analysis()
    E_prev = None
    for it in range(50):
        o.newton_step_init(); E = o.compute_energy(); o.compute_residual_and_Hessian(True)
        if E_prev is not None:
            assert E <= E_prev + 1e-12
        E_prev = E
        d, a = o.newton_step()
        if d < 1e-7:
            break
    assert d < 1e-7 and it < 49


# ------------------------------------------------------------------------------------------------ reference data fixture
def test_balance_state_fixture_is_a_consistent_scene_state():
    """data/balance_state (the reference's only saved state: Scene_balancing.save_all) -- shapes as SURVEY 8c states"""
    import torch
    st = torch.load(os.path.join(GOLD, "balance_state", "state"), weights_only=False)
    assert tuple(st["pos"].shape) == (1332, 3) and st["pos"].dtype == torch.float64
    flag = np.load(os.path.join(GOLD, "balance_state", "proj_flag.npy")); dr = np.load(os.path.join(GOLD, "balance_state", "proj_dir.npy"))
    assert flag.shape == (6, 1332) and dr.shape == (6, 1332)
    assert flag.sum(1).tolist() == [998, 14, 151, 159, 153, 155]


def _balancing_host_scene():
    from thinshelllab_amd.task_scene.Scene_balancing import Scene
    s = Scene(cloth_size=0.06, device="cpu")   # host-side construction only: no engine context is created without a GPU
    s.init_all()
    return s


def test_reference_state_pins_pad_placement_and_gripper_frames():
    """REFERENCE OUTPUT (data/balance_state was written by the reference engine, Scene_balancing.save_all :202-211): the local pad
    coordinates the reference's gripper holds (gripper_tactile.init :103-133 on top of Elastic.init_pos, model_elastic_tactile.py
    :214-230, readfile.py mesh parsing, the scene's pad poses Scene_balancing.py:78-86) equal the restated ones EXACTLY in x, y and
    differ in z by the saved half_gripper_dist (open_gripper moves the upper pad by +d, the lower by -d, :196-218); the world
    frames follow pos + R local; and the saved simulation state carries exactly those world positions on the 49 boundary vertices
    that the restated bound mask selects in each of the four pads (update_bound :244-249), at the restated global vertex offsets,
    while every other pad vertex is deformed."""
    import torch
    s = _balancing_host_scene()
    g = os.path.join(GOLD, "balance_state")
    L = lambda n: np.load(os.path.join(g, n + ".npy"))
    gr = s.gripper
    hd = L("half_gripper_dist")
    assert np.array_equal(gr.pos.to_numpy(), L("pos"))
    du = L("F_x_upper") - gr.F_x_upper.to_numpy(); dl = L("F_x_lower") - gr.F_x_lower.to_numpy()
    assert np.abs(du[..., :2]).max() == 0.0 and np.abs(dl[..., :2]).max() == 0.0
    for j in range(2):
        assert np.abs(du[j, :, 2] - hd[j]).max() < 1e-15 and np.abs(dl[j, :, 2] + hd[j]).max() < 1e-15
    # world frames (identity quaternion in the fixture; rotmat is stored in single precision by the reference)
    from thinshelllab_amd.engine.gripper_tactile import quat_to_rotmat
    R = np.stack([quat_to_rotmat(q) for q in L("rot")]).astype(np.float32)
    assert L("rotmat").dtype == np.float32 and np.array_equal(R, L("rotmat"))
    gr.F_x_upper.from_numpy(L("F_x_upper")); gr.F_x_lower.from_numpy(L("F_x_lower")); gr.rot.from_numpy(L("rot"))
    gr.get_rotmat(); gr.get_vert_pos()
    assert np.array_equal(gr.F_x_upper_world.to_numpy(), L("F_x_upper_world")) and np.array_equal(gr.F_x_lower_world.to_numpy(), L("F_x_lower_world"))
    # the saved state: layout cloth 128 | ball 100 | 4 pads x 276, boundary vertices driven by the gripper
    assert s.tot_NV == 1332 and [(e.offset, e.n_verts) for e in s.elastics] == [(128, 100), (228, 276), (504, 276), (780, 276), (1056, 276)]
    pos = torch.load(os.path.join(g, "state"), weights_only=False)["pos"].numpy()
    b = gr.bound_idx.to_numpy().astype(int)
    assert len(b) == 49
    rest = np.setdiff1d(np.arange(276), b)
    for j in range(2):
        for e, w in ((s.elastics[2 * j + 1], L("F_x_upper_world")), (s.elastics[2 * j + 2], L("F_x_lower_world"))):
            assert np.abs(pos[e.offset + b] - w[j, b]).max() == 0.0
            assert np.abs(pos[e.offset + rest] - w[j, rest]).min() > 0.0 and np.abs(pos[e.offset + rest] - w[j, rest]).max() > 1e-3
    fr = s.frozen.to_numpy().reshape(-1, 3)
    assert [int(fr[e.offset:e.offset + e.n_verts, 0].sum()) for e in s.elastics] == [0, 49, 49, 49, 49]


def test_reference_state_pins_projection_query(oracle):
    """REFERENCE OUTPUT: proj_flag / proj_dir of data/balance_state were produced by the reference's projection_query
    (geometry.py:96-229) at the START of its last time step; the saved positions are those at its END, the saved velocities give the
    start exactly (x - v dt, damping = 1).  The restated query there reproduces the flags of the five FEM bodies exactly (14 / 151 / 159 /
    153 / 155 vertices), proj_dir on every vertex flagged in both, and the cloth-as-target row up to ONE cell-boundary triangle: ten pad
    vertices whose candidate is the cloth triangle [109, 108, 116] (centroid 8.9 um above the z = 0 cell face) are flagged here and not in the
    reference -- asserted as exactly that.  (On the end-of-step positions 16 entries differ: six more vertices crossed their neighbourhood
    during the step.)"""
    from oracle.mirror import oracle_from_scene
    s = _balancing_host_scene()
    o = oracle_from_scene(oracle, s, check_init=True)
    g = os.path.join(GOLD, "balance_state")
    from helpers import start_of_step_positions, assert_cloth_target_mismatches_are_the_cell_boundary_triangle
    x = start_of_step_positions(g, s.dt)
    o.pos[:] = x; o.prev_pos[:] = x; o.push_down_all()
    o.calc_vn(); o.projection_query()
    flag = o.arr("proj_flag").reshape(-1, s.tot_NV); dr = o.arr("proj_dir").reshape(-1, s.tot_NV)
    pidx = o.arr("proj_idx").reshape(6, s.tot_NV, 3)
    F = np.load(os.path.join(g, "proj_flag.npy")); D = np.load(os.path.join(g, "proj_dir.npy"))
    assert flag.shape == F.shape == (6, 1332)
    assert flag[1:].sum(1).tolist() == [14, 151, 159, 153, 155] and np.array_equal(flag[1:], F[1:])
    mm = assert_cloth_target_mismatches_are_the_cell_boundary_triangle(flag[0], F[0], pidx[0], x)
    assert len(mm) == 10 and int(flag[0].sum()) == 998 + 10
    both = (flag == 1) & (F == 1)
    assert both.sum() >= 1600 and np.array_equal(dr[both], D[both])


def test_literal_sign_test_vs_tolerant(oracle):
    """The degenerate sign test `norm_dir[i2] . (x_b - x_a) < 0` (model_fold_offset.py:116,135,144): on the wrongly-tabled slots of
    odd cells the edge lies IN the plane of face i2, the exact value is 0 and its floating-point sign is rounding noise; oracle and
    HIP kernels treat |n . e| <= 1e-10 |e| as zero.  What that replaces, measured: the oracle in its LITERAL
    mode (pyoracle.set_sign_mode(1)) against the tolerant one on the three native scenes at their initial poses and under cloth
    perturbations of 1e-6 and 2e-4 m.  The energy does not depend on it (the sign only enters judge_angle -> mat_M, c_i of those
    slots, and there through the Hessian only); the gradient is unchanged to rounding, the Hessian changes by at most 5e-7 of its
    largest entry on 0.1-1.4 % of its 50k-150k entries."""
    import importlib
    from oracle.mirror import oracle_from_scene
    worst_h = worst_f = 0.0
    for name in ("folding", "lifting", "balancing"):
        s = importlib.import_module(f"thinshelllab_amd.task_scene.Scene_{name}").Scene(device="cpu")
        s.init_all()
        o = oracle_from_scene(oracle, s)
        c = s.cloths[0]
        for amp in (0.0, 1e-6, 2e-4):
            x = s.pos.to_numpy().copy()
            x[c.offset:c.offset + c.NV] += np.random.default_rng(0).normal(0, amp, (c.NV, 3))
            o.pos[:] = x; o.prev_pos[:] = x; o.push_down_all()
            res = {}
            try:
                for mode in (0, 1):
                    oracle.set_sign_mode(mode)
                    o.newton_step_init(); o.compute_residual_and_Hessian(True)
                    res[mode] = (o.H_csr().copy(), o.arr("F").copy(), o.compute_energy())
            finally:
                oracle.set_sign_mode(0)
            dh = abs(res[0][0] - res[1][0]); hm = abs(res[0][0]).max()
            rel_h = (dh.max() if dh.nnz else 0.0) / hm
            rel_f = np.abs(res[0][1] - res[1][1]).max() / max(np.abs(res[0][1]).max(), 1e-300)
            n_diff = int((dh > 1e-12 * hm).sum())
            print(f"{name} amp {amp:g}: max |dH| / |H| = {rel_h:.2e} on {n_diff} of {res[0][0].nnz} entries, max |dF| / |F| = {rel_f:.2e}, dE = {res[1][2] - res[0][2]:.1e}")
            assert abs(res[1][2] - res[0][2]) <= 1e-13 * max(1.0, abs(res[0][2]))
            assert rel_h < 5e-6 and n_diff < 0.02 * res[0][0].nnz
            if amp > 0:
                assert rel_f < 5e-5
            worst_h = max(worst_h, rel_h); worst_f = max(worst_f, rel_f if amp > 0 else 0.0)
    assert worst_h > 0   # the literal mode does differ somewhere: the switch is live
