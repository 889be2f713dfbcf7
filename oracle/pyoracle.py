"""ORACLE (test infrastructure only).

ctypes front-end of ``oracle/_build/libtsl_oracle.so`` -- the fp64 CPU restatement of the
ThinShellLab engine (reference: /root/reference/code/engine, see the .cpp files for file:line
citations).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module; the product package ``thinshelllab_amd`` never does.

Parity status: the reference ships no tests / golden vectors and cannot be imported here (taichi +
cupy are absent).  Pinned by REFERENCE OUTPUT (its saved state data/balance_state): pad placement, gripper
frames, vertex layout, contact-candidate flags of projection_query (tests/test_oracle_pinning.py::
test_reference_state_*).  Energies, derivatives, solver and adjoint remain **unpinned** by the reference
and are pinned by independent checks in tests/test_oracle_*.py (finite differences, numpy eigh, scipy
spsolve, mesh facts from SURVEY.md App. A/C).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libtsl_oracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with g++ (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.tslo_new.restype = C.c_void_p
        L.tslo_version.restype = C.c_char_p
        L.tslo_array.restype = C.c_void_p
        L.tslo_array.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_long), C.POINTER(C.c_int)]
        L.tslo_int.restype = C.c_long
        L.tslo_int.argtypes = [C.c_void_p, C.c_char_p]
        L.tslo_double.restype = C.c_double
        L.tslo_double.argtypes = [C.c_void_p, C.c_char_p]
        L.tslo_compute_energy.restype = C.c_double
        L.tslo_newton_step.restype = C.c_double
        L.tslo_set_scalar.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        L.tslo_set_direct.argtypes = [C.c_void_p, _DIRECT_CB, C.c_int]
        L.tslo_grad_loss.restype = C.c_int
        L.tslo_grad_loss.argtypes = [C.c_void_p, C.c_char_p, C.c_double, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.tslo_reward.restype = C.c_double
        L.tslo_reward.argtypes = [C.c_void_p, C.c_char_p, C.c_double, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        _lib = L
    return _lib


# sparse direct solve for the oracle: the reference's SparseMatrix.solve is cupyx's spsolve (sparse_solver.py:85-105); here scipy's
# SuperLU on the oracle's own assembled block-CSR matrix
_DIRECT_CB = C.CFUNCTYPE(C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))
direct_seconds = [0.0, 0]   # [time spent in SuperLU, calls] (bench.py's cpu_baseline reports it)
direct_permc = ["COLAMD"]    # column ordering handed to SuperLU: scipy's default, or "MMD_AT_PLUS_A" (the operator is structurally symmetric: bench.py times both)
direct_residuals = []        # relative residuals |b - Hx| / |b| of the last SuperLU solves


def _direct_solve(nb, row_ptr, col, vals, b, x):
    try:
        import time
        import scipy.sparse as sp
        import scipy.sparse.linalg as spl
        t0 = time.time()
        rp = np.ctypeslib.as_array(row_ptr, shape=(nb + 1,))
        nnzb = int(rp[nb])
        cc = np.ctypeslib.as_array(col, shape=(nnzb,))
        vv = np.ctypeslib.as_array(vals, shape=(nnzb * 9,)).reshape(nnzb, 3, 3)
        A = sp.bsr_matrix((vv, cc, rp), shape=(3 * nb, 3 * nb)).tocsc()
        bb = np.ctypeslib.as_array(b, shape=(3 * nb,))
        xx = spl.splu(A, permc_spec=direct_permc[0]).solve(bb)
        if not np.isfinite(xx).all():
            return 1
        np.ctypeslib.as_array(x, shape=(3 * nb,))[:] = xx
        direct_seconds[0] += time.time() - t0; direct_seconds[1] += 1
        direct_residuals.append(float(np.linalg.norm(bb - A @ xx) / max(np.linalg.norm(bb), 1e-300)))   # what a pivoted sparse LU attains on this system
        del direct_residuals[:-64]
        return 0
    except Exception:   # noqa: BLE001 -- a Python exception must not unwind through the C frame
        return 2


_direct_cb = _DIRECT_CB(_direct_solve)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def read_node(path):
    """TetGen .node parser (same columns as reference readfile.read_node, readfile.py:1-15)."""
    with open(path) as f:
        n = int(f.readline().split()[0])
        return np.array([[float(t) for t in f.readline().split()][1:4] for _ in range(n)], dtype=np.float64)


def read_ele(path):
    with open(path) as f:
        n = int(f.readline().split()[0])
        return np.array([[int(t) for t in f.readline().split()][1:5] for _ in range(n)], dtype=np.int32)


def read_face(path):
    with open(path) as f:
        n = int(f.readline().split()[0])
        return np.array([[int(t) for t in f.readline().split()][1:4] for _ in range(n)], dtype=np.int32)


class OracleScene:
    """One restated BaseScene.  Build order mirrors the reference constructors:
    set_params -> add_* (bodies in reference order: cloths, then elastics) -> *_init -> finalize."""

    def __init__(self, dt=5e-3, k_contact=1000.0, eps_contact=0.001, eps_v=0.01, damping=1.0, max_n_constraints=10000,
                 newton_cap=1000, plastic=0, effector_cnt=-1, gravity=(0.0, 0.0, -9.8), mu_cloth_elastic=1.0):
        self.L = lib()
        self.h = C.c_void_p(self.L.tslo_new())
        g = np.array(gravity, dtype=np.float64)
        self.L.tslo_set_params(self.h, C.c_double(dt), C.c_double(k_contact), C.c_double(eps_contact), C.c_double(eps_v),
                               C.c_double(damping), int(max_n_constraints), int(newton_cap), int(plastic), int(effector_cnt),
                               _dp(g), C.c_double(mu_cloth_elastic))
        self.dt = dt
        self.set_direct(2)

    def __del__(self):
        try:
            self.L.tslo_free(self.h)
        except Exception:
            pass

    # --- construction
    def add_cloth(self, N, M, Len, rho=40.0, is_square=False):
        return self.L.tslo_add_cloth(self.h, int(N), int(M), C.c_double(Len), C.c_double(rho), int(is_square))

    def add_tactile(self, ratio, nodes, tets, faces):
        nodes = np.ascontiguousarray(nodes, np.float64); tets = np.ascontiguousarray(tets, np.int32); faces = np.ascontiguousarray(faces, np.int32)
        return self.L.tslo_add_tactile(self.h, C.c_double(ratio), len(nodes), _dp(nodes), len(tets), _ip(tets), len(faces), _ip(faces))

    def add_box(self, Len, Nx, Ny, Nz, density=2000.0):
        return self.L.tslo_add_box(self.h, C.c_double(Len), int(Nx), int(Ny), int(Nz), C.c_double(density))

    def add_loaded(self, density, nodes, tets, faces):
        nodes = np.ascontiguousarray(nodes, np.float64); tets = np.ascontiguousarray(tets, np.int32); faces = np.ascontiguousarray(faces, np.int32)
        return self.L.tslo_add_loaded(self.h, C.c_double(density), len(nodes), _dp(nodes), len(tets), _ip(tets), len(faces), _ip(faces))

    def cloth_init(self, ci, ox, oy, oz, fold=False, curv=2, bridge=False):
        self.L.tslo_cloth_init(self.h, ci, 2 if bridge else (1 if fold else 0), C.c_double(ox), C.c_double(oy), C.c_double(oz), int(curv))

    def elastic_init(self, ei, ox, oy, oz, flip=False):
        self.L.tslo_elastic_init(self.h, ei, C.c_double(ox), C.c_double(oy), C.c_double(oz), int(bool(flip)))

    def elastic_init_arch(self, ei, ox, oy, oz, arch):
        self.L.tslo_elastic_init_arch(self.h, ei, C.c_double(ox), C.c_double(oy), C.c_double(oz), C.c_double(arch))

    def finalize(self):
        self.L.tslo_finalize(self.h)

    def add_pair(self, b_idx, v_start, v_end, mu=None, factor=0.0):
        """mu float: fixed; None: the live mu_cloth_elastic parameter; "cloth_cloth": the live mu_cloth_cloth (times ``factor`` when > 0)"""
        kind = 0 if isinstance(mu, (int, float)) else (2 if mu == "cloth_cloth" else 1)
        self.L.tslo_add_pair(self.h, int(b_idx), int(v_start), int(v_end), kind, C.c_double(float(mu) if kind == 0 else float(factor)))

    def gripper_init(self, paired, n_part, pos_array):
        p = np.ascontiguousarray(pos_array, np.float64)
        self.L.tslo_gripper_init(self.h, int(paired), int(n_part), _dp(p))

    def gripper_update_all(self):
        """after writing ``gripper.rot``: rotation matrices, world vertices, and ALL pad vertices overwritten (gripper_single.py:158-162)"""
        self.L.tslo_gripper_update_all(self.h)

    def gripper_reinit(self, pos_array):
        p = np.ascontiguousarray(pos_array, np.float64)
        self.L.tslo_gripper_reinit(self.h, _dp(p))

    def set_body_gravity(self, is_elastic, idx, g):
        g = np.ascontiguousarray(g, np.float64)
        self.L.tslo_set_body_gravity(self.h, int(is_elastic), int(idx), _dp(g))

    def set_solver(self, tol, maxit=20000):
        self.L.tslo_set_solver(self.h, C.c_double(tol), int(maxit))

    def set_direct(self, mode):
        """0: iterative stages only (+ dense LU for n <= 4500); 1: scipy's SuperLU is THE solver (what the reference does:
        spsolve, sparse_solver.py:85-105); 2 (default): SuperLU as the last resort after PCG and BiCGStab"""
        self.L.tslo_set_direct(self.h, _direct_cb, int(mode))

    def set_scalar(self, name, v):
        self.L.tslo_set_scalar(self.h, name.encode(), C.c_double(v))

    # --- views
    def arr(self, name, shape=None):
        cnt = C.c_long(0); typ = C.c_int(0)
        p = self.L.tslo_array(self.h, name.encode(), C.byref(cnt), C.byref(typ))
        if not p:
            raise KeyError(name)
        ct = {0: C.c_double, 1: C.c_int, 2: C.c_float}[typ.value]
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(cnt.value,))
        if shape is not None:
            a = a.reshape(shape)
        return a

    def int(self, name):
        return int(self.L.tslo_int(self.h, name.encode()))

    def double(self, name):
        return float(self.L.tslo_double(self.h, name.encode()))

    @property
    def tot_NV(self):
        return self.int("tot_NV")

    @property
    def pos(self):
        return self.arr("pos", (-1, 3))

    @property
    def vel(self):
        return self.arr("vel", (-1, 3))

    @property
    def prev_pos(self):
        return self.arr("prev_pos", (-1, 3))

    @property
    def frozen(self):
        return self.arr("frozen")

    @property
    def nc(self):
        return int(self.L.tslo_nc(self.h))

    # --- engine calls (names follow BaseScene)
    def push_down_all(self):
        self.L.tslo_push_down_all(self.h)

    def pushup_all(self):
        self.L.tslo_pushup_all(self.h)

    def clear_proj(self):
        self.L.tslo_clear_proj(self.h)

    def compute_energy(self):
        return float(self.L.tslo_compute_energy(self.h))

    def newton_step_init(self):
        self.L.tslo_newton_step_init(self.h)

    def compute_residual_and_Hessian(self, spd=True):
        self.L.tslo_compute_residual_and_Hessian(self.h, int(bool(spd)))

    def compute_Hessian(self, spd=True):
        self.L.tslo_compute_Hessian(self.h, int(bool(spd)))

    def clear_H(self):
        self.L.tslo_clear_H(self.h)

    def solve(self, b):
        b = np.ascontiguousarray(b, np.float64)
        x = np.zeros_like(b)
        flag = self.L.tslo_solve(self.h, _dp(b), _dp(x))
        return x, flag

    def newton_step(self):
        a = C.c_double(0)
        d = float(self.L.tslo_newton_step(self.h, C.byref(a)))
        return d, a.value

    def time_step(self):
        self.L.tslo_time_step(self.h)

    def timestep_init(self):
        self.L.tslo_timestep_init(self.h)

    def timestep_finish(self):
        self.L.tslo_timestep_finish(self.h)

    def calc_vn(self):
        self.L.tslo_calc_vn(self.h)

    def projection_query(self):
        self.L.tslo_projection_query(self.h)

    def set_self_contact(self, body, on=True):
        """geometry_self.projection_query(self_contact=[...]): project the body's vertices onto its own triangles too"""
        self.L.tslo_set_self_contact(self.h, int(body), int(bool(on)))

    def contact_analysis(self):
        self.L.tslo_contact_analysis(self.h)

    def action(self, dpos, drot):
        dp = np.ascontiguousarray(dpos, np.float64); dr = np.ascontiguousarray(drot, np.float64)
        self.L.tslo_action(self.h, _dp(dp), _dp(dr))

    def action_dist(self, delta_pos, delta_rot, delta_dis):
        dp = np.ascontiguousarray(delta_pos, np.float64); dr = np.ascontiguousarray(delta_rot, np.float64); dd = np.ascontiguousarray(delta_dis, np.float64)
        self.L.tslo_action_dist(self.h, _dp(dp), _dp(dr), _dp(dd))

    def prepare_bending(self):
        self.L.tslo_prepare_bending(self.h)

    def stats(self, reset=False):
        out = (C.c_long * 7)()
        self.L.tslo_stats(self.h, out, int(reset))
        return dict(newton=out[0], cg=out[1], ls=out[2], solves=out[3], refine=out[4], flag=out[5], missing=out[6])   # flag 4: sparse direct solve

    def H_csr(self):
        """Assembled system matrix as scipy CSR (n = 3*tot_NV)."""
        import scipy.sparse as sp
        rp = self.arr("H.row_ptr").copy(); col = self.arr("H.col").copy(); vals = self.arr("H.vals").copy().reshape(-1, 3, 3)
        nb = len(rp) - 1
        return sp.bsr_matrix((vals, col, rp), shape=(3 * nb, 3 * nb)).tocsr()

    # --- adjoint (analytic_grad_single.Grad)
    def grad_new(self, T, n_parts):
        self.L.tslo_grad_new(self.h, int(T), int(n_parts)); self.T = T; self.n_parts = n_parts

    def grad_reset(self):
        self.L.tslo_grad_reset(self.h)

    def grad_copy_pos(self, step):
        self.L.tslo_grad_copy_pos(self.h, int(step))

    def grad_transfer(self, step):
        self.L.tslo_grad_transfer(self.h, int(step))
        if self.stats()["flag"] == 3:   # PCG, BiCGStab and (n <= 4500) dense LU all failed: nothing to compare against
            raise RuntimeError(f"oracle: the adjoint solve of step {step} did not converge")

    def grad_loss(self, name, a0=0.0, a1=0.0, rows=(6, 8, 7, 9), target=None):
        """Grad.get_loss_<name> of the reference (analytic_grad_single.py:259-471; name "" = get_loss) on the oracle's own tape;
        a0 / a1: curve7 / curve8 (fold) or sys.target (bounce); rows: hinge rows of the folding seeds; target: NV x 3 (push)"""
        r = np.ascontiguousarray(rows, dtype=np.int32).ravel()
        t = None if target is None else np.ascontiguousarray(target, dtype=np.float64)
        rc = self.L.tslo_grad_loss(self.h, name.encode(), float(a0), float(a1), _ip(r), None if t is None else _dp(t))
        if rc < 0:
            raise ValueError(f"oracle: no loss seed named {name!r}")
        return rc

    def reward(self, name, a0=0.0, a1=0.0, rows=(6, 8, 7, 9), target=None):
        """compute_reward* of the reference's task scenes ("folding", "folding.8", "lifting", "balancing", "balancing.all", ...) from the
        oracle's per-body state (and tape, where the kernel reads pos_buffer)"""
        r = np.ascontiguousarray(rows, dtype=np.int32).ravel()
        t = None if target is None else np.ascontiguousarray(target, dtype=np.float64)
        v = float(self.L.tslo_reward(self.h, name.encode(), float(a0), float(a1), _ip(r), None if t is None else _dp(t)))
        if v != v:
            raise ValueError(f"oracle: no reward named {name!r}")
        return v

    def grad_system(self, system_mode=True, count_kb=True, count_mu_lam=False, count_friction=False):
        """switch the reverse step to analytic_grad_system.Grad semantics (pos_grad clamp +-1, parameter gradients)"""
        self.L.tslo_grad_system(self.h, int(system_mode), int(count_kb), int(count_mu_lam), int(count_friction))

    def grad_params(self, reset=False):
        out = np.zeros(4)
        self.L.tslo_grad_params(self.h, _dp(out), int(reset))
        return dict(kb=out[0], mu=out[1], lam=out[2], friction=out[3])

    def get_paramters_grad(self):
        self.L.tslo_get_paramters_grad(self.h)

    def gather_force(self, n_eff):
        out = np.zeros((n_eff, 3))
        self.L.tslo_gather_force(self.h, _dp(out))
        return out

    def observation(self, dim):
        out = np.zeros(dim)
        self.L.tslo_observation(self.h, _dp(out))
        return out


def set_threads(n):
    lib().tslo_set_threads(int(n))


def set_sign_mode(m):
    """1: the reference's literal sign test `norm_dir[i2] . e < 0` (model_fold_offset.py:116,135,144); 0 (default): values within
    1e-10 |e| of zero count as zero, the exact-arithmetic value on the wrongly-tabled slots"""
    lib().tslo_set_sign_mode(int(m))


def set_spd_mode(m):
    """0: literal reference QR projector (default); 1: converged Jacobi eigen-clamp"""
    lib().tslo_set_spd_mode(int(m))


def spd_project(A, K):
    A = np.array(A, dtype=np.float64, order="C")
    n = A.shape[0]
    sweeps = lib().tslo_spd_project(_dp(A), n, int(K))
    return A, sweeps


def spd_project_jacobi(A):
    A = np.array(A, dtype=np.float64, order="C")
    lib().tslo_spd_project_jacobi(_dp(A), A.shape[0])
    return A


def spd_project_2d(A):
    A = np.array(A, dtype=np.float64, order="C")
    lib().tslo_spd_project_2d(_dp(A))
    return A
