"""Thin object wrapper over the C ABI: owns one ``tsl_ctx`` (one scene on one GPU).

State (pos / prev_pos / vel / ref_angle) stays in caller-owned torch tensors resident in HBM; this
class only passes ``data_ptr()``s.  torch is plumbing here (device memory, streams), not compute.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import Body, ClothDesc, ContactPair, ElasticDesc, SceneDesc, SolveStats, StepStats, check


def _np(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _ptr(t):
    """device/host pointer of a torch tensor or numpy array (or None)."""
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        assert t.is_contiguous(), "tensor must be contiguous"
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)


class TslContext:
    def __init__(self, *, tot_NV, dt, mass, gravity, frozen, cloths=(), elastics=(), faces=None, bodies=(), pairs=(),
                 k_contact=1000.0, eps_contact=1e-3, eps_v=0.01, damping=1.0, max_n_constraints=10000, grid_h=0.003, device="cuda:0"):
        """cloths: dicts with N, M, NV, NF, v_offset, dx, mass, Kl, Ka, Kb, k_angle, f2v, counter_face, counter_point, rest_area, rest_len
        elastics: dicts with kind, n_verts, n_cells, v_offset, mu, lam, alpha, tets, B, W
        bodies: (v_start, v_end, f_start, f_end); pairs: (b_idx, v_start, v_end, mu or None[, factor on mu_cloth_elastic])"""
        self.L = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.TslLibraryError("no HIP device visible: thinshelllab_amd has no CPU path")
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.tot_NV = int(tot_NV)
        self._keep = []
        cl = (ClothDesc * max(len(cloths), 1))()
        for i, c in enumerate(cloths):
            arrs = [_np(c["f2v"], np.int32), _np(c["counter_face"], np.int32), _np(c["counter_point"], np.int32),
                    _np(c["rest_area"], np.float64), _np(c["rest_len"], np.float64)]
            self._keep += arrs
            cl[i] = ClothDesc(c["N"], c["M"], c["NV"], c["NF"], c["v_offset"], c["dx"], c["mass"], c["Kl"], c["Ka"], c["Kb"], c["k_angle"],
                              *[a.ctypes.data for a in arrs])
        el = (ElasticDesc * max(len(elastics), 1))()
        for i, e in enumerate(elastics):
            arrs = [_np(e["tets"], np.int32), _np(e["B"], np.float64), _np(e["W"], np.float64)]
            self._keep += arrs
            el[i] = ElasticDesc(e["kind"], e["n_verts"], e["n_cells"], e["v_offset"], e["mu"], e["lam"], e["alpha"], *[a.ctypes.data for a in arrs])
        bd = (Body * max(len(bodies), 1))()
        for i, b in enumerate(bodies):
            bd[i] = Body(*[int(x) for x in b])
        pr = (ContactPair * max(len(pairs), 1))()
        for i, p in enumerate(pairs):
            mu = p[3]   # float: fixed; None: live mu_cloth_elastic; "cloth_cloth": live mu_cloth_cloth
            factor = float(p[4]) if len(p) > 4 else 0.0
            kind = 0 if isinstance(mu, (int, float)) else (2 if mu == "cloth_cloth" else 1)
            pr[i] = ContactPair(int(p[0]), int(p[1]), int(p[2]), kind, float(mu) if kind == 0 else factor)
        faces = _np(faces if faces is not None else np.zeros((0, 3)), np.int32)
        mass = _np(mass, np.float64); gravity = _np(gravity, np.float64); frozen = _np(frozen, np.int32)
        assert mass.shape == (tot_NV,) and gravity.shape == (tot_NV, 3) and frozen.shape == (3 * tot_NV,)
        d = SceneDesc(tot_NV, len(faces), dt, k_contact, eps_contact, eps_v, damping, int(max_n_constraints),
                      len(cloths), cl, len(elastics), el, len(bodies), bd, len(pairs), pr,
                      mass.ctypes.data, gravity.ctypes.data, faces.ctypes.data, frozen.ctypes.data, grid_h)
        self.h = C.c_void_p()
        check(self.L.tsl_ctx_create(C.byref(d), C.byref(self.h)), "tsl_ctx_create")
        self.n_body = len(bodies)
        self.n_cface = sum(c["NF"] for c in cloths)
        self.dt = dt
        self.max_n_constraints = int(max_n_constraints)
        self.L.tsl_set_stream(self.h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        # solver switches for A/B runs of unchanged drivers: TSL_PARAMS="key=value,key=value" (keys of tsl_set_param)
        for kv in os.environ.get("TSL_PARAMS", "").split(","):
            if "=" in kv:
                k, v = kv.split("=", 1)
                self.set_param(k.strip(), float(v))

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            g = getattr(self, "_group", None)
            if g is not None:   # member of a scene group: the group goes first (the members get buffers of their own again)
                g.close()
            self.L.tsl_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters
    def set_param(self, key, value):
        check(self.L.tsl_set_param(self.h, key.encode(), float(value)), f"tsl_set_param({key})")

    def set_frozen(self, frozen):
        f = _np(frozen, np.int32)
        check(self.L.tsl_set_frozen(self.h, f.ctypes.data), "tsl_set_frozen")

    def set_ext_force(self, f):
        f = _np(f, np.float64)
        check(self.L.tsl_set_ext_force(self.h, f.ctypes.data), "tsl_set_ext_force")

    def set_gravity(self, g):
        g = _np(g, np.float64)
        check(self.L.tsl_set_gravity(self.h, g.ctypes.data), "tsl_set_gravity")

    # ---- engine calls
    def energy(self, pos, prev_pos, vel, ref_angle):
        self.refresh_stream()
        e = C.c_double(0)
        check(self.L.tsl_energy(self.h, _ptr(pos), _ptr(prev_pos), _ptr(vel), _ptr(ref_angle), C.byref(e)), "tsl_energy")
        return e.value

    def assemble(self, pos, prev_pos, vel, ref_angle, spd=True, grad=None):
        self.refresh_stream()
        check(self.L.tsl_assemble(self.h, _ptr(pos), _ptr(prev_pos), _ptr(vel), _ptr(ref_angle), int(bool(spd)), _ptr(grad)), "tsl_assemble")

    def solve(self, rhs, x=None):
        self.refresh_stream()
        if x is None:
            x = torch.empty_like(rhs)
        st = SolveStats()
        check(self.L.tsl_solve(self.h, _ptr(rhs), _ptr(x), C.byref(st)), "tsl_solve")
        return x, st.as_dict()

    def step(self, pos, prev_pos, vel, ref_angle):
        self.refresh_stream()
        st = StepStats()
        check(self.L.tsl_step(self.h, _ptr(pos), _ptr(prev_pos), _ptr(vel), _ptr(ref_angle), C.byref(st)), "tsl_step")
        return st.as_dict()

    def contact_detect(self, pos, prev_pos):
        self.refresh_stream()
        nc = C.c_int32(0)
        check(self.L.tsl_contact_detect(self.h, _ptr(pos), _ptr(prev_pos), C.byref(nc)), "tsl_contact_detect")
        return nc.value

    def contact_reset(self):
        check(self.L.tsl_contact_reset(self.h), "tsl_contact_reset")

    def update_ref_angle(self, pos, ref_angle):
        check(self.L.tsl_update_ref_angle(self.h, _ptr(pos), _ptr(ref_angle)), "tsl_update_ref_angle")

    def adjoint_step(self, step, T, pos_buffer, pos_grad, ref_angle_buffer, angleref_grad, tmp_z_frozen, damping=1.0):
        self.refresh_stream()
        st = SolveStats()
        check(self.L.tsl_adjoint_step(self.h, int(step), int(T), _ptr(pos_buffer), _ptr(pos_grad), _ptr(ref_angle_buffer), _ptr(angleref_grad),
                                      _ptr(tmp_z_frozen), float(damping), C.byref(st)), "tsl_adjoint_step")
        return st.as_dict()

    def elastic_force(self, pos, out):
        check(self.L.tsl_elastic_force(self.h, _ptr(pos), _ptr(out)), "tsl_elastic_force")
        return out

    def friction_grad(self, pos):
        out = C.c_double(0)
        check(self.L.tsl_friction_grad(self.h, _ptr(pos), C.byref(out)), "tsl_friction_grad")
        return out.value

    def param_grad(self, pos, ref_angle):
        """{kb, mu, lam} contributions of the last adjoint_step (system identification)"""
        out = (C.c_double * 3)()
        check(self.L.tsl_param_grad(self.h, _ptr(pos), _ptr(ref_angle), out), "tsl_param_grad")
        return dict(kb=out[0], mu=out[1], lam=out[2])

    # ---- introspection (tests)
    def matrix(self):
        """(row_ptr, col, vals[nnzb,3,3]) of the masked system matrix of the last assemble (static part)."""
        nb = C.c_int32(0); nnzb = C.c_int32(0)
        check(self.L.tsl_matrix_nnzb(self.h, C.byref(nb), C.byref(nnzb)), "tsl_matrix_nnzb")
        rp = np.zeros(nb.value + 1, np.int32); col = np.zeros(nnzb.value, np.int32); vals = np.zeros((nnzb.value, 3, 3), np.float64)
        check(self.L.tsl_matrix_export(self.h, rp.ctypes.data, col.ctypes.data, vals.ctypes.data), "tsl_matrix_export")
        return rp, col, vals

    def matrix_csr(self):
        import scipy.sparse as sp
        rp, col, vals = self.matrix()
        n = 3 * (len(rp) - 1)
        return sp.bsr_matrix((vals, col, rp), shape=(n, n)).tocsr()

    def constraints(self):
        m = self.max_n_constraints
        idx = np.zeros((m, 4), np.int32); w = np.zeros((m, 3)); k = np.zeros(m); dx0 = np.zeros((m, 3)); T = np.zeros((m, 6)); n = np.zeros((m, 3)); mu = np.zeros(m)
        cnt = check(self.L.tsl_constraints_export(self.h, idx.ctypes.data, w.ctypes.data, k.ctypes.data, dx0.ctypes.data, T.ctypes.data, n.ctypes.data,
                                                  mu.ctypes.data, m), "tsl_constraints_export")
        return dict(idx=idx[:cnt], w=w[:cnt], k=k[:cnt], dx0=dx0[:cnt], T=T[:cnt], n=n[:cnt], mu=mu[:cnt])

    def contact_blocks(self, masked=True):
        m = self.max_n_constraints
        blk = np.zeros((m, 12, 12))
        cnt = check(self.L.tsl_contact_blocks_export(self.h, blk.ctypes.data, m, int(masked)), "tsl_contact_blocks_export")
        return blk[:cnt]

    def operator_csr(self):
        """full solve operator: static masked matrix + matrix-free contact blocks, as scipy CSR"""
        import scipy.sparse as sp
        A = self.matrix_csr().tolil()
        cons = self.constraints()
        blk = self.contact_blocks(True)
        for c in range(len(blk)):
            idx = cons["idx"][c]
            dofs = np.concatenate([3 * idx[k] + np.arange(3) for k in range(4)])
            for r in range(12):
                for cc in range(12):
                    if blk[c, r, cc] != 0.0:
                        A[dofs[r], dofs[cc]] += blk[c, r, cc]
        return A.tocsr()

    def proj_export(self):
        nb = max(self.n_body, 1)
        flag = np.zeros((nb, self.tot_NV), np.int32); dr = np.zeros((nb, self.tot_NV), np.int32)
        pidx = np.zeros((nb, self.tot_NV, 3), np.int32); pw = np.zeros((nb, self.tot_NV, 3))
        check(self.L.tsl_proj_export(self.h, flag.ctypes.data, dr.ctypes.data, pidx.ctypes.data, pw.ctypes.data), "tsl_proj_export")
        return flag, dr, pidx, pw

    def proj_import(self, flag, dr):
        flag = _np(flag, np.int32); dr = _np(dr, np.int32)
        check(self.L.tsl_proj_import(self.h, flag.ctypes.data, dr.ctypes.data), "tsl_proj_import")

    def set_border(self, border):
        b = _np(border, np.int32)
        assert b.shape == (self.tot_NV,)
        check(self.L.tsl_set_border(self.h, b.ctypes.data), "tsl_set_border")

    def refresh_stream(self):
        """engine calls are ordered against torch's CURRENT stream of the context's device (it may change between calls)"""
        self.L.tsl_set_stream(self.h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))

    def spd_project(self, blocks, D):
        check(self.L.tsl_spd_project(self.h, _ptr(blocks), blocks.numel() // (D * D), D), "tsl_spd_project")

    def profile_reset(self, enable=True):
        check(self.L.tsl_profile_reset(self.h, int(bool(enable))), "tsl_profile_reset")

    def profile_read(self):
        ms = C.c_double(0); n = C.c_int64(0); b = C.c_int64(0)
        check(self.L.tsl_profile_read(self.h, C.byref(ms), C.byref(n), C.byref(b)), "tsl_profile_read")
        ev = C.c_double(0)
        check(self.L.tsl_profile_read_events(self.h, C.byref(ev)), "tsl_profile_read_events")
        return dict(ms_per_launch=ms.value, launches=n.value, bytes_per_launch=b.value, ms_per_launch_events=ev.value)

    def bench_direct(self, cls, reps=20):
        """kernel class of the sparse direct path replayed back to back (0 Gauss-Jordan inversions on the block-step path, 1 Schur GEMM + extend-add, 2 G = W F12 GEMM, 3 inversions in the LDS kernel, 4 gemv sweeps of one application, 5 inversions in the persistent dataflow kernel)"""
        out = (C.c_double * 4)()
        check(self.L.tsl_bench_direct(self.h, int(cls), int(reps), out), "tsl_bench_direct")
        return dict(us_per_launch=out[0], flops_per_launch=out[1], bytes_per_launch=out[2], launches=int(out[3]))

    def direct_info(self):
        out = (C.c_double * 10)()
        check(self.L.tsl_direct_info(self.h, out), "tsl_direct_info")
        keys = ("plans", "factorizations", "applications", "perturbed_pivots", "plan_seconds", "supernodes", "levels", "batches", "flops_per_factorization", "front_bytes")
        return dict(zip(keys, [float(v) for v in out]))

    def direct_counters(self):
        out = (C.c_double * 13)()
        check(self.L.tsl_direct_counters(self.h, out, 13), "tsl_direct_counters")
        keys = ("flow_launches", "flow_aborts", "plan_cache_hits", "panel_bytes", "schur_bytes", "g_bytes", "schur_entries", "plans_parked",
                "berr_seen", "berr_accepted", "berr_max", "berr_rel_max", "tiles_guarded")
        return dict(zip(keys, [float(v) for v in out]))

    def bench_spmv(self, variant=20, reps=500):
        """microseconds per launch of `reps` back-to-back operator launches between one hipEvent pair (20 = k_pcg_spmv)"""
        us = C.c_double(0)
        check(self.L.tsl_bench_spmv(self.h, int(variant), int(reps), C.byref(us)), "tsl_bench_spmv")
        return us.value
