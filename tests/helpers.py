"""Shared builders: the same scene in the product (thinshelllab_amd, HIP) and in the oracle (CPU restatement)."""
import numpy as np


def oracle_from_scene(po, sys, plastic=None, newton_cap=None):
    """Build an OracleScene with the same bodies / parameters / state as a product scene (after init_all)."""
    import torch
    g = np.asarray(sys.gravity[None], dtype=np.float64)
    o = po.OracleScene(dt=sys.dt, k_contact=sys.k_contact, eps_contact=sys.eps_contact, eps_v=sys.eps_v, damping=sys.damping,
                       max_n_constraints=sys.max_n_constraints, newton_cap=sys._newton_cap if newton_cap is None else newton_cap,
                       plastic=sys._plastic if plastic is None else plastic, effector_cnt=sys.effector_cnt, gravity=tuple(g),
                       mu_cloth_elastic=sys.mu_cloth_elastic.value)
    for c in sys.cloths:
        ci = o.add_cloth(c.N, c.M, c.dx * c.N, rho=c.rho, is_square=False)
        o.L.tslo_cloth_init_mesh(o.h, ci)
    from thinshelllab_amd.engine import readfile
    for e in sys.elastics:
        if e.kind == 0:
            o.add_tactile(e.ratio, e.F_ox_array, e.F_vertices_array, e.f2v_array)
        elif getattr(e, "load", False):
            o.add_loaded(e.density, e.vertex, e.tet_mesh, e.surface_mesh)
        else:
            o.add_box(e.dx * (e.n_cube.max() - 1), *[int(x) for x in e.n_cube], density=e.density)
    return o


def sync_oracle_state(o, sys):
    """copy pos / vel / prev_pos / ref_angle / frozen / params from the product scene into a finalized oracle scene"""
    o.pos[:] = sys.pos.to_numpy(); o.vel[:] = sys.vel.to_numpy(); o.prev_pos[:] = sys.prev_pos.to_numpy()
    o.frozen[:] = sys.frozen.to_numpy()
    for i, c in enumerate(sys.cloths):
        o.arr(f"cloth{i}.ref_angle", (-1, 3))[:] = c.ref_angle.to_numpy()
        o.set_scalar(f"cloth{i}.Kb", c.Kb.value); o.set_scalar(f"cloth{i}.Kl", c.Kl.value); o.set_scalar(f"cloth{i}.Ka", c.Ka.value)
        o.set_scalar(f"cloth{i}.k_angle", c.k_angle.value)
    o.set_scalar("mu_cloth_elastic", sys.mu_cloth_elastic.value)
    o.push_down_all()


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
