"""Test infrastructure (see oracle/README.md): mirror a product scene (thinshelllab_amd, HIP) in the oracle (CPU restatement).
Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only."""
import numpy as np


def oracle_from_scene(po, sys, check_init=True):
    """Build an OracleScene mirroring a product scene (after ``init_all``): same bodies, parameters, contact
    pairs, gripper and state.  With ``check_init`` the oracle runs ITS OWN initialisation code (mesh tables, poses,
    rest matrices, masses, surface orientation) and the result is compared with the product's host code."""
    g = np.asarray(sys.gravity[None], dtype=np.float64)
    o = po.OracleScene(dt=sys.dt, k_contact=sys.k_contact, eps_contact=sys.eps_contact, eps_v=sys.eps_v, damping=sys.damping,
                       max_n_constraints=sys.max_n_constraints, newton_cap=sys._newton_cap, plastic=sys._plastic,
                       effector_cnt=sys.effector_cnt, gravity=tuple(g), mu_cloth_elastic=sys.mu_cloth_elastic.value)
    for i, c in enumerate(sys.cloths):
        ci = o.add_cloth(c.N, c.M, c.dx * c.N, rho=c.rho, is_square=False)
        for k in ("Kb", "Kl", "Ka", "k_angle"):
            o.set_scalar(f"cloth{i}.{k}", getattr(c, k).value)
        kind, ox, oy, oz, curv = c._init_args
        if kind == "flat":
            o.cloth_init(ci, ox, oy, oz)
        elif kind == "bridge":
            o.cloth_init(ci, ox, oy, oz, bridge=True)
        elif kind == "fold":
            o.cloth_init(ci, ox, oy, oz, fold=True, curv=curv)
        else:   # "fold_scaled" (refined folding grids): mesh tables + rest data (V, l_i: model_fold_offset.py:863-868) from the flat
            # initialisation; the pose itself (a generalisation the reference does not have) comes from the product scene below
            o.cloth_init(ci, ox, oy, oz)
    for e in sys.elastics:
        if e.kind == 0:
            ei = o.add_tactile(e.ratio, e.F_ox_array, e.F_vertices_array, e.f2v_array)
        elif getattr(e, "load", False):
            ei = o.add_loaded(e.density, e.vertex, e.tet_mesh, e.surface_mesh)
        else:
            ei = o.add_box(e.dx * (int(e.n_cube.max()) - 1), *[int(x) for x in e.n_cube], density=e.density)
        if getattr(e, "_arch", 0.0):
            o.elastic_init_arch(ei, *e._init_args[:3], e._arch)
        else:
            o.elastic_init(ei, *e._init_args)
    o.finalize()
    if getattr(sys, "grid_extent", None):
        o.set_scalar("grid_extent", sys.grid_extent)
    if getattr(sys, "grid_h", None):
        o.set_scalar("grid_h", sys.grid_h)
    for i, c in enumerate(sys.cloths):
        o.set_body_gravity(0, i, c.gravity.to_numpy())
    for i, e in enumerate(sys.elastics):
        o.set_body_gravity(1, i, e.gravity.to_numpy())
    if hasattr(sys, "gripper") and sys.elastic_cnt > 1:
        o.gripper_init(1 if sys.gripper.paired else 0, sys.gripper.n_part, sys.gripper.pos.to_numpy())
        rot = sys.gripper.rot.to_numpy()
        if not sys.gripper.paired and np.abs(rot[:, 1:]).max() > 0:   # scenes that start with rotated pads (Scene_card.py:89-94)
            o.arr("gripper.rot", (-1, 4))[:] = rot
            o.gripper_update_all()
    if check_init:
        for i, c in enumerate(sys.cloths):
            assert np.array_equal(c.f2v.to_numpy(), o.arr(f"cloth{i}.f2v", (-1, 3)))
            assert np.array_equal(c.counter_face.to_numpy(), o.arr(f"cloth{i}.counter_face", (-1, 3)))
            assert np.array_equal(c.counter_point.to_numpy(), o.arr(f"cloth{i}.counter_point", (-1, 3)))
            if c._init_args[0] != "fold_scaled":
                assert np.abs(c.pos.to_numpy() - o.arr(f"cloth{i}.pos", (-1, 3))).max() < 1e-15
                assert np.abs(c.ref_angle.to_numpy() - o.arr(f"cloth{i}.ref_angle", (-1, 3))).max() < 1e-12
        for i, e in enumerate(sys.elastics):
            assert np.abs(e.F_x.to_numpy() - o.arr(f"elastic{i}.F_x", (-1, 3))).max() < 1e-15, f"elastic {i} pose"
            assert rel_err(e.F_m.to_numpy(), o.arr(f"elastic{i}.F_m")) < 1e-12
            assert rel_err(e.F_W.to_numpy(), o.arr(f"elastic{i}.F_W")) < 1e-12
            assert rel_err(e.F_B.to_numpy().reshape(-1), o.arr(f"elastic{i}.F_B")) < 1e-10
            assert np.array_equal(e.F_vertices.to_numpy(), o.arr(f"elastic{i}.F_vertices", (-1, 4)))
            fo = o.arr(f"elastic{i}.f2v", (-1, 3)); fp = e.f2v.to_numpy()
            assert np.array_equal(np.sort(np.sort(fo, 1), 0), np.sort(np.sort(fp, 1), 0)), f"elastic {i} surface set"
        assert rel_err(sys.mass.to_numpy(), o.arr("mass")) < 1e-12
    # identical surface triangle order / orientation on both sides (the box surface order is arbitrary in the reference)
    o.arr("faces", (-1, 3))[:] = sys.faces.to_numpy()
    for i, e in enumerate(sys.elastics):
        o.arr(f"elastic{i}.f2v", (-1, 3))[:] = e.f2v.to_numpy()
    for p in sys.contact_pairs():
        o.add_pair(*p)
    sync_oracle_state(o, sys)
    return o


def sync_oracle_state(o, sys):
    """copy pos / vel / prev_pos / ref_angle / frozen / params from the product scene into a finalized oracle scene"""
    o.pos[:] = sys.pos.to_numpy(); o.vel[:] = sys.vel.to_numpy(); o.prev_pos[:] = sys.prev_pos.to_numpy()
    o.frozen[:] = sys.frozen.to_numpy()
    for i, c in enumerate(sys.cloths):
        o.arr(f"cloth{i}.ref_angle", (-1, 3))[:] = c.ref_angle.to_numpy()
        for k in ("Kb", "Kl", "Ka", "k_angle"):
            o.set_scalar(f"cloth{i}.{k}", getattr(c, k).value)
    for i, e in enumerate(sys.elastics):   # scenes may retune a body after construction (Scene_sliding.py:27-32)
        o.set_scalar(f"elastic{i}.mu", e.mu.value)
        if e.lam.value != 0:
            o.set_scalar(f"elastic{i}.lam", e.lam.value)
    o.set_scalar("mu_cloth_elastic", sys.mu_cloth_elastic.value)
    if hasattr(sys, "mu_cloth_cloth"):
        o.set_scalar("mu_cloth_cloth", sys.mu_cloth_cloth.value)
    o.push_down_all()


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
