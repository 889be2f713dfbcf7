"""Shared builders for the parity tests (the implementation lives with the oracle: oracle/mirror.py)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.mirror import oracle_from_scene, rel_err, sync_oracle_state  # noqa: E402,F401


def seeds_agree(g, o, T, NV):
    """both sides seeded THEMSELVES (product: Grad.get_loss_*, oracle: tslo_loss.cpp restating the reference kernels): the seed arrays
    must be equal entry for entry before the reverse sweep starts"""
    import numpy as np
    pg_o = o.arr("grad.pos_grad", (T, NV, 3)); ag_o = o.arr("grad.angleref_grad").reshape(g.angleref_grad.shape)
    assert np.array_equal(g.pos_grad.to_numpy(), pg_o), "pos_grad seeds differ"
    assert np.array_equal(g.angleref_grad.to_numpy(), ag_o), "angleref_grad seeds differ"
    assert np.abs(pg_o).max() + np.abs(ag_o).max() > 0


def start_of_step_positions(g, dt):
    """positions at the START of the reference's last step: the fixture holds the state after the step; with damping = 1
    (Scene_balancing / BaseScene.py:872 update_vel: v = (x - x_prev) / dt) the start is exactly x - v dt"""
    import torch
    st = torch.load(os.path.join(g, "state"), weights_only=False)
    return st["pos"].numpy() - st["vel"].numpy() * dt


def assert_cloth_target_mismatches_are_the_cell_boundary_triangle(flag0, F0, pidx0, x, grid_h=0.003):
    """What is left of the cloth-as-target row once the query runs at the start-of-step positions: every mismatching query vertex is flagged
    by the restatement and not by the reference (never the other way round), and its candidate triangle has a centroid within 1e-5 m of a
    face of the broad-phase grid (geometry.py:89-94 `ti.floor(x / grid_h)`, grid_h = 0.003): the reference binned that triangle from positions
    that differ from x - v dt by rounding (v = (x - x_prev) / dt was itself rounded) and found it one cell further, outside the 3 x 3 x 3
    neighbourhood (geometry.py:165-221).  On the fixture: ten pad-1 vertices, all with the cloth triangle [109, 108, 116], centroid z = +8.9 um."""
    mm = np.nonzero(flag0 != F0)[0]
    assert len(mm) <= 12, mm
    assert (flag0[mm] == 1).all() and (F0[mm] == 0).all(), (flag0[mm], F0[mm])
    tris = set()
    for v in mm:
        c = x[pidx0[v]].mean(0)
        d = np.abs(c / grid_h - np.round(c / grid_h)) * grid_h   # distance of the centroid to the nearest cell face, per axis
        assert d.min() < 1e-5, (v, pidx0[v], c, d)
        tris.add(tuple(int(t) for t in pidx0[v]))
    assert len(tris) <= 1, tris
    return mm
