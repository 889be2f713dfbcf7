"""Shared builders for the parity tests (the implementation lives with the oracle: oracle/mirror.py)."""
import os
import sys

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
