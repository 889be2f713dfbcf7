"""Loss seeds and rewards: the product's host code (thinshelllab_amd/engine/analytic_grad_single.py,
task_scene/Scene_*.py) against the ORACLE's own restatement of the reference kernels (oracle/tslo_loss.cpp, loop for loop from
/root/reference/code/engine/analytic_grad_single.py:259-471 and the compute_reward* kernels of /root/reference/code/task_scene).
Both sides hold the same random tape / state and seed THEMSELVES; the arrays must be equal entry for entry.  Runs on the CPU (the seeds
are host code on both sides); the -m gpu sweeps (tests/test_gpu_*.py) repeat the comparison on the tape of a real rollout before they
start the reverse sweep."""
import importlib

import numpy as np
import pytest

from helpers import oracle_from_scene
from oracle import pyoracle as po
from thinshelllab_amd.engine.analytic_grad_single import Grad


def _pair(name, T=6, seed=0, **kw):
    mod = importlib.import_module(f"thinshelllab_amd.task_scene.Scene_{name}")
    s = mod.Scene(device="cpu", **kw)
    s.init_all()
    o = oracle_from_scene(po, s)
    n_part = s.gripper.n_part if hasattr(s, "gripper") else 0
    g = Grad(s, T, n_part)
    o.grad_new(T, n_part)
    rng = np.random.default_rng(seed)
    pb = s.pos.to_numpy()[None] + rng.normal(0, 1e-3, (T, s.tot_NV, 3))
    g.pos_buffer.from_numpy(pb)
    o.arr("grad.pos_buffer", (T, s.tot_NV, 3))[:] = pb
    return s, o, g


def _same_seeds(o, g, T, s):
    pg_o = o.arr("grad.pos_grad", (T, s.tot_NV, 3)); ag_o = o.arr("grad.angleref_grad", (T, s.cloth_cnt, -1, 3))
    assert np.array_equal(g.pos_grad.to_numpy(), pg_o)
    assert np.array_equal(g.angleref_grad.to_numpy().reshape(ag_o.shape), ag_o)
    return np.abs(pg_o).sum() + np.abs(ag_o).sum()


def _randomise_state(s, o, seed=1):
    rng = np.random.default_rng(seed)
    x = s.pos.to_numpy() + rng.normal(0, 1e-3, (s.tot_NV, 3))
    s.pos.from_numpy(x)
    o.pos[:] = x
    o.push_down_all()
    for i, c in enumerate(s.cloths):
        ra = rng.normal(0, 0.3, (c.NF, 3))
        c.ref_angle.from_numpy(ra)
        o.arr(f"cloth{i}.ref_angle", (-1, 3))[:] = ra


# (scene, product seed call, oracle seed name + arguments)
SEEDS = [
    ("folding", lambda g, s: g.get_loss_fold(s, 1.0, -1.0), ("fold", 1.0, -1.0)),
    ("folding", lambda g, s: g.get_loss(s), ("",)),
    ("folding", lambda g, s: g.get_loss_sheet(s), ("sheet",)),
    ("folding", lambda g, s: g.get_loss_book(s), ("book",)),
    ("folding", lambda g, s: g.get_loss_pick(s), ("pick",)),
    ("folding", lambda g, s: g.get_loss_card(s), ("card",)),
    ("folding", lambda g, s: g.get_loss_pick_fold(s), ("pick_fold",)),
    ("folding", lambda g, s: g.get_loss_slide_simple(s), ("slide_simple",)),
    ("lifting", lambda g, s: g.get_loss_lift(s), ("lift",)),
    ("balancing", lambda g, s: g.get_loss_balance(s), ("balance",)),
    ("balancing", lambda g, s: g.get_loss_side(s), ("side",)),
    ("balancing", lambda g, s: g.get_loss_throwing(s), ("throwing",)),
    ("interact", lambda g, s: g.get_loss_interact(s), ("interact",)),
    ("interact", lambda g, s: g.get_loss_interact_1(s), ("interact_1",)),
    ("sliding", lambda g, s: g.get_loss_sep(s), ("sep",)),
]


@pytest.mark.parametrize("scene,prod,orc", SEEDS, ids=[f"{a}-{c[0] or 'get_loss'}" for a, b, c in SEEDS])
def test_loss_seed_product_vs_oracle(scene, prod, orc):
    T = 6
    s, o, g = _pair(scene, T)
    prod(g, s)
    o.grad_loss(*orc)
    assert _same_seeds(o, g, T, s) > 0


def test_folding_seed_with_scaled_rows():
    """the refined folding cloths scale the reference's hard-coded hinge rows (6, 8) / (7, 9) by N / 15 (Scene.fold_rows)"""
    T = 4
    s, o, g = _pair("folding", T, cloth_size=0.1, cloth_N=30, cloth_M=6)
    rows = s.fold_rows()
    assert rows == ((12, 14), (13, 15))
    g.get_loss_fold(s, 0.7, -1.3, rows=rows)
    o.grad_loss("fold", 0.7, -1.3, rows=np.array(rows).ravel())
    assert _same_seeds(o, g, T, s) > 0


def test_push_deliver_and_bounce_seeds():
    T = 75
    s, o, g = _pair("folding", T)
    c = s.cloths[0]
    tgt = np.random.default_rng(3).normal(size=(c.NV, 3))
    g.get_loss_push(s, tgt); o.grad_loss("push", target=tgt)
    assert _same_seeds(o, g, T, s) > 0
    g.reset(); o.grad_reset()
    pb = g.pos_buffer.to_numpy(); o.arr("grad.pos_buffer", pb.shape)[:] = pb   # (reset keeps the tape on neither side's contract: write it again)
    g.pos_buffer.from_numpy(pb)
    g.get_loss_deliver(s); o.grad_loss("deliver")
    assert _same_seeds(o, g, T, s) > 0
    g.reset(); o.grad_reset()
    g.pos_buffer.from_numpy(pb); o.arr("grad.pos_buffer", pb.shape)[:] = pb
    s.target = 0.0123
    tt_p = g.get_loss_bounce(s); tt_o = o.grad_loss("bounce", s.target)
    assert tt_p == tt_o and 40 <= tt_p < T
    assert _same_seeds(o, g, T, s) > 0


REWARDS = [
    ("folding", lambda s, g: s.compute_reward(0.6, -1.1), ("folding", 0.6, -1.1)),
    ("folding", lambda s, g: s.compute_reward_8(), ("folding.8",)),
    ("folding", lambda s, g: s.compute_reward_7(), ("folding.7",)),
    ("lifting", lambda s, g: s.compute_reward(), ("lifting",)),
    ("balancing", lambda s, g: s.compute_reward(), ("balancing",)),
    ("balancing", lambda s, g: s.compute_reward_all(g), ("balancing.all",)),
    ("bouncing", lambda s, g: s.compute_reward(), ("bouncing",)),
    ("card", lambda s, g: s.compute_reward(), ("card",)),
    ("sliding", lambda s, g: s.compute_reward(), ("sliding",)),
    ("interact", lambda s, g: s.compute_reward(), ("interact",)),
    ("interact", lambda s, g: s.compute_reward_1(), ("interact.1",)),
    ("pick", lambda s, g: s.compute_reward(), ("pick",)),
]


@pytest.mark.parametrize("scene,prod,orc", REWARDS, ids=[c[0] for a, b, c in REWARDS])
def test_reward_product_vs_oracle(scene, prod, orc):
    s, o, g = _pair(scene, 6)
    _randomise_state(s, o)
    rp = prod(s, g)
    ro = o.reward(*orc)
    assert np.isfinite(ro) and ro != 0.0
    assert abs(rp - ro) <= 1e-13 * max(1.0, abs(ro)), (rp, ro)


def test_forming_and_pick_tape_rewards():
    s, o, g = _pair("forming", 4)
    _randomise_state(s, o)
    tgt = s.cloths[0].pos.to_numpy() + 1e-3
    assert abs(s.compute_reward(tgt) - o.reward("forming", target=tgt)) < 1e-15
    s, o, g = _pair("pick", 72)
    _randomise_state(s, o)
    rp = s.compute_reward_deliver(g); ro = o.reward("pick.deliver")
    assert abs(rp - ro) <= 1e-13 * abs(ro) and ro != 0.0
    # hinge rewards read the dihedral angle of the CURRENT pose: the oracle's compute_angle needs its face normals
    o.prepare_bending()
    for prod, name in ((s.compute_reward_pick_fold, "pick.pick_fold"), (s.compute_reward_pick_and_fold, "pick.pick_and_fold")):
        rp = prod(); ro = o.reward(name)
        assert abs(rp - ro) <= 1e-11 * max(1.0, abs(ro)), (name, rp, ro)
