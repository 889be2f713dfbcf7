"""Scene constants against tables transcribed HERE from the reference's task scenes (the oracle takes pair lists and frozen masks
from the product's scene objects, oracle/mirror.py -- a transcription slip in a Scene_*.py would otherwise pass every parity test).

contact pairs: /root/reference/code/task_scene/Scene_*.py `contact_analysis` -- (surface body, queried body, friction) in call order;
frozen sets:   `set_frozen_kernel` -- "all" vertices of a body, the gripper-driven "bound" of a tactile pad (is_bottom or
               is_inner_circle: 49 of 276 vertices), the cloth's last grid "row" (M + 1 vertices).
Friction: a number = fixed mu; "ce" = the live mu_cloth_elastic; "ce*10" = ten times that (Scene_card.py:125-127); "cc" = the live
mu_cloth_cloth (Scene_sliding.py:83)."""
import numpy as np
import pytest

C, E = "cloth", "elastic"


def _both(a, b, mu):
    return [(a, b, mu), (b, a, mu)]


def _cloth_vs_bodies(n_el, special=None):
    """for j in range(elastic_cnt): (cloth0 surface <- elastic j vertices), (elastic j surface <- cloth0 vertices)"""
    out = []
    for j in range(n_el):
        out += _both((C, 0), (E, j), (special or {}).get(j, "ce"))
    return out


TABLE = {
    # Scene_balancing.py:98-109 (ball: 0.2), frozen :111-136
    "balancing": dict(cloths=1, elastics=5, pairs=_cloth_vs_bodies(5, {0: 0.2}), frozen=[(E, 1, "bound"), (E, 2, "bound"), (E, 3, "bound"), (E, 4, "bound")]),
    # Scene_bouncing.py:91-97: only the table's triangles against the cloth's vertices; frozen :100-105
    "bouncing": dict(cloths=1, elastics=1, pairs=[((E, 0), (C, 0), "ce")], frozen=[(E, 0, "all")]),
    # Scene_card.py:112-129: i, j loops with abs(i - j) == 1 (each neighbour couple twice), then every pad/table surface against every card; frozen :131-155
    "card": dict(cloths=3, elastics=4,
                 pairs=_both((C, 0), (C, 1), 0.1) + _both((C, 1), (C, 0), 0.1) + _both((C, 1), (C, 2), 0.1) + _both((C, 2), (C, 1), 0.1)
                 + [((E, j), (C, i), "ce" if i == 0 else "ce*10") for i in range(3) for j in range(4)],
                 frozen=[(E, 0, "all"), (E, 1, "bound"), (E, 2, "bound"), (E, 3, "bound")]),
    # Scene_folding.py:99-108, frozen :110-127
    "folding": dict(cloths=1, elastics=2, pairs=_cloth_vs_bodies(2), frozen=[(E, 0, "all"), (E, 1, "bound"), (C, 0, "row")]),
    # Scene_forming.py:95-104, frozen :106-123
    "forming": dict(cloths=1, elastics=2, pairs=_cloth_vs_bodies(2), frozen=[(E, 0, "all"), (E, 1, "bound"), (C, 0, "row")]),
    # Scene_interact.py:109-125 (table and block 0.2; block on table 0.1), frozen :128-147
    "interact": dict(cloths=1, elastics=4, pairs=_cloth_vs_bodies(4, {0: 0.2, 3: 0.2}) + [((E, 0), (E, 3), 0.1), ((E, 3), (E, 0), 0.1)],
                     frozen=[(E, 0, "all"), (E, 1, "bound"), (E, 2, "bound")]),
    # Scene_lifting.py:113-129 (one cloth: the cloth-cloth loop is empty), frozen :131-150
    "lifting": dict(cloths=1, elastics=4, pairs=_cloth_vs_bodies(4), frozen=[(E, 1, "bound"), (E, 2, "bound"), (E, 3, "bound")]),
    # Scene_pick.py:72-88 (table 0.1), frozen :90-108
    "pick": dict(cloths=1, elastics=3, pairs=_cloth_vs_bodies(3, {0: 0.1}), frozen=[(E, 0, "all"), (E, 1, "bound"), (E, 2, "bound")]),
    # Scene_sliding.py:79-99 (sheets: mu_cloth_cloth; table 0.4), frozen :101-113
    "sliding": dict(cloths=3, elastics=2,
                    pairs=_both((C, 0), (C, 1), "cc") + _both((C, 1), (C, 0), "cc") + _both((C, 1), (C, 2), "cc") + _both((C, 2), (C, 1), "cc")
                    + [p for i in range(3) for j in range(2) for p in _both((C, i), (E, j), 0.4 if j == 0 else "ce")],
                    frozen=[(E, 0, "all"), (E, 1, "bound")]),
}


def _scene(name):
    import importlib
    mod = importlib.import_module(f"thinshelllab_amd.task_scene.Scene_{name}")
    s = mod.Scene(device="cpu")
    s.init_all()
    return s


def _mu_spec(p):
    mu = p[3]
    factor = p[4] if len(p) > 4 else 0.0
    if mu is None:
        return "ce*10" if factor == 10.0 else "ce"
    if mu == "cloth_cloth":
        return "cc"
    return float(mu)


@pytest.mark.parametrize("name", sorted(TABLE))
def test_contact_pairs_and_frozen_sets_match_the_reference_tables(name):
    t = TABLE[name]
    s = _scene(name)
    assert s.cloth_cnt == len(s.cloths) == t["cloths"] and s.elastic_cnt == len(s.elastics) == t["elastics"]
    body = {(C, i): c for i, c in enumerate(s.cloths)}
    body.update({(E, j): e for j, e in enumerate(s.elastics)})

    def vrange(b):
        o = body[b]
        return (o.offset, o.offset + (o.NV if b[0] == C else o.n_verts))
    expect = [(body[surf].body_idx, *vrange(q), mu) for surf, q, mu in t["pairs"]]
    got = [(p[0], p[1], p[2], _mu_spec(p)) for p in s.contact_pairs()]
    # the call ORDER matters for scenes that split the constraint list by pair block (nc1 / nc2: Scene_pick.py:85-88, Scene_sliding.py:89)
    assert got == expect
    # bodies appear in the global layout in the reference's order: cloths first, then elastics (BaseScene.py:136-164)
    offs = [vrange((C, i))[0] for i in range(t["cloths"])] + [vrange((E, j))[0] for j in range(t["elastics"])]
    assert offs == sorted(offs) and offs[0] == 0 and vrange((E, t["elastics"] - 1))[1] == s.tot_NV
    fr = s.frozen.to_numpy().reshape(-1, 3)
    assert ((fr == 0) | (fr == 1)).all() and (fr.min(1) == fr.max(1)).all()     # whole vertices
    want = np.zeros(s.tot_NV, bool)
    for kind, idx, what in t["frozen"]:
        o = body[(kind, idx)]
        if what == "all":
            want[o.offset:o.offset + o.n_verts] = True
        elif what == "bound":
            m = np.asarray(o.bound_mask(), bool)
            assert m.sum() == 49 and o.n_verts == 276     # frozen_cnt 49 of the tactile mesh
            want[o.offset:o.offset + o.n_verts][m] = True
        else:
            want[o.offset + o.N * (o.M + 1): o.offset + (o.N + 1) * (o.M + 1)] = True
    assert np.array_equal(fr[:, 0].astype(bool), want)
