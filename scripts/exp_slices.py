"""Probe (not part of the product): row-length / SELL-64 slice-length distribution of the cfg4 system matrix."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from thinshelllab_amd.task_scene.Scene_balancing import Scene

s = Scene(cloth_size=0.12, cloth_N=224, cloth_M=224)
s.init_all()
s.prev_pos.copy_from(s.pos)
ctx = s._ensure_ctx()
ctx.assemble(s.pos.t, s.prev_pos.t, s.vel.t if hasattr(s, "vel") else s.pos.t, s._ref_angle) if False else None
rp, col, vals = ctx.matrix()
ln = np.diff(rp)
print("rows", len(ln), "nnzb", ln.sum(), "row length min/mean/max", ln.min(), ln.mean(), ln.max())
srt = np.sort(ln)[::-1]
sl = np.array([srt[i:i + 64].max() for i in range(0, len(srt), 64)])
print("slices", len(sl), "slice_len max", sl.max(), "mean", sl.mean(), "padded slots", sl.sum() * 64, "fill", ln.sum() / (sl.sum() * 64))
h = np.bincount(sl)
print("slice_len histogram:", {int(k): int(v) for k, v in enumerate(h) if v})
