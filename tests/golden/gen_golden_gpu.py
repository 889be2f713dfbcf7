"""Generates tests/golden/det_<scene>_<grid>.npz ON AN MI355X (python tests/golden/gen_golden_gpu.py): 10 driven steps + the reverse sweep
of cfg4 (balancing, 224 x 224) and cfg3 (folding, 200 x 100) with the engine's deterministic assembly.  The rollout is produced twice and
must agree bit for bit before it is written.  Kept small: digests of the full arrays + a sample of 512 vertices + gripper_grad + the
per-step solver statistics.  (The single evaluations of these states against the oracle are tests/test_gpu_direct_parity.py.)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from test_gpu_determinism import digest, rollout  # noqa: E402

for name, grid, T in (("balancing", 224, 11), ("folding", 200, 11)):
    a = rollout(name, T, grid=grid); b = rollout(name, T, grid=grid)
    for k in a:
        assert np.array_equal(a[k], b[k]), (name, k)
    NV = a["pos_buffer"].shape[1]
    sel = np.linspace(0, NV - 1, 512).astype(np.int64)
    out = os.path.join(sys.argv[1] if len(sys.argv) > 1 else HERE, f"det_{name}_{grid}.npz")
    np.savez_compressed(out, sample_idx=sel, pos_sample=a["pos_buffer"][:, sel], pos_grad_sample=a["pos_grad"][:, sel], gripper_grad=a["gripper_grad"], stats=a["stats"],
                        sha_pos_buffer=digest(a["pos_buffer"]), sha_pos_grad=digest(a["pos_grad"]))
    print(out, os.path.getsize(out), "bytes; unconverged", a["stats"][:, 4].sum(), "nc", a["stats"][:, 0])
