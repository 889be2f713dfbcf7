"""tsl_spd_project (include/tsl_hip.h) on the GPU: batched symmetric eigen-clamp of 2 x 2, 3 x 3 and
9 x 9 blocks against numpy ``eigh``, the oracle's converged Jacobi projector and the oracle's LITERAL restatement of the reference
(Householder + K shifted-QR sweeps, /root/reference/code/engine/linalg.py:15-148; 2 x 2: ti.svd rule :5-12).  The device runs a
converged cyclic Jacobi: it must agree with the literal projector wherever that one converged within its K sweeps."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from thinshelllab_amd.task_scene.Scene_drape import Scene
    s = Scene(cloth_size=0.1 / 15 * 8, N=8, M=8)
    s.init_all()
    return s._ensure_ctx()


def _clamp_eigh(A):
    w, V = np.linalg.eigh(0.5 * (A + A.T))
    return (V * np.maximum(w, 0)) @ V.T


@pytest.mark.parametrize("D,K,scale", [(2, 0, 1.0), (2, 0, 1e4), (3, 10, 1.0), (3, 10, 1e6), (9, 20, 1e3), (9, 20, 1.0)])
def test_spd_project_matches_eigh_and_oracle(oracle, ctx, D, K, scale):
    rng = np.random.default_rng(10 * D + int(np.log10(scale)))
    n = 96
    A = rng.normal(size=(n, D, D)) * scale
    A = 0.5 * (A + A.transpose(0, 2, 1))
    A[0] = np.eye(D) * scale            # already SPD: unchanged
    A[1] = -np.eye(D) * scale           # negative definite: zero
    A[2] = 0.0
    x = rng.normal(size=D); A[3] = np.outer(x, x) * scale - 0.3 * scale * np.eye(D)   # one positive, the rest negative
    dev = torch.as_tensor(A, device="cuda").contiguous()
    ctx.spd_project(dev, D)
    G = dev.cpu().numpy()
    worst_lit = 0.0
    n_conv = 0
    for i in range(n):
        E = _clamp_eigh(A[i])
        tol = 1e-12 * max(np.abs(A[i]).max(), 1e-300) + 1e-300
        assert np.abs(G[i] - E).max() <= 50 * tol, (D, i, np.abs(G[i] - E).max())
        assert np.abs(G[i] - G[i].T).max() <= 50 * tol
        assert np.linalg.eigvalsh(0.5 * (G[i] + G[i].T)).min() >= -1e-9 * max(np.abs(A[i]).max(), 1e-300)
        if D == 2:
            L = oracle.spd_project_2d(A[i])
            assert np.abs(G[i] - L).max() <= 1e-10 * max(np.abs(A[i]).max(), 1e-300), (i, G[i], L)
        else:
            J = oracle.spd_project_jacobi(A[i])
            assert np.abs(G[i] - J).max() <= 50 * tol
            L, sweeps = oracle.spd_project(A[i], K)     # literal reference projector
            if sweeps < K:                                # converged within its sweep limit
                n_conv += 1
                # the literal projector stops at an ABSOLUTE sub-diagonal tolerance of 1e-5 (linalg.py:83): its own error is
                # ~1e-5 |A|^0 .. 1e-8 |A|, which bounds the agreement
                worst_lit = max(worst_lit, np.abs(G[i] - L).max() / max(np.abs(A[i]).max(), 1.0))
    if D > 2:
        assert n_conv >= n // 2
        assert worst_lit < 2e-5, worst_lit
