"""Probe (not part of the product): is a nested-dissection LU WITHOUT pivoting a usable preconditioner for the cfg4 systems?
Runs a cfg4 rollout on the GPU, exports the forward (partly projected) and adjoint (un-projected) operators of the last state,
then factorises them with SuperLU in the geometric nested-dissection order with diagonal pivots only and reports fill, solve
accuracy and the number of refinement steps.  usage: exp_direct_proto.py [steps] [N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
import torch

from thinshelllab_amd.engine.analytic_grad_single import Grad
from thinshelllab_amd.engine.geometry import projection_query
from thinshelllab_amd.task_scene.Scene_balancing import Scene

T = int(sys.argv[1]) if len(sys.argv) > 1 else 12
N = int(sys.argv[2]) if len(sys.argv) > 2 else 224
s = Scene(cloth_size=0.12 * N / 224, cloth_N=N, cloth_M=N)
s.init_all()
s.mu_cloth_elastic[None] = 5.0
s.prev_pos.copy_from(s.pos)
ctx = s._ensure_ctx()
n_part = s.gripper.n_part
g = Grad(s, T + 1, n_part); g.init_mass(s)
g.copy_pos(s, 0)
for f in range(1, T + 1):
    dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3))
    dpos[:, 2] = 1e-4 * np.where(np.arange(n_part) % 2 == 0, 1.0, -1.0)
    s.action(f, dpos, drot)
    t = time.time()
    st = s.time_step(projection_query, f)
    g.copy_pos(s, f)
    print(f"step {f}: {time.time()-t:.2f} s nc={st['nc']} newton={st['newton_iters']} cg={st['cg_iters']} fb={st['fallback']} delta={st['last_delta']:.2e}", flush=True)


def nd_order(N, M, off, leaf=8, width=2):
    """geometric nested dissection of the (N+1) x (M+1) vertex grid, separators `width` lines wide; returns the elimination order"""
    order = []

    def rec(i0, i1, j0, j1):  # inclusive ranges
        ni, nj = i1 - i0 + 1, j1 - j0 + 1
        if ni <= 0 or nj <= 0:
            return
        if ni <= leaf and nj <= leaf:
            for i in range(i0, i1 + 1):
                for j in range(j0, j1 + 1):
                    order.append(off + i * (M + 1) + j)
            return
        if ni >= nj:
            m = (i0 + i1) // 2
            rec(i0, m - 1, j0, j1); rec(m + width, i1, j0, j1)
            for i in range(m, min(m + width, i1 + 1)):
                for j in range(j0, j1 + 1):
                    order.append(off + i * (M + 1) + j)
        else:
            m = (j0 + j1) // 2
            rec(i0, i1, j0, m - 1); rec(i0, i1, m + width, j1)
            for j in range(m, min(m + width, j1 + 1)):
                for i in range(i0, i1 + 1):
                    order.append(off + i * (M + 1) + j)
    rec(0, N, 0, M)
    return order


c = s.cloths[0]
cloth_order = nd_order(c.N, c.M, c.offset)
assert sorted(cloth_order) == list(range(c.offset, c.offset + c.NV))
body_first = [v for e in s.elastics for v in range(e.offset, e.offset + e.n_verts)]
perm_v = np.array(body_first + cloth_order)
perm = (3 * perm_v[:, None] + np.arange(3)[None, :]).ravel()


def test(A, b, name):
    n = A.shape[0]
    Ap = A[perm][:, perm].tocsc()
    bp = b[perm]
    asym = abs(A - A.T).max() / abs(A).max()
    print(f"[{name}] n={n} nnz={A.nnz} asym={asym:.2e}", flush=True)
    for label, kw in (("ND natural, diagonal pivots", dict(permc_spec="NATURAL", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))),
                      ("COLAMD, partial pivoting", dict())):
        t = time.time()
        try:
            lu = spla.splu(Ap, **kw)
        except Exception as e:
            print(f"  {label}: failed {e}")
            continue
        tf = time.time() - t
        x = lu.solve(bp)
        r = bp - Ap @ x
        rel = [np.linalg.norm(r) / np.linalg.norm(bp)]
        for _ in range(4):
            x = x + lu.solve(r)
            r = bp - Ap @ x
            rel.append(np.linalg.norm(r) / np.linalg.norm(bp))
        offd = int((lu.perm_r != np.arange(n)).sum())
        print(f"  {label}: factor {tf:.1f}s fill L+U={lu.L.nnz + lu.U.nnz} ({(lu.L.nnz + lu.U.nnz) / A.nnz:.1f}x) rows pivoted off-diagonal {offd}; rel residual after 0..4 refinements: "
              + " ".join(f"{v:.1e}" for v in rel) + f" |x|={np.linalg.norm(x):.3e}", flush=True)
        # single-precision factors as preconditioner
    try:
        lu32 = spla.splu(Ap.astype(np.float32), permc_spec="NATURAL", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
        x = np.zeros(n); r = bp.copy(); rel = []
        for _ in range(8):
            x = x + lu32.solve(r.astype(np.float32)).astype(np.float64)
            r = bp - Ap @ x
            rel.append(np.linalg.norm(r) / np.linalg.norm(bp))
        print("  fp32 ND factors, refinement residuals: " + " ".join(f"{v:.1e}" for v in rel), flush=True)
    except Exception as e:
        print(f"  fp32: failed {e}")
    # eigenvalue signs
    try:
        w = spla.eigsh(A.astype(np.float64), k=6, sigma=0.0, which="LM", return_eigenvectors=False)
        print("  eigenvalues nearest 0:", np.sort(w), flush=True)
    except Exception as e:
        print("  eigsh failed:", e)


rng = np.random.default_rng(0)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
os.makedirs(out, exist_ok=True)
# forward system at the current state
s.compute_residual_and_Hessian(spd=True)
A = ctx.operator_csr()
b = s.F.to_torch().cpu().numpy().ravel().copy()
print("contacts", len(ctx.constraints()["idx"]))
test(A, b, "forward")
# adjoint system (un-projected) of the last tape step
g.get_loss_balance(s)
g.transfer_grad(T, s, projection_query)
print("adjoint step stats (current solver):", g.last_stats, flush=True)
A2 = ctx.operator_csr()
b2 = rng.standard_normal(A2.shape[0])
fr = s.frozen.t.cpu().numpy().ravel() != 0
b2[fr] = 0
test(A2, b2, "adjoint")
cons = ctx.constraints()
sp.save_npz(os.path.join(out, "cfg4_adjoint_op.npz"), A2.astype(np.float64), compressed=True)
np.savez_compressed(os.path.join(out, "cfg4_adjoint_meta.npz"), idx=cons["idx"], frozen=fr, N=c.N, M=c.M, offset=c.offset,
                    body=np.array([[e.offset, e.n_verts] for e in s.elastics]))
print("saved", flush=True)
