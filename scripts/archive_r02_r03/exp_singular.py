"""Probe (not part of the product): roll the cfg4 scene of bench.py until a time step reports unconverged solves; assemble the forward
operator at the state that step ended in, solve it with the engine and with scipy's SuperLU (partial pivoting), and look at its
spectrum near zero -- is the operator itself singular there, or does the unpivoted factorisation lose it?
usage: exp_singular.py [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from argparse import Namespace
import bench
from thinshelllab_amd.engine.geometry import projection_query

T = int(sys.argv[1]) if len(sys.argv) > 1 else 50
args = Namespace(workload="cfg4", grid=224, cloth_size=0.12)
s = bench.build_scene(args, 0)
ctx = s._ensure_ctx()
for kv in os.environ.get("TSL_PARAMS", "").split(","):
    if "=" in kv:
        ctx.set_param(kv.split("=")[0], float(kv.split("=")[1]))
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
done = 0
for f in range(1, T + 1):
    d = bench._drive(s.gripper.n_part, 1.0, 0, f, int(os.environ.get('IDLE', '0')))
    s.action(f, *d)
    st = s.time_step(projection_query, f)
    print(f"step {f}: nc {st['nc']} its {st['cg_iters']} unconverged {st['unconverged']} max_res {st['max_rel_residual']:.1e} delta {st['last_delta']:.1e}", flush=True)
    if st["unconverged"] > 3 and done < 3:
        done += 1
        # the operator of the LAST Newton iteration of the step is still assembled in the context
        rng = np.random.default_rng(0)
        bn0 = rng.standard_normal(3 * s.tot_NV)
        bn0[s.frozen.t.cpu().numpy().ravel() != 0] = 0
        b = torch.as_tensor(bn0.reshape(-1, 3), device=s.pos.t.device)
        x, ss = ctx.solve(b)
        print("   engine solve at the end state:", ss, flush=True)
        A = ctx.operator_csr().astype(np.float64)
        bn = b.cpu().numpy().ravel()
        n = A.shape[0]
        print(f"   n {n} nnz {A.nnz} asym {abs(A - A.T).max():.3e} |A|max {abs(A).max():.3e} diag min {A.diagonal().min():.3e}")
        A0 = ctx.matrix_csr().astype(np.float64)      # without the contact blocks
        Ac = (A - A0).tocoo()
        print(f"   |static part|max {abs(A0).max():.3e}, |contact part|max {abs(Ac.data).max() if Ac.nnz else 0:.3e}")
        for name, M in (("static", A0.tocoo()), ("contact", Ac)):
            if M.nnz == 0:
                continue
            o = np.argsort(-abs(M.data))[:6]
            print(f"     largest {name} entries:", [(int(M.row[k]) // 3, int(M.col[k]) // 3, float(f"{M.data[k]:.3e}")) for k in o])
        cons = ctx.constraints()
        print("     bodies:", [(e.offset, e.n_verts) for e in s.elastics], "cloth", s.cloths[0].offset, s.cloths[0].NV)
        t0 = time.time()
        lu = spla.splu(A.tocsc())
        xs = lu.solve(bn)
        r = bn - A @ xs
        print(f"   SuperLU (partial pivoting): {time.time() - t0:.1f} s, rel residual {np.linalg.norm(r) / np.linalg.norm(bn):.3e}, |x| {np.linalg.norm(xs):.3e}, |b| {np.linalg.norm(bn):.3e}", flush=True)
        xe = x.cpu().numpy().ravel()
        re = bn - A @ xe
        print(f"   engine x: rel residual {np.linalg.norm(re) / np.linalg.norm(bn):.3e}, |x| {np.linalg.norm(xe):.3e}, |x - x_superlu| / |x| {np.linalg.norm(xe - xs) / np.linalg.norm(xs):.3e}")
        try:
            w, V = spla.eigsh(A, k=8, sigma=0.0, which="LM")
            o = np.argsort(abs(w))
            print("   eigenvalues nearest 0:", w[o])
            for k in o[:3]:
                v = V[:, k].reshape(-1, 3)
                nrm = np.linalg.norm(v, axis=1)
                top = np.argsort(-nrm)[:6]
                print(f"     eigenvalue {w[k]:.3e}: largest vertices {top.tolist()} weights {np.round(nrm[top], 3).tolist()}")
            wl = spla.eigsh(A, k=2, which="LA", return_eigenvectors=False)
            print("   largest eigenvalues:", wl)
        except Exception as e:
            print("   eigsh failed:", e)
        if done == 3:
            break
