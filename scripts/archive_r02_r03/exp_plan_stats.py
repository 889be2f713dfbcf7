"""Probe (not part of the product): plan statistics of the direct solver for a cfg4 constraint set saved by exp_direct_proto.py"""
import ctypes as C, os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from test_direct_plan import cloth_cliques
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(root, "tests", "native", "ds_ref.cpp"); lib = os.path.join(root, "tests", "native", "libdsref.so")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", lib])
L = C.CDLL(lib)
meta = np.load(os.path.join(root, "gpurun_out", "cfg4_adjoint_meta.npz"))
N, M = int(meta["N"]), int(meta["M"]); body = meta["body"]; cons = meta["idx"].astype(np.int32)
NV = (N + 1) * (M + 1) + int(body[:, 1].sum())
_, cliques = cloth_cliques(N, M)
rows = [set([v]) for v in range(NV)]
for c in cliques:
    for a in c:
        rows[a].update(c)
for off, n in body:
    for v in range(off, off + n):
        rows[v] = set(range(off, off + n))
rows = [sorted(r) for r in rows]
rp = np.zeros(NV + 1, np.int32); rp[1:] = np.cumsum([len(r) for r in rows]); col = np.concatenate(rows).astype(np.int32)
grids = np.array([0, N, M], np.int32); blocks = body.astype(np.int32).ravel()
leaf = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for nc in (0, len(cons)):
    out = np.zeros(8)
    rc = L.dsref_plan_stats(NV, rp.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p), 1, grids.ctypes.data_as(C.c_void_p), len(body), blocks.ctypes.data_as(C.c_void_p),
                            nc, cons.ctypes.data_as(C.c_void_p), leaf, 1 if nc else 0, out.ctypes.data_as(C.c_void_p))
    print(f"nc {nc}: rc {rc} supernodes {out[0]:.0f} levels {out[1]:.0f} batches {out[2]:.0f} block steps {out[3]:.0f} flops {out[4]/1e9:.1f} G arena {out[5]/1e9:.2f} GB solve {out[6]/1e9:.2f} GB")
