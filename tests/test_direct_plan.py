"""Host logic of the sparse direct preconditioner (thinshelllab_amd/csrc/direct_sym.hpp, direct_plan.hpp): nested-dissection ordering,
elimination tree with contact cliques, front layout and index maps.  tests/native/ds_ref.cpp executes the plan with plain CPU loops
(test infrastructure, never part of libtsl_hip.so); the solution must agree with scipy's sparse LU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def dsref():
    src = os.path.join(HERE, "native", "ds_ref.cpp")
    lib = os.path.join(HERE, "native", "libdsref.so")
    deps = [src] + [os.path.join(HERE, "..", "thinshelllab_amd", "csrc", f) for f in ("direct_sym.hpp", "direct_plan.hpp")]
    if not os.path.exists(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", lib])
    L = C.CDLL(lib)
    L.dsref_solve.restype = C.c_int
    return L


def cloth_cliques(N, M, off=0):
    """triangles of the alternating-diagonal grid (model_fold_offset.py:936-941) and the 4-vertex hinge cliques"""
    W = M + 1
    tris = []
    for i in range(N):
        for j in range(M):
            a = off + i * W + j; b = a + 1; c = a + W + 1; d = a + W
            tris += [(c, b, a), (a, d, c)] if (i + j) % 2 == 0 else [(b, a, d), (d, c, b)]
    edges = {}
    for t, f in enumerate(tris):
        for k in range(3):
            e = tuple(sorted((f[k], f[(k + 1) % 3])))
            edges.setdefault(e, []).append(t)
    cl = [list(f) for f in tris]
    for e, ts in edges.items():
        if len(ts) == 2:
            cl.append(sorted(set(tris[ts[0]]) | set(tris[ts[1]])))
    return tris, cl


def build_system(N, M, n_body, n_cons, rng, indefinite):
    ncloth = (N + 1) * (M + 1)
    NV = ncloth + n_body
    tris, cliques = cloth_cliques(N, M)
    if n_body:
        cliques.append(list(range(ncloth, NV)))
    rows = [set([v]) for v in range(NV)]
    for c in cliques:
        for a in c:
            rows[a].update(c)
    rows = [sorted(r) for r in rows]
    row_ptr = np.zeros(NV + 1, np.int32)
    row_ptr[1:] = np.cumsum([len(r) for r in rows])
    col = np.concatenate(rows).astype(np.int32)
    # element-wise assembled values: every clique adds a random symmetric PSD block (plus a small non-symmetric part)
    A = sp.lil_matrix((3 * NV, 3 * NV))
    for c in cliques:
        d = np.concatenate([3 * v + np.arange(3) for v in c])
        X = rng.standard_normal((len(d), len(d)))
        Hc = X @ X.T / len(d) + 1e-3 * rng.standard_normal((len(d), len(d)))
        if indefinite:
            Hc -= 0.35 * np.eye(len(d))
        A[np.ix_(d, d)] = A[np.ix_(d, d)].toarray() + Hc
    A = A.tocsr() + 0.05 * sp.identity(3 * NV)
    bsr = sp.bsr_matrix(A, blocksize=(3, 3))
    # values in the order of `rows`
    vals = np.zeros((len(col), 3, 3))
    dense_lookup = {}
    bsr.sort_indices()
    for r in range(NV):
        for q in range(bsr.indptr[r], bsr.indptr[r + 1]):
            dense_lookup[(r, bsr.indices[q])] = bsr.data[q]
    for r in range(NV):
        for k, cc in enumerate(rows[r]):
            vals[row_ptr[r] + k] = dense_lookup.get((r, cc), np.zeros((3, 3)))
    cons = np.zeros((n_cons, 4), np.int32)
    conH = np.zeros((n_cons, 12, 12))
    for e in range(n_cons):
        t = tris[rng.integers(len(tris))]
        if n_body and e % 2 == 0:
            cons[e] = [t[0], t[1], t[2], ncloth + rng.integers(n_body)]          # body vertex against a cloth triangle
        elif n_body:
            bv = ncloth + rng.choice(n_body, 3, replace=False)
            cons[e] = [bv[0], bv[1], bv[2], rng.integers(ncloth)]                # cloth vertex against a body triangle
        else:
            cons[e] = [t[0], t[1], t[2], rng.integers(ncloth)]                   # cloth against cloth
            if cons[e][3] in t:
                cons[e][3] = (max(t) + 7) % ncloth
        X = rng.standard_normal((12, 3))
        conH[e] = 5.0 * X @ X.T
    Afull = A.tolil()
    for e in range(n_cons):
        d = np.concatenate([3 * v + np.arange(3) for v in cons[e]])
        Afull[np.ix_(d, d)] = Afull[np.ix_(d, d)].toarray() + conH[e]
    return NV, ncloth, row_ptr, col, vals, cons, conH, Afull.tocsc()


@pytest.mark.parametrize("N,M,n_body,n_cons,leaf,indefinite", [(12, 9, 0, 0, 8, False), (24, 17, 9, 14, 12, False), (33, 40, 20, 30, 16, True), (20, 20, 0, 10, 32, True)])
def test_multifrontal_plan_matches_sparse_lu(dsref, N, M, n_body, n_cons, leaf, indefinite):
    rng = np.random.default_rng(N * 100 + M)
    NV, ncloth, row_ptr, col, vals, cons, conH, A = build_system(N, M, n_body, n_cons, rng, indefinite)
    b = rng.standard_normal(3 * NV)
    x = np.zeros(3 * NV)
    grids = np.array([0, N, M], np.int32)
    blocks = np.array([ncloth, n_body], np.int32)
    stats = np.zeros(8)
    rc = dsref.dsref_solve(NV, row_ptr.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p), 1, grids.ctypes.data_as(C.c_void_p),
                           1 if n_body else 0, blocks.ctypes.data_as(C.c_void_p), n_cons, cons.ctypes.data_as(C.c_void_p), conH.ctypes.data_as(C.c_void_p), leaf,
                           b.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), stats.ctypes.data_as(C.c_void_p))
    assert rc == 0
    xr = spla.splu(A).solve(b)
    assert stats[1] >= 2  # a real tree
    assert np.linalg.norm(x - xr) <= 1e-8 * np.linalg.norm(xr), (np.linalg.norm(x - xr) / np.linalg.norm(xr), stats)
    assert np.linalg.norm(A @ x - b) <= 1e-9 * np.linalg.norm(b)
    # the gather form of the extend-add: every child sits on a LOWER level than its parent (its Schur complement is stored before the
    # parent reads it -- the reference run starts from a NaN Schur arena, a value read before it was stored would poison x), and the
    # parents did gather something
    assert stats[5] == 0 and stats[6] > 0, stats


@pytest.mark.parametrize("N,M,n_body,n_cons,leaf", [(24, 17, 9, 14, 12), (33, 40, 20, 30, 16), (64, 48, 12, 40, 16), (20, 20, 0, 10, 32)])
def test_lookahead_tables_follow_their_definitions(dsref, N, M, n_body, n_cons, leaf):
    """The host tables of the look-ahead (direct_plan.hpp, round 6): DsFrontDesc.lead -- the boundary dofs of a front that are OWN dofs of its parent are exactly its first
    `lead` boundary dofs (so the leading lead x lead block of S is all the parent's pivot block receives) --, the level-ordered block list and the contact groups with
    the entries inside F11 first (blk_lmid, cgr_lmid), and la_from: every level from there on is one batch."""
    rng = np.random.default_rng(7 * N + M)
    NV, ncloth, row_ptr, col, vals, cons, conH, A = build_system(N, M, n_body, n_cons, rng, False)
    grids = np.array([0, N, M], np.int32)
    blocks = np.array([ncloth, n_body], np.int32)
    out = np.zeros(9)
    dsref.dsref_check_lookahead.restype = C.c_int
    rc = dsref.dsref_check_lookahead(NV, row_ptr.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p), 1, grids.ctypes.data_as(C.c_void_p),
                                     1 if n_body else 0, blocks.ctypes.data_as(C.c_void_p), n_cons, cons.ctypes.data_as(C.c_void_p), leaf, out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    with_parent, lead_viol, nblk, blk_viol, ngrp, grp_viol, la_from, levels, multi = out
    assert with_parent > 0 and lead_viol == 0, out
    assert nblk == len(col) and blk_viol == 0, out          # every block of the pattern is listed once, on the right side of its level's mid pointer
    assert (ngrp > 0) == (n_cons > 0) and grp_viol == 0, out
    assert multi == 0 and (la_from < 0 or 1 <= la_from <= levels - 2), out
