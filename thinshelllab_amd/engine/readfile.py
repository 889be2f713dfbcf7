"""TetGen mesh readers (.node / .ele / .face) and a cloth mesh writer.

Counterpart of /root/reference/code/engine/readfile.py:1-51 (same return convention: ``(count, rows)``)
and :117-128 (``save_cloth_mesh``; written as ASCII PLY here, open3d is not required).
Default paths point at the package's ``data/`` copy of the reference mesh files.
"""
import os

import numpy as np

DATA_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")


def _resolve(filename, default):
    if filename is None:
        return os.path.join(DATA_DIR, default)
    if os.path.exists(filename):
        return filename
    cand = os.path.join(DATA_DIR, os.path.basename(filename))  # "../data/ball.node" style paths of the reference
    return cand if os.path.exists(cand) else filename


def _rows(path, conv, lo, hi):
    with open(path, encoding="utf-8") as f:
        n = int(f.readline().split()[0])
        out = []
        for _ in range(n):
            tok = f.readline().split()
            out.append([conv(t) for t in tok[lo:hi]])
    return n, out


def read_node(filename=None):
    return _rows(_resolve(filename, "tactile.node"), float, 1, 4)


def read_smesh(filename=None):
    return _rows(_resolve(filename, "tactile.face"), int, 1, 4)


def read_ele(filename=None):
    return _rows(_resolve(filename, "tactile.ele"), int, 1, 5)


def save_cloth_mesh(cloth, path):
    v = cloth.pos.to_numpy(dtype="float64")
    f = cloth.f2v.to_numpy().astype(np.int64)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as fh:
        fh.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty double x\nproperty double y\nproperty double z\n" % len(v))
        fh.write("element face %d\nproperty list uchar int vertex_indices\nend_header\n" % len(f))
        for p in v:
            fh.write("%.17g %.17g %.17g\n" % tuple(p))
        for t in f:
            fh.write("3 %d %d %d\n" % tuple(t))
