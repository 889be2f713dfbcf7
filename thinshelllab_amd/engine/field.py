"""Taichi-field look-alikes backed by torch tensors (HBM resident when a GPU is present).

The reference's task scenes / scripts touch engine state through ``ti.field`` objects
(``x[None] = v``, ``x[i]``, ``.fill``, ``.from_numpy``, ``.to_numpy``, ``.to_torch``, ``.copy_from``;
SURVEY.md section 8b).  ``Field`` gives the same surface over a tensor -- possibly a *view* into a larger
tensor, which is how per-body arrays (``cloth.pos``, ``elastic.F_x``) alias the single global node array
instead of being copied back and forth like ``pushup_property`` / ``pushdown_property`` do
(/root/reference/code/engine/BaseScene.py:317-330).
"""
import numpy as np
import torch


class Field:
    def __init__(self, tensor, on_write=None):
        self.t = tensor
        self._on_write = on_write

    # -- taichi-like API
    @property
    def shape(self):
        return tuple(self.t.shape)

    def fill(self, v):
        self.t.fill_(v)
        self._written()

    def from_numpy(self, a):
        self.t.copy_(torch.as_tensor(np.ascontiguousarray(a), dtype=self.t.dtype).reshape(self.t.shape))
        self._written()

    def from_torch(self, a):
        self.t.copy_(a.to(self.t.dtype).reshape(self.t.shape))
        self._written()

    def to_numpy(self, dtype=None):
        a = self.t.detach().cpu().numpy().copy()
        return a if dtype is None else a.astype(dtype)

    def to_torch(self, device=None):
        return self.t.detach().clone() if device is None else self.t.detach().to(device).clone()

    def copy_from(self, other):
        self.t.copy_(other.t if isinstance(other, Field) else other)
        self._written()

    def __getitem__(self, idx):
        if idx is None:
            v = self.t
            return v.item() if v.numel() == 1 else v.detach().cpu().numpy().copy()
        v = self.t[idx]
        if v.numel() == 1:
            return v.item()
        return v.detach().cpu().numpy().copy()

    def __setitem__(self, idx, value):
        if idx is None:
            if torch.is_tensor(value):
                self.t.copy_(value)
            else:
                self.t.copy_(torch.as_tensor(np.asarray(value), dtype=self.t.dtype).reshape(self.t.shape))
        else:
            self.t[idx] = torch.as_tensor(np.asarray(value), dtype=self.t.dtype, device=self.t.device)
        self._written()

    def _written(self):
        if self._on_write is not None:
            self._on_write(self)


class ScalarField(Field):
    """0-d field (``Kb[None] = 400.0``) kept on the host; writes are forwarded to the engine context."""

    def __init__(self, value, on_write=None, dtype=torch.float64):
        super().__init__(torch.tensor(value, dtype=dtype), on_write)

    @property
    def value(self):
        return self.t.item()
