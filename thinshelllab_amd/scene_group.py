"""Several scenes of one GPU advanced in lock step (``tsl_group_*`` of the C ABI).

The reference's trajectory-optimisation drivers roll out independent copies of one scene (``training/trajopt_*.py``: one rollout per
candidate trajectory); ``SceneGroup`` is the per-GPU half of that batch: the members stay ordinary scene objects -- grippers, ``Grad`` tapes,
``action``, ``transfer_grad`` work on them as before -- and ``SceneGroup.time_step`` replaces the members' ``time_step`` calls of one frame.
Per member the engine runs exactly the kernels and decisions of ``BaseScene.time_step`` (BaseScene.py:1327-1370); the sparse factorisation and
the first application of the factors of ALL members are one set of launches (csrc/direct_group.hpp).  A member's state after the step is
bit-identical to what its own ``time_step`` gives.
"""
import ctypes as C

from . import _lib
from ._lib import SolveStats, StepStats, check
from .context import _ptr


class SceneGroup:
    def __init__(self, scenes):
        self.scenes = list(scenes)
        assert len(self.scenes) >= 1
        self.L = _lib.load()
        ctxs = [s._ensure_ctx() for s in self.scenes]
        for c in ctxs:
            assert getattr(c, "_group", None) is None, "scene is a member of another group"
        n = len(ctxs)
        arr = (C.c_void_p * n)(*[c.h for c in ctxs])
        self.h = C.c_void_p()
        check(self.L.tsl_group_create(arr, n, C.byref(self.h)), "tsl_group_create")
        self._ctxs = ctxs
        for c in ctxs:
            c._group = self

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.tsl_group_destroy(self.h)
            self.h = None
            for c in self._ctxs:
                c._group = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return len(self.scenes)

    def time_step(self, f_contact, frame_idx, force_stick=True):
        """one implicit step of every member (``BaseScene.time_step`` for each of them); returns the members' statistics"""
        n = len(self.scenes)
        cols = [[], [], [], []]
        for s in self.scenes:
            ctx = s._ensure_ctx()
            ctx.set_param("contact", 0.0 if f_contact is None else 1.0)
            ctx.refresh_stream()
            for k, t in enumerate(s._state()):
                cols[k].append(_ptr(t))
        arrs = [(C.c_void_p * n)(*col) for col in cols]
        st = (StepStats * n)()
        check(self.L.tsl_group_step(self.h, *arrs, st), "tsl_group_step")
        out = []
        for s, r in zip(self.scenes, st):
            d = r.as_dict()
            s.last_stats = d
            s.nc[None] = d["nc"]
            s.E[None] = d["energy"]
            out.append(d)
        return out

    def transfer_grad(self, step, grads, f_contact):
        """``Grad.transfer_grad(step, scene, f_contact)`` (analytic_grad_single.py:217-257) of every member for the same reverse step: the members'
        adjoint systems go through one merged factorisation; ``grads[i]`` is the tape of ``scenes[i]``."""
        n = len(self.scenes)
        assert len(grads) == n
        T = grads[0].tot_timestep
        cols = [[], [], [], [], []]
        damp = (C.c_double * n)()
        for i, (s, g) in enumerate(zip(self.scenes, grads)):
            assert g.tot_timestep == T, "the members' tapes must have one length"
            ctx = s._ensure_ctx()
            ctx.set_param("contact", 0.0 if f_contact is None else 1.0)
            ctx.refresh_stream()
            for k, t in enumerate((g.pos_buffer.t, g.pos_grad.t, g.ref_angle_buffer.t, g.angleref_grad.t, s.tmp_z_frozen.t)):
                cols[k].append(_ptr(t))
            damp[i] = float(g.damping)
        arrs = [(C.c_void_p * n)(*col) for col in cols]
        st = (SolveStats * n)()
        check(self.L.tsl_group_adjoint_step(self.h, int(step), int(T), *arrs, damp, st), "tsl_group_adjoint_step")
        for s, g, r in zip(self.scenes, grads, st):
            g.last_stats = r.as_dict()
            g.check_solve(step)
            s.copy_pos_and_refangle(g, step)
            if g.n_part > 0 and hasattr(s, "gripper"):
                s.gripper.set(g.gripper_pos_buffer, g.gripper_rot_buffer, step)
                if step > 0:
                    g.get_gripper_grad(step, s)

    def info(self):
        v = (C.c_double * 9)()
        check(self.L.tsl_group_info(self.h, v), "tsl_group_info")
        keys = ("plan_merges", "arena_relayouts", "merge_seconds", "arena_bytes", "merged_factorizations", "merged_applications", "member_solves_on_own_path",
                "merged_flow_launches", "merged_flow_aborts")
        return dict(zip(keys, [float(x) for x in v]))
