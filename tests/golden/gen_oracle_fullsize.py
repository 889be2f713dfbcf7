"""Generates tests/golden/oracle_cfg{3,4}.npz in the BUILD CONTAINER (no GPU):  python tests/golden/gen_oracle_fullsize.py cfg3|cfg4 [out_dir]

An ORACLE rollout at the full sizes of BASELINE configs[2] / configs[3] -- the CPU restatement of the reference stepping by itself
(BaseScene.time_step, BaseScene.py:1327-1370; Grad.transfer_grad, analytic_grad_single.py:217-257; the linear solves by scipy's
SuperLU as the reference calls spsolve, sparse_solver.py:85-105), from the deterministic initial state of the product scene's HOST
initialisation (built on device "cpu": no kernel runs) + the sub-micron ripple of the parity tests:
    cfg4  Scene_balancing 224 x 224 (100,352 triangles), bench drive (+-1e-4 m on the paired grippers): 2 steps + the reverse step of the last, loss get_loss_balance
    cfg3  Scene_folding 200 x 100 (40,000 triangles), pad -2e-4 m in z per step: 2 steps + the reverse step of the last, loss get_loss_fold
tests/test_gpu_fullsize_oracle.py steps the HIP engine from the same state and compares.  Kept small: a strided sample of the cloth
vertices + every body vertex, SHA-256 digests of the oracle's full arrays (fixture integrity), Newton / contact / line-search counts
per step, gripper_grad, and the wall time of every oracle step on this container's cores (a MEASURED complete CPU step at bench size).
This is test infrastructure: data (inputs and expected outputs), produced by oracle/ -- nothing of the reference travels with it."""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ripple(x, c):
    x[c.offset:c.offset + c.NV, 2] += 2e-6 * np.sin(0.7 * np.arange(c.NV) + 0.3)
    return x


def build(which, device="cpu"):
    """the product scene at t = 0 (host initialisation only) and the drive of step f; shared with the GPU test"""
    shrink = int(os.environ.get("TSL_GOLDEN_SHRINK", "1"))   # (> 1: a dry run of this script on a coarser grid; never written next to the fixtures)
    if which == "cfg4":
        from thinshelllab_amd.task_scene.Scene_balancing import Scene
        s = Scene(cloth_size=0.12, cloth_N=224 // shrink, cloth_M=224 // shrink, device=device)
        s.init_all()

        def drive(f, n_part):
            dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3))
            dpos[:, 2] = 1e-4 * np.where(np.arange(n_part) % 2 == 0, 1.0, -1.0)   # bench.py _drive, rank 0
            return dpos, drot
        steps = int(os.environ.get("TSL_GOLDEN_STEPS", "2"))   # step 1 meets one contact and converges in 7 Newton iterations; step 2 has ~76 contacts and sits at the cap of 50
    else:
        from thinshelllab_amd.task_scene.Scene_folding import Scene
        s = Scene(cloth_size=0.1, cloth_N=200 // shrink, cloth_M=100 // shrink, device=device)
        s.cloths[0].Kb[None] = 400.0
        s.init_all()

        def drive(f, n_part):
            dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3))
            dpos[:, 2] = -2e-4
            return dpos, drot
        steps = 2
    s.mu_cloth_elastic[None] = 5.0
    s.prev_pos.copy_from(s.pos)
    return s, drive, steps


def apply_ripple(s, o=None):
    """the ripple goes on AFTER the oracle was mirrored (check_init compares the un-rippled initial poses of both sides)"""
    x = ripple(s.pos.to_numpy(), s.cloths[0])
    s.pos.from_numpy(x); s.prev_pos.from_numpy(x)
    if o is not None:
        o.pos[:] = x; o.prev_pos[:] = x; o.push_down_all()


def sample_index(s):
    """every 7th cloth vertex (7 is coprime to the row lengths 225 / 201 / 101) + every vertex of the other bodies"""
    c = s.cloths[0]
    cloth = c.offset + np.arange(0, c.NV, 7)
    rest = np.setdiff1d(np.arange(s.tot_NV), np.arange(c.offset, c.offset + c.NV))
    return np.sort(np.concatenate([cloth, rest])).astype(np.int64)


def main():
    which = sys.argv[1]
    out_dir = sys.argv[2] if len(sys.argv) > 2 else HERE
    threads = int(os.environ.get("TSL_ORACLE_THREADS", "4"))
    from oracle import pyoracle as po
    from oracle.mirror import oracle_from_scene
    po.set_threads(threads)
    s, drive, steps = build(which)
    o = oracle_from_scene(po, s, check_init=True)
    apply_ripple(s, o)
    o.set_solver(1e-10); o.set_direct(1)
    T = steps + 1
    n_part = s.gripper.n_part
    NV = s.tot_NV
    o.grad_new(T, n_part)
    o.grad_copy_pos(0)
    sel = sample_index(s)
    stats, secs = [], []
    for f in range(1, steps + 1):
        o.action(*drive(f, n_part))
        o.stats(reset=True); po.direct_seconds[:] = [0.0, 0]
        t0 = time.time()
        o.time_step()
        dt_ = time.time() - t0
        o.grad_copy_pos(f)
        st = o.stats()
        stats.append([o.nc, st["newton"], st["ls"], st["flag"]])
        secs.append([dt_, po.direct_seconds[0], po.direct_seconds[1]])
        print(f"{which} step {f}: nc {o.nc} newton {st['newton']} line-search evaluations {st['ls']} flag {st['flag']}: {dt_:.1f} s, of which SuperLU {po.direct_seconds[0]:.1f} s in {po.direct_seconds[1]} solves", flush=True)
    if which == "cfg3":
        rows = np.array(s.fold_rows()).ravel()
        o.grad_loss("fold", 1.0, -1.0, rows=rows)
        reward = o.reward("folding", 1.0, -1.0, rows=rows)
    else:
        o.grad_loss("balance")
        reward = o.reward("balancing.all")
    pg = o.arr("grad.pos_grad", (T, NV, 3))
    seed_last = pg[T - 1].copy()
    po.direct_seconds[:] = [0.0, 0]
    t0 = time.time()
    o.grad_transfer(T - 1)
    t_adj = time.time() - t0
    print(f"{which} reverse step {T - 1}: {t_adj:.1f} s (SuperLU {po.direct_seconds[0]:.1f} s), oracle solver flag {o.stats()['flag']}", flush=True)
    pb = o.arr("grad.pos_buffer", (T, NV, 3))
    gg = o.arr("grad.gripper_grad", (T, n_part, 6))
    ag = o.arr("grad.angleref_grad")
    tz = o.arr("tmp_z_frozen")
    out = os.path.join(out_dir, f"oracle_{which}.npz")
    np.savez_compressed(
        out, sample_idx=sel, pos_buffer_sample=pb[:, sel].copy(), pos_grad_prev_sample=pg[T - 2, sel].copy(), seed_last_sample=seed_last[sel],
        gripper_grad=gg.copy(), tmp_z_frozen_sample=np.asarray(tz).reshape(-1, 3)[sel].copy() if np.asarray(tz).size == 3 * NV else np.asarray(tz).copy(),
        angleref_grad_prev_absmax=np.array(np.abs(ag.reshape(T, -1)[T - 2]).max()), angleref_grad_prev_sample=ag.reshape(T, -1)[T - 2, ::7].copy(),
        stats=np.array(stats, dtype=np.int64), seconds=np.array(secs), adjoint_seconds=np.array([t_adj, po.direct_seconds[0]]), threads=np.array(threads),
        reward=np.array(reward), tot_NV=np.array(NV), triangles=np.array(s.cloths[0].NF),
        sha_pos_buffer=digest(pb), sha_pos_grad_prev=digest(pg[T - 2]), sha_gripper_grad=digest(gg),
        pos_absmax=np.array(np.abs(pb).max()), pos_grad_prev_absmax=np.array(np.abs(pg[T - 2]).max()))
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
