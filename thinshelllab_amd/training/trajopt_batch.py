"""Batched trajectory optimisation (BASELINE.json configs[4]): independent folding / lifting / balancing
rollouts, one scene per GPU (scene s on rank s % world), every scene with its own trajectory; per optimisation iteration the ranks
exchange nothing but ``(reward, gripper_grad)`` -- one RCCL ``all_gather`` of a few KB (``Batch.gather_results``) -- so that rank 0
can log / keep the best trajectory of the whole batch.  The per-scene loop is the reference's (forward rollout -> reward -> loss
seed -> reverse sweep -> Adam, /root/reference/code/training/trajopt_folding.py:74-142); nothing of a scene crosses GPUs.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m thinshelllab_amd.training.trajopt_batch \
        --env folding --scenes 8 --iter 400 --tot_step 50 --lr 3e-5

Initial trajectories: scene s draws ``numpy.random.default_rng(1000 + s)`` N(0, --init_sigma) offsets.
"""
import os
import time
from argparse import ArgumentParser

import numpy as np
import torch


def run_batch(batch, n_scenes, iters, make_problem, out_dir=None, log=print, rollout_many=None):
    """Generic driver.  ``make_problem(scene_id)`` returns an object with
         traj: torch tensor (T, n_part, 6), updated in place by ``step(grad)``;
         rollout() -> (reward: float, gripper_grad: tensor (T, n_part, 6))   -- forward rollout + reverse sweep of one scene;
         step(gripper_grad)                                                  -- optimiser update (+ action limits).
    Scene s runs on rank s % world (``batch.scene_ids``); after every iteration all ranks hold the rewards of the whole batch.
    ``rollout_many(list of problems) -> list of (reward, gripper_grad)``, if given, rolls out ALL scenes of this rank together (a scene group:
    lock-step forward steps with merged factorisations, ``rollout_scene_group``) instead of one after the other.
    Returns {scene: [reward per iteration]} (identical on every rank) and the best (reward, scene, iteration)."""
    local = batch.scene_ids(n_scenes)
    slots = (n_scenes + batch.world - 1) // batch.world
    problems = {s: make_problem(s) for s in local}
    history = {s: [] for s in range(n_scenes)}
    best = (-float("inf"), -1, -1)
    shape = None
    for p in problems.values():
        shape = tuple(p.traj.shape)
    if batch.dist is not None:   # ranks without a scene still take part in the gathers
        t = torch.tensor(list(shape or (0, 0, 0)), dtype=torch.float64, device=batch.device)
        batch.dist.all_reduce(t, op=batch.dist.ReduceOp.MAX)
        shape = tuple(int(v) for v in t.tolist())
    for it in range(iters):
        t0 = time.time()
        together = None
        if rollout_many is not None and len(local) > 1:
            try:
                together = dict(zip(local, rollout_many([problems[s] for s in local])))
            except Exception as e:  # noqa: BLE001 -- as below: the other ranks wait in the gathers
                log(f"rank {batch.rank} iter {it}: grouped rollout failed ({e!r}); its scenes are reported as NaN, no update")
                together = {s: (float("nan"), torch.zeros(shape, dtype=torch.float64)) for s in local}
        for slot in range(slots):
            s = slot * batch.world + batch.rank
            ok = False
            if s in problems and together is not None:
                reward, g = together[s]
                ok = bool(np.isfinite(reward))
            elif s in problems:
                # a rollout that fails (an unconverged adjoint solve, a constraint overflow) must not take the job down: the other
                # ranks would wait in the gather below.  The scene reports NaN for this iteration and skips its update.
                try:
                    reward, g = problems[s].rollout()
                    ok = bool(np.isfinite(reward))
                except Exception as e:  # noqa: BLE001
                    log(f"rank {batch.rank} scene {s} iter {it}: rollout failed ({e!r}); reported as NaN, no update")
                    reward, g = float("nan"), torch.zeros(shape, dtype=torch.float64)
            else:
                reward, g = float("nan"), torch.zeros(shape, dtype=torch.float64)
            rewards, grads = batch.gather_results(reward, g)          # the only exchange of an iteration
            for r in range(batch.world):
                sr = slot * batch.world + r
                if sr < n_scenes:
                    history[sr].append(rewards[r])
                    if rewards[r] > best[0]:
                        best = (rewards[r], sr, it)
                        if out_dir is not None and batch.rank == 0:
                            np.save(os.path.join(out_dir, "best_gripper_grad.npy"), grads[r].numpy())
            if s in problems and ok:
                problems[s].step(g)
        if batch.rank == 0:
            log(f"iter {it}: rewards {[round(history[s][-1], 6) for s in range(n_scenes)]} best {best} ({time.time() - t0:.2f} s)")
            if out_dir is not None:
                np.save(os.path.join(out_dir, "plot_data.npy"), np.array([history[s] for s in range(n_scenes)]))
    if out_dir is not None:
        for s, p in problems.items():
            np.save(os.path.join(out_dir, f"traj_scene{s}.npy"), p.traj.cpu().numpy())
    return history, best


class SceneProblem:
    """One reference-style trajectory optimisation (scene + Grad + agent + Adam) behind the rollout / step surface of run_batch."""

    def __init__(self, env, scene_id, tot_step, lr, device, init_sigma=1e-4, cloth_N=None):
        from ..agent.traj_opt_single import agent_trajopt
        from ..engine.analytic_grad_single import Grad
        from ..engine.geometry import projection_query
        from ..optimizer.optim import Adam_single
        self.env, self.T, self.contact = env, tot_step, projection_query
        kw = {}
        if env == "folding":
            from ..task_scene.Scene_folding import Scene
            if cloth_N:
                kw = dict(cloth_N=cloth_N, cloth_M=cloth_N // 2)
            self.sys = Scene(cloth_size=0.1, device=device, **kw)
            self.sys.cloths[0].Kb[None] = 400.0
            self.mu, self.limit_grad, self.fix = 5.0, False, True
        elif env == "lifting":
            from ..task_scene.Scene_lifting import Scene
            self.sys = Scene(cloth_size=0.06, device=device)
            self.mu, self.limit_grad, self.fix = 1.0, True, False
        elif env == "balancing":
            from ..task_scene.Scene_balancing import Scene
            if cloth_N:
                kw = dict(cloth_N=cloth_N, cloth_M=cloth_N)
            self.sys = Scene(cloth_size=0.06 if not cloth_N else 0.12, device=device, **kw)
            self.mu, self.limit_grad, self.fix = 5.0, True, False
        else:
            raise ValueError(f"unknown env {env}")
        n_part = self.sys.gripper.n_part
        self.grad = Grad(self.sys, tot_step, n_part)
        self.adam = Adam_single((tot_step, n_part, 6), lr, 0.9, 0.9999, 1e-8)
        self.agent = agent_trajopt(tot_step, n_part, max_moving_dist=0.001)
        self.sys.init_all()
        self.grad.init_mass(self.sys)
        rng = np.random.default_rng(1000 + scene_id)
        tr = np.cumsum(rng.normal(0.0, init_sigma, (tot_step, n_part, 6)), axis=0)
        tr[0] = 0.0
        tr[:, :, 3:] = 0.0
        self.agent.traj.from_numpy(tr)
        if self.fix:
            self.agent.fix_action(0.015)
        self.traj = self.agent.traj.t

    def begin(self):
        s, g = self.sys, self.grad
        s.reset()
        s.mu_cloth_elastic[None] = self.mu
        g.copy_pos(s, 0)

    def pre_step(self, frame):
        self.agent.get_action(frame)
        self.sys.action(frame, self.agent.delta_pos, self.agent.delta_rot)

    def post_step(self, frame):
        self.grad.copy_pos(self.sys, frame)

    def seed(self):
        """reward and loss seed at the end of the forward sweep (trajopt_folding.py:100-118)"""
        s, g = self.sys, self.grad
        if self.env == "folding":
            self.reward = s.compute_reward(1.0, -1.0)
            g.get_loss_fold(s, 1.0, -1.0, rows=s.fold_rows())
        elif self.env == "lifting":
            self.reward = s.compute_reward()
            g.get_loss_lift(s)
        else:
            self.reward = s.compute_reward_all(g)
            g.get_loss_balance(s)
        g.allow_unconverged = True   # counted instead of raised in the middle of the sweep (run_batch keeps the ranks in lock-step)
        g.unconverged = 0

    def collect(self):
        """(reward, gripper gradient) after the reverse sweep (trajopt_folding.py:119-135)"""
        g, T = self.grad, self.T
        if g.unconverged:
            print(f"scene rollout: {g.unconverged} adjoint solves of {T - 1} did not converge; the gradient of this iteration is discarded", flush=True)
            return float("nan"), torch.zeros_like(g.gripper_grad.t)
        if self.limit_grad:
            g.apply_action_limit_grad(self.agent, 0.015)
        return float(self.reward), g.gripper_grad.t.clone()

    def finish(self):
        """reward, loss seed, reverse sweep (trajopt_folding.py:100-135)"""
        self.seed()
        for step in range(self.T - 1, 0, -1):
            self.grad.transfer_grad(step, self.sys, self.contact)
        return self.collect()

    def rollout(self):
        self.begin()
        for frame in range(1, self.T):
            self.pre_step(frame)
            self.sys.time_step(self.contact, frame)
            self.post_step(frame)
        return self.finish()

    def step(self, gripper_grad):
        self.adam.step(self.agent.traj, self.grad.gripper_grad)
        if self.fix:
            self.agent.fix_action(0.015)
        self.grad.reset()


_GROUPS = {}


def rollout_scene_group(problems):
    """the scenes of one rank rolled out together: every frame's time steps as ONE SceneGroup.time_step (thinshelllab_amd/scene_group.py: lock step,
    merged factorisations, each scene bit-identical to its own time_step), the reverse sweep as SceneGroup.transfer_grad per step, reward / seed per scene"""
    from ..scene_group import SceneGroup
    key = tuple(id(p) for p in problems)
    if key not in _GROUPS:
        for p in problems:
            p.sys._ensure_ctx().set_param("direct", 1)     # a group shares the sparse direct solves of its members
        _GROUPS[key] = SceneGroup([p.sys for p in problems])
    group = _GROUPS[key]
    for p in problems:
        p.begin()
    for frame in range(1, problems[0].T):
        for p in problems:
            p.pre_step(frame)
        group.time_step(problems[0].contact, frame)
        for p in problems:
            p.post_step(frame)
    for p in problems:
        p.seed()
    for step in range(problems[0].T - 1, 0, -1):   # the reverse sweep in lock step as well: one merged factorisation per adjoint step
        group.transfer_grad(step, [p.grad for p in problems], problems[0].contact)
    return [p.collect() for p in problems]


def main(argv=None):
    ap = ArgumentParser()
    ap.add_argument("--env", choices=["folding", "lifting", "balancing"], default="folding")
    ap.add_argument("--scenes", type=int, default=0, help="number of independent scenes (default: one per rank)")
    ap.add_argument("--iter", type=int, default=10)
    ap.add_argument("--tot_step", type=int, default=5)
    ap.add_argument("--lr", type=float, default=3e-5)
    ap.add_argument("--init_sigma", type=float, default=1e-4)
    ap.add_argument("--cloth_N", type=int, default=0, help="refined cloth grid (0: the task's native grid)")
    ap.add_argument("--out", type=str, default=None)
    ap.add_argument("--group", type=int, default=0, help="1: the scenes of a rank step in lock step as one scene group (more scenes than GPUs; needs --cloth_N large enough "
                                                         "for the sparse direct solve or forces it)")
    args = ap.parse_args(argv)
    from ..batch import Batch
    batch = Batch()
    n_scenes = args.scenes or batch.world
    dev = f"cuda:{batch.local_rank}"
    out_dir = args.out or os.environ.get("TSL_OUT")
    if out_dir and batch.rank == 0:
        os.makedirs(out_dir, exist_ok=True)
    batch.barrier()
    history, best = run_batch(batch, n_scenes, args.iter,
                              lambda s: SceneProblem(args.env, s, args.tot_step, args.lr, dev, args.init_sigma, args.cloth_N or None), out_dir=out_dir,
                              rollout_many=rollout_scene_group if args.group else None)
    for g in _GROUPS.values():
        g.close()
    _GROUPS.clear()
    batch.close()
    return history, best


if __name__ == "__main__":
    main()
