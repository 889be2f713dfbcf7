"""Gradient-free baseline on the forward path only: counterpart of /root/reference/code/training/run_cmaes_all.py (same flags,
same candidate -> trajectory decoding :98-114, same reward shaping :116-163, outputs plot_Data.npy and traj_<iter>.npy; the
ti.ui window / GIF part of the reference is rendering and is left out).  `cma` is used when importable, otherwise the in-tree
restatement thinshelllab_amd/optimizer/cmaes.py.  e.g.
python -m thinshelllab_amd.training.run_cmaes_all --env folding --tot_step 50 --abs_step 10 --pop_size 8 --iter 10 --Kb 400 --mu 5"""
import importlib
import os
from argparse import ArgumentParser

import numpy as np


def decode(agent, x, args, gripper_cnt, sub_steps, scaling, scaling_angle):
    """candidate vector (values around 5) -> piecewise-linear trajectory (run_cmaes_all.py:98-114)"""
    t = agent.traj.t
    t.zero_()
    for ii in range(args.abs_step):
        for jj in range(sub_steps):
            if ii == 0 and jj == 0:
                continue
            i = ii * sub_steps + jj
            if i < 5 and args.env == "interact":
                continue
            for j in range(gripper_cnt):
                for k in range(3):
                    t[i, j, k] = t[i - 1, j, k] + (x[ii * 6 * gripper_cnt + j * 6 + k] - 5) / sub_steps / scaling
                    t[i, j, k + 3] = t[i - 1, j, k + 3] + (x[ii * 6 * gripper_cnt + j * 6 + k + 3] - 5) / sub_steps / scaling_angle
    agent.fix_action(0.015)


def main(argv=None):
    parser = ArgumentParser()
    parser.add_argument('--pop_size', type=int, default=8)
    parser.add_argument('--iter', type=int, default=10)
    parser.add_argument('--tot_step', type=int, default=60)
    parser.add_argument('--abs_step', type=int, default=60)
    parser.add_argument('--sigma', type=float, default=1.0)
    parser.add_argument('--trial', type=str, default="0")
    parser.add_argument('--env', type=str, default="folding")
    parser.add_argument('--Kb', type=float, default=100.0)
    parser.add_argument('--mu', type=float, default=1.0)
    parser.add_argument('--reward_name', type=str, default=None)
    parser.add_argument('--load_dir', type=str, default=None)
    parser.add_argument('--max_dist', type=float, default=0.002)
    parser.add_argument('--curve7', type=float, default=1.0)
    parser.add_argument('--curve8', type=float, default=1.0)
    parser.add_argument('--dense', type=float, default=10000.0)
    parser.add_argument('--target_dir', type=str, default=None)
    parser.add_argument('--seed', type=int, default=None)
    args = parser.parse_args(argv)

    from ..agent.traj_opt_single import agent_trajopt
    from ..engine.analytic_grad_single import Grad
    from ..engine.geometry import projection_query
    try:
        import cma
        Strategy = cma.CMAEvolutionStrategy
    except ImportError:
        from ..optimizer.cmaes import CMAEvolutionStrategy as Strategy
    Scene = importlib.import_module(f"thinshelllab_amd.task_scene.Scene_{args.env}")

    tot_timestep = args.tot_step
    cloth_size = 0.1 if args.env in ("folding_2", "forming") else 0.06
    dev = f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}"   # one scene per GPU under torch.distributed.run
    if args.env == "interact":
        sys = Scene.Scene(cloth_size=cloth_size, soft=args.Kb < 2, dense=args.dense, device=dev)
    else:
        sys = Scene.Scene(cloth_size=cloth_size, device=dev)
    sys.cloths[0].Kb[None] = args.Kb
    sys.init_all()
    sys.mu_cloth_elastic[None] = args.mu

    save_path = os.path.join(os.environ.get("TSL_OUT", os.path.join(os.getcwd(), "data")), f"cmaes_traj_{args.env}_{args.trial}")
    os.makedirs(save_path, exist_ok=True)

    gripper_cnt = sys.elastic_cnt - 1
    if sys.enable_gripper:
        gripper_cnt = int((sys.effector_cnt - 1) // 2)
    analy_grad = Grad(sys, tot_timestep, gripper_cnt)
    x0 = (args.abs_step * 6 * gripper_cnt) * [5]
    opts = {'popsize': args.pop_size}
    if args.seed is not None:
        opts['seed'] = args.seed
    es = Strategy(x0, args.sigma, opts)
    agent = agent_trajopt(tot_timestep, gripper_cnt, max_moving_dist=args.max_dist)
    sub_steps = int(tot_timestep / args.abs_step)
    scaling = 5.0 / (sub_steps * 0.0003)
    scaling_angle = 5.0 / (sub_steps * 0.01)
    target = np.load(args.target_dir) if args.env == "forming" else None
    tape = args.env in ("balancing", "bounce", "bouncing")

    def reward_of(sys):
        """run_cmaes_all.py:137-162"""
        if args.reward_name is None or args.env == "balancing":
            if args.env == "balancing":
                if args.reward_name == "compute_reward_throwing":
                    return sys.compute_reward_throwing(analy_grad) + 10
                func = getattr(sys, args.reward_name or "compute_reward_all")
                return func(analy_grad) + 5
            if args.env == "forming":
                return sys.compute_reward(target) + 5
            if args.env in ("bounce", "bouncing"):
                return sys.compute_reward(analy_grad) + 5
            if args.env == "folding":
                return sys.compute_reward(args.curve7, args.curve8) + 5
            return sys.compute_reward() + 5
        func = getattr(sys, args.reward_name)
        if not callable(func):
            raise SystemExit(f"{args.reward_name}, not a callable function!!")
        return func() + 5

    def evaluate(x):
        sys.reset()
        if args.load_dir is not None:
            sys.load_all(args.load_dir)
        decode(agent, x, args, gripper_cnt, sub_steps, scaling, scaling_angle)
        early_stop = False
        stop_step = 0
        if tape:
            analy_grad.copy_pos(sys, 0)
        for frame in range(1, tot_timestep):
            agent.get_action(frame)
            sys.action(frame, agent.delta_pos, agent.delta_rot)
            sys.time_step(projection_query, frame)
            early_stop = sys.check_early_stop(frame)
            if early_stop:
                break
            stop_step = frame + 1
            if tape:
                analy_grad.copy_pos(sys, frame)
        reward = stop_step / tot_timestep * 0.1
        if not early_stop:
            reward += reward_of(sys)
        return -reward

    # Under torch.distributed.run the population is the embarrassingly parallel batch of SURVEY.md section 8e: candidate k of a
    # generation is rolled out on rank k % world (one scene and one engine context per GPU), one all_reduce of the fitness vector
    # per generation; every rank keeps an identical strategy state, which needs a common --seed.
    from ..batch import Batch
    batch = Batch(device=sys.device if hasattr(sys, "device") else None)
    if batch.world > 1 and args.seed is None:
        raise SystemExit("run_cmaes_all: --seed is required when the population is shared between ranks")
    plot_y = []
    for ww in range(args.iter):
        X = es.ask()
        tell_list = batch.share_population(len(X), lambda k: evaluate(X[k]))
        plot_y.extend(tell_list)
        es.tell(X, tell_list)
        if batch.rank == 0:
            es.disp()
            np.save(os.path.join(save_path, "plot_Data.npy"), np.array(plot_y))
            decode(agent, es.result.xbest, args, gripper_cnt, sub_steps, scaling, scaling_angle)
            np.save(os.path.join(save_path, f"traj_{ww}.npy"), agent.traj.to_numpy())
    batch.close()
    return dict(fbest=es.result.fbest, history=plot_y, save_path=save_path)


if __name__ == "__main__":
    main()
