"""Gradient-free baseline of the system-identification tasks (forward path only): counterpart of
/root/reference/code/training/run_cmaes_parameter.py (same flags; CMA-ES over a 2-vector whose first entry offsets one physical
parameter: the cloth-cloth friction for `--env slide`, :91-93, else the bending stiffness Kb + 200 x, :95-96; fitness = -compute_reward
after a scripted rollout, :98-105; output plot_Data.npy).  The reference's `--env slide` names no module of its task_scene package
(the scene file is Scene_sliding.py): `slide` and `sliding` both select it here.  e.g.
python -m thinshelllab_amd.training.run_cmaes_parameter --tot_step 50 --iter 5 --pop_size 10 --sigma 0.1 --env card --Kb 1000 --mu 1.0 --traj init_traj_card"""
import importlib
import os
from argparse import ArgumentParser

import numpy as np


def main(argv=None):
    parser = ArgumentParser()
    parser.add_argument('--pop_size', type=int, default=5)
    parser.add_argument('--iter', type=int, default=10)
    parser.add_argument('--tot_step', type=int, default=60)
    parser.add_argument('--sigma', type=float, default=1.0)
    parser.add_argument('--trial', type=str, default="0")
    parser.add_argument('--env', type=str, default="card")
    parser.add_argument('--traj', type=str, default="")
    parser.add_argument('--Kb', type=float, default=100.0)
    parser.add_argument('--max_dist', type=float, default=0.002)
    parser.add_argument('--mu', type=float, default=1.0)
    parser.add_argument('--mu_cloth', type=float, default=1.0)
    parser.add_argument('--seed', type=int, default=None)
    args = parser.parse_args(argv)

    from ..agent.traj_opt_single import agent_trajopt
    from ..batch import Batch
    from ..engine.geometry import projection_query
    try:
        import cma
        Strategy = cma.CMAEvolutionStrategy
    except ImportError:
        from ..optimizer.cmaes import CMAEvolutionStrategy as Strategy
    slide = args.env in ("slide", "sliding")
    Scene = importlib.import_module(f"thinshelllab_amd.task_scene.Scene_{'sliding' if slide else args.env}")
    tot_timestep = args.tot_step
    sys = Scene.Scene(cloth_size=0.06, device=f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}")
    sys.cloths[0].Kb[None] = args.Kb
    sys.init_all()
    sys.mu_cloth_elastic[None] = args.mu
    save_path = os.path.join(os.environ.get("TSL_OUT", os.path.join(os.getcwd(), "data")), f"cmaes_traj_{args.env}_{args.trial}")
    os.makedirs(save_path, exist_ok=True)
    gripper_cnt = sys.elastic_cnt - 1
    if sys.enable_gripper:
        gripper_cnt = int((sys.effector_cnt - 1) // 2)
    opts = {'popsize': args.pop_size}
    if args.seed is not None:
        opts['seed'] = args.seed
    es = Strategy([0, 0], args.sigma, opts)
    agent = agent_trajopt(tot_timestep, max(gripper_cnt, 1), max_moving_dist=args.max_dist)
    if gripper_cnt > 0 and args.traj:
        func = getattr(agent, args.traj)
        if callable(func):
            func()

    def evaluate(x):
        sys.reset()
        if slide:
            sys.mu_cloth_cloth[None] = max(0.0001, args.mu_cloth + x[0])
        else:
            sys.cloths[0].Kb[None] = max(0.0001, args.Kb + x[0] * 200)
        for frame in range(1, tot_timestep):
            if gripper_cnt > 0:
                agent.get_action(frame)
                sys.action(frame, agent.delta_pos, agent.delta_rot)
            sys.time_step(projection_query, frame)
        return -sys.compute_reward()

    batch = Batch(device=sys.device)   # population shared between ranks under torch.distributed.run (needs --seed), as in run_cmaes_all
    if batch.world > 1 and args.seed is None:
        raise SystemExit("run_cmaes_parameter: --seed is required when the population is shared between ranks")
    plot_y = []
    for ww in range(args.iter):
        X = es.ask()
        tell_list = batch.share_population(len(X), lambda k: evaluate(X[k]))
        plot_y.extend(tell_list)
        es.tell(X, tell_list)
        if batch.rank == 0:
            es.disp()
            np.save(os.path.join(save_path, "plot_Data.npy"), np.array(plot_y))
    batch.close()
    return dict(fbest=es.result.fbest, xbest=es.result.xbest, history=plot_y, save_path=save_path)


if __name__ == "__main__":
    main()
