"""Forming trajectory optimisation: counterpart of /root/reference/code/training/trajopt_forming.py (same flags;
scripts/run_trajopt_forming.sh is empty in the reference).  Without ``--target_dir`` the driver only records the final cloth
pose of the current trajectory (``cloth_pos.npy``: how the reference produces a target, trajopt_forming.py:99-101); with it the
reward is the negative squared distance to that pose and the loss seed is ``get_loss_push``."""
import os
from argparse import ArgumentParser

import numpy as np


def main(argv=None):
    parser = ArgumentParser()
    parser.add_argument('--l', type=int, default=0)
    parser.add_argument('--r', type=int, default=5)
    parser.add_argument('--iter', type=int, default=10)
    parser.add_argument('--lr', type=float, default=0.001)
    parser.add_argument('--tot_step', type=int, default=5)
    parser.add_argument('--target_dir', type=str, default=None)
    parser.add_argument('--load_traj', type=str, default=None)
    parser.add_argument('--render_option', type=str, default="None")
    args = parser.parse_args(argv)

    from ..agent.traj_opt_single import agent_trajopt
    from ..engine.analytic_grad_single import Grad
    from ..engine.geometry import projection_query
    from ..engine.render_engine import Renderer
    from ..optimizer.optim import Adam_single
    from ..task_scene.Scene_forming import Scene
    from ._common import optimise

    tot_timestep = args.tot_step
    sys = Scene(cloth_size=0.1)
    sys.cloths[0].Kb[None] = 200.0
    analy_grad = Grad(sys, tot_timestep, sys.elastic_cnt - 1)
    adam = Adam_single((tot_timestep, sys.elastic_cnt - 1, 6), args.lr, 0.9, 0.9999, 1e-8)
    agent = agent_trajopt(args.tot_step, sys.elastic_cnt - 1, max_moving_dist=0.001)
    sys.init_all()
    analy_grad.init_mass(sys)
    renderer = Renderer(sys, "forming", option=args.render_option)

    def before(s):
        s.mu_cloth_elastic[None] = 5.0

    if args.target_dir is None:
        # record the pose reached by the given (or zero) trajectory: the target of a later optimisation run
        out_root = os.environ.get("TSL_OUT", os.path.join(os.getcwd(), "data"))
        pos_save_dir = os.path.join(out_root, "forming_pos_save")
        os.makedirs(pos_save_dir, exist_ok=True)
        sys.reset(); before(sys)
        if args.load_traj is not None:
            agent.traj.from_numpy(np.load(args.load_traj)); agent.fix_action(0.015)
        for frame in range(1, tot_timestep):
            agent.get_action(frame)
            sys.action(frame, agent.delta_pos, agent.delta_rot)
            sys.time_step(projection_query, frame)
        np.save(os.path.join(pos_save_dir, "cloth_pos.npy"), sys.cloths[0].pos.to_numpy())
        print("saved", os.path.join(pos_save_dir, "cloth_pos.npy"))
        return {}

    target_pos = np.load(args.target_dir)
    return optimise(args, sys, analy_grad, adam, agent, renderer, projection_query, tag="forming",
                    reward_fn=lambda s, g: s.compute_reward(target_pos), loss_fn=lambda s, g: g.get_loss_push(s, target_pos),
                    limit_grad=False, fix_action=True, before_rollout=before)


if __name__ == "__main__":
    main()
