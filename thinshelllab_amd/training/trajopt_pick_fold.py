"""Pick-and-fold trajectory optimisation: counterpart of /root/reference/code/training/trajopt_pick_fold.py (same flags;
scripts/run_trajopt_pickfolding.sh is empty in the reference).  Reward ``compute_reward_pick_fold``, loss seed
``get_loss_pick_fold``, reverse sweep over the steps ``tot_step-1 .. 9`` (:117), action-limit gradient, Adam on the trajectory."""
import os
import time
from argparse import ArgumentParser

import numpy as np


def main(argv=None):
    parser = ArgumentParser()
    parser.add_argument('--l', type=int, default=0)
    parser.add_argument('--r', type=int, default=5)
    parser.add_argument('--iter', type=int, default=10)
    parser.add_argument('--lr', type=float, default=0.001)
    parser.add_argument('--tot_step', type=int, default=5)
    parser.add_argument('--load_traj', type=str, default=None)
    parser.add_argument('--render', type=int, default=10)
    parser.add_argument('--max_dist', type=float, default=0.001)
    parser.add_argument('--render_option', type=str, default="None")
    parser.add_argument('--first_backprop_step', type=int, default=8, help="the reverse sweep stops above this step (trajopt_pick_fold.py:117)")
    args = parser.parse_args(argv)

    from ..agent.traj_opt_single import agent_trajopt
    from ..engine.analytic_grad_single import Grad
    from ..engine.geometry import projection_query
    from ..engine.render_engine import Renderer
    from ..optimizer.optim import Adam_single
    from ..task_scene.Scene_pick import Scene

    tot_timestep = args.tot_step
    sys = Scene(cloth_size=0.06)
    sys.cloths[0].Kb[None] = 200.0
    analy_grad = Grad(sys, tot_timestep, sys.elastic_cnt - 1)
    adam = Adam_single((tot_timestep, sys.elastic_cnt - 1, 6), args.lr, 0.9, 0.9999, 1e-8)
    agent = agent_trajopt(tot_timestep, sys.elastic_cnt - 1, max_moving_dist=args.max_dist)
    sys.init_all()
    analy_grad.init_mass(sys)
    renderer = Renderer(sys, "pick", option=args.render_option)
    out_root = os.environ.get("TSL_OUT", os.path.join(os.getcwd(), "imgs"))
    now_reward = -100000
    history = {}
    for ww in range(args.l, args.r):
        save_path = os.path.join(out_root, f"traj_opt_pick_fold_{ww}")
        renderer.set_save_dir(save_path)
        print(f"Saving Path: {save_path}")
        sys.reset()
        sys.mu_cloth_elastic[None] = 10.0
        plot_y = []
        if args.load_traj is not None:
            agent.traj.from_numpy(np.load(args.load_traj))
        else:
            agent.init_traj_pick_fold()
        adam.reset()
        for i in range(args.iter):
            print("iter: ", i)
            analy_grad.copy_pos(sys, 0)
            start_time = time.time()
            for frame in range(1, tot_timestep):
                agent.get_action(frame)
                sys.action(frame, agent.delta_pos, agent.delta_rot)
                sys.time_step(projection_query, frame)
                analy_grad.copy_pos(sys, frame)
            print("tot_time:", time.time() - start_time)
            tot_reward = sys.compute_reward_pick_fold()
            plot_y.append(tot_reward)
            print("total_reward:", plot_y)
            if tot_reward > now_reward:
                now_reward = tot_reward
                np.save(os.path.join(save_path, "best_traj.npy"), agent.traj.to_numpy())
            np.save(os.path.join(save_path, "plot_data.npy"), np.array(plot_y))
            analy_grad.get_loss_pick_fold(sys)
            for s in range(tot_timestep - 1, args.first_backprop_step, -1):
                analy_grad.transfer_grad(s, sys, projection_query)
            print("done grad")
            analy_grad.apply_action_limit_grad(agent, 0.015)
            sys.reset()
            adam.step(agent.traj, analy_grad.gripper_grad)
            analy_grad.reset()
        history[ww] = plot_y
    return history


if __name__ == "__main__":
    main()
