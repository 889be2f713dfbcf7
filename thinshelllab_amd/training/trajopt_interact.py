"""Interaction ("separating") trajectory optimisation: counterpart of /root/reference/code/training/trajopt_interact.py (same
flags; scripts/run_trajopt_separating.sh is empty in the reference).  Reward ``compute_reward_1`` (block only) or, with ``--sep``,
``compute_reward`` (sheet against block); loss seeds ``get_loss_interact_1`` / ``get_loss_interact``; reverse sweep over the steps
``tot_step-1 .. 6`` (:151), action-limit gradient, Adam with learning-rate discount."""
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
    parser.add_argument('--sep', action="store_true", default=False)
    parser.add_argument('--Kb', type=float, default=100)
    parser.add_argument('--mu', type=float, default=5.0)
    parser.add_argument('--discount', type=float, default=0.9)
    parser.add_argument('--load_traj', type=str, default=None)
    parser.add_argument('--soft', action="store_true", default=False)
    parser.add_argument('--dense', type=float, default=10000.0)
    parser.add_argument('--render', type=int, default=10)
    parser.add_argument('--render_option', type=str, default="None")
    parser.add_argument('--first_backprop_step', type=int, default=5, help="the reverse sweep stops above this step (trajopt_interact.py:151)")
    args = parser.parse_args(argv)

    from ..agent.traj_opt_single import agent_trajopt
    from ..engine.analytic_grad_single import Grad
    from ..engine.geometry import projection_query
    from ..engine.render_engine import Renderer
    from ..optimizer.optim import Adam_single
    from ..task_scene.Scene_interact import Scene

    tot_timestep = args.tot_step
    sys = Scene(cloth_size=0.06, soft=args.soft, dense=args.dense)
    sys.cloths[0].Kb[None] = 0.1 if args.soft else args.Kb
    gripper_cnt = int((sys.effector_cnt - 1) // 2) if sys.enable_gripper else sys.elastic_cnt - 1
    analy_grad = Grad(sys, tot_timestep, gripper_cnt)
    adam = Adam_single((tot_timestep, gripper_cnt, 6), args.lr, 0.9, 0.9999, 1e-8, discount=args.discount)
    agent = agent_trajopt(tot_timestep, gripper_cnt, max_moving_dist=0.002)
    sys.init_all()
    analy_grad.init_mass(sys)
    renderer = Renderer(sys, "interact", option=args.render_option)
    out_root = os.environ.get("TSL_OUT", os.path.join(os.getcwd(), "imgs"))
    reward = (lambda: sys.compute_reward()) if args.sep else (lambda: sys.compute_reward_1())
    now_reward = -100000
    history = {}
    for ww in range(args.l, args.r):
        save_path = os.path.join(out_root, f"traj_opt_interact_{ww}")
        renderer.set_save_dir(save_path)
        print(f"Saving Path: {save_path}")
        sys.reset()
        sys.mu_cloth_elastic[None] = args.mu
        plot_y = []
        if args.load_traj is not None:
            agent.traj.from_numpy(np.load(args.load_traj))
            agent.fix_action(0.015)
        adam.reset()
        print("init reward:", reward())
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
            tot_reward = reward()
            plot_y.append(tot_reward)
            print("total_reward:", plot_y)
            if tot_reward > now_reward:
                now_reward = tot_reward
                np.save(os.path.join(save_path, "best_traj.npy"), agent.traj.to_numpy())
            np.save(os.path.join(save_path, "plot_data.npy"), np.array(plot_y))
            if args.sep:
                print("min traj x:", min(0.0, float(agent.traj.to_numpy()[:, 0, 0].min())))
                analy_grad.get_loss_interact(sys)
            else:
                analy_grad.get_loss_interact_1(sys)
            for s in range(tot_timestep - 1, args.first_backprop_step, -1):
                analy_grad.transfer_grad(s, sys, projection_query)
            print("done grad")
            sys.reset()
            analy_grad.apply_action_limit_grad(agent, 0.015)
            adam.step(agent.traj, analy_grad.gripper_grad)
            analy_grad.reset()
        history[ww] = plot_y
    return history


if __name__ == "__main__":
    main()
