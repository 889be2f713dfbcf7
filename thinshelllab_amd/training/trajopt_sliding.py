"""Friction-coefficient identification on the sliding stack: counterpart of /root/reference/code/training/trajopt_silding.py
(scripts/run_dp_slide.sh is empty in the reference).

Per iteration: scripted pad trajectory (``init_traj_slide``), forward rollout, reward, loss seed ``get_loss_slide``, full
reverse sweep of ``analytic_grad_system.Grad`` with ``count_friction_grad``, gradient step on ``mu_cloth_cloth``."""
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
    parser.add_argument('--mu', type=float, default=1.0)
    parser.add_argument('--render_option', type=str, default="None")
    args = parser.parse_args(argv)

    from ..agent.traj_opt_single import agent_trajopt
    from ..engine.analytic_grad_system import Grad
    from ..engine.geometry import projection_query
    from ..engine.render_engine import Renderer
    from ..task_scene.Scene_sliding import Scene

    tot_timestep = args.tot_step
    sys = Scene(cloth_size=0.06)
    sys.cloths[0].Kb[None] = 1000.0
    sys.mu_cloth_cloth[None] = args.mu
    analy_grad = Grad(sys, tot_timestep, sys.elastic_cnt - 1)
    analy_grad.count_friction_grad = True
    analy_grad.count_kb_grad = False
    agent = agent_trajopt(tot_timestep, sys.elastic_cnt - 1, max_moving_dist=0.001)
    sys.init_all()
    analy_grad.init_mass(sys)
    renderer = Renderer(sys, "sliding", option=args.render_option)
    out_root = os.environ.get("TSL_OUT", os.path.join(os.getcwd(), "imgs"))
    history = {}
    for ww in range(args.l, args.r):
        save_path = os.path.join(out_root, f"traj_opt_slide_{ww}")
        renderer.set_save_dir(save_path)
        print(f"Saving Path: {save_path}")
        sys.reset()
        sys.mu_cloth_elastic[None] = 1.0
        plot_y, mu_list = [], []
        agent.init_traj_slide()
        agent.fix_action(0.015)
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
            tot_reward = sys.compute_reward()
            plot_y.append(tot_reward)
            np.save(os.path.join(save_path, "plot_data.npy"), np.array(plot_y))
            print("total_reward:", plot_y)
            analy_grad.get_loss_slide(sys)
            for s in range(tot_timestep - 1, 0, -1):
                analy_grad.transfer_grad(s, sys, projection_query)
            print("done grad")
            mu_list.append(sys.mu_cloth_cloth.value)
            print("mu: ", mu_list)
            step = analy_grad.grad_friction_coef.value * args.lr
            print("friction grad:", step)
            sys.mu_cloth_cloth[None] = sys.mu_cloth_cloth.value - step
            sys.reset()
            analy_grad.reset()
            np.save(os.path.join(save_path, "mu.npy"), np.array(mu_list))
        history[ww] = (plot_y, mu_list)
    return history


if __name__ == "__main__":
    main()
