"""Bending-stiffness identification on the bouncing sheet: counterpart of /root/reference/code/training/trajopt_bouncing.py
(scripts/run_dp_bouncing.sh is empty in the reference; flags as in that script's siblings).

Per iteration: forward rollout (no actuator), reward, loss seed ``get_loss_table``, full reverse sweep of
``analytic_grad_system.Grad``, clamped gradient step on ``cloths[0].Kb`` (+-100), learning rate decayed by 0.95."""
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
    parser.add_argument('--Kb', type=float, default=1000.0)
    parser.add_argument('--render_option', type=str, default="None")
    args = parser.parse_args(argv)

    from ..engine.analytic_grad_system import Grad
    from ..engine.geometry import projection_query
    from ..engine.render_engine import Renderer
    from ..task_scene.Scene_bouncing import Scene

    tot_timestep = args.tot_step
    sys = Scene(cloth_size=0.06)
    sys.cloths[0].Kb[None] = args.Kb
    analy_grad = Grad(sys, tot_timestep, sys.elastic_cnt - 1)
    sys.init_all()
    analy_grad.init_mass(sys)
    renderer = Renderer(sys, "bouncing", option=args.render_option)
    out_root = os.environ.get("TSL_OUT", os.path.join(os.getcwd(), "imgs"))
    lr = args.lr
    history = {}
    for ww in range(args.l, args.r):
        save_path = os.path.join(out_root, f"traj_opt_table_{ww}")
        renderer.set_save_dir(save_path)
        print(f"Saving Path: {save_path}")
        sys.reset()
        sys.mu_cloth_elastic[None] = 0.5
        plot_y, kb_list = [], []
        for i in range(args.iter):
            print("iter: ", i)
            analy_grad.copy_pos(sys, 0)
            start_time = time.time()
            for frame in range(1, tot_timestep):
                sys.time_step(projection_query, frame)
                analy_grad.copy_pos(sys, frame)
            print("tot_time:", time.time() - start_time)
            tot_reward = sys.compute_reward()
            plot_y.append(tot_reward)
            np.save(os.path.join(save_path, "plot_data.npy"), np.array(plot_y))
            print("total_reward:", plot_y)
            analy_grad.get_loss_table(sys)
            for j in range(tot_timestep - 1, 0, -1):
                analy_grad.transfer_grad(j, sys, projection_query)
            loss_grad = min(max(analy_grad.grad_kb.value * lr, -100.0), 100.0)
            sys.cloths[0].Kb[None] = sys.cloths[0].Kb.value - loss_grad
            kb_list.append(sys.cloths[0].Kb.value)
            print("done grad")
            sys.reset()
            print("prev kbs", kb_list)
            print("now kb", sys.cloths[0].Kb.value, "now grad", analy_grad.grad_kb.value * lr)
            analy_grad.reset()
            lr *= 0.95
            np.save(os.path.join(save_path, "kb.npy"), np.array(kb_list))
        history[ww] = (plot_y, kb_list)
    return history


if __name__ == "__main__":
    main()
