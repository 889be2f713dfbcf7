"""Bending-stiffness identification on the card scene: counterpart of /root/reference/code/training/trajopt_card.py
(scripts/run_dp_card.sh: --l 0 --r 1 --iter 50 --tot_step 80 --lr 20000 --Kb 1400).

Per iteration: scripted gripper trajectory (``init_traj_card``), forward rollout onto the tape, reward, loss seed
``get_loss_card``, reverse sweep of ``analytic_grad_system.Grad`` over the steps ``tot_step-1 .. 51`` (trajopt_card.py:103),
gradient step on ``cloths[0].Kb`` with the learning rate decayed by 0.95.  Outputs as the reference: ``best_traj.npy``,
``plot_data.npy`` and the per-frame ``faces_*.npy`` / ``verts_*.npy`` of the first card; ``kb.npy`` (history) in addition."""
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
    parser.add_argument('--first_backprop_step', type=int, default=50, help="the reverse sweep stops above this step (trajopt_card.py:103)")
    args = parser.parse_args(argv)

    from ..agent.traj_opt_single import agent_trajopt
    from ..engine.analytic_grad_system import Grad
    from ..engine.geometry import projection_query
    from ..engine.render_engine import Renderer
    from ..task_scene.Scene_card import Scene

    tot_timestep = args.tot_step
    sys = Scene(cloth_size=0.06)
    sys.cloths[0].Kb[None] = args.Kb
    analy_grad = Grad(sys, tot_timestep, sys.elastic_cnt - 1)
    agent = agent_trajopt(tot_timestep, sys.elastic_cnt - 1, max_moving_dist=0.001)
    sys.init_all()
    analy_grad.init_mass(sys)
    renderer = Renderer(sys, "card", option=args.render_option)
    out_root = os.environ.get("TSL_OUT", os.path.join(os.getcwd(), "imgs"))
    lr = args.lr
    history = {}
    for ww in range(args.l, args.r):
        save_path = os.path.join(out_root, f"traj_opt_card_{ww}")
        renderer.set_save_dir(save_path)
        print(f"Saving Path: {save_path}")
        sys.reset()
        sys.mu_cloth_elastic[None] = 1.0
        plot_y, kb_list = [], []
        agent.init_traj_card()
        agent.fix_action(0.015)
        np.save(os.path.join(save_path, "best_traj.npy"), agent.traj.to_numpy())
        for i in range(args.iter):
            print("iter: ", i)
            analy_grad.copy_pos(sys, 0)
            start_time = time.time()
            for frame in range(1, tot_timestep):
                agent.get_action(frame)
                sys.action(frame, agent.delta_pos, agent.delta_rot)
                sys.time_step(projection_query, frame)
                analy_grad.copy_pos(sys, frame)
                np.save(os.path.join(save_path, f"faces_{frame}.npy"), sys.cloths[0].f2v.to_numpy())
                np.save(os.path.join(save_path, f"verts_{frame}.npy"), sys.cloths[0].pos.to_numpy())
            print("tot_time:", time.time() - start_time)
            tot_reward = sys.compute_reward()
            plot_y.append(tot_reward)
            np.save(os.path.join(save_path, "plot_data.npy"), np.array(plot_y))
            print("total_reward:", plot_y)
            analy_grad.get_loss_card(sys)
            for s in range(tot_timestep - 1, args.first_backprop_step, -1):
                analy_grad.transfer_grad(s, sys, projection_query)
            grad_kb = analy_grad.grad_kb.value
            sys.cloths[0].Kb[None] = sys.cloths[0].Kb.value - grad_kb * lr
            kb_list.append(sys.cloths[0].Kb.value)
            print("done grad")
            sys.reset()
            lr *= 0.95
            print("prev kbs", kb_list)
            print("now kb", sys.cloths[0].Kb.value, "now grad", grad_kb)
            agent.fix_action(0.015)
            analy_grad.reset()
            np.save(os.path.join(save_path, "kb.npy"), np.array(kb_list))
        history[ww] = (plot_y, kb_list)
    return history


if __name__ == "__main__":
    main()
