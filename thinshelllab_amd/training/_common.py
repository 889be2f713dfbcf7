"""Shared trajectory-optimisation loop of the three drivers (forward rollout -> reward -> loss seed -> reverse sweep ->
Adam), order of operations as in /root/reference/code/training/trajopt_folding.py:59-142 (lifting / balancing differ
only in scene, reward, loss seed and the action-limit handling)."""
import os
import time

import numpy as np


def optimise(args, sys, analy_grad, adam, agent, renderer, projection_query, *, tag, reward_fn, loss_fn, limit_grad, fix_action,
             before_rollout=None, out_root=None):
    tot_timestep = args.tot_step
    out_root = out_root or os.environ.get("TSL_OUT", os.path.join(os.getcwd(), "imgs"))
    history = {}
    now_reward = -100000
    for ww in range(args.l, args.r):
        save_path = os.path.join(out_root, f"traj_opt_{tag}_{ww}")
        renderer.set_save_dir(save_path)
        print(f"Saving Path: {save_path}")
        sys.reset()
        plot_y = []
        if args.load_traj is not None:
            agent.traj.from_numpy(np.load(args.load_traj))
        adam.reset()
        for i in range(args.iter):
            print("iter: ", i)
            if before_rollout is not None:
                before_rollout(sys)
            analy_grad.copy_pos(sys, 0)
            start_time = time.time()
            for frame in range(1, tot_timestep):
                agent.get_action(frame)
                sys.action(frame, agent.delta_pos, agent.delta_rot)
                sys.time_step(projection_query, frame)
                analy_grad.copy_pos(sys, frame)
            print("tot_time:", time.time() - start_time)
            tot_reward = reward_fn(sys, analy_grad)
            plot_y.append(tot_reward)
            print("total_reward:", plot_y)
            if tot_reward > now_reward:
                now_reward = tot_reward
                np.save(os.path.join(save_path, "best_traj.npy"), agent.traj.to_numpy())
            np.save(os.path.join(save_path, "plot_data.npy"), np.array(plot_y))
            loss_fn(sys, analy_grad)
            for s in range(tot_timestep - 1, 0, -1):
                analy_grad.transfer_grad(s, sys, projection_query)
            if limit_grad:
                analy_grad.apply_action_limit_grad(agent, 0.015)
            print("done grad")
            sys.reset()
            adam.step(agent.traj, analy_grad.gripper_grad)
            if fix_action:
                agent.fix_action(0.015)
            analy_grad.reset()
        history[ww] = plot_y
    return history
