"""Shared trajectory-optimisation loop of the three drivers (forward rollout -> reward -> loss seed -> reverse sweep ->
Adam), order of operations as in /root/reference/code/training/trajopt_folding.py:59-142 (lifting / balancing differ
only in scene, reward, loss seed and the action-limit handling)."""
import os
import time

import numpy as np


def optimise(args, sys, analy_grad, adam, agent, renderer, projection_query, *, tag, reward_fn, loss_fn, limit_grad, fix_action,
             before_rollout=None, out_root=None, first_backprop_step=0, init_traj=None, announce_init_reward=False):
    """first_backprop_step: the reverse sweep covers the steps tot_step-1 .. first_backprop_step+1 (pick-and-fold stops at 9,
    interact at 6); init_traj: scripted start trajectory when none is loaded."""
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
        elif init_traj is not None:
            init_traj(agent)
        adam.reset()
        if announce_init_reward:
            if before_rollout is not None:
                before_rollout(sys)
            print("init reward:", reward_fn(sys, analy_grad))
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
            for s in range(tot_timestep - 1, first_backprop_step, -1):
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


def identify(args, sys, analy_grad, agent, renderer, projection_query, *, tag, get_param, set_param, get_grad, loss_fn, name,
             mu_cloth_elastic, init_traj=None, fix_action_each_iter=False, first_backprop_step=0, clamp_step=None, lr_decay=1.0,
             save_first_cloth=False, out_root=None):
    """Shared loop of the system-identification drivers (card / bouncing / sliding): rollout -> reward -> loss seed -> reverse sweep
    of analytic_grad_system.Grad -> gradient step on ONE physical parameter.  Order of operations as in
    /root/reference/code/training/trajopt_card.py:59-118 (bouncing and sliding differ in the clamping of the step, the
    learning-rate decay and whether a trajectory exists)."""
    tot_timestep = args.tot_step
    out_root = out_root or os.environ.get("TSL_OUT", os.path.join(os.getcwd(), "imgs"))
    lr = args.lr
    history = {}
    for ww in range(args.l, args.r):
        save_path = os.path.join(out_root, f"traj_opt_{tag}_{ww}")
        renderer.set_save_dir(save_path)
        print(f"Saving Path: {save_path}")
        sys.reset()
        sys.mu_cloth_elastic[None] = mu_cloth_elastic
        plot_y, values = [], []
        if init_traj is not None:
            init_traj(agent)
            agent.fix_action(0.015)
            np.save(os.path.join(save_path, "best_traj.npy"), agent.traj.to_numpy())
        for i in range(args.iter):
            print("iter: ", i)
            analy_grad.copy_pos(sys, 0)
            start_time = time.time()
            for frame in range(1, tot_timestep):
                if agent is not None:
                    agent.get_action(frame)
                    sys.action(frame, agent.delta_pos, agent.delta_rot)
                sys.time_step(projection_query, frame)
                analy_grad.copy_pos(sys, frame)
                if save_first_cloth:
                    np.save(os.path.join(save_path, f"faces_{frame}.npy"), sys.cloths[0].f2v.to_numpy())
                    np.save(os.path.join(save_path, f"verts_{frame}.npy"), sys.cloths[0].pos.to_numpy())
            print("tot_time:", time.time() - start_time)
            plot_y.append(sys.compute_reward())
            np.save(os.path.join(save_path, "plot_data.npy"), np.array(plot_y))
            print("total_reward:", plot_y)
            loss_fn(analy_grad, sys)
            for s in range(tot_timestep - 1, first_backprop_step, -1):
                analy_grad.transfer_grad(s, sys, projection_query)
            print("done grad")
            step = get_grad(analy_grad) * lr
            if clamp_step is not None:
                step = min(max(step, -clamp_step), clamp_step)
            values.append(get_param(sys))
            set_param(sys, get_param(sys) - step)
            print(f"{name} history:", values, f"-> {get_param(sys)} (step {step})")
            sys.reset()
            lr *= lr_decay
            if fix_action_each_iter and agent is not None:
                agent.fix_action(0.015)
            analy_grad.reset()
            np.save(os.path.join(save_path, f"{name}.npy"), np.array(values + [get_param(sys)]))
        history[ww] = (plot_y, values)
    return history
