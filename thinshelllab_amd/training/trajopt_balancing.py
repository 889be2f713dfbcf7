"""Balancing trajectory optimisation: counterpart of /root/reference/code/training/trajopt_balancing.py
(scripts/run_trajopt_balancing.sh: --l 0 --r 1 --iter 400 --tot_step 50 --lr 0.00001).  ``--load_state`` points at a
directory written by Scene.save_all (the reference ships one: data/balance_state); omit it to start from rest."""
from argparse import ArgumentParser


def main(argv=None):
    parser = ArgumentParser()
    parser.add_argument('--l', type=int, default=0)
    parser.add_argument('--r', type=int, default=5)
    parser.add_argument('--iter', type=int, default=10)
    parser.add_argument('--lr', type=float, default=0.001)
    parser.add_argument('--tot_step', type=int, default=5)
    parser.add_argument('--throwing', action="store_true", default=False)
    parser.add_argument('--save', action="store_true", default=False)
    parser.add_argument('--load_traj', type=str, default=None)
    parser.add_argument('--Kb', type=float, default=100)
    parser.add_argument('--load_state', type=str, default=None)
    parser.add_argument('--render_option', type=str, default="None")
    args = parser.parse_args(argv)

    from ..agent.traj_opt_single import agent_trajopt
    from ..engine.analytic_grad_single import Grad
    from ..engine.geometry import projection_query
    from ..engine.render_engine import Renderer
    from ..optimizer.optim import Adam_single
    from ..task_scene.Scene_balancing import Scene
    from ._common import optimise

    tot_timestep = args.tot_step
    sys = Scene(cloth_size=0.06)
    sys.cloths[0].Kb[None] = args.Kb
    n_part = sys.gripper.n_part
    analy_grad = Grad(sys, tot_timestep, n_part)
    adam = Adam_single((tot_timestep, n_part, 6), args.lr, 0.9, 0.9999, 1e-8)
    agent = agent_trajopt(tot_timestep, n_part, max_moving_dist=0.001)
    sys.init_all()
    analy_grad.init_mass(sys)
    renderer = Renderer(sys, "balancing", option=args.render_option)

    def before(s):
        s.mu_cloth_elastic[None] = 5.0
        if args.load_state and not args.save:
            s.load_all(args.load_state)

    loss = (lambda s, g: g.get_loss_throwing(s)) if args.throwing else (lambda s, g: g.get_loss_balance(s))
    return optimise(args, sys, analy_grad, adam, agent, renderer, projection_query, tag="balancing",
                    reward_fn=lambda s, g: s.compute_reward_all(g), loss_fn=loss, limit_grad=True, fix_action=False, before_rollout=before)


if __name__ == "__main__":
    main()
