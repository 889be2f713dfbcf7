"""Lifting trajectory optimisation: counterpart of /root/reference/code/training/trajopt_lifting.py
(scripts/run_trajopt_lifting.sh: --l 0 --r 1 --iter 400 --tot_step 50 --lr 0.00001)."""
from argparse import ArgumentParser


def main(argv=None):
    parser = ArgumentParser()
    parser.add_argument('--l', type=int, default=0)
    parser.add_argument('--r', type=int, default=5)
    parser.add_argument('--iter', type=int, default=10)
    parser.add_argument('--lr', type=float, default=0.001)
    parser.add_argument('--tot_step', type=int, default=5)
    parser.add_argument('--Kb', type=float, default=100)
    parser.add_argument('--mu', type=float, default=1.0)
    parser.add_argument('--load_traj', type=str, default=None)
    parser.add_argument('--mode', type=str, default="grad")
    parser.add_argument('--render_option', type=str, default="None")
    args = parser.parse_args(argv)

    from ..agent.traj_opt_single import agent_trajopt
    from ..engine.analytic_grad_single import Grad
    from ..engine.geometry import projection_query
    from ..engine.render_engine import Renderer
    from ..optimizer.optim import Adam_single
    from ..task_scene.Scene_lifting import Scene
    from ._common import optimise

    tot_timestep = args.tot_step
    sys = Scene(cloth_size=0.06)
    sys.cloths[0].Kb[None] = args.Kb
    analy_grad = Grad(sys, tot_timestep, sys.elastic_cnt - 1)
    adam = Adam_single((tot_timestep, sys.elastic_cnt - 1, 6), args.lr, 0.9, 0.9999, 1e-8)
    agent = agent_trajopt(tot_timestep, sys.elastic_cnt - 1, max_moving_dist=0.001)
    sys.init_all()
    analy_grad.init_mass(sys)
    renderer = Renderer(sys, "lifting", option=args.render_option)

    def before(s):
        s.mu_cloth_elastic[None] = args.mu

    return optimise(args, sys, analy_grad, adam, agent, renderer, projection_query, tag="lift",
                    reward_fn=lambda s, g: s.compute_reward(), loss_fn=lambda s, g: g.get_loss_lift(s),
                    limit_grad=True, fix_action=False, before_rollout=before)


if __name__ == "__main__":
    main()
