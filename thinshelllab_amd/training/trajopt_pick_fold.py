"""Pick-and-fold trajectory optimisation: counterpart of /root/reference/code/training/trajopt_pick_fold.py (same flags;
scripts/run_trajopt_pickfolding.sh is empty in the reference).  Reward ``compute_reward_pick_fold``, loss seed
``get_loss_pick_fold``, reverse sweep over the steps ``tot_step-1 .. 9`` (:117), action-limit gradient, Adam on the trajectory."""
from argparse import ArgumentParser


def main(argv=None):
    parser = ArgumentParser()
    for flag, typ, default in (('--l', int, 0), ('--r', int, 5), ('--iter', int, 10), ('--lr', float, 0.001), ('--tot_step', int, 5),
                               ('--load_traj', str, None), ('--render', int, 10), ('--max_dist', float, 0.001), ('--render_option', str, "None")):
        parser.add_argument(flag, type=typ, default=default)
    args = parser.parse_args(argv)

    from ..agent.traj_opt_single import agent_trajopt
    from ..engine.analytic_grad_single import Grad
    from ..engine.geometry import projection_query
    from ..engine.render_engine import Renderer
    from ..optimizer.optim import Adam_single
    from ..task_scene.Scene_pick import Scene
    from ._common import optimise

    sys = Scene(cloth_size=0.06)
    sys.cloths[0].Kb[None] = 200.0
    n_part = sys.elastic_cnt - 1
    analy_grad = Grad(sys, args.tot_step, n_part)
    adam = Adam_single((args.tot_step, n_part, 6), args.lr, 0.9, 0.9999, 1e-8)
    agent = agent_trajopt(args.tot_step, n_part, max_moving_dist=args.max_dist)
    sys.init_all()
    analy_grad.init_mass(sys)

    def before(s):
        s.mu_cloth_elastic[None] = 10.0

    return optimise(args, sys, analy_grad, adam, agent, Renderer(sys, "pick", option=args.render_option), projection_query, tag="pick_fold",
                    reward_fn=lambda s, g: s.compute_reward_pick_fold(), loss_fn=lambda s, g: g.get_loss_pick_fold(s),
                    limit_grad=True, fix_action=False, before_rollout=before, first_backprop_step=8,
                    init_traj=lambda a: a.init_traj_pick_fold())


if __name__ == "__main__":
    main()
