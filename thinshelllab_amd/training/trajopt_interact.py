"""Interaction ("separating") trajectory optimisation: counterpart of /root/reference/code/training/trajopt_interact.py (same
flags; scripts/run_trajopt_separating.sh is empty in the reference).  Reward ``compute_reward_1`` (block only) or, with ``--sep``,
``compute_reward`` (sheet against block); loss seeds ``get_loss_interact_1`` / ``get_loss_interact``; reverse sweep over the steps
``tot_step-1 .. 6`` (:151), action-limit gradient, Adam with learning-rate discount."""
from argparse import ArgumentParser


def main(argv=None):
    parser = ArgumentParser()
    for flag, typ, default in (('--l', int, 0), ('--r', int, 5), ('--iter', int, 10), ('--lr', float, 0.001), ('--tot_step', int, 5),
                               ('--Kb', float, 100), ('--mu', float, 5.0), ('--discount', float, 0.9), ('--load_traj', str, None),
                               ('--dense', float, 10000.0), ('--render', int, 10), ('--render_option', str, "None")):
        parser.add_argument(flag, type=typ, default=default)
    parser.add_argument('--sep', action="store_true", default=False)
    parser.add_argument('--soft', action="store_true", default=False)
    args = parser.parse_args(argv)

    from ..agent.traj_opt_single import agent_trajopt
    from ..engine.analytic_grad_single import Grad
    from ..engine.geometry import projection_query
    from ..engine.render_engine import Renderer
    from ..optimizer.optim import Adam_single
    from ..task_scene.Scene_interact import Scene
    from ._common import optimise

    sys = Scene(cloth_size=0.06, soft=args.soft, dense=args.dense)
    sys.cloths[0].Kb[None] = 0.1 if args.soft else args.Kb
    gripper_cnt = int((sys.effector_cnt - 1) // 2) if sys.enable_gripper else sys.elastic_cnt - 1
    analy_grad = Grad(sys, args.tot_step, gripper_cnt)
    adam = Adam_single((args.tot_step, gripper_cnt, 6), args.lr, 0.9, 0.9999, 1e-8, discount=args.discount)
    agent = agent_trajopt(args.tot_step, gripper_cnt, max_moving_dist=0.002)
    sys.init_all()
    analy_grad.init_mass(sys)

    def before(s):
        s.mu_cloth_elastic[None] = args.mu

    if args.sep:
        reward_fn, loss_fn = (lambda s, g: s.compute_reward()), (lambda s, g: g.get_loss_interact(s))
    else:
        reward_fn, loss_fn = (lambda s, g: s.compute_reward_1()), (lambda s, g: g.get_loss_interact_1(s))
    return optimise(args, sys, analy_grad, adam, agent, Renderer(sys, "interact", option=args.render_option), projection_query, tag="interact",
                    reward_fn=reward_fn, loss_fn=loss_fn, limit_grad=True, fix_action=False, before_rollout=before,
                    first_backprop_step=5, announce_init_reward=True)


if __name__ == "__main__":
    main()
