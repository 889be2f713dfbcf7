"""Friction-coefficient identification on the sliding stack: counterpart of /root/reference/code/training/trajopt_silding.py
(scripts/run_dp_slide.sh is empty in the reference).  Scripted pad trajectory (``init_traj_slide``), forward rollout, reward, loss
seed ``get_loss_slide``, full reverse sweep of ``analytic_grad_system.Grad`` with ``count_friction_grad``, gradient step on
``mu_cloth_cloth``."""
from argparse import ArgumentParser


def main(argv=None):
    parser = ArgumentParser()
    for flag, typ, default in (('--l', int, 0), ('--r', int, 5), ('--iter', int, 10), ('--lr', float, 0.001), ('--tot_step', int, 5),
                               ('--mu', float, 1.0), ('--render_option', str, "None")):
        parser.add_argument(flag, type=typ, default=default)
    args = parser.parse_args(argv)

    from ..agent.traj_opt_single import agent_trajopt
    from ..engine.analytic_grad_system import Grad
    from ..engine.geometry import projection_query
    from ..engine.render_engine import Renderer
    from ..task_scene.Scene_sliding import Scene
    from ._common import identify

    sys = Scene(cloth_size=0.06)
    sys.cloths[0].Kb[None] = 1000.0
    sys.mu_cloth_cloth[None] = args.mu
    analy_grad = Grad(sys, args.tot_step, sys.elastic_cnt - 1)
    analy_grad.count_friction_grad = True
    analy_grad.count_kb_grad = False
    agent = agent_trajopt(args.tot_step, sys.elastic_cnt - 1, max_moving_dist=0.001)
    sys.init_all()
    analy_grad.init_mass(sys)

    def set_mu(s, v):
        s.mu_cloth_cloth[None] = v

    return identify(args, sys, analy_grad, agent, Renderer(sys, "sliding", option=args.render_option), projection_query, tag="slide", name="mu",
                    get_param=lambda s: s.mu_cloth_cloth.value, set_param=set_mu, get_grad=lambda g: g.grad_friction_coef.value,
                    loss_fn=lambda g, s: g.get_loss_slide(s), mu_cloth_elastic=1.0, init_traj=lambda a: a.init_traj_slide())


if __name__ == "__main__":
    main()
