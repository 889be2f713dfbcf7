"""Bending-stiffness identification on the bouncing sheet: counterpart of /root/reference/code/training/trajopt_bouncing.py
(scripts/run_dp_bouncing.sh is empty in the reference).  No actuator: forward rollout, reward, loss seed ``get_loss_table``, full
reverse sweep of ``analytic_grad_system.Grad``, gradient step on ``cloths[0].Kb`` clamped to +-100, learning rate decayed by 0.95."""
from argparse import ArgumentParser


def main(argv=None):
    parser = ArgumentParser()
    for flag, typ, default in (('--l', int, 0), ('--r', int, 5), ('--iter', int, 10), ('--lr', float, 0.001), ('--tot_step', int, 5),
                               ('--Kb', float, 1000.0), ('--render_option', str, "None")):
        parser.add_argument(flag, type=typ, default=default)
    args = parser.parse_args(argv)

    from ..engine.analytic_grad_system import Grad
    from ..engine.geometry import projection_query
    from ..engine.render_engine import Renderer
    from ..task_scene.Scene_bouncing import Scene
    from ._common import identify

    sys = Scene(cloth_size=0.06)
    sys.cloths[0].Kb[None] = args.Kb
    analy_grad = Grad(sys, args.tot_step, sys.elastic_cnt - 1)
    sys.init_all()
    analy_grad.init_mass(sys)

    def set_kb(s, v):
        s.cloths[0].Kb[None] = v

    return identify(args, sys, analy_grad, None, Renderer(sys, "bouncing", option=args.render_option), projection_query, tag="table", name="Kb",
                    get_param=lambda s: s.cloths[0].Kb.value, set_param=set_kb, get_grad=lambda g: g.grad_kb.value,
                    loss_fn=lambda g, s: g.get_loss_table(s), mu_cloth_elastic=0.5, clamp_step=100.0, lr_decay=0.95)


if __name__ == "__main__":
    main()
