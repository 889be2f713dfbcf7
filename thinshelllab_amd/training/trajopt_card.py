"""Bending-stiffness identification on the card scene: counterpart of /root/reference/code/training/trajopt_card.py
(scripts/run_dp_card.sh: --l 0 --r 1 --iter 50 --tot_step 80 --lr 20000 --Kb 1400).

Per iteration: scripted gripper trajectory (``init_traj_card``), forward rollout onto the tape, reward, loss seed ``get_loss_card``,
reverse sweep of ``analytic_grad_system.Grad`` over the steps ``tot_step-1 .. 51`` (trajopt_card.py:103), gradient step on
``cloths[0].Kb`` with the learning rate decayed by 0.95.  Outputs as the reference (``best_traj.npy``, ``plot_data.npy``, per-frame
``faces_*.npy`` / ``verts_*.npy`` of the first card) plus the parameter history ``Kb.npy``."""
from argparse import ArgumentParser


def main(argv=None):
    parser = ArgumentParser()
    for flag, typ, default in (('--l', int, 0), ('--r', int, 5), ('--iter', int, 10), ('--lr', float, 0.001), ('--tot_step', int, 5),
                               ('--Kb', float, 1000.0), ('--render_option', str, "None"), ('--first_backprop_step', int, 50)):
        parser.add_argument(flag, type=typ, default=default)
    args = parser.parse_args(argv)

    from ..agent.traj_opt_single import agent_trajopt
    from ..engine.analytic_grad_system import Grad
    from ..engine.geometry import projection_query
    from ..engine.render_engine import Renderer
    from ..task_scene.Scene_card import Scene
    from ._common import identify

    sys = Scene(cloth_size=0.06)
    sys.cloths[0].Kb[None] = args.Kb
    analy_grad = Grad(sys, args.tot_step, sys.elastic_cnt - 1)
    agent = agent_trajopt(args.tot_step, sys.elastic_cnt - 1, max_moving_dist=0.001)
    sys.init_all()
    analy_grad.init_mass(sys)

    def set_kb(s, v):
        s.cloths[0].Kb[None] = v

    return identify(args, sys, analy_grad, agent, Renderer(sys, "card", option=args.render_option), projection_query, tag="card", name="Kb",
                    get_param=lambda s: s.cloths[0].Kb.value, set_param=set_kb, get_grad=lambda g: g.grad_kb.value,
                    loss_fn=lambda g, s: g.get_loss_card(s), mu_cloth_elastic=1.0, init_traj=lambda a: a.init_traj_card(),
                    fix_action_each_iter=True, first_backprop_step=args.first_backprop_step, lr_decay=0.95, save_first_cloth=True)


if __name__ == "__main__":
    main()
