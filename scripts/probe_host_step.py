"""Probe (not part of the product): where the HOST spends a bench step -- wall time of every call of the rollout loop with a device synchronisation behind
it (action, time_step, copy_pos; loss seed; transfer_grad split into the C call and the Python around it), and cProfile of the same loop."""
import argparse, cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from thinshelllab_amd.engine.analytic_grad_single import Grad
from thinshelllab_amd.engine.geometry import projection_query as contact

args = argparse.Namespace(workload="cfg4", grid=224, cloth_size=None, idle=0, cg_tol=1e-10, param=[])
s = bench.build_scene(args, 0)
ctx = s._ensure_ctx(); ctx.set_param("cg_tol", 1e-10)
K = 6
g = Grad(s, K + 1, s.gripper.n_part); g.init_mass(s)
bench.run_rollout(s, g, 3, args)   # warm-up
T = {}
def lap(name, t0):
    torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0
g.copy_pos(s, 0)
torch.cuda.synchronize(); t_all = time.perf_counter()
for f in range(1, K + 1):
    s._bench_frame = getattr(s, "_bench_frame", 0) + 1
    t0 = time.perf_counter(); s.action(f, *bench._drive(s.gripper.n_part, s._bench_gs, 0, s._bench_frame, 0)); lap("action", t0)
    t0 = time.perf_counter(); st = s.time_step(contact, f); lap("time_step", t0)
    t0 = time.perf_counter(); g.copy_pos(s, f); lap("copy_pos", t0)
t0 = time.perf_counter(); g.pos_grad.t.zero_(); g.angleref_grad.t.zero_(); g.get_loss_balance(s); lap("loss seed", t0)
for k in range(K, 0, -1):
    t0 = time.perf_counter(); g.transfer_grad(k, s, contact); lap("transfer_grad", t0)
torch.cuda.synchronize(); t_all = time.perf_counter() - t_all
print(f"{K} steps: {t_all / K * 1e3:.2f} ms per step (with a synchronisation behind every call)")
for k, v in T.items():
    print(f"  {k:14s} {v / K * 1e3:8.3f} ms per step")
ci = ctx.direct_info()
print("plan seconds total", ci["plan_seconds"], "plans", ci["plans"])
pr = cProfile.Profile(); pr.enable()
bench.run_rollout(s, g, 4, args)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(18)
