import sys, types, time
sys.path.insert(0, ".")
import numpy as np, torch
import bench
from thinshelllab_amd.engine.geometry import projection_query as contact
from thinshelllab_amd.engine.analytic_grad_single import Grad
args = types.SimpleNamespace(workload="cfg4", grid=224, cloth_size=None, idle=0)
s = bench.build_scene(args, 0)
ctx = s._ensure_ctx(); ctx.set_param("direct", 1)
K = 12
g = Grad(s, K + 1, s.gripper.n_part); g.allow_unconverged = True
g.copy_pos(s, 0)
tf = []; 
for f in range(1, K + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s.action(f, *bench._drive(s.gripper.n_part, s._bench_gs, 0, f, 0))
    torch.cuda.synchronize(); t1 = time.perf_counter()
    st = s.time_step(contact, f)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    g.copy_pos(s, f)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    tf.append((t1 - t0, t2 - t1, t3 - t2, st["plans"]))
g.pos_grad.t.zero_(); g.angleref_grad.t.zero_()
t0 = time.perf_counter(); g.get_loss_balance(s); torch.cuda.synchronize(); tl = time.perf_counter() - t0
ta = []
info0 = ctx.direct_info()
for q in range(K, 0, -1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g.transfer_grad(q, s, contact)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    i1 = ctx.direct_info()
    ta.append((t1 - t0, i1["plans"] - info0["plans"], g.last_stats["iters"])); info0 = i1
print("forward: action / time_step / copy_pos [ms], plans")
for x in tf: print("  %.2f  %.2f  %.2f  %d" % (1e3 * x[0], 1e3 * x[1], 1e3 * x[2], x[3]))
print("loss seed %.2f ms" % (1e3 * tl))
print("adjoint: transfer_grad [ms], plans built, refinement iterations")
for x in ta: print("  %.2f  %d  %d" % (1e3 * x[0], x[1], x[2]))
