"""Scaling probe (not part of the product): Newton / PCG iteration counts and wall time of the drape workload.
usage: exp_scale.py N[:cloth_size] ..."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from thinshelllab_amd.task_scene.Scene_drape import Scene

for a in sys.argv[1:] or ["32", "71"]:
    N = int(a.split(":")[0]); size = float(a.split(":")[1]) if ":" in a else 0.1 / 15 * N
    s = Scene(cloth_size=size, N=N)
    s.init_all()
    ctx = s._ensure_ctx()
    ctx.set_param("cg_maxit", 400000)
    for kv in os.environ.get("TSL_PARAMS", "").split(","):
        if "=" in kv:
            ctx.set_param(kv.split("=")[0], float(kv.split("=")[1]))
    ctx.profile_reset(True)
    for step in range(2):
        torch.cuda.synchronize(); t = time.time()
        st = s.time_step(None, step + 1)
        torch.cuda.synchronize(); dt = time.time() - t
        pr = ctx.profile_read()
        print(f"N={N} size={size:.3f} dx={size/N:.2e} T={2*N*N} step={step} {dt*1e3:.1f} ms newton={st['newton_iters']} cg={st['cg_iters']} ls={st['ls_evals']} "
              f"restarts={st['restarts']} fb={st['fallback']} delta={st['last_delta']:.2e} | spmv {pr['ms_per_launch']*1e3:.2f} us "
              f"{pr['bytes_per_launch']/max(pr['ms_per_launch'],1e-9)/1e6:.1f} GB/s", flush=True)
