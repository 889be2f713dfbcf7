import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from thinshelllab_amd.task_scene.Scene_drape import Scene
N = int(sys.argv[1]) if len(sys.argv) > 1 else 224
s = Scene(cloth_size=0.1 / 15 * N, N=N); s.init_all()
ctx = s._ensure_ctx(); s.compute_residual_and_Hessian(spd=True)
ctx.set_param('cg_maxit', 64); ctx.solve(s.F.t)  # fills v_b with a real right-hand side
L = ctx.L; L.tsl_bench_spmv.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
b = ctx.profile_read()["bytes_per_launch"]
for v in [0, 3, 10, 11, 12, 13, 14]:
    for rep in range(2):
        us = C.c_double(0); L.tsl_bench_spmv(ctx.h, v, 500, C.byref(us))
    print(f"variant {v}: {us.value:.2f} us/launch  {b/us.value/1e3:.0f} GB/s")
