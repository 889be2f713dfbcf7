"""One-off measurement (not a test: 8-10 minutes of host time): ONE complete time step of the oracle at the full cfg4 size next to the
HIP engine's.  The GPU scene is driven 12 steps (contacts active), the oracle is mirrored from that state (positions, velocities, previous
positions, latched side flags, gripper frames), both take the 13th step with the same gripper action; the oracle's linear solves are scipy's
SuperLU (the reference calls spsolve).  Reports: Newton iterations of both (does the reference's stop rule trigger at dx = 0.54 mm?),
contact counts, max |x_gpu - x_oracle|.  Result committed as profiles/r03_cfg4_full_step_parity.json.
    gpurun -- 'python scripts/exp_fullsize_step.py > gpurun_out/full_step.json'"""
import json, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as po
from oracle.mirror import oracle_from_scene, sync_oracle_state
from thinshelllab_amd.task_scene.Scene_balancing import Scene
from thinshelllab_amd.engine.geometry import projection_query

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 224
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 12
po.set_threads(min(os.cpu_count() or 4, 32))
s = Scene(cloth_size=0.12 * grid / 224, cloth_N=grid, cloth_M=grid)
s.init_all(); s.mu_cloth_elastic[None] = 5.0; s.prev_pos.copy_from(s.pos)
n_part = s.gripper.n_part
dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3)); dpos[:, 2] = [1e-4, -1e-4][:n_part]
o = oracle_from_scene(po, s, check_init=False)    # at t = 0: gripper frames from the initial poses on both sides
for f in range(1, pre + 1):
    s.action(f, dpos, drot); o.action(dpos, drot)  # the oracle's gripper follows; its bodies are overwritten by the sync below
    st = s.time_step(projection_query, f)
ctx = s._ctx
nb = len(s.body_list)
flag, dr, _, _ = ctx.proj_export()
sync_oracle_state(o, s)
o.arr("proj_flag", (nb, -1))[:] = flag; o.arr("proj_dir", (nb, -1))[:] = dr
o.set_solver(1e-10); o.set_direct(1)
ctx.set_param("cg_tol", 1e-10)
s.action(pre + 1, dpos, drot); o.action(dpos, drot)
t0 = time.time(); st = s.time_step(projection_query, pre + 1); t_gpu = time.time() - t0
o.stats(reset=True)
t0 = time.time(); o.time_step(); t_cpu = time.time() - t0
so = o.stats()
err = float(np.abs(s.pos.to_numpy() - o.pos).max())
move = float(np.abs(s.pos.to_numpy() - s.prev_pos.to_numpy()).max())
print(json.dumps({"what": f"one time step of the oracle (SuperLU) and of the HIP engine from the same state: cfg4 scene with a {grid}x{grid} cloth after {pre} driven steps",
                  "triangles": 2 * grid * grid, "newton_gpu": st["newton_iters"], "newton_oracle": so["newton"], "nc_gpu": st["nc"], "nc_oracle": o.nc,
                  "max_abs_dx_m": err, "largest_displacement_of_the_step_m": move, "gpu_last_delta": st["last_delta"], "gpu_unconverged": st["unconverged"],
                  "line_search_evals_gpu": st["ls_evals"], "line_search_evals_oracle": so["ls"], "seconds_gpu": t_gpu, "seconds_oracle": t_cpu,
                  "oracle_threads": min(os.cpu_count() or 4, 32), "sparse_lu_calls": po.direct_seconds[1], "sparse_lu_seconds": po.direct_seconds[0]}))
