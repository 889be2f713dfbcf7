"""Probe (not part of the product): host time of the symbolic plan per step of the bench scene (verbose 2 lines of the engine)."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from thinshelllab_amd.engine.geometry import projection_query
args = types.SimpleNamespace(workload="cfg4", grid=224, cloth_size=None, idle=0)
s = bench.build_scene(args, 0)
ctx = s._ensure_ctx(); ctx.set_param("direct", 1); ctx.set_param("verbose", 2)
for f in range(1, int(sys.argv[1]) + 1 if len(sys.argv) > 1 else 9):
    s.action(f, *bench._drive(s.gripper.n_part, s._bench_gs, 0, f, 0))
    s.time_step(projection_query, f)
