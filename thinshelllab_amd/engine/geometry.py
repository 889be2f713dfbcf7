"""Contact query entry point: counterpart of ``projection_query``
(/root/reference/code/engine/geometry.py:223-229).

The scripts pass this function to ``Scene.time_step`` / ``Grad.transfer_grad`` as ``f_contact``.  The
broad phase (uniform 3 mm grid, geometry.py:8-19, :96-163) and the closest-triangle narrow phase
(:23-87, :165-221) run on the GPU inside the engine context (csrc/k_contact.hpp); calling the function
directly performs vertex normals + projection + constraint build at the scene's current state.
"""
grid_h = 0.003
grid_n = int(0.2 // grid_h) * 2
grid_bound = grid_h * (grid_n - 1) / 2
max_n_particles = 100000


def projection_query(sys, debug=False):
    ctx = sys._ensure_ctx()
    nc = ctx.contact_detect(sys.pos.t, sys.prev_pos.t)
    sys.nc[None] = nc
    return nc
