"""Contact query with self-contact: counterpart of ``projection_query(sys, debug, self_contact)`` in
/root/reference/code/engine/geometry_self.py:290-297 (a module the reference ships but imports nowhere).

Same broad / narrow phase as ``geometry.projection_query`` on the module's own, coarser grid (``grid_h = 0.1``, :8-10) plus, for every
body listed in ``self_contact``, ``project_pair_self`` (:166-230): the body's vertices against its own triangles, skipping the
triangles a vertex belongs to and keeping only projections that fall inside a triangle.  Runs on the GPU inside the engine context
(``k_project_pair<G, true>`` in csrc/k_contact.hpp); which vertex ranges become constraints is the scene's ``contact_pairs()``.
"""
grid_h = 0.1
grid_n = int(0.2 // grid_h) * 2
grid_bound = grid_h * (grid_n - 1) / 2
max_n_particles = 100000


def configure(sys, self_contact=()):
    """switch the scene's context to this module's grid and self-contact bodies (persistent until ``geometry.configure`` style reset)"""
    ctx = sys._ensure_ctx()
    ctx.set_param("grid_h", grid_h)
    ctx.set_param("grid_extent", 0.2)
    for b in range(len(sys.body_list)):
        ctx.set_param(f"self_contact{b}", 1.0 if b in self_contact else 0.0)
    return ctx


def projection_query(sys, debug=False, self_contact=()):
    ctx = configure(sys, self_contact)
    nc = ctx.contact_detect(sys.pos.t, sys.prev_pos.t)
    sys.nc[None] = nc
    return nc
