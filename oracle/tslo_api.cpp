// ORACLE (test infrastructure only -- see oracle/README.md).
// Plain C entry points over the CPU restatement, loaded with ctypes by oracle/pyoracle.py.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
#include <cstdio>
#include <cstring>
#include <string>

#include "tslo_engine.h"

using namespace tslo;
namespace tslo {
int grad_get_loss(Grad& g, Scene& sys, const char* name, double a0, double a1, const int* rows, const double* target);   // tslo_loss.cpp
double scene_reward(Scene& sys, const Grad* g, const char* name, double a0, double a1, const int* rows, const double* target);
int& sign_mode();   // tslo_cloth.cpp
}

struct Handle {
  Scene sys;
  Grad grad;
  bool finalized = false;
};

#define S(h) (((Handle*)(h))->sys)
#define G(h) (((Handle*)(h))->grad)

extern "C" {

const char* tslo_version() { return "tsl-oracle 0.1 (restates ThinShellLab engine, fp64)"; }

void* tslo_new() { return new Handle(); }
void tslo_free(void* h) { delete (Handle*)h; }
void tslo_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
int tslo_get_max_threads() { return omp_get_max_threads(); }

void tslo_set_params(void* h, double dt, double k_contact, double eps_contact, double eps_v, double damping, int max_n_constraints,
                     int newton_cap, int plastic, int effector_cnt, const double* gravity, double mu_cloth_elastic) {
  Scene& s = S(h);
  s.dt = dt; s.h = dt; s.k_contact = k_contact; s.eps_contact = eps_contact; s.eps_v = eps_v; s.damping = damping;
  s.max_n_constraints = max_n_constraints; s.newton_cap = newton_cap; s.plastic = plastic; s.effector_cnt = effector_cnt;
  s.gravity = V3(gravity[0], gravity[1], gravity[2]); s.mu_cloth_elastic = mu_cloth_elastic;
}
void tslo_set_solver(void* h, double cg_tol, int cg_maxit) { S(h).cg_tol = cg_tol; S(h).cg_maxit = cg_maxit; }
void tslo_set_direct(void* h, Scene::direct_cb_t cb, int mode) { S(h).direct_cb = cb; S(h).direct_mode = cb ? mode : 0; }
void tslo_set_scalar(void* h, const char* name, double v) {
  Scene& s = S(h);
  std::string n(name);
  if (n == "mu_cloth_elastic") s.mu_cloth_elastic = v;
  else if (n == "mu_cloth_cloth") s.mu_cloth_cloth = v;
  else if (n == "k_contact") s.k_contact = v;
  else if (n == "eps_contact") s.eps_contact = v;
  else if (n == "damping") s.damping = v;
  else if (n == "newton_cap") s.newton_cap = (int)v;
  else if (n == "plastic") s.plastic = (int)v;
  else if (n == "grid_h") { s.grid_h = v; s.grid_n = (int)std::floor(s.grid_extent / v) * 2; s.grid_bound = v * (s.grid_n - 1) / 2; }
  else if (n == "grid_extent") { s.grid_extent = v; s.grid_n = (int)std::floor(v / s.grid_h) * 2; s.grid_bound = s.grid_h * (s.grid_n - 1) / 2; }
  else if (n.rfind("cloth", 0) == 0) {
    int ci = n[5] - '0';
    std::string f = n.substr(7);
    Cloth& c = s.cloths[ci];
    if (f == "Kb") c.Kb = v; else if (f == "Kl") c.Kl = v; else if (f == "Ka") c.Ka = v; else if (f == "k_angle") c.k_angle = v;
  } else if (n.rfind("elastic", 0) == 0) {
    int ei = n[7] - '0';
    std::string f = n.substr(9);
    Elastic& e = s.elastics[ei];
    if (f == "mu") e.mu = v; else if (f == "lam") e.lam = v;
    if (e.kind == 0) e.alpha = 1 + e.mu / e.lam;  // model_elastic_tactile.py:20-23
  }
}

int tslo_add_cloth(void* h, int N, int M, double Len, double rho, int is_square) {
  Scene& s = S(h);
  Cloth c;
  c.construct(N, s.dt, Len, rho, s.tot_NV, is_square != 0, M);
  s.tot_NV += c.NV;
  s.cloths.push_back(std::move(c));
  return (int)s.cloths.size() - 1;
}
int tslo_add_tactile(void* h, double ratio, int nv, const double* nodes, int nc, const int* tets, int ns, const int* faces) {
  Scene& s = S(h);
  Elastic e;
  e.construct_tactile(s.dt, s.tot_NV, ratio, nv, nodes, nc, tets, ns, faces);
  s.tot_NV += e.n_verts;
  s.elastics.push_back(std::move(e));
  return (int)s.elastics.size() - 1;
}
int tslo_add_box(void* h, double Len, int Nx, int Ny, int Nz, double density) {
  Scene& s = S(h);
  Elastic e;
  e.construct_box(s.dt, Len, s.tot_NV, Nx, Ny, Nz, density);
  s.tot_NV += e.n_verts;
  s.elastics.push_back(std::move(e));
  return (int)s.elastics.size() - 1;
}
int tslo_add_loaded(void* h, double density, int nv, const double* nodes, int nc, const int* tets, int ns, const int* faces) {
  Scene& s = S(h);
  Elastic e;
  e.construct_loaded(s.dt, s.tot_NV, density, nv, nodes, nc, tets, ns, faces);
  s.tot_NV += e.n_verts;
  s.elastics.push_back(std::move(e));
  return (int)s.elastics.size() - 1;
}
// mode 0: Cloth.init (flat), 1: Cloth.init_fold
void tslo_cloth_init(void* h, int ci, int mode, double ox, double oy, double oz, int curv) {
  Cloth& c = S(h).cloths[ci];
  if (mode == 0) c.init(ox, oy, oz); else if (mode == 2) { c.init(ox, oy, oz); c.init_ref_angle_bridge(); } else c.init_fold(ox, oy, oz, curv);
}
void tslo_cloth_init_mesh(void* h, int ci) { S(h).cloths[ci].init_mesh(); }
void tslo_elastic_init(void* h, int ei, double ox, double oy, double oz, int flip) { S(h).elastics[ei].init(ox, oy, oz, flip); }
void tslo_elastic_init_arch(void* h, int ei, double ox, double oy, double oz, double arch) { S(h).elastics[ei].arch = arch; S(h).elastics[ei].init(ox, oy, oz, 0); }
void tslo_finalize(void* h) { S(h).finalize(); S(h).init_property(); ((Handle*)h)->finalized = true; }
void tslo_init_property(void* h) { S(h).init_property(); }
void tslo_add_pair(void* h, int b_idx, int v_start, int v_end, int mu_is_param, double mu) { S(h).pairs.push_back(PairSpec{b_idx, v_start, v_end, mu_is_param, mu}); }
void tslo_gripper_init(void* h, int paired, int n_part, const double* pos_array) {
  Scene& s = S(h);
  const Elastic& e1 = s.elastics[1];
  s.gripper.construct(paired, e1.n_verts, e1.frozen_cnt, e1.surf_point, n_part);
  s.gripper.init(s, pos_array);
  s.has_gripper = 1;
}
void tslo_gripper_reinit(void* h, const double* pos_array) { S(h).gripper.init(S(h), pos_array); }
void tslo_gripper_update_all(void* h) {
  Scene& s = S(h);
  s.gripper.get_rotmat(); s.gripper.get_vert_pos(); s.gripper.update_all(s);
  for (size_t j = 1; j < s.elastics.size(); j++) {
    auto& e = s.elastics[j];
    if (e.kind != 0) continue;
    for (int i = 0; i < e.n_verts; i++) s.pos[e.offset + i] = e.F_x[i];
  }
}

// per-body gravity override (scene-specific init_property, e.g. Scene_lifting.py:87-103)
void tslo_set_body_gravity(void* h, int is_elastic, int idx, const double* g) {
  Scene& s = S(h);
  V3 gv(g[0], g[1], g[2]);
  if (is_elastic) s.elastics[idx].gravity = gv; else s.cloths[idx].gravity = gv;
}

// state sync helpers
void tslo_pushup_all(void* h) { S(h).pushup_all(); }
void tslo_push_down_all(void* h) { S(h).push_down_pos(); S(h).push_down_vel(); S(h).push_down_prev(); }
void tslo_clear_proj(void* h) { std::fill(S(h).proj_flag.begin(), S(h).proj_flag.end(), 0); }

// engine calls
void tslo_compute_normal_dir(void* h) { for (auto& c : S(h).cloths) c.compute_normal_dir(); }
double tslo_compute_energy(void* h) { S(h).compute_energy(); return S(h).E; }
void tslo_newton_step_init(void* h) { S(h).newton_step_init(); }
void tslo_compute_residual_and_Hessian(void* h, int spd) { S(h).compute_residual_and_Hessian(spd); }
void tslo_compute_Hessian(void* h, int spd) { S(h).compute_Hessian(spd); }
void tslo_clear_H(void* h) { S(h).H.clear_all(); }
int tslo_solve(void* h, const double* b, double* x) { return S(h).solve(b, x); }
double tslo_newton_step(void* h, double* alpha) { return S(h).newton_step(alpha); }
void tslo_time_step(void* h) { S(h).time_step(); }
void tslo_timestep_init(void* h) { S(h).timestep_init(); }
void tslo_timestep_finish(void* h) { S(h).timestep_finish(); }
void tslo_calc_vn(void* h) { S(h).calc_vn(); }
void tslo_projection_query(void* h) { S(h).projection_query(); }
void tslo_set_self_contact(void* h, int body, int on) {
  Scene& s = S(h);
  if ((int)s.self_contact.size() < (int)s.body_list.size()) s.self_contact.assign(s.body_list.size(), 0);
  if (body >= 0 && body < (int)s.self_contact.size()) s.self_contact[body] = on;
}
void tslo_contact_analysis(void* h) { S(h).contact_analysis(); }
int tslo_nc(void* h) { return S(h).nc; }
void tslo_action(void* h, const double* dpos, const double* drot) { S(h).action(dpos, drot); }
void tslo_action_dist(void* h, const double* dpos, const double* drot, const double* ddis) { S(h).action_dist(dpos, drot, ddis); }
void tslo_update_ref_angle(void* h) { for (auto& c : S(h).cloths) c.update_ref_angle(); }
void tslo_prepare_bending(void* h) { for (auto& c : S(h).cloths) { c.compute_normal_dir(); c.prepare_bending(); } }

// adjoint
void tslo_grad_new(void* h, int T, int n_parts) { G(h).construct(S(h), T, n_parts); }
void tslo_grad_reset(void* h) { G(h).reset(); }
void tslo_grad_copy_pos(void* h, int step) { G(h).copy_pos(S(h), step); }
void tslo_grad_transfer(void* h, int step) { G(h).transfer_grad(step, S(h)); }
// analytic_grad_system.Grad: mode switch, flags, accumulated parameter gradients (kb, mu, lam)
void tslo_grad_system(void* h, int system_mode, int count_kb, int count_mu_lam, int count_friction) {
  G(h).system_mode = system_mode; G(h).count_kb_grad = count_kb; G(h).count_mu_lam_grad = count_mu_lam; G(h).count_friction_grad = count_friction;
}
void tslo_grad_params(void* h, double* out, int reset) {
  out[0] = G(h).grad_kb; out[1] = G(h).grad_mu; out[2] = G(h).grad_lam; out[3] = G(h).grad_friction_coef;
  if (reset) { G(h).grad_kb = 0; G(h).grad_mu = 0; G(h).grad_lam = 0; G(h).grad_friction_coef = 0; }
}
void tslo_get_paramters_grad(void* h) { S(h).get_paramters_grad(); }
// BaseScene.gather_force after Elastic.get_force of the effector pads (BaseScene.py:1541-1549, :1566-1570); out: (effector_cnt - 1) x 3
void tslo_gather_force(void* h, double* out) {
  Scene& s = S(h);
  for (int j = 1; j < s.effector_cnt; j++) {
    Elastic& e = s.elastics[j];
    e.get_force();
    V3 t;
    for (int i = 0; i < e.n_verts; i++)
      if (e.is_bottom(i) || e.is_inner_circle(i)) t += e.F_f[i];
    for (int k = 0; k < 3; k++) out[(j - 1) * 3 + k] = t[k];
  }
}
// BaseScene.get_observation_kernel (BaseScene.py:1586-1619) with n_obs_cloth = 4, n_obs_elastic = 16 (:184-188)
void tslo_observation(void* h, double* obs) {
  Scene& s = S(h);
  const int no = 4, ne = 16;
  const int ns = s.cloths[0].N / 4, ms = s.cloths[0].M / 4, cN = s.cloths[0].N;
  const int ccnt = (int)s.cloths.size(), ecnt = (int)s.elastics.size();
  for (int j = 0; j < no; j++)
    for (int k = 0; k < no; k++)
      for (int i = 0; i < ccnt; i++) {
        int xx = i * no * no + j * no + k;
        int jj = ns / 2 + j * ns, kk = ms / 2 + k * ms;
        const int q = jj * cN + kk;  // runs past NV on non-square cloths (unchecked read in the reference): zeros
        for (int d = 0; d < 3; d++) { obs[xx * 6 + d] = q < s.cloths[i].NV ? s.cloths[i].pos[q][d] : 0.0; obs[xx * 6 + 3 + d] = q < s.cloths[i].NV ? s.cloths[i].vel[q][d] : 0.0; }
      }
  for (int j = 0; j < ne; j++)
    for (int i = 0; i < ecnt; i++) {
      int xx = no * no * ccnt + i * ne + j;
      int nv = s.elastics[i].n_verts;
      int ii = (nv / ne) * j - 1;
      if (ii < 0) ii += nv;  // Taichi / numpy negative index: the last vertex
      for (int d = 0; d < 3; d++) { obs[xx * 6 + d] = s.elastics[i].F_x[ii][d]; obs[xx * 6 + 3 + d] = s.elastics[i].F_v[ii][d]; }
    }
  int base = (no * no * ccnt + ecnt * ne) * 6;
  for (int j = 0; j < s.gripper.n_part; j++) {
    for (int d = 0; d < 3; d++) obs[base + j * 7 + d] = s.gripper.pos[j][d];
    for (int d = 0; d < 4; d++) obs[base + j * 7 + 3 + d] = s.gripper.rot[j * 4 + d];
  }
}

// stats: [newton, cg, ls, solves, refine, last_solve_flag, H.missing]
void tslo_stats(void* h, long* out, int reset) {
  Scene& s = S(h);
  out[0] = s.stat_newton; out[1] = s.stat_cg; out[2] = s.stat_ls; out[3] = s.stat_solves; out[4] = s.stat_refine;
  out[5] = s.last_solve_flag; out[6] = s.H.missing;
  if (reset) s.stat_newton = s.stat_cg = s.stat_ls = s.stat_solves = s.stat_refine = 0;
}

void tslo_set_spd_mode(int m) { spd_mode() = m; }
void tslo_set_sign_mode(int m) { sign_mode() = m; }   // 1: literal `n2 . e < 0` of model_fold_offset.py:116,135,144; 0: |n2 . e| <= 1e-10 |e| counts as zero

// SPD projections for unit tests
int tslo_spd_project(double* A, int n, int K) {
  double T[81], Q[81];
  return spd_project(A, T, Q, n, n, K);
}
void tslo_spd_project_jacobi(double* A, int n) { spd_project_jacobi(A, n, n); }
void tslo_spd_project_2d(double* A) {
  double hm[2][2] = {{A[0], A[1]}, {A[2], A[3]}};
  spd_project_2d(hm);
  A[0] = hm[0][0]; A[1] = hm[0][1]; A[2] = hm[1][0]; A[3] = hm[1][1];
}

// BSR export
int tslo_bsr_nnzb(void* h) { return (int)S(h).H.col.size(); }

// generic array view: returns pointer, element count and type code (0 f64, 1 i32, 2 f32)
void* tslo_array(void* h, const char* name, long* count, int* type) {
  Scene& s = S(h);
  Grad& g = G(h);
  std::string n(name);
  *type = 0;
#define RET_V3(vec) do { *count = (long)(vec).size() * 3; return (void*)(vec).data(); } while (0)
#define RET_D(vec) do { *count = (long)(vec).size(); return (void*)(vec).data(); } while (0)
#define RET_I(vec, k) do { *count = (long)(vec).size() * (k); *type = 1; return (void*)(vec).data(); } while (0)
  if (n == "pos") RET_V3(s.pos);
  if (n == "vel") RET_V3(s.vel);
  if (n == "prev_pos") RET_V3(s.prev_pos);
  if (n == "x1") RET_V3(s.x1);
  if (n == "vn") RET_V3(s.vn);
  if (n == "ext_force") RET_V3(s.ext_force);
  if (n == "d_ka") RET_V3(s.d_ka);
  if (n == "d_kl") RET_V3(s.d_kl);
  if (n == "d_kb") RET_V3(s.d_kb);
  if (n == "d_mu") RET_V3(s.d_mu);
  if (n == "mass") RET_D(s.mass);
  if (n == "F") RET_D(s.F);
  if (n == "frozen") RET_I(s.frozen, 1);
  if (n == "border_flag") RET_I(s.border_flag, 1);
  if (n == "faces") RET_I(s.faces, 3);
  if (n == "proj_flag") RET_I(s.proj_flag, 1);
  if (n == "proj_dir") RET_I(s.proj_dir, 1);
  if (n == "proj_idx") RET_I(s.proj_idx, 3);
  if (n == "proj_w") RET_V3(s.proj_w);
  if (n == "const_idx") RET_I(s.const_idx, 4);
  if (n == "const_w") RET_V3(s.const_w);
  if (n == "const_n") RET_V3(s.const_n);
  if (n == "const_dx0") RET_V3(s.const_dx0);
  if (n == "const_k") RET_D(s.const_k);
  if (n == "const_mu") RET_D(s.const_mu);
  if (n == "const_T") RET_D(s.const_T);
  if (n == "tmp_z_frozen") RET_D(s.tmp_z_frozen);
  if (n == "tmp_z_not_frozen") RET_D(s.tmp_z_not_frozen);
  if (n == "H.row_ptr") RET_I(s.H.row_ptr, 1);
  if (n == "H.col") RET_I(s.H.col, 1);
  if (n == "H.vals") RET_D(s.H.vals);
  if (n == "gripper.pos") RET_V3(s.gripper.pos);
  if (n == "gripper.rot") RET_D(s.gripper.rot);
  if (n == "gripper.bound_idx") RET_I(s.gripper.bound_idx, 1);
  if (n == "gripper.F_x") RET_V3(s.gripper.F_x);
  if (n == "gripper.F_x_lower") RET_V3(s.gripper.F_x_lower);
  if (n == "gripper.rotmat") { *count = (long)s.gripper.rotmat.size(); *type = 2; return (void*)s.gripper.rotmat.data(); }
  if (n == "grad.pos_buffer") RET_D(g.pos_buffer);
  if (n == "grad.pos_grad") RET_D(g.pos_grad);
  if (n == "grad.ref_angle_buffer") RET_D(g.ref_angle_buffer);
  if (n == "grad.angleref_grad") RET_D(g.angleref_grad);
  if (n == "grad.gripper_grad") RET_D(g.gripper_grad);
  if (n == "grad.x_hat_grad") RET_D(g.x_hat_grad);
  if (n == "grad.gripper_pos_buffer") RET_D(g.gripper_pos_buffer);
  if (n == "grad.gripper_rot_buffer") RET_D(g.gripper_rot_buffer);
  if (n.rfind("cloth", 0) == 0 && n.size() > 7) {
    int ci = n[5] - '0';
    if (ci < 0 || ci >= (int)s.cloths.size()) return nullptr;
    Cloth& c = s.cloths[ci];
    std::string f = n.substr(7);
    if (f == "pos") RET_V3(c.pos);
    if (f == "prev_pos") RET_V3(c.prev_pos);
    if (f == "vel") RET_V3(c.vel);
    if (f == "F_b") RET_V3(c.F_b);
    if (f == "norm_dir") RET_V3(c.norm_dir);
    if (f == "manipulate_force") RET_V3(c.manipulate_force);
    if (f == "f2v") RET_I(c.f2v, 3);
    if (f == "counter_face") RET_I(c.counter_face, 3);
    if (f == "counter_point") RET_I(c.counter_point, 3);
    if (f == "ref_angle") { *count = (long)c.ref_angle.size() * 3; return (void*)c.ref_angle.data(); }
    if (f == "heights") { *count = (long)c.heights.size() * 3; return (void*)c.heights.data(); }
    if (f == "angle") { *count = (long)c.angle.size() * 3; return (void*)c.angle.data(); }
    if (f == "c_i") { *count = (long)c.c_i.size() * 3; return (void*)c.c_i.data(); }
    if (f == "d_i") { *count = (long)c.d_i.size() * 3; return (void*)c.d_i.data(); }
    if (f == "V") RET_D(c.V);
    if (f == "l_i") { *count = (long)c.l_i.size() * 3; return (void*)c.l_i.data(); }
  }
  if (n.rfind("elastic", 0) == 0 && n.size() > 9) {
    int ei = n[7] - '0';
    if (ei < 0 || ei >= (int)s.elastics.size()) return nullptr;
    Elastic& e = s.elastics[ei];
    std::string f = n.substr(9);
    if (f == "F_x") RET_V3(e.F_x);
    if (f == "F_x_prev") RET_V3(e.F_x_prev);
    if (f == "F_v") RET_V3(e.F_v);
    if (f == "F_ox") RET_V3(e.F_ox);
    if (f == "F_f") RET_V3(e.F_f);
    if (f == "F_b") RET_V3(e.F_b);
    if (f == "ext_force") RET_V3(e.ext_force);
    if (f == "F_m") RET_D(e.F_m);
    if (f == "F_W") RET_D(e.F_W);
    if (f == "F_B") { *count = (long)e.F_B.size() * 9; return (void*)e.F_B.data(); }
    if (f == "F_vertices") RET_I(e.F_vertices, 4);
    if (f == "f2v") RET_I(e.f2v, 3);
    if (f == "is_surface") RET_I(e.is_surface, 1);
  }
  return nullptr;
}

// integer scene facts: tot_NV, tot_NF, n bodies; per cloth/elastic sizes
long tslo_int(void* h, const char* name) {
  Scene& s = S(h);
  std::string n(name);
  if (n == "tot_NV") return s.tot_NV;
  if (n == "tot_NF") return s.tot_NF;
  if (n == "n_bodies") return (long)s.body_list.size();
  if (n == "nc") return s.nc;
  if (n.rfind("cloth", 0) == 0) {
    Cloth& c = s.cloths[n[5] - '0'];
    std::string f = n.substr(7);
    if (f == "NV") return c.NV; if (f == "NF") return c.NF; if (f == "N") return c.N; if (f == "M") return c.M;
    if (f == "offset") return c.offset; if (f == "offset_faces") return c.offset_faces;
  }
  if (n.rfind("elastic", 0) == 0) {
    Elastic& e = s.elastics[n[7] - '0'];
    std::string f = n.substr(9);
    if (f == "n_verts") return e.n_verts; if (f == "n_cells") return e.n_cells; if (f == "n_surfaces") return e.n_surfaces;
    if (f == "offset") return e.offset; if (f == "offset_faces") return e.offset_faces;
    if (f == "frozen_cnt") return e.frozen_cnt; if (f == "surf_point") return e.surf_point;
  }
  return -1;
}
double tslo_double(void* h, const char* name) {
  Scene& s = S(h);
  std::string n(name);
  if (n == "E") return s.E;
  if (n.rfind("cloth", 0) == 0) {
    Cloth& c = s.cloths[n[5] - '0'];
    std::string f = n.substr(7);
    if (f == "U") return c.U; if (f == "mass") return c.mass; if (f == "dx") return c.dx;
  }
  if (n.rfind("elastic", 0) == 0) {
    Elastic& e = s.elastics[n[7] - '0'];
    std::string f = n.substr(9);
    if (f == "U") return e.U; if (f == "mu") return e.mu; if (f == "lam") return e.lam; if (f == "alpha") return e.alpha;
  }
  return 0.0 / 0.0;
}

// loss seeds (analytic_grad_single.py:259-471) and rewards (task_scene/Scene_*.py compute_reward*), tslo_loss.cpp
int tslo_grad_loss(void* h, const char* name, double a0, double a1, const int* rows, const double* target) { return grad_get_loss(G(h), S(h), name, a0, a1, rows, target); }
double tslo_reward(void* h, const char* name, double a0, double a1, const int* rows, const double* target) { return scene_reward(S(h), &G(h), name, a0, a1, rows, target); }

}  // extern "C"
