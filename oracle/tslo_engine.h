// ORACLE (test infrastructure only -- see oracle/README.md).
// Data model of the CPU restatement.  Field names follow the reference so every member can be
// checked against its origin:
//   Cloth    <- /root/reference/code/engine/model_fold_offset.py:10-107
//   Elastic  <- model_elastic_tactile.py:12-80 (kind 0) and model_elastic_offset.py:11-92 (kind 1/2)
//   Gripper  <- gripper_single.py:28-49 (paired=0) and gripper_tactile.py:10-35 (paired=1)
//   Scene    <- BaseScene.py:31-195
//   Grad     <- analytic_grad_single.py:5-26
#pragma once
#include <omp.h>

#include <string>
#include <vector>

#include "tslo_linalg.h"
#include "tslo_math.h"
#include "tslo_matrix.h"

namespace tslo {

struct Scene;
struct Grad;

struct Cloth {
  // model_fold_offset.py:11-40
  int N = 0, M = 0, NV = 0, NF = 0, offset = 0, offset_faces = 0, body_idx = 0;
  double dt = 0, dx = 0, rho = 0, mass = 0, grid_len = 0;
  double Kl = 1000.0, Ka = 1000.0, Kb = 100.0, k_angle = 3.14;
  V3 gravity = V3(0, 0, -9.8);
  double U = 0;
  std::vector<V3> pos, prev_pos, vel, F_b, manipulate_force, norm_dir;
  std::vector<I3> f2v, counter_face, counter_point;
  std::vector<double> V;        // rest area per face
  std::vector<D3> l_i;          // rest length per (face, edge slot)
  std::vector<V3> f_deri;       // NF*3
  std::vector<D3> heights, angle, c_i, d_i, ref_angle;
  std::vector<M3> mat_M, mat_N;  // NF*3
  std::vector<double> H_me;      // (NF*3) x 3 x 3   (+ projector scratch)
  std::vector<double> T_me, Q_me;

  void construct(int N_, double dt_, double Len, double rho_, int offset_, bool is_square, int M_);
  // model_fold_offset.py:928-1025 / :825-868 / :787-797
  void init_mesh();
  void init_pos_offset(double ox, double oy, double oz);
  void init_pos_offset_fold(double ox, double oy, double oz, int half_curv_num);
  void init_ref_angle();
  void init_ref_angle_bridge();  // model_fold_offset.py:812-822
  void init(double ox, double oy, double oz) { init_mesh(); init_pos_offset(ox, oy, oz); for (auto& r : ref_angle) r = D3{{0, 0, 0}}; }
  void init_fold(double ox, double oy, double oz, int curv) { init_mesh(); init_pos_offset_fold(ox, oy, oz, curv); compute_normal_dir(); init_ref_angle(); }

  double compute_angle(int i1, int i2, int l) const;
  bool judge_angle(int i1, int i2, int l) const;
  void compute_normal_dir();
  void update_ref_angle();
  void compute_energy();
  void prepare_bending();
  void compute_bending_grad(int i1, int l, V3& a, V3& b, V3& c, V3& d) const;
  void compute_residual();
  void compute_Hessian_me(Scene& H, int spd);
  void compute_Hessian_ma(Scene& H);
  void compute_Hessian_bending(Scene& H);
  void ref_angle_backprop_x2a(Grad& g, int step, const double* p, int cnt);
  void ref_angle_backprop_a2ax(Grad& g, int step, int cnt);
  // model_fold_offset.py:1082-1127: force per unit stiffness (system identification)
  std::vector<V3> d_ka, d_kl, d_kb;
  void compute_deri();

  // scalar helpers model_fold_offset.py:260-377
  double compute_membrane_dl(double l_tau, double l_base) const { return -Kl * 2.0 * (1.0 - l_tau / l_base); }
  double compute_membrane_dl2(double l_base) const { return Kl * 2.0 / l_base; }
  double compute_membrane_darea(double area, double base) const { return -Ka * 2.0 * (1.0 - area / base); }
  double compute_membrane_darea2(double base) const { return Ka * 2.0 / base; }
  double compute_bending_dtheta_ref(double theta, double ref) const { return 2.0 * Kb * (theta - ref) * dx * dx * 1.0 / 3.0; }
  double compute_bending_dtheta2() const { return 2.0 * Kb * dx * dx * 1.0 / 3.0; }
  double dtheta_ref() const { return -2.0 * Kb * dx * dx * 1.0 / 3.0; }
};

struct Elastic {
  int kind = 0;  // 0 tactile pad (stable Neo-Hookean), 1 box (5-tet cubes), 2 loaded mesh ("ball")
  double E = 0, nu = 0, mu = 0, lam = 0, alpha = 0, density = 0, dt = 0, ratio = 1, dx = 0;
  int offset = 0, offset_faces = 0, body_idx = 0;
  int n_verts = 0, n_cells = 0, n_surfaces = 0, frozen_cnt = 0, surf_point = 0;
  int n_cube[3] = {0, 0, 0};
  V3 gravity = V3(0, 0, -9.8);
  double U = 0;
  std::vector<I4> F_vertices;
  std::vector<V3> F_x, F_x_prev, F_ox, F_v, F_f, F_b, ext_force;
  std::vector<double> F_m, F_W;
  std::vector<M3> F_B;
  std::vector<I3> f2v, f2v_array;
  std::vector<int> is_surface;
  std::vector<double> H_e, T_e, Q_e;  // n_cells x 9 x 9

  // tactile: model_elastic_tactile.py:13-80, :302-326 ; box/ball: model_elastic_offset.py:12-92, :395-405
  void construct_tactile(double dt_, int offset_, double ratio_, int nv, const double* nodes, int nc, const int* tets, int ns, const int* faces);
  void construct_box(double dt_, double Len, int offset_, int Nx, int Ny, int Nz, double density_);
  void construct_loaded(double dt_, int offset_, double density_, int nv, const double* nodes, int nc, const int* tets, int ns, const int* faces);
  void init(double ox, double oy, double oz, int flip);
  double arch = 0;  // model_elastic_offset.py:253-270 init_pos_arch: z += arch sin(pi x / (Nx - 1)) before the rest matrices
  M3 Ds(const I4& verts) const { return from_cols(F_x[verts[0]] - F_x[verts[3]], F_x[verts[1]] - F_x[verts[3]], F_x[verts[2]] - F_x[verts[3]]); }
  bool is_bottom(int i) const { return F_ox[i][2] < 0.001 && is_surface[i]; }
  bool is_inner_circle(int i) const { return norm(F_ox[i]) < 0.0076 && is_surface[i]; }
  bool is_surf(int i) const { return norm(F_ox[i]) > 0.0148 && is_surface[i]; }
  void compute_energy();
  void get_force();
  void compute_residual();
  void compute_Hessian(Scene& A, int spd);
  // model_elastic_tactile.py:329-347 / model_elastic_offset.py:415-431 (the latter never clears d_mu / d_lam)
  std::vector<V3> d_mu, d_lam;
  void compute_deri();
};

struct Gripper {
  int paired = 0;  // 0: gripper_single.gripper, 1: gripper_tactile.gripper
  int n_verts = 0, n_bound = 0, n_surf = 0, n_part = 0;
  std::vector<V3> F_x, F_x_world;            // single: (cnt, n_verts) ; paired: "upper"
  std::vector<V3> F_x_lower, F_x_lower_world;  // paired only
  std::vector<int> bound_idx, surface_idx;
  std::vector<V3> pos, d_pos, d_angle;
  std::vector<double> rot;     // cnt x 4 (s,x,y,z)
  std::vector<float> rotmat;   // cnt x 9, stored f32 like the reference (gripper_single.py:48)
  void construct(int paired_, int n_verts_, int n_bound_, int n_surf_, int cnt);
  void init(Scene& sys, const double* pos_array);
  void get_rotmat();
  void get_vert_pos();
  void step_simple(const double* delta_pos, const double* delta_rot);
  void update_bound(Scene& sys);
  void update_all(Scene& sys);  // gripper_single.py:158-162
  std::vector<double> half_gripper_dist;
  void step(const double* delta_pos, const double* delta_rot, const double* delta_dis);  // gripper_tactile.py:196-218
  void gather_grad(const double* grad, Scene& sys);
};

struct Body { int v_start, v_end, f_start, f_end; };
struct PairSpec { int b_idx, v_start, v_end; int mu_is_param; double mu; };  // one contact_pair_analysis call

struct Scene {
  // BaseScene.py:31-60 / scene init_scene_parameters
  double dt = 5e-3, h = 5e-3;
  double k_contact = 1000, eps_contact = 0.001, eps_v = 0.01, damping = 1.0;
  int max_n_constraints = 100000;
  int newton_cap = 1000;        // BaseScene 1000, folding/balancing 50, lifting 15
  int plastic = 0;              // folding-style timestep_finish calls update_ref_angle (Scene_folding.py:227-231)
  int effector_cnt = -1;
  V3 gravity = V3(0, 0, -9.8);
  double mu_cloth_elastic = 1.0;
  double mu_cloth_cloth = 1.0;  // Scene_sliding.py:25-27: second live friction parameter, nc1 = constraints of the cloth-cloth pairs
  int nc1 = 0;
  std::vector<Cloth> cloths;
  std::vector<Elastic> elastics;
  Gripper gripper;
  int has_gripper = 0;
  int tot_NV = 0, tot_NF = 0;
  // BaseScene.py:69-88
  std::vector<V3> pos, vel, ext_force, vn, prev_pos, x1, x_hat, F_b;
  std::vector<double> mass, F;
  std::vector<int> frozen, border_flag;
  std::vector<I3> faces;
  std::vector<Body> body_list;
  std::vector<PairSpec> pairs;  // scene contact_analysis
  // BaseScene.py:101-134
  std::vector<int> proj_flag, proj_dir;  // bodies x tot_NV
  std::vector<I3> proj_idx;
  std::vector<V3> proj_w;
  std::vector<double> contact_force;
  int nc = 0;
  std::vector<I4> const_idx;
  std::vector<V3> const_w, const_n, const_dx0;
  std::vector<double> const_k, const_mu, const_T;  // const_T: nc x 6 (2x3)
  std::vector<double> det_H, det_G, cross_H, cross_G, d_H, d_G, projT, projQ;
  std::vector<V3> force_T, force_f;
  double E = 0;
  Bsr H;
  long H_static_cliques = 0;
  std::vector<std::vector<int>> static_cliques;
  std::vector<double> tmp_z_not_frozen, tmp_z_frozen;
  std::vector<V3> d_ka, d_kl, d_kb, d_mu, d_lam;  // BaseScene.py:160-164
  void get_paramters_grad();                      // BaseScene.py:1513-1525 (sic)
  int counting_z_frozen = 0;
  // solver controls (the reference calls cupyx spsolve, sparse_solver.py:85-105)
  double cg_tol = 1e-12;
  int cg_maxit = 20000;
  long stat_newton = 0, stat_cg = 0, stat_ls = 0, stat_solves = 0, stat_refine = 0;
  int last_solve_flag = 0;
  // sparse direct solve supplied by the test harness (scipy's SuperLU): the reference's SparseMatrix.solve IS a direct solve
  // (cupyx spsolve, sparse_solver.py:85-105).  direct_mode 0: never, 1: primary solver, 2: last resort after PCG and BiCGStab
  // (replaces the dense LU above n = 4500).  Arguments: block rows, row_ptr, col, 3x3 block values, rhs, solution; returns 0 on success.
  typedef int (*direct_cb_t)(int nb, const int* row_ptr, const int* col, const double* vals, const double* b, double* x);
  direct_cb_t direct_cb = nullptr;
  int direct_mode = 0;
  // geometry.py:8-19 (uniform grid)
  double grid_h = 0.003;
  double grid_extent = 0.2;  // half-width of the broad-phase box (geometry.py:8-19 hard-codes 0.2 m; scaled scenes enlarge it)
  int grid_n = 0;
  double grid_bound = 0;

  void finalize();           // BaseScene.__init__ tail + init_property + build pattern
  void init_property();      // BaseScene.py:361-383
  void rebuild_pattern();    // static cliques + current constraints
  void pushup_all();         // cloth/elastic pos,vel -> global
  void push_down_pos();
  void push_down_vel();
  void push_down_prev();
  // BaseScene.py:392-405
  inline void add_F(int i, double v) {
    if (!frozen[i]) {
#pragma omp atomic
      F[i] += v;
    }
  }
  inline void add_H(int i, int j, double v) {
    if (!frozen[i] && !frozen[j]) H.add(i, j, v);
    else if (counting_z_frozen && frozen[j] && !frozen[i]) {
#pragma omp atomic
      tmp_z_frozen[j] -= v * tmp_z_not_frozen[i];
    }
  }
  double f0(double x) const;
  double f1(double x) const;
  double f2(double x) const;
  void compute_energy();
  void contact_energy(int diff, int spd);
  void contact_energy_backprop(Grad& g, int step, const double* p);
  void contact_energy_backprop_friction(Grad& g, int step, const double* p);  // Scene_sliding.py:139-176
  void contact_pair_analysis(int b_idx, int v_start, int v_end, double mu);
  void contact_analysis();
  void calc_vn();
  void projection_query();
  std::vector<int> self_contact;  // geometry_self.projection_query(self_contact=[...]): bodies queried against their own triangles
  void compute_residual_and_Hessian(int spd);
  void compute_Hessian(int spd);
  void newton_step_init();
  int solve(const double* b, double* x);  // H x = b
  double newton_step(double* alpha_out);
  void timestep_init();
  void timestep_finish();
  void time_step();
  void update_vel();
  void action(const double* delta_pos, const double* delta_rot);
  void action_dist(const double* delta_pos, const double* delta_rot, const double* delta_dis);  // Scene_interact.py:176-180 (step < 5)
};

struct Grad {
  // analytic_grad_single.py:5-26
  int n_part = 0, tot_NV = 0, tot_timestep = 0, cloth_cnt = 0, NF = 0;
  double dt = 0, damping = 1.0;
  std::vector<double> pos_buffer, pos_grad;            // T x NV x 3
  std::vector<double> gripper_pos_buffer, gripper_rot_buffer;  // T x n_part x 3 / 4
  std::vector<double> ref_angle_buffer, angleref_grad;  // T x cloth_cnt x NF x 3
  std::vector<double> x_hat_grad, gripper_grad, mass, F;
  void construct(Scene& sys, int T, int n_parts);
  void reset();
  void copy_pos(Scene& sys, int step);
  void transfer_grad(int step, Scene& sys);
  // analytic_grad_system.py: same reverse step with pos_grad clamped to +-1, no gripper gradient, and the parameter gradients
  int system_mode = 0, count_kb_grad = 1, count_mu_lam_grad = 0;
  double grad_kb = 0, grad_mu = 0, grad_lam = 0, grad_friction_coef = 0;
  int count_friction_grad = 0;
  double& PG(int s, int i, int j) { return pos_grad[((size_t)s * tot_NV + i) * 3 + j]; }
  double& PB(int s, int i, int j) { return pos_buffer[((size_t)s * tot_NV + i) * 3 + j]; }
  double& AG(int s, int c, int f, int l) { return angleref_grad[(((size_t)s * cloth_cnt + c) * NF + f) * 3 + l]; }
  double& RB(int s, int c, int f, int l) { return ref_angle_buffer[(((size_t)s * cloth_cnt + c) * NF + f) * 3 + l]; }
};

}  // namespace tslo
