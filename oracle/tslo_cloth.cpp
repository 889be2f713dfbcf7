// ORACLE (test infrastructure only -- see oracle/README.md).
// CPU restatement of the thin-shell model /root/reference/code/engine/model_fold_offset.py.
// Every function cites the lines it follows; quirks listed in SURVEY.md section 8(a)/App. A are
// reproduced literally (counter_face table errors, the factor 2 in compute_area_dxy_p12, the
// c_i[l]/mat_N[l*3+..] indexing in compute_Hessian_bending, the +sign in compute_l_dxy).
#include "tslo_engine.h"

namespace tslo {

// Sign test of Cloth.compute_angle / judge_angle / compute_bending_energy (model_fold_offset.py:116,:135,:144):
//   norm_dir[i2].dot(pos[f2v[i1][(l+1)%2]] - pos[f2v[i1][l]]) < 0
// For the table slots init_mesh fills wrongly (odd-cell slot 2 -> face k-2) the edge vector lies exactly in the
// plane of face i2, so the dot product is 0 in exact arithmetic and its floating-point sign is rounding noise
// (it decides judge_angle -> sign of mat_M and c_i of those slots).  The restatement evaluates the test as in
// exact arithmetic: values within 1e-10 |e| of zero count as zero (not negative).  Documented in DESIGN.md.
// sign_mode() = 1 switches to the LITERAL test `< 0` of the reference (tslo_set_sign_mode; tests measure what the tolerance changes).
int& sign_mode() { static int m = 0; return m; }
static inline bool sign_test_negative(const V3& n2, const V3& e) { return sign_mode() == 1 ? dot(n2, e) < 0 : dot(n2, e) < -1e-10 * norm(e); }

// model_fold_offset.py:11-33
void Cloth::construct(int N_, double dt_, double Len, double rho_, int offset_, bool is_square, int M_) {
  N = N_;
  M = is_square ? N_ : M_;
  dt = dt_;
  dx = Len / N;
  NF = 2 * N * M;
  NV = (N + 1) * (M + 1);
  offset = offset_;
  rho = rho_;
  grid_len = dx;
  mass = rho * (dx * dx);
  pos.assign(NV, V3()); prev_pos.assign(NV, V3()); vel.assign(NV, V3()); F_b.assign(NV, V3());
  manipulate_force.assign(NV, V3());
  f2v.assign(NF, I3{{0, 0, 0}}); counter_face.assign(NF, I3{{0, 0, 0}}); counter_point.assign(NF, I3{{0, 0, 0}});
  V.assign(NF, 0.0); l_i.assign(NF, D3{{0, 0, 0}}); f_deri.assign((size_t)NF * 3, V3());
  norm_dir.assign(NF, V3());
  heights.assign(NF, D3{{0, 0, 0}}); angle.assign(NF, D3{{0, 0, 0}}); c_i.assign(NF, D3{{0, 0, 0}});
  d_i.assign(NF, D3{{0, 0, 0}}); ref_angle.assign(NF, D3{{0, 0, 0}});
  mat_M.assign((size_t)NF * 3, M3()); mat_N.assign((size_t)NF * 3, M3());
  H_me.assign((size_t)NF * 3 * 9, 0.0); T_me.assign((size_t)NF * 3 * 9, 0.0); Q_me.assign((size_t)NF * 3 * 9, 0.0);
}

// model_fold_offset.py:928-1018 (Taichi fields are zero-initialised; entries the kernel never
// writes keep 0 -- e.g. counter_face[k][0] of odd cells)
void Cloth::init_mesh() {
  for (int i = 0; i < N; i++)
    for (int j = 0; j < M; j++) {
      int k = (i * M + j) * 2;
      int a = i * (M + 1) + j;
      int b = a + 1;
      int c = a + M + 2;
      int d = a + M + 1;
      if ((i + j) % 2 == 0) {
        f2v[k + 0] = I3{{c, b, a}};
        f2v[k + 1] = I3{{a, d, c}};
      } else {
        f2v[k + 0] = I3{{b, a, d}};
        f2v[k + 1] = I3{{d, c, b}};
      }
      if ((i + j) % 2 == 0) {
        if (i > 0) { counter_face[k][0] = ((i - 1) * M + j) * 2 + 1; counter_point[k][0] = 2; }
        else counter_face[k][0] = -1;
        if (j < M - 1) { counter_face[k][2] = k + 2; counter_point[k][2] = 0; }
        else counter_face[k][2] = -1;
        if (i < N - 1) { counter_face[k + 1][0] = ((i + 1) * M + j) * 2; counter_point[k + 1][0] = 2; }
        else counter_face[k + 1][0] = -1;
        if (j > 0) { counter_face[k + 1][2] = k - 2; counter_point[k + 1][2] = 0; }
        else counter_face[k + 1][2] = -1;
        counter_face[k][1] = k + 1; counter_point[k][1] = 1;
        counter_face[k + 1][1] = k; counter_point[k + 1][1] = 1;
      } else {
        if (i > 0) { counter_face[k][2] = ((i - 1) * M + j) * 2 + 1; counter_point[k][2] = 0; }
        else counter_face[k][2] = -1;
        if (j < M - 1) { counter_face[k + 1][0] = k + 3; counter_point[k + 1][0] = 2; }
        else counter_face[k + 1][0] = -1;
        if (i < N - 1) { counter_face[k + 1][2] = ((i + 1) * M + j) * 2; counter_point[k + 1][2] = 0; }
        else counter_face[k + 1][2] = -1;
        if (j > 0) { counter_face[k][2] = k - 2; counter_point[k][2] = 2; }
        else counter_face[k][2] = -1;
        counter_face[k][1] = k + 1; counter_point[k][1] = 1;
        counter_face[k + 1][1] = k; counter_point[k + 1][1] = 1;
      }
    }
}

// model_fold_offset.py:825-838
void Cloth::init_pos_offset(double ox, double oy, double oz) {
  for (auto& r : ref_angle) r = D3{{0, 0, 0}};
  for (int i = 0; i <= N; i++)
    for (int j = 0; j <= M; j++) {
      int k = i * (M + 1) + j;
      pos[k] = V3(i * grid_len + ox, j * grid_len + oy, oz);
      vel[k] = V3(0, 0, 0);
    }
  for (int i = 0; i < NF; i++) {
    V[i] = grid_len * grid_len * 0.5;
    l_i[i][0] = grid_len; l_i[i][1] = grid_len; l_i[i][2] = grid_len * std::sqrt(2.0);
  }
}

// model_fold_offset.py:840-868
void Cloth::init_pos_offset_fold(double ox, double oy, double oz, int half_curv_num) {
  for (auto& r : ref_angle) r = D3{{0, 0, 0}};
  double r = grid_len;
  if (half_curv_num != 2) r = grid_len * (half_curv_num * 2 - 1) / 3.1415;
  int L = 7 - half_curv_num + 1;
  int R = 7 + half_curv_num;
  for (int i = 0; i <= N; i++)
    for (int j = 0; j <= M; j++) {
      int k = i * (M + 1) + j;
      if (i <= L) { pos[k] = V3((15 - i) * grid_len + ox, j * grid_len + oy, oz + 2 * r); vel[k] = V3(); }
      if (i >= L + 1 && i <= R - 1) {
        double x = (15 - L) * grid_len;
        double ang = (double)(i - L) / (half_curv_num * 2 - 1) * 3.1415;
        pos[k] = V3(x - r * std::sin(ang) + ox, j * grid_len + oy, oz + r * (1 + std::cos(ang)));
        vel[k] = V3();
      }
      if (i >= R) { pos[k] = V3(i * grid_len + ox, j * grid_len + oy, oz); vel[k] = V3(); }
    }
  for (int i = 0; i < NF; i++) {
    V[i] = grid_len * grid_len * 0.5;
    l_i[i][0] = grid_len; l_i[i][1] = grid_len; l_i[i][2] = grid_len * std::sqrt(2.0);
  }
}

// model_fold_offset.py:126-138
double Cloth::compute_angle(int i1, int i2, int l) const {
  double theta = 0.0;
  if (i2 != -1) {
    double cos_theta = dot(norm_dir[i1], norm_dir[i2]);
    if (cos_theta < 0.999999) theta = std::acos(cos_theta);
    else theta = 2 * std::sqrt(std::fabs(1.0 - cos_theta)) / std::sqrt(1 + cos_theta);
    if (sign_test_negative(norm_dir[i2], pos[f2v[i1][(l + 1) % 2]] - pos[f2v[i1][l]])) theta = -theta;
  }
  return theta;
}

// model_fold_offset.py:140-147
bool Cloth::judge_angle(int i1, int i2, int l) const {
  bool ret = true;
  if (i2 != -1)
    if (sign_test_negative(norm_dir[i2], pos[f2v[i1][(l + 1) % 2]] - pos[f2v[i1][l]])) ret = false;
  return ret;
}

// model_fold_offset.py:169-174
void Cloth::compute_normal_dir() {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < NF; i++) {
    V3 a = pos[f2v[i][0]], b = pos[f2v[i][1]], c = pos[f2v[i][2]];
    norm_dir[i] = normalized(cross(b - a, c - b));
  }
}

// model_fold_offset.py:787-797
void Cloth::init_ref_angle() {
  for (int i = 0; i < NF; i++)
    for (int l = 0; l < 3; l++)
      if (counter_face[i][l] > i) {
        double theta = compute_angle(i, counter_face[i][l], l);
        double theta_dis = theta - ref_angle[i][l];
        double abs_dis = std::fabs(theta_dis);
        if (abs_dis > k_angle) ref_angle[i][l] += (abs_dis - k_angle) * theta_dis / abs_dis;
      }
}

// model_fold_offset.py:812-822
void Cloth::init_ref_angle_bridge() {
  for (int i = 0; i < NF; i++)
    for (int l = 0; l < 3; l++)
      if (counter_face[i][l] > i) {
        int p = f2v[counter_face[i][l]][counter_point[i][l]];
        int r1 = f2v[i][l] / (M + 1), r2 = p / (M + 1);
        if (r1 == 4 && r2 == 6) ref_angle[i][l] = 1.7;
        if (r1 == 9 && r2 == 11) ref_angle[i][l] = 1.7;
      }
}

// model_fold_offset.py:176-185
void Cloth::update_ref_angle() {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < NF; i++)
    for (int l = 0; l < 3; l++)
      if (counter_face[i][l] > i) {
        double theta = compute_angle(i, counter_face[i][l], l);
        double theta_dis = theta - ref_angle[i][l];
        double abs_dis = std::fabs(theta_dis);
        if (abs_dis > k_angle) ref_angle[i][l] += (abs_dis - k_angle) * theta_dis / abs_dis;
      }
}

// model_fold_offset.py:190-218 with :108-120, :149-167
void Cloth::compute_energy() {
  double Usum = 0;
#pragma omp parallel for reduction(+ : Usum) schedule(static)
  for (int i = 0; i < NV; i++) {
    Usum += -dot(manipulate_force[i], pos[i]);
    Usum += -dot(pos[i], gravity) * mass;
    V3 X = pos[i] - prev_pos[i] - vel[i] * dt;
    Usum += 0.5 * mass * dot(X, X) / (dt * dt);
  }
  auto membrane_energy = [&](int i) {
    V3 a = pos[f2v[i][0]], b = pos[f2v[i][1]], c = pos[f2v[i][2]];
    double area = norm(cross(b - a, c - a)) * 0.5;
    return Ka * (1 - area / V[i]) * (1 - area / V[i]) * V[i];
  };
  auto membrane_energy_edge = [&](int p1, int p2, int edge_idx) {
    double l = norm(pos[p2] - pos[p1]);
    double base_len = (edge_idx == 2) ? dx * std::sqrt(2.0) : dx;
    return Kl * (1 - l / base_len) * (1 - l / base_len) * base_len;
  };
#pragma omp parallel for reduction(+ : Usum) schedule(static)
  for (int ij = 0; ij < N * M; ij++) {
    int k = ij * 2;
    Usum += membrane_energy(k);
    Usum += membrane_energy(k + 1);
    Usum += membrane_energy_edge(f2v[k][1], f2v[k][2], 1);
    Usum += membrane_energy_edge(f2v[k + 1][0], f2v[k + 1][1], 0);
    Usum += membrane_energy_edge(f2v[k][0], f2v[k][2], 2);
    Usum += membrane_energy_edge(f2v[k][0], f2v[k][2], 2);
    Usum += membrane_energy_edge(f2v[k + 1][1], f2v[k + 1][2], 1);
    Usum += membrane_energy_edge(f2v[k][0], f2v[k][1], 0);
  }
#pragma omp parallel for reduction(+ : Usum) schedule(static)
  for (int i = 0; i < NF; i++)
    for (int l = 0; l < 3; l++)
      if (counter_face[i][l] > i) {
        // compute_bending_energy (:108-120)
        int i2 = counter_face[i][l];
        double cos_theta = dot(norm_dir[i], norm_dir[i2]);
        double theta;
        if (cos_theta < 0.999999) theta = std::acos(cos_theta);
        else theta = 2 * std::sqrt(std::fabs(1.0 - cos_theta)) / std::sqrt(1 + cos_theta);
        if (sign_test_negative(norm_dir[i2], pos[f2v[i][(l + 1) % 2]] - pos[f2v[i][l]])) theta = -theta;
        double dth = theta - ref_angle[i][l];
        Usum += Kb * dth * dth * dx * dx * 1.0 / 3.0;
      }
  U = Usum;
}

// model_fold_offset.py:288-377 (literal closed forms; note the `2 *` at :369)
static inline double compute_l_dx2(const V3& p1, const V3& p2, double l_tau, int dim) {
  return (l_tau * l_tau - (p1[dim] - p2[dim]) * (p1[dim] - p2[dim])) / (l_tau * l_tau * l_tau);
}
static inline double compute_l_dxy(const V3& p1, const V3& p2, double l_tau, int dim, int d1) {
  return (p1[dim] - p2[dim]) * (p1[d1] - p2[d1]) / (l_tau * l_tau * l_tau);
}
static inline double sq(double x) { return x * x; }
static inline double cube(double x) { return x * x * x; }
static double compute_area_dx2(double area, const V3& p1, const V3& p2, const V3& p3, int dim) {
  area = area * 2.0;
  int d1 = 0;
  if (dim == 0) d1 = 1;
  int d2 = 3 - d1 - dim;
  double deri = (sq(p2[d1] - p3[d1]) + sq(p2[d2] - p3[d2])) / area -
                sq((p2[d1] - p3[d1]) * ((p2[dim] - p1[dim]) * (p3[d1] - p1[d1]) - (p3[dim] - p1[dim]) * (p2[d1] - p1[d1])) +
                   (p2[d2] - p3[d2]) * ((p2[dim] - p1[dim]) * (p3[d2] - p1[d2]) - (p3[dim] - p1[dim]) * (p2[d2] - p1[d2]))) /
                    cube(area);
  return deri * 0.5;
}
static double compute_area_dx(double area, const V3& p1, const V3& p2, const V3& p3, int dim) {
  area = area * 2.0;
  int d1 = 0;
  if (dim == 0) d1 = 1;
  int d2 = 3 - d1 - dim;
  double deri = 0.5 *
                (p1[dim] * (sq(p2[d1] - p3[d1]) + sq(p2[d2] - p3[d2])) -
                 p2[dim] * (p1[d1] * (p2[d1] - p3[d1]) - p2[d1] * p3[d1] + sq(p3[d1]) + p1[d2] * p2[d2] - p1[d2] * p3[d2] -
                            p2[d2] * p3[d2] + sq(p3[d2])) +
                 p3[dim] * (p1[d1] * (p2[d1] - p3[d1]) - sq(p2[d1]) + p2[d1] * p3[d1] + (p1[d2] - p2[d2]) * (p2[d2] - p3[d2]))) /
                area;
  return deri;
}
static double compute_area_dxy_p1(double area, const V3& p1, const V3& p2, const V3& p3, int dim, int d1) {
  area = area * 2.0;
  int d2 = 3 - d1 - dim;
  double deri = ((p3[dim] - p2[dim]) * (p2[d1] - p3[d1])) / area -
                (((p3[dim] - p2[dim]) * ((p2[dim] - p1[dim]) * (p3[d1] - p1[d1]) - (p3[dim] - p1[dim]) * (p2[d1] - p1[d1])) +
                  (p2[d2] - p3[d2]) * ((p2[d1] - p1[d1]) * (p3[d2] - p1[d2]) - (p3[d1] - p1[d1]) * (p2[d2] - p1[d2]))) *
                 ((p2[d1] - p3[d1]) * ((p2[dim] - p1[dim]) * (p3[d1] - p1[d1]) - (p3[dim] - p1[dim]) * (p2[d1] - p1[d1])) +
                  (p2[d2] - p3[d2]) * ((p2[dim] - p1[dim]) * (p3[d2] - p1[d2]) - (p3[dim] - p1[dim]) * (p2[d2] - p1[d2])))) /
                    cube(area);
  return deri * 0.5;
}
static double compute_area_dx2_p12(double area, const V3& p1, const V3& p2, const V3& p3, int dim) {
  area = area * 2.0;
  int d1 = 0;
  if (dim == 0) d1 = 1;
  int d2 = 3 - d1 - dim;
  double deri = ((p3[d1] - p1[d1]) * (p2[d1] - p3[d1]) + (p3[d2] - p1[d2]) * (p2[d2] - p3[d2])) / area -
                (((p2[d1] - p3[d1]) * ((p2[dim] - p1[dim]) * (p3[d1] - p1[d1]) - (p3[dim] - p1[dim]) * (p2[d1] - p1[d1])) +
                  (p2[d2] - p3[d2]) * ((p2[dim] - p1[dim]) * (p3[d2] - p1[d2]) - (p3[dim] - p1[dim]) * (p2[d2] - p1[d2]))) *
                 ((p3[d1] - p1[d1]) * ((p2[dim] - p1[dim]) * (p3[d1] - p1[d1]) - (p3[dim] - p1[dim]) * (p2[d1] - p1[d1])) +
                  (p3[d2] - p1[d2]) * ((p2[dim] - p1[dim]) * (p3[d2] - p1[d2]) - (p3[dim] - p1[dim]) * (p2[d2] - p1[d2])))) /
                    cube(area);
  return deri * 0.5;
}
static double compute_area_dxy_p12(double area, const V3& p1, const V3& p2, const V3& p3, int dim, int d1) {
  area = area * 2.0;
  int d2 = 3 - d1 - dim;
  double deri = (((p2[dim] - p1[dim]) * (p3[d1] - p1[d1]) - (p3[dim] - p1[dim]) * (p2[d1] - p1[d1])) + (p1[dim] - p3[dim]) * (p2[d1] - p3[d1])) / area -
                ((2 * (p1[dim] - p3[dim]) * ((p2[dim] - p1[dim]) * (p3[d1] - p1[d1]) - (p3[dim] - p1[dim]) * (p2[d1] - p1[d1])) +
                  (p3[d2] - p1[d2]) * ((p2[d1] - p1[d1]) * (p3[d2] - p1[d2]) - (p3[d1] - p1[d1]) * (p2[d2] - p1[d2]))) *
                 ((p2[d1] - p3[d1]) * ((p2[dim] - p1[dim]) * (p3[d1] - p1[d1]) - (p3[dim] - p1[dim]) * (p2[d1] - p1[d1])) +
                  (p2[d2] - p3[d2]) * ((p2[dim] - p1[dim]) * (p3[d2] - p1[d2]) - (p3[dim] - p1[dim]) * (p2[d2] - p1[d2])))) /
                    cube(area);
  return deri * 0.5;
}

// model_fold_offset.py:379-402
void Cloth::compute_bending_grad(int i1, int l, V3& a, V3& b, V3& c, V3& d) const {
  int i2 = counter_face[i1][l];
  int p11 = (l + 1) % 3;
  int p12 = (l + 2) % 3;
  int p4 = counter_point[i1][l];
  int p21 = (p4 + 1) % 3;
  if (f2v[i1][p11] != f2v[i2][p21]) p21 = (p4 + 2) % 3;
  int p22 = 3 - p21 - p4;
  a = -1.0 / heights[i1][l] * norm_dir[i1];
  d = -1.0 / heights[i2][p4] * norm_dir[i2];
  b = angle[i1][p12] / heights[i1][p11] * norm_dir[i1] + angle[i2][p22] / heights[i2][p21] * norm_dir[i2];
  c = angle[i1][p11] / heights[i1][p12] * norm_dir[i1] + angle[i2][p21] / heights[i2][p22] * norm_dir[i2];
}

// model_fold_offset.py:415-448
void Cloth::prepare_bending() {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < NF; i++) {
    for (int l = 0; l < 3; l++) {
      int pi = f2v[i][l], ai = f2v[i][(l + 1) % 3], bi = f2v[i][(l + 2) % 3];
      V3 p = pos[pi], a = pos[ai], b = pos[bi];
      V3 edge = b - a;
      V3 nd = norm_dir[i];
      if (judge_angle(i, counter_face[i][l], l)) nd = -nd;
      V3 edge_norm = cross(nd, edge);
      V3 edge1 = a - p;
      double ang = dot(edge_norm, edge1);
      if (ang > 0) edge_norm = -edge_norm;
      mat_M[i * 3 + l] = outer(nd, edge_norm);
      mat_N[i * 3 + l] = mat_M[i * 3 + l] / norm(edge);
      angle[i][l] = dot(normalized(a - p), normalized(b - p));
      heights[i][l] = std::fabs(dot(p - a, edge_norm)) / norm(edge_norm);
      if (counter_face[i][l] != -1) {
        double theta = compute_angle(i, counter_face[i][l], l);
        c_i[i][l] = compute_bending_dtheta_ref(theta, ref_angle[i][l]);
      } else c_i[i][l] = 0;
    }
    for (int l = 0; l < 3; l++)
      d_i[i][l] = c_i[i][(l + 1) % 3] * angle[i][(l + 2) % 3] + c_i[i][(l + 2) % 3] * angle[i][(l + 1) % 3] - c_i[i][l];
  }
}

static inline void atomic_add_v3(V3& dst, const V3& v) {
  for (int j = 0; j < 3; j++) {
#pragma omp atomic
    dst.v[j] += v.v[j];
  }
}

// model_fold_offset.py:639-687
void Cloth::compute_residual() {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < NV; i++) {
    F_b[i] = -mass * gravity;
    F_b[i] -= manipulate_force[i];
    F_b[i] += mass * (pos[i] - prev_pos[i] - vel[i] * dt) / (dt * dt);
  }
#pragma omp parallel for schedule(static)
  for (int i = 0; i < NF; i++) {
    V3 a = pos[f2v[i][0]], b = pos[f2v[i][1]], c = pos[f2v[i][2]];
    for (int l = 0; l < 3; l++) {
      int xx = f2v[i][l], yy = f2v[i][(l + 1) % 3];
      double base_len = l_i[i][l];
      V3 delta = pos[xx] - pos[yy];
      double l_tau = norm(delta);
      atomic_add_v3(F_b[xx], delta * compute_membrane_dl(l_tau, base_len) / l_tau);
      atomic_add_v3(F_b[yy], -delta * compute_membrane_dl(l_tau, base_len) / l_tau);
    }
    double base_area = V[i];
    double area = 0.5 * norm(cross(b - a, c - a));
    for (int l = 0; l < 3; l++)
      for (int j = 0; j < 3; j++) {
        double v = compute_membrane_darea(area, base_area) *
                   compute_area_dx(area, pos[f2v[i][l]], pos[f2v[i][(l + 1) % 3]], pos[f2v[i][(l + 2) % 3]], j);
#pragma omp atomic
        F_b[f2v[i][l]].v[j] += v;
      }
    for (int l = 0; l < 3; l++)
      if (counter_face[i][l] > i) {
        V3 ga, gb, gc, gd;
        compute_bending_grad(i, l, ga, gb, gc, gd);
        double theta = compute_angle(i, counter_face[i][l], l);
        double d_theta = compute_bending_dtheta_ref(theta, ref_angle[i][l]);
        atomic_add_v3(F_b[f2v[i][l]], d_theta * ga);
        atomic_add_v3(F_b[f2v[i][(l + 1) % 3]], d_theta * gb);
        atomic_add_v3(F_b[f2v[i][(l + 2) % 3]], d_theta * gc);
        atomic_add_v3(F_b[f2v[counter_face[i][l]][counter_point[i][l]]], d_theta * gd);
      }
  }
}

// model_fold_offset.py:1082-1127: elastic force per unit stiffness, d_k* = -(d E_* / d x) / K_*
void Cloth::compute_deri() {
  d_ka.assign(NV, V3()); d_kl.assign(NV, V3()); d_kb.assign(NV, V3());
  for (int i = 0; i < NF; i++) {
    V3 a = pos[f2v[i][0]], b = pos[f2v[i][1]], c = pos[f2v[i][2]];
    for (int l = 0; l < 3; l++) {
      int xx = f2v[i][l], yy = f2v[i][(l + 1) % 3];
      double base_len = l_i[i][l];
      V3 delta = pos[xx] - pos[yy];
      double l_tau = norm(delta);
      d_kl[xx] += -delta * compute_membrane_dl(l_tau, base_len) / l_tau;
      d_kl[yy] += delta * compute_membrane_dl(l_tau, base_len) / l_tau;
    }
    double base_area = V[i];
    double area = 0.5 * norm(cross(b - a, c - a));
    for (int l = 0; l < 3; l++)
      for (int j = 0; j < 3; j++)
        d_ka[f2v[i][l]].v[j] -= compute_membrane_darea(area, base_area) *
                                 compute_area_dx(area, pos[f2v[i][l]], pos[f2v[i][(l + 1) % 3]], pos[f2v[i][(l + 2) % 3]], j);
    for (int l = 0; l < 3; l++)
      if (counter_face[i][l] > i) {
        V3 ga, gb, gc, gd;
        compute_bending_grad(i, l, ga, gb, gc, gd);
        double theta = compute_angle(i, counter_face[i][l], l);
        double d_theta = compute_bending_dtheta_ref(theta, ref_angle[i][l]);
        d_kb[f2v[i][l]] -= d_theta * ga;
        d_kb[f2v[i][(l + 1) % 3]] -= d_theta * gb;
        d_kb[f2v[i][(l + 2) % 3]] -= d_theta * gc;
        d_kb[f2v[counter_face[i][l]][counter_point[i][l]]] -= d_theta * gd;
      }
  }
  for (int i = 0; i < NV; i++) { d_ka[i] = d_ka[i] / Ka; d_kl[i] = d_kl[i] / Kl; d_kb[i] = d_kb[i] / Kb; }
}

// model_fold_offset.py:466-524
void Cloth::compute_Hessian_me(Scene& H, int spd) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < NV; i++)
    for (int j = 0; j < 3; j++) H.H.add(3 * (i + offset) + j, 3 * (i + offset) + j, mass / (dt * dt));
#pragma omp parallel for schedule(static)
  for (int i = 0; i < NF; i++) {
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++)
        for (int l = 0; l < 3; l++) {
          int xx = f2v[i][l], yy = f2v[i][(l + 1) % 3];
          V3 delta = pos[xx] - pos[yy];
          V3 a = pos[xx], b = pos[yy];
          double l_tau = norm(delta);
          V3 dldx = delta / l_tau;
          double base_len = l_i[i][l];
          double* hm = &H_me[(size_t)(i * 3 + l) * 9];
          if (j == k)
            hm[j * 3 + k] = compute_membrane_dl(l_tau, base_len) * compute_l_dx2(a, b, l_tau, j) + compute_membrane_dl2(base_len) * dldx[j] * dldx[k];
          else
            hm[j * 3 + k] = compute_membrane_dl(l_tau, base_len) * compute_l_dxy(a, b, l_tau, j, k) + compute_membrane_dl2(base_len) * dldx[j] * dldx[k];
        }
    for (int l = 0; l < 3; l++) {
      int idx = i * 3 + l;
      double* hm = &H_me[(size_t)idx * 9];
      if (spd) spd_project(hm, &T_me[(size_t)idx * 9], &Q_me[(size_t)idx * 9], 3, 3, 10);
      int xx = f2v[i][l] + offset, yy = f2v[i][(l + 1) % 3] + offset;
      for (int j = 0; j < 3; j++)
        for (int k = 0; k < 3; k++) {
          H.add_H(xx * 3 + j, xx * 3 + k, hm[j * 3 + k]);
          H.add_H(xx * 3 + j, yy * 3 + k, -hm[j * 3 + k]);
          H.add_H(yy * 3 + j, xx * 3 + k, -hm[j * 3 + k]);
          H.add_H(yy * 3 + j, yy * 3 + k, hm[j * 3 + k]);
        }
    }
  }
}

// model_fold_offset.py:526-580
void Cloth::compute_Hessian_ma(Scene& H) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < NF; i++) {
    V3 a = pos[f2v[i][0]], b = pos[f2v[i][1]], c = pos[f2v[i][2]];
    double base_area = V[i];
    double darea2 = compute_membrane_darea2(base_area);
    double area = 0.5 * norm(cross(b - a, c - a));
    for (int l = 0; l < 3; l++)
      for (int j = 0; j < 3; j++)
        f_deri[i * 3 + l][j] = compute_area_dx(area, pos[f2v[i][l]], pos[f2v[i][(l + 1) % 3]], pos[f2v[i][(l + 2) % 3]], j);
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++)
        for (int l = 0; l < 3; l++)
          for (int m = 0; m < 3; m++) {
            int xx = f2v[i][l] + offset, yy = f2v[i][m] + offset;
            H.add_H(xx * 3 + j, yy * 3 + k, f_deri[i * 3 + l][j] * f_deri[i * 3 + m][k] * darea2);
            double da = compute_membrane_darea(area, base_area);
            if (j == k) {
              if (l == m)
                H.add_H(xx * 3 + j, yy * 3 + k, da * compute_area_dx2(area, pos[f2v[i][l]], pos[f2v[i][(l + 1) % 3]], pos[f2v[i][(l + 2) % 3]], j));
              else
                H.add_H(xx * 3 + j, yy * 3 + k, da * compute_area_dx2_p12(area, pos[f2v[i][l]], pos[f2v[i][m]], pos[f2v[i][3 - l - m]], j));
            } else {
              if (l == m)
                H.add_H(xx * 3 + j, yy * 3 + k, da * compute_area_dxy_p1(area, pos[f2v[i][l]], pos[f2v[i][(l + 1) % 3]], pos[f2v[i][(l + 2) % 3]], j, k));
              else
                H.add_H(xx * 3 + j, yy * 3 + k, da * compute_area_dxy_p12(area, pos[f2v[i][l]], pos[f2v[i][m]], pos[f2v[i][3 - l - m]], j, k));
            }
          }
  }
}

// model_fold_offset.py:582-637 (c_i[l][..] and mat_N[l*3+..] index FACE l in {0,1,2}: literal)
void Cloth::compute_Hessian_bending(Scene& H) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < NF; i++) {
    for (int l = 0; l < 3; l++)
      for (int lm = l; lm < l + 2; lm++) {
        int m = lm % 3;
        M3 H_lm;
        if (l == m) {
          int i1 = (l + 1) % 3, i2 = (l + 2) % 3;
          H_lm = 1.0 / (heights[i][l] * heights[i][m]) * (d_i[i][l] * transpose(mat_M[i * 3 + m]) + d_i[i][m] * mat_M[i * 3 + l]) -
                 c_i[l][i1] * mat_N[l * 3 + i1] - c_i[l][i2] * mat_N[l * 3 + i2];
        } else {
          int i3 = 3 - l - m;
          H_lm = 1.0 / (heights[i][l] * heights[i][m]) * (d_i[i][l] * transpose(mat_M[i * 3 + m]) + d_i[i][m] * mat_M[i * 3 + l]);
          H_lm = H_lm + c_i[l][i3] * mat_N[l * 3 + i3];
        }
        int xx = f2v[i][l] + offset, yy = f2v[i][m] + offset;
        for (int j = 0; j < 3; j++)
          for (int k = 0; k < 3; k++) {
            H.add_H(xx * 3 + j, yy * 3 + k, H_lm[j][k]);
            if (l != m) H.add_H(yy * 3 + j, xx * 3 + k, H_lm[k][j]);
          }
      }
    for (int l = 0; l < 3; l++)
      if (counter_face[i][l] > i) {
        V3 g[4];
        compute_bending_grad(i, l, g[0], g[1], g[2], g[3]);
        int pt[4] = {f2v[i][l], f2v[i][(l + 1) % 3], f2v[i][(l + 2) % 3], f2v[counter_face[i][l]][counter_point[i][l]]};
        double d2_theta = compute_bending_dtheta2();
        for (int j = 0; j < 4; j++)
          for (int k = 0; k < 4; k++) {
            int pj = pt[j] + offset, pk = pt[k] + offset;
            M3 gridmat = d2_theta * outer(g[j], g[k]);
            for (int jj = 0; jj < 3; jj++)
              for (int kk = 0; kk < 3; kk++) H.add_H(pj * 3 + jj, pk * 3 + kk, gridmat[jj][kk]);
          }
      }
  }
}

// model_fold_offset.py:1154-1168
void Cloth::ref_angle_backprop_x2a(Grad& g, int step, const double* p, int cnt) {
  for (int i = 0; i < NF; i++)
    for (int l = 0; l < 3; l++)
      if (counter_face[i][l] > i) {
        V3 a, b, c, d;
        compute_bending_grad(i, l, a, b, c, d);
        double d_ref = dtheta_ref();
        for (int j = 0; j < 3; j++) {
          g.AG(step - 1, cnt, i, l) += -p[(f2v[i][l] + offset) * 3 + j] * d_ref * a[j];
          g.AG(step - 1, cnt, i, l) += -p[(f2v[i][(l + 1) % 3] + offset) * 3 + j] * d_ref * b[j];
          g.AG(step - 1, cnt, i, l) += -p[(f2v[i][(l + 2) % 3] + offset) * 3 + j] * d_ref * c[j];
          g.AG(step - 1, cnt, i, l) += -p[(f2v[counter_face[i][l]][counter_point[i][l]] + offset) * 3 + j] * d_ref * d[j];
        }
      }
}

// model_fold_offset.py:1179-1206
void Cloth::ref_angle_backprop_a2ax(Grad& g, int step, int cnt) {
  for (int i = 0; i < NF; i++)
    for (int l = 0; l < 3; l++)
      if (counter_face[i][l] > i) {
        V3 a, b, c, d;
        compute_bending_grad(i, l, a, b, c, d);
        double theta = compute_angle(i, counter_face[i][l], l);
        g.AG(step - 1, cnt, i, l) += g.AG(step, cnt, i, l);
        double theta_dis = theta - ref_angle[i][l];
        double abs_dis = std::fabs(theta_dis);
        double sign = (abs_dis > k_angle) ? g.AG(step, cnt, i, l) : g.AG(step, cnt, i, l) * 0.1;
        int wing = f2v[counter_face[i][l]][counter_point[i][l]];
        for (int j = 0; j < 3; j++) {
          g.PG(step, f2v[i][l] + offset, j) += sign * a[j];
          g.PG(step, f2v[i][(l + 1) % 3] + offset, j) += sign * b[j];
          g.PG(step, f2v[i][(l + 2) % 3] + offset, j) += sign * c[j];
          g.PG(step, wing + offset, j) += sign * d[j];
        }
      }
}

}  // namespace tslo
