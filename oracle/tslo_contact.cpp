// ORACLE (test infrastructure only -- see oracle/README.md).
// CPU restatement of the contact path:
//   contact_diff.det / .cross          /root/reference/code/engine/contact_diff.py:4-130
//   BaseScene.f0/f1/f2                 BaseScene.py:453-478
//   BaseScene.contact_energy           BaseScene.py:487-598
//   BaseScene.contact_energy_backprop  BaseScene.py:682-730
//   BaseScene.contact_pair_analysis    BaseScene.py:778-816
//   BaseScene.calc_vn                  BaseScene.py:837-850
//   geometry.pt2tri / p2g / project_pair / projection_query   geometry.py:23-229
#include <algorithm>
#include <array>

#include "tslo_engine.h"

namespace tslo {

// contact_diff.py:4-25.  H: 9x9 row-major (persistent scratch; only the listed entries are written)
static double cd_det(const V3& a, const V3& b, const V3& c, int diff, double* H, double* G) {
  double d = a[0] * b[1] * c[2] + a[1] * b[2] * c[0] + a[2] * b[0] * c[1] - a[2] * b[1] * c[0] - a[1] * b[0] * c[2] - a[0] * b[2] * c[1];
  if (diff) {
    V3 grad_a(b[1] * c[2] - b[2] * c[1], b[2] * c[0] - b[0] * c[2], b[0] * c[1] - b[1] * c[0]);
    V3 grad_b(c[1] * a[2] - c[2] * a[1], c[2] * a[0] - c[0] * a[2], c[0] * a[1] - c[1] * a[0]);
    V3 grad_c(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
    for (int i = 0; i < 3; i++) {
      G[0 * 3 + i] = grad_a[i];
      G[1 * 3 + i] = grad_b[i];
      G[2 * 3 + i] = grad_c[i];
    }
    for (int i = 0; i < 3; i++) {
      int j = i < 2 ? i + 1 : 0;
      int k = i > 0 ? i - 1 : 2;
      H[(0 * 3 + i) * 9 + 1 * 3 + j] = H[(1 * 3 + j) * 9 + 0 * 3 + i] = c[k];
      H[(1 * 3 + i) * 9 + 2 * 3 + j] = H[(2 * 3 + j) * 9 + 1 * 3 + i] = a[k];
      H[(2 * 3 + i) * 9 + 0 * 3 + j] = H[(0 * 3 + j) * 9 + 2 * 3 + i] = b[k];
      H[(1 * 3 + i) * 9 + 0 * 3 + j] = H[(0 * 3 + j) * 9 + 1 * 3 + i] = -c[k];
      H[(2 * 3 + i) * 9 + 1 * 3 + j] = H[(1 * 3 + j) * 9 + 2 * 3 + i] = -a[k];
      H[(0 * 3 + i) * 9 + 2 * 3 + j] = H[(2 * 3 + j) * 9 + 0 * 3 + i] = -b[k];
    }
  }
  return d;
}

// contact_diff.py:27-130
static double cd_cross(const V3& a, const V3& b, int diff, double* H, double* G) {
  double a0 = a[0], a1 = a[1], a2 = a[2];
  double b0 = b[0], b1 = b[1], b2 = b[2];
  double c1 = a[0] * b[1] - b[0] * a[1];
  double c2 = a[0] * b[2] - b[0] * a[2];
  double c3 = a[1] * b[2] - b[1] * a[2];
  double c0 = c1 * c1 + c2 * c2 + c3 * c3;
  double c = std::sqrt(c0);
  if (diff) {
    V3 grad_a = V3(c1 * b[1] + c2 * b[2], -c1 * b[0] + c3 * b[2], -c2 * b[0] - c3 * b[1]) / c;
    V3 grad_b = V3(-c1 * a[1] - c2 * a[2], c1 * a[0] - c3 * a[2], c2 * a[0] + c3 * a[1]) / c;
    for (int i = 0; i < 3; i++) {
      G[0 * 3 + i] = grad_a[i];
      G[1 * 3 + i] = grad_b[i];
    }
    double x0 = b1 * b1;
    double x1 = b2 * b2;
    double x2 = a0 * b1;
    double x3 = a1 * b0;
    double x4 = x2 - x3;
    double x5 = a0 * b2;
    double x6 = a2 * b0;
    double x7 = x5 - x6;
    double x8 = a1 * b2;
    double x9 = a2 * b1;
    double x10 = x8 - x9;
    double x11 = x10 * x10 + x4 * x4 + x7 * x7;
    double x12 = std::pow(x11, -0.5);
    double x13 = b1 * x4 + b2 * x7;
    double x14 = std::pow(x11, -1.5);
    double x15 = b0 * x12;
    double x16 = b0 * x4 - b2 * x10;
    double x17 = -b1 * x15 + x13 * x14 * x16;
    double x18 = b0 * x7 + b1 * x10;
    double x19 = -b2 * x15 + x13 * x14 * x18;
    double x20 = a1 * b1;
    double x21 = a2 * b2;
    double x22 = a1 * x4 + a2 * x7;
    double x23 = -x12 * (x20 + x21) + x13 * x14 * x22;
    double x24 = a0 * x4 - a2 * x10;
    double x25 = x13 * x14;
    double x26 = x12 * (2.0 * x2 - x3) - x24 * x25;
    double x27 = a0 * x7 + a1 * x10;
    double x28 = x12 * (2.0 * x5 - x6) - x25 * x27;
    double x29 = b0 * b0;
    double x30 = x14 * x16;
    double x31 = -b1 * b2 * x12 - x18 * x30;
    double x32 = x30;
    double x33 = -x12 * (x2 - 2.0 * x3) - x22 * x32;
    double x34 = a0 * b0;
    double x35 = -x12 * (x21 + x34) + x14 * x16 * x24;
    double x36 = x12 * (2.0 * x8 - x9) + x27 * x32;
    double x37 = x18;
    double x38 = -x12 * (x5 - 2.0 * x6) - x14 * x22 * x37;
    double x39 = -x12 * (x8 - 2.0 * x9) + x14 * x24 * x37;
    double x40 = -x12 * (x20 + x34) + x14 * x18 * x27;
    double x41 = a1 * a1;
    double x42 = a2 * a2;
    double x43 = a0 * x12;
    double x44 = -a1 * x43 + x14 * x22 * x24;
    double x45 = -a2 * x43 + x14 * x22 * x27;
    double x46 = a0 * a0;
    double x47 = -a1 * a2 * x12 - x14 * x24 * x27;
#define HH(i, j) H[(i) * 9 + (j)]
    HH(0, 0) = x12 * (x0 + x1) - x13 * x13 * x14;
    HH(0, 1) = x17; HH(0, 2) = x19; HH(0, 3) = x23; HH(0, 4) = x26; HH(0, 5) = x28;
    HH(1, 0) = x17;
    HH(1, 1) = x12 * (x1 + x29) - x14 * x16 * x16;
    HH(1, 2) = x31; HH(1, 3) = x33; HH(1, 4) = x35; HH(1, 5) = x36;
    HH(2, 0) = x19; HH(2, 1) = x31;
    HH(2, 2) = x12 * (x0 + x29) - x14 * x18 * x18;
    HH(2, 3) = x38; HH(2, 4) = x39; HH(2, 5) = x40;
    HH(3, 0) = x23; HH(3, 1) = x33; HH(3, 2) = x38;
    HH(3, 3) = x12 * (x41 + x42) - x14 * x22 * x22;
    HH(3, 4) = x44; HH(3, 5) = x45;
    HH(4, 0) = x26; HH(4, 1) = x35; HH(4, 2) = x39; HH(4, 3) = x44;
    HH(4, 4) = x12 * (x42 + x46) - x14 * x24 * x24;
    HH(4, 5) = x47;
    HH(5, 0) = x28; HH(5, 1) = x36; HH(5, 2) = x40; HH(5, 3) = x45; HH(5, 4) = x47;
    HH(5, 5) = x12 * (x41 + x46) - x14 * x27 * x27;
#undef HH
  }
  return c;
}

// BaseScene.py:453-478
double Scene::f0(double x) const {
  if (x > eps_v * h) return x;
  return -x / (3.0 * eps_v * eps_v) * x / (h * h) * x + x / (eps_v * h) * x + eps_v * h / 3.0;
}
double Scene::f1(double x) const {
  if (x > eps_v * h) return 1.0 / x;
  return -x / ((eps_v * h) * (eps_v * h)) + 2.0 / (eps_v * h);
}
double Scene::f2(double x) const {
  if (x > eps_v * h) return -1.0 / (x * x);
  return -1.0 / ((eps_v * h) * (eps_v * h));
}

// BaseScene.py:487-598
void Scene::contact_energy(int diff, int spd) {
  double Esum = 0;
#pragma omp parallel for reduction(+ : Esum) schedule(static)
  for (int i = 0; i < nc; i++) {
    const I4& idx = const_idx[i];
    V3 p1 = pos[idx[1]] - pos[idx[0]];
    V3 p2 = pos[idx[2]] - pos[idx[0]];
    V3 p = pos[idx[3]] - pos[idx[0]];
    double* dH = &det_H[(size_t)i * 81]; double* dG = &det_G[(size_t)i * 9];
    double* cH = &cross_H[(size_t)i * 81]; double* cG = &cross_G[(size_t)i * 9];
    double* H9 = &d_H[(size_t)i * 81]; double* G9 = &d_G[(size_t)i * 9];
    double d = cd_det(p1, p2, p, diff, dH, dG);
    double c = cd_cross(p1, p2, diff, cH, cG);
    if (d / c < eps_contact) {
      if (diff) {
        for (int j = 0; j < 9; j++) G9[j] = dG[j] / c - d * cG[j] / (c * c);
        for (int j = 0; j < 9; j++)
          for (int k = 0; k < 9; k++)
            H9[j * 9 + k] = dH[j * 9 + k] / c - dG[j] * cG[k] / (c * c) - dG[k] * cG[j] / (c * c) - d * cH[j * 9 + k] / (c * c) +
                            2 * d * cG[j] * cG[k] / (c * c * c);
      }
      d /= c;
      double e = 0.5 * k_contact * (d - eps_contact) * (d - eps_contact);
      double pe_pd = k_contact * (d - eps_contact);
      if (diff) {
        for (int j = 0; j < 9; j++)
          for (int k = 0; k < 9; k++) H9[j * 9 + k] = k_contact * G9[j] * G9[k] + pe_pd * H9[j * 9 + k];
        for (int j = 0; j < 9; j++) G9[j] *= pe_pd;
        if (spd) spd_project(H9, &projT[(size_t)i * 81], &projQ[(size_t)i * 81], 9, 9, 20);
        force_T[i] = V3(0, 0, 0);
        for (int j = 0; j < 3; j++)
          for (int k = 0; k < 3; k++) {
            double g = G9[k * 3 + j];
            add_F(idx[k + 1] * 3 + j, g);
            add_F(idx[0] * 3 + j, -g);
            for (int j2 = 0; j2 < 3; j2++)
              for (int k2 = 0; k2 < 3; k2++) {
                double hh = H9[(k * 3 + j) * 9 + k2 * 3 + j2];
                add_H(idx[k + 1] * 3 + j, idx[k2 + 1] * 3 + j2, hh);
                add_H(idx[k + 1] * 3 + j, idx[0] * 3 + j2, -hh);
                add_H(idx[0] * 3 + j, idx[k2 + 1] * 3 + j2, -hh);
                add_H(idx[0] * 3 + j, idx[0] * 3 + j2, hh);
              }
            force_T[i][j] += g;
          }
      } else Esum += e;
    }
  }
  // friction
#pragma omp parallel for reduction(+ : Esum) schedule(static)
  for (int i = 0; i < nc; i++) {
    const I4& idx = const_idx[i];
    const V3& w = const_w[i];
    double k = const_k[i];
    const double* T = &const_T[(size_t)i * 6];
    V3 x_c = pos[idx[0]] * w[0] + pos[idx[1]] * w[1] + pos[idx[2]] * w[2];
    V3 dxv = pos[idx[3]] - x_c - const_dx0[i];
    double u[2] = {T[0] * dxv[0] + T[1] * dxv[1] + T[2] * dxv[2], T[3] * dxv[0] + T[4] * dxv[1] + T[5] * dxv[2]};
    double r = std::sqrt(u[0] * u[0] + u[1] * u[1]);
    if (diff) {
      double g[2] = {u[0] * k * f1(r), u[1] * k * f1(r)};
      double g1[3];
      for (int j = 0; j < 3; j++) g1[j] = g[0] * T[j] + g[1] * T[3 + j];
      double hm[2][2] = {{f1(r), 0}, {0, f1(r)}};
      if (r > 1e-9)
        for (int a = 0; a < 2; a++)
          for (int b = 0; b < 2; b++) hm[a][b] += f2(r) * (u[a] / r) * u[b];
      if (spd) spd_project_2d(hm);
      double h1[3][3];
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
          double s = 0;
          for (int p = 0; p < 2; p++)
            for (int q = 0; q < 2; q++) s += T[p * 3 + a] * hm[p][q] * T[q * 3 + b];
          h1[a][b] = k * s;
        }
      force_f[i] = V3(g1[0], g1[1], g1[2]);
      double w1[4] = {-w[0], -w[1], -w[2], 1};
      for (int i1 = 0; i1 < 4; i1++)
        for (int j1 = 0; j1 < 3; j1++) add_F(idx[i1] * 3 + j1, w1[i1] * g1[j1]);
      for (int i1 = 0; i1 < 4; i1++)
        for (int i2 = 0; i2 < 4; i2++)
          for (int j1 = 0; j1 < 3; j1++)
            for (int j2 = 0; j2 < 3; j2++) add_H(idx[i1] * 3 + j1, idx[i2] * 3 + j2, w1[i1] * w1[i2] * h1[j1][j2]);
    } else Esum += k * f0(r);
  }
  if (!diff) E += Esum;
}

// BaseScene.py:682-730
void Scene::contact_energy_backprop(Grad& g, int step, const double* p_array) {
  for (int i = 0; i < nc; i++) {
    const I4& idx = const_idx[i];
    const V3& w = const_w[i];
    double k = const_k[i];
    const double* T = &const_T[(size_t)i * 6];
    V3 x_c = pos[idx[0]] * w[0] + pos[idx[1]] * w[1] + pos[idx[2]] * w[2];
    V3 dxv = pos[idx[3]] - x_c - const_dx0[i];
    double u[2] = {T[0] * dxv[0] + T[1] * dxv[1] + T[2] * dxv[2], T[3] * dxv[0] + T[4] * dxv[1] + T[5] * dxv[2]};
    double r = std::sqrt(u[0] * u[0] + u[1] * u[1]);
    double pressure = k / const_mu[i];
    double gg[2] = {u[0] * k * f1(r), u[1] * k * f1(r)};
    double g1[3];
    for (int j = 0; j < 3; j++) g1[j] = gg[0] * T[j] + gg[1] * T[3 + j];
    V3 n_c = const_n[i];
    {
      double w1[4] = {w[0], w[1], w[2], -1};
      for (int i1 = 0; i1 < 4; i1++)
        for (int j1 = 0; j1 < 3; j1++) {
          double dfdp = w1[i1] * g1[j1] / pressure;
          double zT = p_array[idx[i1] * 3 + j1];
          for (int i2 = 0; i2 < 4; i2++)
            for (int j2 = 0; j2 < 3; j2++) g.PG(step, idx[i2], j2) += zT * dfdp * w1[i2] * n_c[j2] * k_contact;
        }
    }
    double hm[2][2] = {{f1(r), 0}, {0, f1(r)}};
    if (r > 1e-9)
      for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++) hm[a][b] += f2(r) * (u[a] / r) * u[b];
    double h1[3][3];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        double s = 0;
        for (int pp = 0; pp < 2; pp++)
          for (int q = 0; q < 2; q++) s += T[pp * 3 + a] * hm[pp][q] * T[q * 3 + b];
        h1[a][b] = k * s;
      }
    double w1[4] = {-w[0], -w[1], -w[2], 1};
    for (int i1 = 0; i1 < 4; i1++)
      for (int i2 = 0; i2 < 4; i2++)
        for (int j1 = 0; j1 < 3; j1++)
          for (int j2 = 0; j2 < 3; j2++) {
            double zT = p_array[idx[i1] * 3 + j1];
            g.PG(step, idx[i2], j2) += zT * w1[i1] * w1[i2] * h1[j1][j2];
          }
  }
}

// BaseScene.py:778-816 (serial: the reference's atomic counter gives an arbitrary constraint
// Scene_sliding.py:139-176: d(loss)/d(mu_cloth_cloth) from the first nc1 (cloth-cloth) constraints
void Scene::contact_energy_backprop_friction(Grad& g, int step, const double* p_array) {
  (void)step;
  for (int i = 0; i < nc1; i++) {
    const I4& idx = const_idx[i];
    const V3& w = const_w[i];
    double k = const_k[i];
    const double* T = &const_T[(size_t)i * 6];
    V3 x_c = pos[idx[0]] * w[0] + pos[idx[1]] * w[1] + pos[idx[2]] * w[2];
    V3 dxv = pos[idx[3]] - x_c - const_dx0[i];
    double u[2] = {T[0] * dxv[0] + T[1] * dxv[1] + T[2] * dxv[2], T[3] * dxv[0] + T[4] * dxv[1] + T[5] * dxv[2]};
    double r = std::sqrt(u[0] * u[0] + u[1] * u[1]);
    double gg[2] = {u[0] * k * f1(r), u[1] * k * f1(r)};
    double g1[3];
    for (int j = 0; j < 3; j++) g1[j] = gg[0] * T[j] + gg[1] * T[3 + j];
    const double w1[4] = {w[0], w[1], w[2], -1.0};
    for (int i1 = 0; i1 < 4; i1++)
      for (int j1 = 0; j1 < 3; j1++) {
        double dfdmu = w1[i1] * g1[j1] / mu_cloth_cloth;
        double zT = p_array[idx[i1] * 3 + j1];
        if (!frozen[idx[i1] * 3 + j1]) g.grad_friction_coef += zT * dfdmu;
      }
  }
}

// order; vertex order is used here)
void Scene::contact_pair_analysis(int b_idx, int v_start, int v_end, double mu) {
  for (int i = v_start; i < v_end; i++) {
    size_t bi = (size_t)b_idx * tot_NV + i;
    if (proj_flag[bi]) {
      I3 idx = proj_idx[bi];
      V3 w = proj_w[bi];
      V3 x_c = pos[idx[0]] * w[0] + pos[idx[1]] * w[1] + pos[idx[2]] * w[2];
      V3 x0_c = prev_pos[idx[0]] * w[0] + prev_pos[idx[1]] * w[1] + prev_pos[idx[2]] * w[2];
      V3 n_c = normalized(cross(pos[idx[1]] - pos[idx[0]], pos[idx[2]] - pos[idx[0]]));
      if (proj_dir[bi] == 0) {
        n_c = -n_c;
        idx = I3{{idx[0], idx[2], idx[1]}};
        w = V3(w[0], w[2], w[1]);
      }
      if (dot(pos[i] - x_c, n_c) < eps_contact) {
        int c_idx = nc++;
        if (c_idx >= max_n_constraints) { nc = max_n_constraints; return; }
        contact_force[bi] = k_contact * (dot(pos[i] - x_c, n_c) - eps_contact);
        const_idx[c_idx] = I4{{idx[0], idx[1], idx[2], i}};
        const_w[c_idx] = w;
        const_k[c_idx] = -mu * contact_force[bi];
        const_mu[c_idx] = mu;
        const_dx0[c_idx] = prev_pos[i] - x0_c;
        V3 t1;
        if (std::fabs(n_c[0]) < 0.5) t1 = V3(n_c[0], n_c[2], -n_c[1]);
        else t1 = V3(n_c[1], -n_c[0], n_c[2]);
        V3 t2 = cross(n_c, t1);
        t1 = cross(n_c, t2);
        double* T = &const_T[(size_t)c_idx * 6];
        T[0] = t1[0]; T[1] = t1[1]; T[2] = t1[2]; T[3] = t2[0]; T[4] = t2[1]; T[5] = t2[2];
        const_n[c_idx] = n_c;
      } else contact_force[bi] = 0;
    } else contact_force[bi] = 0;
  }
}

void Scene::contact_analysis() {
  nc = 0;
  nc1 = 0;
  for (const auto& ps : pairs) {
    const double live = ps.mu_is_param == 2 ? mu_cloth_cloth : mu_cloth_elastic;  // Scene_sliding.py:78-85
    contact_pair_analysis(ps.b_idx, ps.v_start, ps.v_end, ps.mu_is_param ? live * (ps.mu > 0 ? ps.mu : 1.0) : ps.mu);  // factor: Scene_card.py:122-126
    if (ps.mu_is_param == 2) nc1 = nc;  // the cloth-cloth pairs come first in Scene_sliding.contact_analysis (:87)
  }
  rebuild_pattern();
}

// BaseScene.py:837-850
void Scene::calc_vn() {
  for (int i = 0; i < tot_NV; i++) vn[i] = V3();
  for (int i = 0; i < tot_NF; i++) {
    V3 v1 = pos[faces[i][0]], v2 = pos[faces[i][1]], v3 = pos[faces[i][2]];
    V3 n = cross(v2 - v1, v3 - v1);
    vn[faces[i][0]] += n; vn[faces[i][1]] += n; vn[faces[i][2]] += n;
  }
  for (int i = 0; i < tot_NV; i++) vn[i] = normalized(vn[i]);
}

// geometry.py:23-87
static void pt2tri(const V3& x, const V3& p1, const V3& p2, const V3& p3, int& c, double& d, V3& w) {
  const double eps = 0;
  V3 e1 = normalized(p2 - p1);
  V3 e2 = normalized(p3 - p2);
  V3 e3 = normalized(p1 - p3);
  V3 n = -normalized(cross(e1, e3));
  V3 x1 = x - dot(x - p1, n) * n;
  d = 0.0; c = 0; w = V3();
  if (dot(cross(x1 - p1, e1), n) > eps) {
    if (dot(x1 - p1, e1) < -eps) { c = 1; d = norm(x - p1); w = V3(1, 0, 0); }
    else if (dot(x1 - p2, e1) > eps) { c = 2; d = norm(x - p2); w = V3(0, 1, 0); }
    else {
      c = -3;
      double alpha = dot(x1 - p1, e1) / dot(p2 - p1, e1);
      V3 x2 = p1 + alpha * (p2 - p1);
      d = norm(x - x2);
      w = V3(1 - alpha, alpha, 0);
    }
  } else if (dot(cross(x1 - p2, e2), n) > eps) {
    if (dot(x1 - p2, e2) < -eps) { c = 2; d = norm(x - p2); w = V3(0, 1, 0); }
    else if (dot(x1 - p3, e2) > eps) { c = 3; d = norm(x - p3); w = V3(0, 0, 1); }
    else {
      c = -1;
      double alpha = dot(x1 - p2, e2) / dot(p3 - p2, e2);
      V3 x2 = p2 + alpha * (p3 - p2);
      d = norm(x - x2);
      w = V3(0, 1 - alpha, alpha);
    }
  } else if (dot(cross(x1 - p3, e3), n) > eps) {
    if (dot(x1 - p3, e3) < -eps) { c = 3; d = norm(x - p3); w = V3(0, 0, 1); }
    else if (dot(x1 - p1, e3) > eps) { c = 1; d = norm(x - p1); w = V3(1, 0, 0); }
    else {
      c = -2;
      double alpha = dot(x1 - p3, e3) / dot(p1 - p3, e3);
      V3 x2 = p3 + alpha * (p1 - p3);
      d = norm(x - x2);
      w = V3(alpha, 0, 1 - alpha);
    }
  } else {
    d = norm(x - x1);
    double S = norm(cross(p3 - p1, p2 - p1));
    double w1 = dot(cross(p3 - p2, x1 - p2), n) / S;
    double w2 = dot(cross(p1 - p3, x1 - p3), n) / S;
    double w3 = dot(cross(p2 - p1, x1 - p1), n) / S;
    w = V3(w1, w2, w3);
  }
}

// geometry.py:89-94 ; ti.floor(x / h, i32): floor of the quotient
static inline void grid_idx(const Scene& s, const V3& x, int out[3]) {
  for (int a = 0; a < 3; a++) {
    double v = std::min(std::max(x[a], -s.grid_bound), s.grid_bound);
    out[a] = (int)std::floor(v / s.grid_h) + s.grid_n / 2;
  }
}

// geometry.py:96-229.  The counting sort (p2g) is restated as a stable sort of the triangles of one
// body by cell id (triangle order inside a cell = face order; the reference's order inside a cell
// depends on atomic scheduling), the 27-cell scan and all tests are literal.
void Scene::projection_query() {
  int nbody = (int)body_list.size();
  std::vector<int> cell_of, order, cell_start;
  for (int body_idx = 0; body_idx < nbody; body_idx++) {
    const Body& body = body_list[body_idx];
    int nf = body.f_end - body.f_start;
    // p2g
    int amin[3] = {grid_n, grid_n, grid_n}, amax[3] = {0, 0, 0};
    cell_of.resize(nf);
    std::vector<std::array<int, 3>> cidx(nf);
    for (int t = 0; t < nf; t++) {
      const I3& f = faces[body.f_start + t];
      V3 mid_v = (pos[f[0]] + pos[f[1]] + pos[f[2]]) / 3;
      int id[3];
      grid_idx(*this, mid_v, id);
      cidx[t] = {id[0], id[1], id[2]};
      for (int a = 0; a < 3; a++) { amin[a] = std::min(amin[a], id[a]); amax[a] = std::max(amax[a], id[a]); }
    }
    int ext[3] = {amax[0] - amin[0] + 1, amax[1] - amin[1] + 1, amax[2] - amin[2] + 1};
    if (nf == 0) { ext[0] = ext[1] = ext[2] = 0; }
    size_t ncell = (size_t)std::max(ext[0], 0) * std::max(ext[1], 0) * std::max(ext[2], 0);
    cell_start.assign(ncell + 1, 0);
    for (int t = 0; t < nf; t++) {
      cell_of[t] = ((cidx[t][0] - amin[0]) * ext[1] + (cidx[t][1] - amin[1])) * ext[2] + (cidx[t][2] - amin[2]);
      cell_start[cell_of[t] + 1]++;
    }
    for (size_t c = 0; c < ncell; c++) cell_start[c + 1] += cell_start[c];
    order.resize(nf);
    {
      std::vector<int> fill(cell_start.begin(), cell_start.end() - 1);
      for (int t = 0; t < nf; t++) order[fill[cell_of[t]]++] = t;
    }
    // particle_v snapshot (geometry.py:153-158): triangle vertex positions at p2g time == current pos
    for (int body_idx2 = 0; body_idx2 < nbody; body_idx2++) {
      if (body_idx2 == body_idx) continue;
      const Body& body2 = body_list[body_idx2];
#pragma omp parallel for schedule(dynamic, 64)
      for (int i = body2.v_start; i < body2.v_end; i++) {
        V3 xq = pos[i];
        int q[3];
        grid_idx(*this, xq, q);
        int r0[3], r1[3];
        for (int a = 0; a < 3; a++) {
          r0[a] = std::max(q[a] - 1, amin[a]);
          r1[a] = std::min(q[a] + 1, amax[a]) + 1;
        }
        double d_min = 1e6, cos_max = -1e6;
        int pflag = 0;
        I3 pidx{{0, 0, 0}};
        V3 pw;
        for (int gi = r0[0]; gi < r1[0]; gi++)
          for (int gj = r0[1]; gj < r1[1]; gj++)
            for (int gk = r0[2]; gk < r1[2]; gk++) {
              int cell = ((gi - amin[0]) * ext[1] + (gj - amin[1])) * ext[2] + (gk - amin[2]);
              for (int s = cell_start[cell]; s < cell_start[cell + 1]; s++) {
                const I3& f = faces[body.f_start + order[s]];
                V3 v1 = pos[f[0]], v2 = pos[f[1]], v3 = pos[f[2]];
                int c; double d; V3 w;
                pt2tri(xq, v1, v2, v3, c, d, w);
                V3 vt = v1 * w[0] + v2 * w[1] + v3 * w[2];
                V3 nt = normalized(cross(v2 - v1, v3 - v1));
                double cs = dot(xq - vt, nt);
                if (d < d_min - 1e-5 || (d < d_min + 1e-5 && cs > cos_max)) {
                  d_min = d; cos_max = cs;
                  pidx = f; pw = w;
                  if (c == 0) pflag = 1;
                  else if (c > 0) pflag = !border_flag[f[c - 1]];
                  else {
                    int p1 = (c != -3) ? f[2] : f[0];
                    int p2 = (c != -3) ? f[2 + c] : f[1];
                    pflag = !(border_flag[p1] && border_flag[p2]);
                  }
                }
              }
            }
        V3 v1 = pos[pidx[0]], v2 = pos[pidx[1]], v3 = pos[pidx[2]];
        V3 n1 = vn[pidx[0]], n2 = vn[pidx[1]], n3 = vn[pidx[2]];
        V3 v = pw[0] * v1 + pw[1] * v2 + pw[2] * v3;
        V3 n = pw[0] * n1 + pw[1] * n2 + pw[2] * n3;
        size_t bi = (size_t)body_idx * tot_NV + i;
        if (proj_flag[bi] == 0 && pflag == 1) proj_dir[bi] = dot(xq - v, n) > 0;
        proj_flag[bi] = pflag;
        proj_idx[bi] = pidx;
        proj_w[bi] = pw;
      }
    }
    // geometry_self.project_pair_self (geometry_self.py:166-230), called for the bodies in self_contact (:295-296)
    if (body_idx < (int)self_contact.size() && self_contact[body_idx]) {
#pragma omp parallel for schedule(dynamic, 64)
      for (int i = body.v_start; i < body.v_end; i++) {
        V3 xq = pos[i];
        int q[3];
        grid_idx(*this, xq, q);
        int r0[3], r1[3];
        for (int a = 0; a < 3; a++) {
          r0[a] = std::max(q[a] - 1, amin[a]);
          r1[a] = std::min(q[a] + 1, amax[a]) + 1;
        }
        double d_min = 1e6, cos_max = -1e6;
        int pflag = 0;
        I3 pidx{{0, 0, 0}};
        V3 pw;
        for (int gi = r0[0]; gi < r1[0]; gi++)
          for (int gj = r0[1]; gj < r1[1]; gj++)
            for (int gk = r0[2]; gk < r1[2]; gk++) {
              int cell = ((gi - amin[0]) * ext[1] + (gj - amin[1])) * ext[2] + (gk - amin[2]);
              for (int s = cell_start[cell]; s < cell_start[cell + 1]; s++) {
                const I3& f = faces[body.f_start + order[s]];
                if (i == f[0] || i == f[1] || i == f[2]) continue;   // :186-187
                V3 v1 = pos[f[0]], v2 = pos[f[1]], v3 = pos[f[2]];
                int c; double d; V3 w;
                pt2tri(xq, v1, v2, v3, c, d, w);
                if (c != 0) continue;                                  // :194-195
                V3 vt = v1 * w[0] + v2 * w[1] + v3 * w[2];
                V3 nt = normalized(cross(v2 - v1, v3 - v1));
                double cs = dot(xq - vt, nt);
                if (d < d_min - 1e-5 || (d < d_min + 1e-5 && cs > cos_max)) {
                  d_min = d; cos_max = cs;
                  pidx = f; pw = w;
                  pflag = 1;
                }
              }
            }
        V3 v1 = pos[pidx[0]], v2 = pos[pidx[1]], v3 = pos[pidx[2]];
        V3 n1 = vn[pidx[0]], n2 = vn[pidx[1]], n3 = vn[pidx[2]];
        V3 v = pw[0] * v1 + pw[1] * v2 + pw[2] * v3;
        V3 n = pw[0] * n1 + pw[1] * n2 + pw[2] * n3;
        size_t bi = (size_t)body_idx * tot_NV + i;
        if (proj_flag[bi] == 0 && pflag == 1) proj_dir[bi] = dot(xq - v, n) > 0;
        proj_flag[bi] = pflag;
        proj_idx[bi] = pidx;
        proj_w[bi] = pw;
      }
    }
  }
}

}  // namespace tslo
