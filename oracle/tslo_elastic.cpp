// ORACLE (test infrastructure only -- see oracle/README.md).
// CPU restatement of the tetrahedral FEM bodies:
//   kind 0  /root/reference/code/engine/model_elastic_tactile.py  (tactile pad, stable Neo-Hookean)
//   kind 1  /root/reference/code/engine/model_elastic_offset.py   (box split into 5 tets per cube)
//   kind 2  same file with load=True (mesh read from data/ball.*)
#include "tslo_engine.h"

namespace tslo {

// model_elastic_tactile.py:13-80 and count() :302-321
void Elastic::construct_tactile(double dt_, int offset_, double ratio_, int nv, const double* nodes, int nc, const int* tets,
                                int ns, const int* faces) {
  kind = 0;
  E = 300000; nu = 0.2;
  mu = E / (2 * (1 + nu));
  lam = E * nu / ((1 + nu) * (1 - 2 * nu));
  alpha = 1 + mu / lam;
  density = 2000.0;
  dt = dt_; offset = offset_; ratio = ratio_;
  gravity = V3(0, 0, -9.8);
  n_verts = nv; n_cells = nc; n_surfaces = ns;
  F_ox.resize(nv);
  for (int i = 0; i < nv; i++) F_ox[i] = V3(nodes[i * 3], nodes[i * 3 + 1], nodes[i * 3 + 2]);
  F_vertices.resize(nc);
  for (int i = 0; i < nc; i++) F_vertices[i] = I4{{tets[i * 4], tets[i * 4 + 1], tets[i * 4 + 2], tets[i * 4 + 3]}};
  f2v_array.resize(ns); f2v.assign(ns, I3{{0, 0, 0}});
  for (int i = 0; i < ns; i++) f2v_array[i] = I3{{faces[i * 3], faces[i * 3 + 1], faces[i * 3 + 2]}};
  F_x.assign(nv, V3()); F_x_prev.assign(nv, V3()); F_v.assign(nv, V3()); F_f.assign(nv, V3()); F_b.assign(nv, V3());
  ext_force.assign(nv, V3()); F_m.assign(nv, 0.0); F_B.assign(nc, M3()); F_W.assign(nc, 0.0);
  H_e.assign((size_t)nc * 81, 0.0); T_e.assign((size_t)nc * 81, 0.0); Q_e.assign((size_t)nc * 81, 0.0);
  // count()
  is_surface.assign(nv, 0);
  for (int i = 0; i < ns; i++) for (int k = 0; k < 3; k++) is_surface[f2v_array[i][k]] = 1;
  frozen_cnt = 0; surf_point = 0;
  for (int i = 0; i < nv; i++)
    if (is_surface[i]) {
      bool bottom = F_ox[i][2] < 0.001, inner = norm(F_ox[i]) < 0.0076, surf = norm(F_ox[i]) > 0.0148;
      if (bottom || inner) frozen_cnt++;
      else if (surf) surf_point++;
    }
}

// model_elastic_offset.py:12-92 (load=False)
void Elastic::construct_box(double dt_, double Len, int offset_, int Nx, int Ny, int Nz, double density_) {
  kind = 1;
  E = 5e5; nu = 0.0;
  mu = E / (2 * (1 + nu));
  lam = E * nu / ((1 + nu) * (1 - 2 * nu));
  density = density_;
  dt = dt_; offset = offset_;
  gravity = V3(0, 0, -9.8);
  n_cube[0] = Nx; n_cube[1] = Ny; n_cube[2] = Nz;
  n_verts = Nx * Ny * Nz;
  n_cells = 5 * (Nx - 1) * (Ny - 1) * (Nz - 1);
  int mx = std::max(Nx, std::max(Ny, Nz));
  dx = Len / (mx - 1);
  int su = 0;
  for (int i = 0; i < 3; i++) su += (n_cube[i] - 1) * (n_cube[(i + 1) % 3] - 1);
  n_surfaces = 2 * su * 2;
  int nv = n_verts, nc = n_cells, ns = n_surfaces;
  F_ox.assign(nv, V3()); F_vertices.assign(nc, I4{{0, 0, 0, 0}}); f2v.assign(ns, I3{{0, 0, 0}});
  F_x.assign(nv, V3()); F_x_prev.assign(nv, V3()); F_v.assign(nv, V3()); F_f.assign(nv, V3()); F_b.assign(nv, V3());
  ext_force.assign(nv, V3()); F_m.assign(nv, 0.0); F_B.assign(nc, M3()); F_W.assign(nc, 0.0);
  is_surface.assign(nv, 0);
}

// model_elastic_offset.py:38-48 (load=True)
void Elastic::construct_loaded(double dt_, int offset_, double density_, int nv, const double* nodes, int nc, const int* tets,
                               int ns, const int* faces) {
  kind = 2;
  E = 5e5; nu = 0.0;
  mu = E / (2 * (1 + nu));
  lam = E * nu / ((1 + nu) * (1 - 2 * nu));
  density = density_;
  dt = dt_; offset = offset_;
  gravity = V3(0, 0, -9.8);
  n_verts = nv; n_cells = nc; n_surfaces = ns;
  F_ox.resize(nv);
  for (int i = 0; i < nv; i++) F_ox[i] = V3(nodes[i * 3], nodes[i * 3 + 1], nodes[i * 3 + 2]);
  F_vertices.resize(nc);
  for (int i = 0; i < nc; i++) F_vertices[i] = I4{{tets[i * 4], tets[i * 4 + 1], tets[i * 4 + 2], tets[i * 4 + 3]}};
  f2v.resize(ns); f2v_array.resize(ns);
  for (int i = 0; i < ns; i++) f2v_array[i] = I3{{faces[i * 3], faces[i * 3 + 1], faces[i * 3 + 2]}};
  F_x.assign(nv, V3()); F_x_prev.assign(nv, V3()); F_v.assign(nv, V3()); F_f.assign(nv, V3()); F_b.assign(nv, V3());
  ext_force.assign(nv, V3()); F_m.assign(nv, 0.0); F_B.assign(nc, M3()); F_W.assign(nc, 0.0);
  is_surface.assign(nv, 0);
}

void Elastic::init(double ox, double oy, double oz, int flip) {
  if (kind == 0) {
    // model_elastic_tactile.py:323-326 -> init_pos (:214-230), init_surface_indices (:265-291)
    for (int i = 0; i < n_verts; i++) { F_v[i] = V3(); F_f[i] = V3(); F_m[i] = 0; }
    for (int i = 0; i < n_verts; i++) {
      F_x[i] = ratio * F_ox[i];
      if (flip) F_x[i] = -F_x[i];
      F_x[i] += V3(ox, oy, oz);
    }
    for (int c = 0; c < n_cells; c++) {
      M3 F = Ds(F_vertices[c]);
      F_B[c] = inverse(F);
      F_W[c] = std::fabs(det(F)) / 6;
      for (int i = 0; i < 4; i++) F_m[F_vertices[c][i]] += F_W[c] / 4 * density;
    }
    for (int i = 0; i < n_surfaces; i++) {
      f2v[i] = f2v_array[i];
      V3 p1 = F_x[f2v[i][0]], p2 = F_x[f2v[i][1]], p3 = F_x[f2v[i][2]];
      V3 n = normalized(cross(p2 - p1, p3 - p1));
      V3 inner_point(ox, oy, oz + 0.002 * ratio);
      if (flip) inner_point = V3(ox, oy, oz - 0.002 * ratio);
      bool all_inner = is_inner_circle(f2v[i][0]) && is_inner_circle(f2v[i][1]) && is_inner_circle(f2v[i][2]);
      if (dot(n, inner_point - p1) > 0) {
        if (!all_inner) std::swap(f2v[i][1], f2v[i][2]);
      } else {
        if (all_inner) std::swap(f2v[i][1], f2v[i][2]);
      }
    }
    return;
  }
  if (kind == 1) {
    // model_elastic_offset.py:292-304 get_vertices: 5 tets per cube, parity-flipped corner codes
    auto i2p = [&](int x, int y, int z) { return (x * n_cube[1] + y) * n_cube[2] + z; };
    for (int Ix = 0; Ix < n_cube[0] - 1; Ix++)
      for (int Iy = 0; Iy < n_cube[1] - 1; Iy++)
        for (int Iz = 0; Iz < n_cube[2] - 1; Iz++) {
          int e = ((Ix * (n_cube[1] - 1) + Iy) * (n_cube[2] - 1) + Iz) * 5;
          auto set_element = [&](int ee, int v0, int v1, int v2, int v3) {
            int vs[4] = {v0, v1, v2, v3};
            for (int i = 0; i < 4; i++) {
              int bx = ((vs[i] >> 0) ^ Ix) & 1, by = ((vs[i] >> 1) ^ Iy) & 1, bz = ((vs[i] >> 2) ^ Iz) & 1;
              F_vertices[ee][i] = i2p(Ix + bx, Iy + by, Iz + bz);
            }
          };
          int js[4] = {0, 3, 5, 6};
          for (int i = 0; i < 4; i++) { int j = js[i]; set_element(e + i, j, j ^ 1, j ^ 2, j ^ 4); }
          set_element(e + 4, 1, 2, 4, 7);
        }
    for (int x = 0; x < n_cube[0]; x++)
      for (int y = 0; y < n_cube[1]; y++)
        for (int z = 0; z < n_cube[2]; z++) F_ox[i2p(x, y, z)] = V3(x * dx, y * dx, z * dx);
  }
  // model_elastic_offset.py:232-250 init_pos
  for (int u = 0; u < n_verts; u++) { F_x[u] = F_ox[u]; F_v[u] = V3(); F_f[u] = V3(); F_m[u] = 0.0; }
  if (kind == 1 && arch != 0.0)  // init_pos_arch (:259-260)
    for (int x = 0; x < n_cube[0]; x++)
      for (int y = 0; y < n_cube[1]; y++)
        for (int z = 0; z < n_cube[2]; z++) F_x[(x * n_cube[1] + y) * n_cube[2] + z][2] += arch * std::sin((double)x / (double)(n_cube[0] - 1) * 3.1415926);
  for (int c = 0; c < n_cells; c++) {
    M3 F = Ds(F_vertices[c]);
    F_B[c] = inverse(F);
    F_W[c] = std::fabs(det(F)) / 6;
    for (int i = 0; i < 4; i++) F_m[F_vertices[c][i]] += F_W[c] / 4 * density;
  }
  for (int u = 0; u < n_verts; u++) F_x[u] += V3(ox, oy, oz);
  if (kind == 1) {
    // model_elastic_offset.py:333-376 get_surface_indices (order of the atomic counter is
    // nondeterministic in the reference; here: cell order)
    auto check = [&](int u) {
      int ans = 0, rest = u;
      for (int i = 0; i < 3; i++) {
        int k = rest % n_cube[2 - i];
        rest = rest / n_cube[2 - i];
        if (k == 0) ans |= (1 << (i * 2));
        if (k == n_cube[2 - i] - 1) ans |= (1 << (i * 2 + 1));
      }
      return ans;
    };
    int cnt = 0;
    for (int c = 0; c < n_cells; c++)
      if (c % 5 != 4) {
        int is[3] = {0, 2, 3};
        for (int t = 0; t < 3; t++) {
          int i = is[t];
          int verts[3] = {F_vertices[c][(i + 0) % 4], F_vertices[c][(i + 1) % 4], F_vertices[c][(i + 2) % 4]};
          int sum_ = check(verts[0]) & check(verts[1]) & check(verts[2]);
          if (sum_) {
            int m = cnt++;
            int verts3 = F_vertices[c][(i + 3) % 4];
            V3 normal = cross(F_x[verts[1]] - F_x[verts[0]], F_x[verts[2]] - F_x[verts[0]]);
            if (dot(normal, F_x[verts3] - F_x[verts[0]]) > 0) std::swap(verts[1], verts[2]);
            if (m < n_surfaces) f2v[m] = I3{{verts[0], verts[1], verts[2]}};
          }
        }
      }
  } else {
    // model_elastic_offset.py:378-393 init_normal (load=True)
    for (int i = 0; i < n_surfaces; i++) {
      f2v[i] = f2v_array[i];
      V3 p1 = F_x[f2v[i][0]], p2 = F_x[f2v[i][1]], p3 = F_x[f2v[i][2]];
      V3 n = normalized(cross(p2 - p1, p3 - p1));
      V3 inner_point(ox, oy, oz);
      if (dot(n, inner_point - p1) > 0) std::swap(f2v[i][1], f2v[i][2]);
    }
  }
}

// model_elastic_tactile.py:183-201 / model_elastic_offset.py:314-331
void Elastic::compute_energy() {
  double Usum = 0;
  for (int c = 0; c < n_verts; c++) {
    Usum += -F_m[c] * dot(gravity, F_x[c]);
    Usum += -dot(ext_force[c], F_x[c]);
  }
  for (int c = 0; c < n_verts; c++) {
    V3 X = F_x[c] - F_x_prev[c] - F_v[c] * dt;
    Usum += 0.5 * F_m[c] * dot(X, X) / (dt * dt);
  }
#pragma omp parallel for reduction(+ : Usum) schedule(static)
  for (int c = 0; c < n_cells; c++) {
    M3 F_i = Ds(F_vertices[c]) * F_B[c];
    double phi_i;
    if (kind == 0) {
      double J_i = det(F_i);
      double I_i = trace(transpose(F_i) * F_i);
      phi_i = mu / 2 * (I_i - 3);
      phi_i += lam / 2 * (J_i - alpha) * (J_i - alpha);
    } else {
      double log_J_i = std::log(std::max(0.01, det(F_i)));
      phi_i = mu / 2 * (trace(transpose(F_i) * F_i) - 3);
      phi_i -= mu * log_J_i;
      phi_i += lam / 2 * log_J_i * log_J_i;
    }
    Usum += F_W[c] * phi_i;
  }
  U = Usum;
}

// model_elastic_tactile.py:144-164 / model_elastic_offset.py:188-208
void Elastic::get_force() {
  for (int i = 0; i < n_verts; i++) F_f[i] = V3();
  for (int c = 0; c < n_cells; c++) {
    const I4& verts = F_vertices[c];
    M3 F = Ds(verts) * F_B[c];
    M3 F_T = transpose(inverse(F));
    M3 P;
    if (kind == 0) {
      double J = det(F);
      P = mu * F + lam * (J - alpha) * J * F_T;
    } else {
      double J = std::max(det(F), 0.01);
      P = mu * (F - F_T) + lam * std::log(J) * F_T;
    }
    M3 Hm = -F_W[c] * (P * transpose(F_B[c]));
    for (int i = 0; i < 3; i++) {
      V3 force(Hm[0][i], Hm[1][i], Hm[2][i]);
      F_f[verts[i]] += force;
      F_f[verts[3]] -= force;
    }
  }
  for (int u = 0; u < n_verts; u++) {
    F_f[u] += gravity * F_m[u];
    F_f[u] += ext_force[u];
  }
}

// model_elastic_tactile.py:329-347 (clears the fields, P1 = mu (F - J F^-T), P2 = lam (J - 1) J F^-T) and
// model_elastic_offset.py:415-431 (clears F_f instead of d_mu / d_lam, so they accumulate over the calls; lam = 0 there
// makes d_lam 0/0)
void Elastic::compute_deri() {
  if (d_mu.size() != (size_t)n_verts) { d_mu.assign(n_verts, V3()); d_lam.assign(n_verts, V3()); }
  if (kind == 0) { d_mu.assign(n_verts, V3()); d_lam.assign(n_verts, V3()); }
  else for (int i = 0; i < n_verts; i++) F_f[i] = V3();
  for (int c = 0; c < n_cells; c++) {
    const I4& verts = F_vertices[c];
    M3 F = Ds(verts) * F_B[c];
    M3 F_T = transpose(inverse(F));
    M3 P1, P2;
    if (kind == 0) {
      double J = det(F);
      P1 = mu * (F - J * F_T);
      P2 = lam * (J - 1) * J * F_T;
    } else {
      double J = std::max(det(F), 0.01);
      P1 = mu * (F - F_T);
      P2 = lam * std::log(J) * F_T;
    }
    M3 H1 = -F_W[c] * (P1 * transpose(F_B[c]));
    M3 H2 = -F_W[c] * (P2 * transpose(F_B[c]));
    for (int i = 0; i < 3; i++) {
      V3 f1(H1[0][i], H1[1][i], H1[2][i]);
      d_mu[verts[i]] += f1 / mu; d_mu[verts[3]] -= f1 / mu;
      V3 f2(H2[0][i], H2[1][i], H2[2][i]);
      d_lam[verts[i]] += f2 / lam; d_lam[verts[3]] -= f2 / lam;
    }
  }
}

// model_elastic_tactile.py:166-169 / model_elastic_offset.py:210-213
void Elastic::compute_residual() {
  for (int i = 0; i < n_verts; i++) F_b[i] = F_m[i] * (F_x[i] - F_x_prev[i] - F_v[i] * dt) / (dt * dt) - F_f[i];
}

void Elastic::compute_Hessian(Scene& A, int spd) {
  for (int i = 0; i < n_verts; i++)
    for (int j = 0; j < 3; j++) A.H.add(3 * (i + offset) + j, 3 * (i + offset) + j, F_m[i] / (dt * dt));
  if (kind == 0) {
    // model_elastic_tactile.py:81-124
#pragma omp parallel for schedule(static)
    for (int e = 0; e < n_cells; e++) {
      M3 F = Ds(F_vertices[e]) * F_B[e];
      M3 F_inv = inverse(F);
      M3 F_inv_T = transpose(F_inv);
      double J = det(F);
      double* He = &H_e[(size_t)e * 81];
      for (int n = 0; n < 3; n++)
        for (int dim = 0; dim < 3; dim++) {
          M3 dD;
          dD[dim][n] = 1;
          M3 dF = dD * F_B[e];
          M3 dF_T = transpose(dF);
          double dTr = trace(F_inv * dF);
          M3 dP = -mu * dF;
          dP = dP - lam * 2 * J * J * dTr * F_inv_T;
          dP = dP + lam * alpha * J * dTr * F_inv_T;
          dP = dP + lam * (J - alpha) * J * (F_inv_T * dF_T * F_inv_T);
          M3 dH = -F_W[e] * (dP * transpose(F_B[e]));
          for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) He[(n * 3 + dim) * 9 + i * 3 + j] = dH[j][i];
        }
      if (spd) spd_project(He, &T_e[(size_t)e * 81], &Q_e[(size_t)e * 81], 9, 9, 20);
      int idx[4];
      for (int k = 0; k < 4; k++) idx[k] = F_vertices[e][k] + offset;
      for (int j = 0; j < 3; j++)
        for (int k = 0; k < 3; k++)
          for (int j2 = 0; j2 < 3; j2++)
            for (int k2 = 0; k2 < 3; k2++) {
              double h = He[(k * 3 + j) * 9 + k2 * 3 + j2];
              A.add_H(idx[k] * 3 + j, idx[k2] * 3 + j2, h);
              A.add_H(idx[k] * 3 + j, idx[3] * 3 + j2, -h);
              A.add_H(idx[3] * 3 + j, idx[k2] * 3 + j2, -h);
              A.add_H(idx[3] * 3 + j, idx[3] * 3 + j2, h);
            }
    }
  } else {
    // model_elastic_offset.py:94-168 (spd ignored)
#pragma omp parallel for schedule(static)
    for (int e = 0; e < n_cells; e++) {
      M3 dD[4][3], dFm[4][3], dPm[4][3], dHm[4][3];
      for (int n = 0; n < 3; n++)
        for (int dim = 0; dim < 3; dim++) dD[n][dim][dim][n] = 1;
      for (int dim = 0; dim < 3; dim++) dD[3][dim] = (dD[0][dim] + dD[1][dim] + dD[2][dim]) * -1.0;
      for (int n = 0; n < 4; n++)
        for (int dim = 0; dim < 3; dim++) dFm[n][dim] = dD[n][dim] * F_B[e];
      M3 F = Ds(F_vertices[e]) * F_B[e];
      M3 F_1 = inverse(F);
      M3 F_1_T = transpose(F_1);
      double J = std::max(det(F), 0.01);
      for (int n = 0; n < 4; n++)
        for (int dim = 0; dim < 3; dim++)
          for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
              M3 dF;
              dF[i][j] = 1;
              M3 dF_T = transpose(dF);
              double dTr = F_1_T[i][j];
              M3 dP_dFij = mu * dF + (mu - lam * std::log(J)) * (F_1_T * dF_T * F_1_T) + lam * dTr * F_1_T;
              double dFij_ndim = dFm[n][dim][i][j];
              dPm[n][dim] = dPm[n][dim] + dP_dFij * dFij_ndim;
            }
      for (int n = 0; n < 4; n++)
        for (int dim = 0; dim < 3; dim++) dHm[n][dim] = -F_W[e] * (dPm[n][dim] * transpose(F_B[e]));
      for (int n = 0; n < 4; n++) {
        int i = F_vertices[e][n];
        for (int dim = 0; dim < 3; dim++) {
          int ind = (i + offset) * 3 + dim;
          for (int j = 0; j < 3; j++) {
            int idx = F_vertices[e][j] + offset;
            A.add_H(idx * 3 + 0, ind, -dHm[n][dim][0][j]);
            A.add_H(idx * 3 + 1, ind, -dHm[n][dim][1][j]);
            A.add_H(idx * 3 + 2, ind, -dHm[n][dim][2][j]);
          }
          int idx = F_vertices[e][3] + offset;
          A.add_H(idx * 3 + 0, ind, dHm[n][dim][0][0] + dHm[n][dim][0][1] + dHm[n][dim][0][2]);
          A.add_H(idx * 3 + 1, ind, dHm[n][dim][1][0] + dHm[n][dim][1][1] + dHm[n][dim][1][2]);
          A.add_H(idx * 3 + 2, ind, dHm[n][dim][2][0] + dHm[n][dim][2][1] + dHm[n][dim][2][2]);
        }
      }
    }
  }
}

}  // namespace tslo
