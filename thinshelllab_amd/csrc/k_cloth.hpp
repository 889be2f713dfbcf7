// Thin-shell (cloth) kernels for gfx950.  What each kernel computes is the sum the reference accumulates with Taichi atomics; how it is organised
// is native to this engine:
//   * faces / hinges are independent work items (one lane each, 64-wide waves, no divergence inside a work-item class), positions are gathered
//     straight from the global node array (L2 resident: 24 B per node);
//   * NO atomics: element gradients go to staging slots (three-vectors per element and slot) that k_vertex_gather sums per vertex, element blocks
//     to records (faces: their 9 x 9 block, hinges: four vertex gradients and a scale) that k_cloth_gather sums per 3 x 3 block of the SELL-64
//     matrix -- every sum in a fixed order, the assembled gradient and matrix are the same bits in every run;
//   * closed forms are derived independently of the reference's expanded expressions (vector calculus for the triangle-area Hessian, explicit
//     quirk terms where the reference deviates) -- the parity tests compare them with the literal CPU restatement in oracle/.
// Reference: /root/reference/code/engine/model_fold_offset.py (line numbers at each kernel).
#pragma once
#include "tsl_ctx.hpp"
#include "tsl_device.hpp"

struct FaceGeom {
  d3 n;          // unit normal (b-a) x (c-b)
  double h[3];   // altitude of vertex slot l over its opposite edge (Cloth.heights)
  double ca[3];  // cosine of the interior angle at slot l        (Cloth.angle)
};

// Cloth.compute_normal_dir (:169-174) + the position-only parts of prepare_bending (:415-437)
TSL_DEV FaceGeom face_geom(const d3 P[3]) {
  FaceGeom g;
  g.n = normalized(cross(P[1] - P[0], P[2] - P[1]));
#pragma unroll
  for (int l = 0; l < 3; l++) {
    const d3 p = P[l], a = P[(l + 1) % 3], b = P[(l + 2) % 3];
    const d3 en = cross(g.n, b - a);
    g.h[l] = fabs(dot(p - a, en)) / norm(en);
    g.ca[l] = dot(normalized(a - p), normalized(b - p));
  }
  return g;
}

// Cloth.compute_angle (:126-138): signed dihedral between face normals n1 (own) and n2 (neighbour);
// e = pos[f2v[i1][(l+1)%2]] - pos[f2v[i1][l]] decides the sign.
// The sign test n2.e < 0 is evaluated as in exact arithmetic: for the wrongly-tabled slots of init_mesh the edge e
// lies in the plane of the "neighbour" and n2.e is pure rounding noise; |n2.e| <= 1e-10 |e| counts as zero.
TSL_DEV bool sign_neg(const d3& n2, const d3& e) { return dot(n2, e) < -1e-10 * norm(e); }
TSL_DEV double dihedral(const d3& n1, const d3& n2, const d3& e) {
  const double c = dot(n1, n2);
  double theta = (c < 0.999999) ? acos(c) : 2.0 * sqrt(fabs(1.0 - c)) / sqrt(1.0 + c);
  if (sign_neg(n2, e)) theta = -theta;
  return theta;
}

// Cloth.compute_bending_grad (:379-402): d(theta)/dx for the 4 hinge vertices
// slot l of a three-entry array by selects: an index computed at run time makes the array an alloca, which the compiler parks in LDS (288 bytes per lane:
// 72 KB per workgroup, two workgroups per CU for the hinge kernels -- k_cloth_grad_hinge 129 us for 150k hinges next to the other assembly kernels)
TSL_DEV double pick3(const double a[3], int i) { return i == 0 ? a[0] : (i == 1 ? a[1] : a[2]); }
TSL_DEV d3 pick3(const d3 a[3], int i) {   // (component by component: a select between structs becomes a select between their ADDRESSES)
  return d3(i == 0 ? a[0].x : (i == 1 ? a[1].x : a[2].x), i == 0 ? a[0].y : (i == 1 ? a[1].y : a[2].y), i == 0 ? a[0].z : (i == 1 ? a[1].z : a[2].z));
}
TSL_DEV int pick3(const int a[3], int i) { return i == 0 ? a[0] : (i == 1 ? a[1] : a[2]); }
TSL_DEV void hinge_grad(const FaceGeom& g1, const FaceGeom& g2, int l, int p4, int p21, d3 g[4]) {
  const int p11 = (l + 1) % 3, p12 = (l + 2) % 3, p22 = 3 - p21 - p4;
  g[0] = (-1.0 / pick3(g1.h, l)) * g1.n;
  g[3] = (-1.0 / pick3(g2.h, p4)) * g2.n;
  g[1] = (pick3(g1.ca, p12) / pick3(g1.h, p11)) * g1.n + (pick3(g2.ca, p22) / pick3(g2.h, p21)) * g2.n;
  g[2] = (pick3(g1.ca, p11) / pick3(g1.h, p12)) * g1.n + (pick3(g2.ca, p21) / pick3(g2.h, p22)) * g2.n;
}

TSL_DEV void load_face(const double* __restrict__ pos, const int* __restrict__ f2v, int F, int v[3], d3 P[3]) {
#pragma unroll
  for (int k = 0; k < 3; k++) { v[k] = f2v[3 * F + k]; P[k] = ld3(pos, v[k]); }
}

// ---------------------------------------------------------------------------------------------
__global__ void k_cloth_normals(int n_cface, const double* __restrict__ pos, const int* __restrict__ f2v, double* __restrict__ norm_dir) {
  const int F = blockIdx.x * blockDim.x + threadIdx.x;
  if (F >= n_cface) return;
  int v[3]; d3 P[3];
  load_face(pos, f2v, F, v, P);
  st3(norm_dir, F, normalized(cross(P[1] - P[0], P[2] - P[1])));
}

// per-face membrane energy: 3 edge springs + area term (Cloth.compute_energy :202-213 visits the same
// six edge terms and two area terms per cell; :149-167)
TSL_DEV double cface_energy(const ClothDev& c, const d3 P[3], double V, const double* li) {
  double e = 0;
#pragma unroll
  for (int l = 0; l < 3; l++) {
    const double len = norm(P[l] - P[(l + 1) % 3]);
    const double s = 1.0 - len / li[l];
    e += c.Kl * s * s * li[l];
  }
  const double area = 0.5 * norm(cross(P[1] - P[0], P[2] - P[0]));
  const double sa = 1.0 - area / V;
  e += c.Ka * sa * sa * V;
  return e;
}

struct ClothArgs {
  int n_cface, n_hinge;
  const ClothDev* cloth;
  const int *f2v, *cf, *cp, *cid;
  const double *V, *li;
  const int *hg_info, *hg_v;
  const double* norm_dir;
  const int* f_order;   // processing order of the faces: faces of one stencil class in a row (coalesced record writes)
  // element gradients go to a staging array of 3-vectors (face f, slot l: 3 f + l; hinge h, slot j: gs_hinge + 4 h + j) that k_vertex_gather sums per
  // vertex in a fixed order
  double* gstage;
  int gs_hinge;
};

// hinge energy (Cloth.compute_bending_energy :108-120)
TSL_DEV double hinge_energy(const ClothArgs& A, int h, const double* __restrict__ pos, const double* __restrict__ ref_angle) {
  const int f1 = A.hg_info[8 * h], l = A.hg_info[8 * h + 1], f2 = A.hg_info[8 * h + 2];
  const ClothDev c = A.cloth[A.cid[f1]];
  const d3 n1 = ld3(A.norm_dir, f1), n2 = ld3(A.norm_dir, f2);
  const d3 e = ld3(pos, A.f2v[3 * f1 + (l + 1) % 2]) - ld3(pos, A.f2v[3 * f1 + l]);
  const double th = dihedral(n1, n2, e) - ref_angle[3 * f1 + l];
  return c.Kb * th * th * c.dx * c.dx * (1.0 / 3.0);
}

// ---------------------------------------------------------------------------------------------
// gradient: per face (edges + area, Cloth.compute_residual :653-677), per hinge (:679-687)
__global__ void k_cloth_grad_face(ClothArgs A, const double* __restrict__ pos) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= A.n_cface) return;
  const int f = A.f_order[t];
  const ClothDev c = A.cloth[A.cid[f]];
  int v[3]; d3 P[3];
  load_face(pos, A.f2v, f, v, P);
  d3 g[3] = {d3(), d3(), d3()};
#pragma unroll
  for (int l = 0; l < 3; l++) {
    const int m = (l + 1) % 3;
    const d3 delta = P[l] - P[m];
    const double len = norm(delta);
    const double dl = -c.Kl * 2.0 * (1.0 - len / A.li[3 * f + l]);
    const d3 t = delta * (dl / len);
    g[l] = g[l] + t;
    g[m] = g[m] - t;
  }
  const d3 Nn = cross(P[1] - P[0], P[2] - P[0]);
  const double nN = norm(Nn);
  const double area = 0.5 * nN;
  const double da = -c.Ka * 2.0 * (1.0 - area / A.V[f]);
  const d3 nh = Nn / nN;
#pragma unroll
  for (int l = 0; l < 3; l++) {
    // dA/dx_l = 1/2 n x (x_{l+2} - x_{l+1})  (== compute_area_dx :312-325)
    const d3 ga = 0.5 * cross(nh, P[(l + 2) % 3] - P[(l + 1) % 3]);
    g[l] = g[l] + da * ga;
  }
#pragma unroll
  for (int l = 0; l < 3; l++) st3(A.gstage, 3 * f + l, g[l]);
}

__global__ void __launch_bounds__(256) k_cloth_grad_hinge(ClothArgs A, const double* __restrict__ pos, const double* __restrict__ ref_angle) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= A.n_hinge) return;
  const int f1 = A.hg_info[8 * h], l = A.hg_info[8 * h + 1], f2 = A.hg_info[8 * h + 2], p4 = A.hg_info[8 * h + 3], p21 = A.hg_info[8 * h + 4];
  const ClothDev c = A.cloth[A.cid[f1]];
  int v1[3], v2[3]; d3 P1[3], P2[3];
  load_face(pos, A.f2v, f1, v1, P1);
  load_face(pos, A.f2v, f2, v2, P2);
  const FaceGeom g1 = face_geom(P1), g2 = face_geom(P2);
  d3 g[4];
  hinge_grad(g1, g2, l, p4, p21, g);
  const double theta = dihedral(g1.n, g2.n, pick3(P1, (l + 1) % 2) - pick3(P1, l));
  const double dth = 2.0 * c.Kb * (theta - ref_angle[3 * f1 + l]) * c.dx * c.dx * (1.0 / 3.0);
#pragma unroll
  for (int j = 0; j < 4; j++) st3(A.gstage, A.gs_hinge + 4 * h + j, dth * g[j]);
}

// ---------------------------------------------------------------------------------------------
// Hessian: the element kernels leave RECORDS, k_cloth_gather below sums them into the 3 x 3 blocks of the SELL-64 value array (element (r, c) of a block
// lives at +64 (3 r + c) from its base).

// Quirk constants: compute_Hessian_bending (:597, :606) reads c_i[l][.] and mat_N[l*3+.] with l the VERTEX SLOT,
// i.e. the bending data of faces 0,1,2 of the cloth, for every face.  One tiny launch evaluates prepare_bending
// (:415-448) for those three faces: Q[cloth][face l][slot s] = {c_i, mat_N(9)}.
__global__ void k_cloth_quirk(ClothArgs A, int n_cloth, const double* __restrict__ pos, const double* __restrict__ ref_angle, double* __restrict__ Q) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_cloth * 9) return;
  const int cid = t / 9, fl = (t % 9) / 3, s = t % 3;
  const ClothDev c = A.cloth[cid];
  double* out = Q + (size_t)t * 10;
  if (fl >= c.NF) { for (int k = 0; k < 10; k++) out[k] = 0; return; }
  const int f = c.face_start + fl;
  int v[3]; d3 P[3];
  load_face(pos, A.f2v, f, v, P);
  const d3 n = ld3(A.norm_dir, f);
  const int nb = A.cf[3 * f + s];
  const d3 e = P[(s + 1) % 2] - P[s];
  bool judge = true;
  double ci = 0;
  if (nb != -1) {
    const d3 n2 = ld3(A.norm_dir, nb);
    if (sign_neg(n2, e)) judge = false;
    const double theta = dihedral(n, n2, e);
    ci = 2.0 * c.Kb * (theta - ref_angle[3 * f + s]) * c.dx * c.dx * (1.0 / 3.0);
  }
  const d3 nd = judge ? -n : n;
  const d3 p = P[s], a = P[(s + 1) % 3], b = P[(s + 2) % 3];
  const d3 edge = b - a;
  d3 en = cross(nd, edge);
  if (dot(en, a - p) > 0) en = -en;
  const m3 Mm = m3_outer(nd, en);
  const double inv = 1.0 / norm(edge);
  out[0] = ci;
  for (int k = 0; k < 9; k++) out[1 + k] = Mm.m[k] * inv;
}

// One lane per face: edge springs (compute_Hessian_me :472-522), area term (compute_Hessian_ma :526-580),
// second-order bending part H_lm (compute_Hessian_bending :585-614); all land in the face's own 3x3 grid
// of blocks, accumulated in registers and flushed once.
// CLAMP_ALL: the whole 9x9 face block (springs + un-projected area and bending parts) is projected as well -- used only to
// build an SPD preconditioner when the reference's partially projected Hessian turns out indefinite.
template <bool CLAMP_ALL>
__global__ void __launch_bounds__(128)
k_cloth_hess_face(ClothArgs A, const double* __restrict__ pos, const double* __restrict__ ref_angle, const double* __restrict__ Q, int spd, double* __restrict__ rec) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= A.n_cface) return;
  const int f = A.f_order[t];
  const int cid = A.cid[f];
  const ClothDev c = A.cloth[cid];
  int v[3]; d3 P[3];
  load_face(pos, A.f2v, f, v, P);
  double L[81];  // [(l*3+j)*9 + (m*3+k)]
#pragma unroll
  for (int i = 0; i < 81; i++) L[i] = 0;

  // ---- edge springs: K = dl * G + dl2 * d d^T ; G_jj = (1 - d_j^2)/len, G_jk = + d_j d_k / len (sign as in :292-294)
#pragma unroll
  for (int l = 0; l < 3; l++) {
    const int m = (l + 1) % 3;
    const d3 delta = P[l] - P[m];
    const double len = norm(delta);
    const d3 d = delta / len;
    const double base = A.li[3 * f + l];
    const double dl = -c.Kl * 2.0 * (1.0 - len / base);
    const double dl2 = c.Kl * 2.0 / base;
    double K[9];
    const double dd[3] = {d.x, d.y, d.z};
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const double G = (j == k) ? (1.0 - dd[j] * dd[j]) / len : dd[j] * dd[k] / len;
        K[j * 3 + k] = dl * G + dl2 * dd[j] * dd[k];
      }
    if (spd) spd_clamp<3>(K);
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const double kv = K[j * 3 + k];
        L[(l * 3 + j) * 9 + l * 3 + k] += kv;
        L[(l * 3 + j) * 9 + m * 3 + k] -= kv;
        L[(m * 3 + j) * 9 + l * 3 + k] -= kv;
        L[(m * 3 + j) * 9 + m * 3 + k] += kv;
      }
  }

  // ---- area term: darea2 g g^T + darea * d2A, d2A from  A = |N|/2, N = sum_l x_l x x_{l+1}:
  //   d2A_lm = 1/2 [ ((e_l.e_m) I - e_m e_l^T - 4 g_l g_m^T)/|N| + sigma_lm [n]x ],  e_l = x_{l+2} - x_{l+1},
  //   sigma = +1 (m = l+2), -1 (m = l+1), 0 (m = l); minus the reference's factor-2 slip in compute_area_dxy_p12 (:369)
  //   on the (l != m, j != k) entries.
  {
    const d3 Nn = cross(P[1] - P[0], P[2] - P[0]);
    const double nN = norm(Nn);
    const double area = 0.5 * nN;
    const double base_area = A.V[f];
    const double da = -c.Ka * 2.0 * (1.0 - area / base_area);
    const double da2 = c.Ka * 2.0 / base_area;
    const d3 nh = Nn / nN;
    d3 e[3], g[3];
#pragma unroll
    for (int l = 0; l < 3; l++) { e[l] = P[(l + 2) % 3] - P[(l + 1) % 3]; g[l] = 0.5 * cross(nh, e[l]); }
    const double nx[9] = {0, -nh.z, nh.y, nh.z, 0, -nh.x, -nh.y, nh.x, 0};
    const double a2 = 2.0 * area, a2c = a2 * a2 * a2;
#pragma unroll
    for (int l = 0; l < 3; l++)
#pragma unroll
      for (int m = 0; m < 3; m++) {
        const double elm = dot(e[l], e[m]);
        const double sigma = (m == l) ? 0.0 : ((m == (l + 2) % 3) ? 1.0 : -1.0);
        const double el[3] = {e[l].x, e[l].y, e[l].z}, em[3] = {e[m].x, e[m].y, e[m].z};
        const double gl[3] = {g[l].x, g[l].y, g[l].z}, gm[3] = {g[m].x, g[m].y, g[m].z};
        const int o = 3 - l - m;  // third vertex when l != m
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
          for (int k = 0; k < 3; k++) {
            double h2 = 0.5 * ((((j == k) ? elm : 0.0) - em[j] * el[k] - 4.0 * gl[j] * gm[k]) / nN + sigma * nx[j * 3 + k]);
            if (l != m && j != k) {
              const d3 p1 = P[l], p2 = P[m], p3 = P[o];
              const int d2 = 3 - j - k;
              const double Cjk = (p2[j] - p1[j]) * (p3[k] - p1[k]) - (p3[j] - p1[j]) * (p2[k] - p1[k]);
              const double Cjd = (p2[j] - p1[j]) * (p3[d2] - p1[d2]) - (p3[j] - p1[j]) * (p2[d2] - p1[d2]);
              const double S = (p2[k] - p3[k]) * Cjk + (p2[d2] - p3[d2]) * Cjd;
              h2 -= 0.5 * (p1[j] - p3[j]) * Cjk * S / a2c;
            }
            L[(l * 3 + j) * 9 + m * 3 + k] += da2 * gl[j] * gm[k] + da * h2;
          }
      }
  }

  // ---- bending, second-order part (literal structure of :585-614)
  {
    const d3 n = ld3(A.norm_dir, f);
    double hgt[3], ca[3], ci[3];
    m3 Mm[3];
#pragma unroll
    for (int l = 0; l < 3; l++) {
      const int nb = A.cf[3 * f + l];
      const d3 e = P[(l + 1) % 2] - P[l];
      bool judge = true;
      ci[l] = 0;
      if (nb != -1) {
        const d3 n2 = ld3(A.norm_dir, nb);
        if (sign_neg(n2, e)) judge = false;
        const double theta = dihedral(n, n2, e);
        ci[l] = 2.0 * c.Kb * (theta - ref_angle[3 * f + l]) * c.dx * c.dx * (1.0 / 3.0);
      }
      const d3 nd = judge ? -n : n;
      const d3 p = P[l], a = P[(l + 1) % 3], b = P[(l + 2) % 3];
      d3 en = cross(nd, b - a);
      if (dot(en, a - p) > 0) en = -en;
      Mm[l] = m3_outer(nd, en);
      ca[l] = dot(normalized(a - p), normalized(b - p));
      hgt[l] = fabs(dot(p - a, en)) / norm(en);
    }
    double di[3];
#pragma unroll
    for (int l = 0; l < 3; l++) di[l] = ci[(l + 1) % 3] * ca[(l + 2) % 3] + ci[(l + 2) % 3] * ca[(l + 1) % 3] - ci[l];
    const double* Qc = Q + (size_t)cid * 90;
#pragma unroll
    for (int l = 0; l < 3; l++)
#pragma unroll
      for (int lm = 0; lm < 2; lm++) {
        const int m = (l + lm) % 3;
        const double inv = 1.0 / (hgt[l] * hgt[m]);
        double Hlm[9];
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
          for (int k = 0; k < 3; k++) Hlm[j * 3 + k] = inv * (di[l] * Mm[m].m[k * 3 + j] + di[m] * Mm[l].m[j * 3 + k]);
        if (l == m) {
          const int i1 = (l + 1) % 3, i2 = (l + 2) % 3;
          const double* q1 = Qc + (l * 3 + i1) * 10;
          const double* q2 = Qc + (l * 3 + i2) * 10;
#pragma unroll
          for (int e9 = 0; e9 < 9; e9++) Hlm[e9] -= q1[0] * q1[1 + e9] + q2[0] * q2[1 + e9];
        } else {
          const int i3 = 3 - l - m;
          const double* q3 = Qc + (l * 3 + i3) * 10;
#pragma unroll
          for (int e9 = 0; e9 < 9; e9++) Hlm[e9] += q3[0] * q3[1 + e9];
        }
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
          for (int k = 0; k < 3; k++) {
            L[(l * 3 + j) * 9 + m * 3 + k] += Hlm[j * 3 + k];
            if (l != m) L[(m * 3 + j) * 9 + l * 3 + k] += Hlm[k * 3 + j];
          }
      }
  }

  if (CLAMP_ALL) spd_clamp<9>(L);
  // the 9 x 9 block as a record of nine contiguous 3 x 3 vertex blocks for k_cloth_gather, addressed by the processing index t (what the gather lists
  // refer to): a lane fills its own 648 bytes, a gather reads 72 of them in a row
  double* R = rec + (size_t)t * 81;
#pragma unroll
  for (int l = 0; l < 3; l++)
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
      for (int j = 0; j < 3; j++)
#pragma unroll
        for (int k = 0; k < 3; k++) R[(l * 3 + m) * 9 + j * 3 + k] = L[(l * 3 + j) * 9 + m * 3 + k];
}

// One lane per hinge: Gauss-Newton block d2theta * grad grad^T (compute_Hessian_bending :616-637)
__global__ void __launch_bounds__(256) k_cloth_hess_hinge(ClothArgs A, const double* __restrict__ pos, double* __restrict__ rec) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= A.n_hinge) return;
  const int f1 = A.hg_info[8 * h], l = A.hg_info[8 * h + 1], f2 = A.hg_info[8 * h + 2], p4 = A.hg_info[8 * h + 3], p21 = A.hg_info[8 * h + 4];
  const ClothDev c = A.cloth[A.cid[f1]];
  int v1[3], v2[3]; d3 P1[3], P2[3];
  load_face(pos, A.f2v, f1, v1, P1);
  load_face(pos, A.f2v, f2, v2, P2);
  const FaceGeom g1 = face_geom(P1), g2 = face_geom(P2);
  d3 g[4];
  hinge_grad(g1, g2, l, p4, p21, g);
  const double d2 = 2.0 * c.Kb * c.dx * c.dx * (1.0 / 3.0);
  // the record k_cloth_gather forms the blocks from: the four vertex gradients and the scale (128 bytes); block (j, k) = d2 g_j g_k^T
  double* R = rec + (size_t)h * 16;
#pragma unroll
  for (int j = 0; j < 4; j++) { R[3 * j] = g[j].x; R[3 * j + 1] = g[j].y; R[3 * j + 2] = g[j].z; }
  R[12] = d2;
}

// Gather assembly of the cloth Hessian (round 3): one lane per 3 x 3 matrix block that some face or hinge contributes to.  The
// element kernels above leave records (faces: their 9 x 9 block as nine 3 x 3 blocks; hinges: four vertex gradients + scale) and this kernel sums, for
// its block, the contributions of the incident elements from a list built once per context (ent[ptr[b] .. ptr[b + 1]): bit 31
// hinge / face, bits 4..30 element, bits 0..3 local vertex pair) and adds the sum to the matrix with ONE plain read-modify-write
// per entry -- the scatter versions of rounds 1-2 issued 144 (hinge) + 81 (face) f64 atomics per element, 30 M per assembly at 100k triangles,
// which bound both kernels (0.27 ms of an assembly's 0.45).  Blocks are sorted by their address in the SELL-64 value array: the
// lanes of a wave write consecutive lanes of one slice.  Fixed summation order: the assembled cloth blocks are the same bits every run.
#define CG_BPW 28   // blocks per workgroup of k_cloth_gather (28 x 9 = 252 of 256 threads)
__global__ void __launch_bounds__(256) k_cloth_gather(int n_blk, const int* __restrict__ base, const int* __restrict__ ptr, const unsigned* __restrict__ ent, int n_hinge, int n_cface,
                                                      const double* __restrict__ hrec, const double* __restrict__ frec, const double* __restrict__ trec, double* __restrict__ vals) {
  // Round 6: a workgroup takes 28 consecutive blocks.  SUMS: one lane per ENTRY of a block (nine lanes per block, entry index fastest: a record is read as one
  // contiguous run of 72 bytes, nine times the loads in flight of the one-lane-per-block form that walked its list with one record at a time -- 84 MB of records
  // at 0.8 TB/s).  STORES: through LDS, block index fastest -- the blocks are sorted by their address in the SELL-64 value array, so for one entry index the
  // lanes write consecutive lanes of a slice (the nine entries of a block lie 512 bytes apart: written by the nine lanes of the block, every read-modify-write
  // was a 64-byte sector of its own).  Every entry is still summed by ONE lane in the order of the list: the same bits.
  __shared__ double tile[9][CG_BPW + 1];
  const int b0 = blockIdx.x * CG_BPW, t = threadIdx.x;
  if (t < 9 * CG_BPW) {
    const int bl = t / 9, e9 = t - 9 * bl, b = b0 + bl, r = e9 / 3, c = e9 - 3 * r;
    double acc = 0.0;
    if (b < n_blk) {
      for (int q = ptr[b]; q < ptr[b + 1]; q++) {
        const unsigned e = ent[q];
        const int el = (int)((e >> 4) & 0x7ffffff), pr = (int)(e & 15);
        if ((e >> 30) == 1) {   // tetrahedron: its 16 vertex-pair blocks as stored by k_tet_hess_coop (144 doubles per element)
          acc += trec[(size_t)(el & 0x3ffffff) * 144 + pr * 9 + e9];
        } else if (e >> 31) {   // hinge: block (j, k) = d2 g_j g_k^T from the four vertex gradients and the scale
          const int j = pr >> 2, k = pr & 3;
          const double* R = hrec + (size_t)el * 16;
          acc += R[12] * (R[3 * j + r] * R[3 * k + c]);
        } else {
          acc += frec[(size_t)el * 81 + pr * 9 + e9];   // el: the face's processing index, pr = 3 l + m
        }
      }
    }
    tile[e9][bl] = acc;
  }
  __syncthreads();
  if (t < 9 * CG_BPW) {
    const int e9 = t / CG_BPW, bl = t - CG_BPW * e9, b = b0 + bl;
    if (b < n_blk) vals[(size_t)base[b] + 64 * e9] += tile[e9][bl];
  }
}

// Cloth.update_ref_angle (:176-185), one lane per hinge
__global__ void k_cloth_update_ref(ClothArgs A, const double* __restrict__ pos, double* __restrict__ ref_angle) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= A.n_hinge) return;
  const int f1 = A.hg_info[8 * h], l = A.hg_info[8 * h + 1], f2 = A.hg_info[8 * h + 2];
  const ClothDev c = A.cloth[A.cid[f1]];
  const d3 n1 = ld3(A.norm_dir, f1), n2 = ld3(A.norm_dir, f2);
  const d3 e = ld3(pos, A.f2v[3 * f1 + (l + 1) % 2]) - ld3(pos, A.f2v[3 * f1 + l]);
  const double theta = dihedral(n1, n2, e);
  const double dis = theta - ref_angle[3 * f1 + l];
  const double ad = fabs(dis);
  if (ad > c.k_angle) ref_angle[3 * f1 + l] += (ad - c.k_angle) * dis / ad;
}
