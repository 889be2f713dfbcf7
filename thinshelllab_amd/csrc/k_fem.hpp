// Per-vertex (inertia / gravity / external force) and per-tetrahedron kernels.
// Reference: /root/reference/code/engine/model_elastic_tactile.py (kind 0) and model_elastic_offset.py (kind 1),
// vertex terms also model_fold_offset.py:193-200, :641-648, :468-470.
#pragma once
#include "tsl_ctx.hpp"
#include "tsl_device.hpp"

struct VertArgs {
  int NV;
  const double *mass, *grav, *fext;
  double dt;
};

// E_v = -f_ext.x - m g.x + 1/2 m |x - x_prev - v dt|^2 / dt^2   (model_fold_offset.py:193-200, model_elastic_tactile.py:186-192)
TSL_DEV double vert_energy(const VertArgs& A, int i, const double* __restrict__ pos, const double* __restrict__ prev, const double* __restrict__ vel) {
  const d3 x = ld3(pos, i);
  const double m = A.mass[i];
  const d3 X = x - ld3(prev, i) - ld3(vel, i) * A.dt;
  return -dot(ld3(A.fext, i), x) - m * dot(ld3(A.grav, i), x) + 0.5 * m * dot(X, X) / (A.dt * A.dt);
}

// gradient of the above, written (not accumulated) as the first contribution to F
__global__ void k_vert_grad(VertArgs A, const double* __restrict__ pos, const double* __restrict__ prev, const double* __restrict__ vel, double* __restrict__ F) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A.NV) return;
  const double m = A.mass[i];
  const d3 X = ld3(pos, i) - ld3(prev, i) - ld3(vel, i) * A.dt;
  const d3 g = -m * ld3(A.grav, i) - ld3(A.fext, i) + X * (m / (A.dt * A.dt));
  st3(F, i, g);   // (the first contribution: the gradient array holds nothing else yet)
}

// mass diagonal m/dt^2 on every dof, frozen or not (H.H.add without frozen test, model_fold_offset.py:468-470,
// model_elastic_tactile.py:84-86)
__global__ void k_vert_hess(VertArgs A, const int* __restrict__ diag_blk, double* __restrict__ vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A.NV) return;
  const double d = A.mass[i] / (A.dt * A.dt);
  const int base = diag_blk[i];
  atomicAdd(&vals[(size_t)base + 64 * 0], d);
  atomicAdd(&vals[(size_t)base + 64 * 4], d);
  atomicAdd(&vals[(size_t)base + 64 * 8], d);
}

struct TetArgs {
  int n_tet;
  const ElasticDev* el;
  const int *tv, *tel;
  const double *B, *W;
  double* gstage;   // deterministic assembly: element gradients of tet t at gstage[3 (4 t + j)] (null: atomics)
};

TSL_DEV m3 tet_F(const TetArgs& A, int t, const double* __restrict__ pos, int v[4], m3& B) {
#pragma unroll
  for (int k = 0; k < 4; k++) v[k] = A.tv[4 * t + k];
  const d3 x3 = ld3(pos, v[3]);
  const d3 c0 = ld3(pos, v[0]) - x3, c1 = ld3(pos, v[1]) - x3, c2 = ld3(pos, v[2]) - x3;
  m3 Ds;
  Ds.m[0] = c0.x; Ds.m[1] = c1.x; Ds.m[2] = c2.x;
  Ds.m[3] = c0.y; Ds.m[4] = c1.y; Ds.m[5] = c2.y;
  Ds.m[6] = c0.z; Ds.m[7] = c1.z; Ds.m[8] = c2.z;
#pragma unroll
  for (int k = 0; k < 9; k++) B.m[k] = A.B[9 * (size_t)t + k];
  return m3_mul(Ds, B);
}

// model_elastic_tactile.py:194-201 / model_elastic_offset.py:325-331
TSL_DEV double tet_energy(const TetArgs& A, int t, const double* __restrict__ pos) {
  int v[4]; m3 B;
  const m3 F = tet_F(A, t, pos, v, B);
  const ElasticDev e = A.el[A.tel[t]];
  double I1 = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) I1 += F.m[k] * F.m[k];
  const double J = m3_det(F);
  double phi;
  if (e.kind == 0) {
    phi = e.mu / 2 * (I1 - 3) + e.lam / 2 * (J - e.alpha) * (J - e.alpha);
  } else {
    const double lj = log(fmax(0.01, J));
    phi = e.mu / 2 * (I1 - 3) - e.mu * lj + e.lam / 2 * lj * lj;
  }
  return A.W[t] * phi;
}

// forces: model_elastic_tactile.py:144-154 / model_elastic_offset.py:188-198 ; residual contribution is -force
__global__ void k_tet_grad(TetArgs A, const double* __restrict__ pos, double* __restrict__ Fg) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= A.n_tet) return;
  int v[4]; m3 B;
  const m3 F = tet_F(A, t, pos, v, B);
  const ElasticDev e = A.el[A.tel[t]];
  m3 P;
  if (e.kind == 0) {
    // P = mu F + lam (J - alpha) J F^-T with J F^-T taken as the cofactor matrix: the same polynomial in F, without the division
    // by a determinant that passes through zero when a pad element is crushed flat (the energy is finite there)
    const double J = m3_det(F);
    const m3 C = m3_cof2(F, F);
    const double s = e.lam * (J - e.alpha);
#pragma unroll
    for (int k = 0; k < 9; k++) P.m[k] = e.mu * F.m[k] + s * C.m[k];
  } else {
    const m3 FiT = m3_T(m3_inv(F));
    const double J = fmax(m3_det(F), 0.01);
    const double s = e.lam * log(J);
#pragma unroll
    for (int k = 0; k < 9; k++) P.m[k] = e.mu * (F.m[k] - FiT.m[k]) + s * FiT.m[k];
  }
  const m3 Hm = m3_mul(P, m3_T(B));  // force on vertex i = -W * column i
  const double W = A.W[t];
  d3 f3 = d3();
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const d3 gi = d3(W * Hm.m[i], W * Hm.m[3 + i], W * Hm.m[6 + i]);  // +W*col = -(force) = dE/dx_i
    if (A.gstage) st3(A.gstage, 4 * t + i, gi); else atomic_add3(Fg, v[i], gi);
    f3 = f3 - gi;
  }
  if (A.gstage) st3(A.gstage, 4 * t + 3, f3); else atomic_add3(Fg, v[3], f3);
}

// F_f = -(elastic gradient) + m g + f_ext on the vertices of the FEM bodies (Elastic.get_force, model_elastic_tactile.py:144-164 /
// model_elastic_offset.py:188-208); f holds the elastic gradient on entry
__global__ void k_elastic_force_finish(VertArgs A, int v0, int v1, double* __restrict__ f) {
  const int i = v0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= v1) return;
  const d3 g = ld3(A.grav, i), e = ld3(A.fext, i);
  st3(f, i, A.mass[i] * g + e - ld3(f, i));
}

// d(force)/d(mu): model_elastic_tactile.py:329-347 (P1 / mu = F - J F^-T) and model_elastic_offset.py:415-431 (P1 / mu = F - F^-T).
// Tactile contributions go to d_tact (cleared by the caller on every call), box / ball contributions to d_accum, which the
// reference never clears (it zeroes F_f instead), so it keeps growing over the calls.
// stA / stB (deterministic): the four vertex contributions of tet t go to slots 4 t + j of stA (tactile material) or stB (the others), zeros to
// the other array; k_vertex_gather sums them per vertex in a fixed order
__global__ void k_tet_deri_mu(TetArgs A, const double* __restrict__ pos, double* __restrict__ d_tact, double* __restrict__ d_accum, double* __restrict__ stA, double* __restrict__ stB) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= A.n_tet) return;
  int v[4]; m3 B;
  const m3 F = tet_F(A, t, pos, v, B);
  const ElasticDev e = A.el[A.tel[t]];
  const m3 JFiT = (e.kind == 0) ? m3_cof2(F, F) : m3_T(m3_inv(F));   // J F^-T as the cofactor matrix (tactile) / F^-T (J = 1 in the formula of the box model)
  m3 P;
#pragma unroll
  for (int k = 0; k < 9; k++) P.m[k] = F.m[k] - JFiT.m[k];
  const m3 Hm = m3_mul(P, m3_T(B));
  const double W = A.W[t];
  double* out = (e.kind == 0) ? d_tact : d_accum;
  double* sto = (e.kind == 0) ? stA : stB;
  double* stz = (e.kind == 0) ? stB : stA;
  d3 f3 = d3();
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const d3 fi = d3(-W * Hm.m[i], -W * Hm.m[3 + i], -W * Hm.m[6 + i]);
    if (sto) { st3(sto, 4 * t + i, fi); st3(stz, 4 * t + i, d3()); }
    else atomic_add3(out, v[i], fi);
    f3 = f3 - fi;
  }
  if (sto) { st3(sto, 4 * t + 3, f3); st3(stz, 4 * t + 3, d3()); }
  else atomic_add3(out, v[3], f3);
}

// dP(dF) for the two materials (energy Hessian direction), returns dE-Hessian column block dH = W * dP * B^T
TSL_DEV m3 tet_dH(const ElasticDev& e, const m3& F, const m3& Fi, const m3& FiT, double J, double logJ, const m3& dF, const m3& BT, double W) {
  m3 dP;
  if (e.kind == 0) {
    // P = mu F + lam (J - alpha) C, C = J F^-T = cof F  (model_elastic_tactile.py:104-107 with the sign folded in).  The reference
    // writes dP = mu dF + lam (2 J^2 - alpha J) tr(F^-1 dF) F^-T - lam (J - alpha) J F^-T dF^T F^-T; with dJ = C : dF = J tr(F^-1 dF)
    // and dC = (dJ C - C dF^T C) / J that is dP = mu dF + lam dJ C + lam (J - alpha) dC -- the same polynomial, evaluated here
    // from cofactors (Fi, FiT unused): the two 1 / J terms of the reference's form cancel only analytically, and pad elements
    // do pass through J = 0 late in the cfg4 rollout (entries of 1e15 in the operator before this form).
    const m3 C = m3_cof2(F, F);
    const m3 dC1 = m3_cof2(F, dF), dC2 = m3_cof2(dF, F);
    double dJ = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) dJ += C.m[k] * dF.m[k];
    const double a = e.lam * dJ, b = e.lam * (J - e.alpha);
#pragma unroll
    for (int k = 0; k < 9; k++) dP.m[k] = e.mu * dF.m[k] + a * C.m[k] + b * (dC1.m[k] + dC2.m[k]);
  } else {
    // tr(F^-1 dF)
    const m3 FidF = m3_mul(Fi, dF);
    const double dTr = FidF.m[0] + FidF.m[4] + FidF.m[8];
    const m3 X = m3_mul(m3_mul(FiT, m3_T(dF)), FiT);  // F^-T dF^T F^-T
    // P = mu (F - F^-T) + lam log J F^-T  (model_elastic_offset.py:141-142)
#pragma unroll
    for (int k = 0; k < 9; k++) dP.m[k] = e.mu * dF.m[k] + (e.mu - e.lam * logJ) * X.m[k] + e.lam * dTr * FiT.m[k];
  }
  m3 r = m3_mul(dP, BT);
#pragma unroll
  for (int k = 0; k < 9; k++) r.m[k] *= W;
  return r;
}

// Element Hessians: kind 0 = 9x9 over vertices 0..2 with optional SPD projection, vertex 3 = minus row/col sums
// (model_elastic_tactile.py:88-124); kind 1 = direct 12x12 (model_elastic_offset.py:101-167, no projection).
__global__ void __launch_bounds__(64)
k_tet_hess(TetArgs A, const int* __restrict__ blk, const double* __restrict__ pos, int spd, double* __restrict__ vals, double* __restrict__ Vws, int warm, double* __restrict__ rec) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= A.n_tet) return;
  int v[4]; m3 B;
  const m3 F = tet_F(A, t, pos, v, B);
  const ElasticDev e = A.el[A.tel[t]];
  const m3 Fi = (e.kind == 0) ? F : m3_inv(F), FiT = m3_T(Fi), BT = m3_T(B);   // the tactile material works from cofactors (tet_dH)
  const double W = A.W[t];
  const double Jraw = m3_det(F);
  const double J = (e.kind == 0) ? Jraw : fmax(Jraw, 0.01);
  const double logJ = (e.kind == 0) ? 0.0 : log(J);
  // He[(n*3+dim)*9 + (i*3+j)] = d(grad of vertex i, comp j)/d(x_n,dim), n,i in 0..2
  double He[81];
  for (int n = 0; n < 3; n++)
    for (int dim = 0; dim < 3; dim++) {
      m3 dF;  // dD @ B with dD[dim][n] = 1  -> row dim of dF = row n of B
#pragma unroll
      for (int k = 0; k < 9; k++) dF.m[k] = 0;
#pragma unroll
      for (int c = 0; c < 3; c++) dF.m[dim * 3 + c] = B.m[n * 3 + c];
      const m3 dH = tet_dH(e, F, Fi, FiT, J, logJ, dF, BT, W);
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) He[(n * 3 + dim) * 9 + i * 3 + j] = dH.m[j * 3 + i];
    }
  if ((e.kind == 0 && spd) || spd == 2) {   // spd 2: preconditioner-only assembly, every element block projected
    if (Vws) spd_clamp_warm<9>(He, Vws + t, (size_t)A.n_tet, warm != 0);   // eigenvector basis of the element's previous assembly as the start
    else spd_clamp<9>(He);
  }
  if (e.kind != 0) {
    // model_elastic_offset.py:151-167 scatters row = (vertex j, comp r), column = (n, dim): the transpose of the
    // tactile convention (identical whenever the block is symmetric, i.e. J > 0.01)
    for (int a = 0; a < 9; a++)
      for (int b = a + 1; b < 9; b++) { const double tmp = He[a * 9 + b]; He[a * 9 + b] = He[b * 9 + a]; He[b * 9 + a] = tmp; }
  }
  // scatter: 16 blocks; block (a,b) element (j,j2): a,b<3: He[(a*3+j)*9 + b*3+j2]; vertex 3 gets minus sums
  for (int a = 0; a < 4; a++)
    for (int b = 0; b < 4; b++) {
      const int base = blk[16 * t + a * 4 + b];
      for (int j = 0; j < 3; j++)
        for (int j2 = 0; j2 < 3; j2++) {
          double s = 0;
          if (a < 3 && b < 3) s = He[(a * 3 + j) * 9 + b * 3 + j2];
          else if (a < 3) { for (int bb = 0; bb < 3; bb++) s -= He[(a * 3 + j) * 9 + bb * 3 + j2]; }
          else if (b < 3) { for (int aa = 0; aa < 3; aa++) s -= He[(aa * 3 + j) * 9 + b * 3 + j2]; }
          else { for (int aa = 0; aa < 3; aa++) for (int bb = 0; bb < 3; bb++) s += He[(aa * 3 + j) * 9 + bb * 3 + j2]; }
          if (rec) rec[(size_t)t * 144 + (a * 4 + b) * 9 + 3 * j + j2] = s;   // gather assembly: k_cloth_gather adds the block
          else atomicAdd(&vals[(size_t)base + 64 * (3 * j + j2)], s);
        }
    }
}
