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
  vals[(size_t)base + 64 * 0] += d;   // (one writer per diagonal block, behind the clear and in front of the gathers on the same stream)
  vals[(size_t)base + 64 * 4] += d;
  vals[(size_t)base + 64 * 8] += d;
}

struct TetArgs {
  int n_tet;
  const ElasticDev* el;
  const int *tv, *tel;
  const double *B, *W;
  double* gstage;   // element gradients of tet t go to gstage[3 (4 t + j)], summed per vertex by k_vertex_gather
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
__global__ void k_tet_grad(TetArgs A, const double* __restrict__ pos) {
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
    st3(A.gstage, 4 * t + i, gi);
    f3 = f3 - gi;
  }
  st3(A.gstage, 4 * t + 3, f3);
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
// stA / stB: the four vertex contributions of tet t go to slots 4 t + j of stA (tactile material) or stB (the others), zeros to the other array;
// k_vertex_gather sums them per vertex in a fixed order (into d_tact / d_accum)
__global__ void k_tet_deri_mu(TetArgs A, const double* __restrict__ pos, double* __restrict__ stA, double* __restrict__ stB) {
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
  double* sto = (e.kind == 0) ? stA : stB;
  double* stz = (e.kind == 0) ? stB : stA;
  d3 f3 = d3();
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const d3 fi = d3(-W * Hm.m[i], -W * Hm.m[3 + i], -W * Hm.m[6 + i]);
    st3(sto, 4 * t + i, fi); st3(stz, 4 * t + i, d3());
    f3 = f3 - fi;
  }
  st3(sto, 4 * t + 3, f3); st3(stz, 4 * t + 3, d3());
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

// Element Hessians, cooperative form (round 6): 16 lanes per tetrahedron, the 9 x 9 block in LDS.  Lane c < 9 forms ROW c = (vertex n, axis dim) of the
// block (one directional derivative tet_dH each instead of nine per lane), the eigen-clamp is spd_clamp9_par (tsl_device.hpp: nine rounds of four simultaneous
// Jacobi rotations per sweep) started from the eigenvector basis of the element's previous assembly (Vws, as spd_clamp_warm<9>: A' = V^T A V is nearly diagonal
// when the element moved little; `warm` false or a stored basis that does not look like one starts from the identity), and lane b of the group writes block
// (b / 4, b % 4) of the element record that k_cloth_gather sums into the matrix.  (Rounds 1-5: one lane per element with the block and the basis in
// private arrays -- 512 registers, 1414 spilled, 130-170 us for the 5.8k elements of cfg4, the longest kernel of an assembly.)  kind 0 = 9 x 9 over vertices 0..2 with optional SPD projection, vertex 3 = minus row / column sums
// (model_elastic_tactile.py:88-124); kind 1 = direct 12 x 12 (model_elastic_offset.py:101-167, no projection).
__global__ void __launch_bounds__(256)
k_tet_hess_coop(TetArgs A, const double* __restrict__ pos, int spd, double* __restrict__ Vws, int warm, double* __restrict__ rec) {
  __shared__ double sA[16][81], sV[16][81], sT[16][81];
  const int l = threadIdx.x & 15, g = threadIdx.x >> 4;
  int t = blockIdx.x * 16 + g;
  const bool valid = t < A.n_tet;
  if (!valid) t = A.n_tet - 1;   // whole groups beyond the list still take part in the wave-wide votes
  int v[4]; m3 B;
  const m3 F = tet_F(A, t, pos, v, B);
  const ElasticDev e = A.el[A.tel[t]];
  const m3 Fi = (e.kind == 0) ? F : m3_inv(F), FiT = m3_T(Fi), BT = m3_T(B);   // the tactile material works from cofactors (tet_dH)
  const double W = A.W[t];
  const double Jraw = m3_det(F);
  const double J = (e.kind == 0) ? Jraw : fmax(Jraw, 0.01);
  const double logJ = (e.kind == 0) ? 0.0 : log(J);
  double* sa = sA[g]; double* sv = sV[g]; double* st = sT[g];
  if (l < 9) {   // He[(n * 3 + dim) * 9 + (i * 3 + j)] = d(grad of vertex i, comp j) / d(x_n, dim), n, i in 0..2: row l = (n, dim)
    const int n = l / 3, dim = l % 3;
    m3 dF;  // dD @ B with dD[dim][n] = 1  -> row dim of dF = row n of B
#pragma unroll
    for (int k = 0; k < 9; k++) dF.m[k] = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const double bnc = n == 0 ? B.m[c] : (n == 1 ? B.m[3 + c] : B.m[6 + c]);
      dF.m[c] = dim == 0 ? bnc : 0.0; dF.m[3 + c] = dim == 1 ? bnc : 0.0; dF.m[6 + c] = dim == 2 ? bnc : 0.0;
    }
    const m3 dH = tet_dH(e, F, Fi, FiT, J, logJ, dF, BT, W);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) sa[l * 9 + i * 3 + j] = dH.m[j * 3 + i];
  }
  spd_grp_sync();
  const bool clamp = (e.kind == 0 && spd) || spd == 2;   // spd 2: preconditioner-only assembly, every element block projected
  if (__any(clamp)) {
    // symmetrise (spd_clamp_warm does so before the basis change)
    if (l < 9 && clamp) {
      for (int k = l + 1; k < 9; k++) { const double s2 = 0.5 * (sa[l * 9 + k] + sa[k * 9 + l]); sa[l * 9 + k] = s2; sa[k * 9 + l] = s2; }
    }
    bool w_ok = false;
    if (Vws != nullptr && warm != 0) {   // the stored basis must look like one: finite entries and squared Frobenius norm 9
      double ss = 0.0;
      for (int q = l; q < 81; q += 16) { const double x = Vws[(size_t)q * A.n_tet + t]; sv[q] = x; ss += x * x; }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 16);
      w_ok = fabs(ss - 9.0) <= 9e-6;
    }
    spd_grp_sync();
    if (w_ok) {   // A <- V^T A V
      if (l < 9) {
#pragma unroll
        for (int j = 0; j < 9; j++) { double s2 = 0; for (int k = 0; k < 9; k++) s2 += sa[l * 9 + k] * sv[k * 9 + j]; st[l * 9 + j] = s2; }
      }
      spd_grp_sync();
      double ar[9];
      if (l < 9) {
#pragma unroll
        for (int j = 0; j < 9; j++) { double s2 = 0; for (int k = 0; k < 9; k++) s2 += sv[k * 9 + l] * st[k * 9 + j]; ar[j] = s2; }
      }
      spd_grp_sync();
      if (l < 9 && clamp) {   // (upper part of row l and its mirror, as the serial routine: the product is symmetric up to rounding)
#pragma unroll
        for (int j = 0; j < 9; j++) if (j >= l) { sa[l * 9 + j] = ar[j]; sa[j * 9 + l] = ar[j]; }
      }
    } else if (l < 9) {
#pragma unroll
      for (int k = 0; k < 9; k++) sv[l * 9 + k] = (k == l) ? 1.0 : 0.0;
    }
    spd_grp_sync();
    spd_clamp9_par(sa, sv, l, clamp);
    if (Vws != nullptr && valid && clamp) for (int q = l; q < 81; q += 16) Vws[(size_t)q * A.n_tet + t] = sv[q];   // the new basis for the next call
  }
  if (!valid) return;
  // element record: 16 blocks of 9; block (a, b) element (j, j2): a, b < 3: He[(a * 3 + j) * 9 + b * 3 + j2]; vertex 3 gets minus sums.  model_elastic_offset.py:151-167
  // scatters row = (vertex j, comp r), column = (n, dim): the transpose of the tactile convention (identical whenever the block is symmetric, i.e. J > 0.01)
  const bool tr = e.kind != 0;
  auto He = [&](int x, int y) { return tr ? sa[y * 9 + x] : sa[x * 9 + y]; };
  const int a = l >> 2, b = l & 3;
  double* R = rec + (size_t)t * 144 + l * 9;
#pragma unroll
  for (int j = 0; j < 3; j++)
#pragma unroll
    for (int j2 = 0; j2 < 3; j2++) {
      double s2 = 0;
      if (a < 3 && b < 3) s2 = He(a * 3 + j, b * 3 + j2);
      else if (a < 3) { for (int bb = 0; bb < 3; bb++) s2 -= He(a * 3 + j, bb * 3 + j2); }
      else if (b < 3) { for (int aa = 0; aa < 3; aa++) s2 -= He(aa * 3 + j, b * 3 + j2); }
      else { for (int aa = 0; aa < 3; aa++) for (int bb = 0; bb < 3; bb++) s2 += He(aa * 3 + j, bb * 3 + j2); }
      R[3 * j + j2] = s2;
    }
}

