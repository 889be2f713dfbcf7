// Geometric multigrid preconditioner for the cloth block of the system matrix (no reference counterpart: the
// reference factorises the matrix with cupyx spsolve, sparse_solver.py:85-105; block-Jacobi PCG needs O(N) iterations on
// an N x N cloth because the membrane stiffness 2Kl/dx dominates the inertia m/dt^2 by ~0.01/dx^3).
//
// The cloth is a regular (N+1) x (M+1) vertex grid, so the hierarchy is geometric: level l+1 keeps every second
// vertex, prolongation P is bilinear interpolation of the 3-vector displacement field, coarse operators are Galerkin
// products A_{l+1} = P^T A_l P recomputed after every assembly.  Level 0 is the global SELL-64 operator (all bodies +
// matrix-free contact blocks); FEM vertices and contact couplings are only smoothed there.  Levels >= 1 are stored as a
// radius-2 block stencil (25 slots x 9 doubles per vertex, slot-major / SoA so that every load of a wave is contiguous).
// Cycle: V(nu,nu) with damped block-Jacobi smoothing (symmetric => the preconditioner is SPD and PCG stays valid).
#pragma once
#include "k_solver.hpp"
#include "tsl_ctx.hpp"
#include "tsl_device.hpp"

// bilinear weights of fine index i onto coarse indices: i even -> (i/2, 1); i odd -> (i/2, 1/2), (i/2 + 1, 1/2)
TSL_DEV int mg_coarse(int i, int q, double& w) {
  if ((i & 1) == 0) { w = (q == 0) ? 1.0 : 0.0; return i >> 1; }
  w = 0.5;
  return (i >> 1) + q;
}

struct MgGrid { int N, M; };  // cells; vertices (N+1) x (M+1), index i*(M+1)+j

// ---- stencil-level kernels (levels >= 1).  A[(s*9+e)*n + row], s = (dI+2)*5 + (dJ+2)
__global__ void k_st_spmv(MgGrid g, const double* __restrict__ A, const double* __restrict__ x, double* __restrict__ y) {
  const int n = (g.N + 1) * (g.M + 1);
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  const int I = row / (g.M + 1), J = row % (g.M + 1);
  double y0 = 0, y1 = 0, y2 = 0;
#pragma unroll
  for (int dI = -2; dI <= 2; dI++) {
    const int I2 = I + dI;
    if (I2 < 0 || I2 > g.N) continue;
#pragma unroll
    for (int dJ = -2; dJ <= 2; dJ++) {
      const int J2 = J + dJ;
      if (J2 < 0 || J2 > g.M) continue;
      const int s = (dI + 2) * 5 + (dJ + 2);
      const double* a = A + (size_t)s * 9 * n + row;
      const d3 xj = ld3(x, I2 * (g.M + 1) + J2);
      y0 += a[0] * xj.x + a[(size_t)n] * xj.y + a[2 * (size_t)n] * xj.z;
      y1 += a[3 * (size_t)n] * xj.x + a[4 * (size_t)n] * xj.y + a[5 * (size_t)n] * xj.z;
      y2 += a[6 * (size_t)n] * xj.x + a[7 * (size_t)n] * xj.y + a[8 * (size_t)n] * xj.z;
    }
  }
  st3(y, row, d3(y0, y1, y2));
}

// 5 lanes per row (one per stencil row dI): 64 rows x 5 = 320 threads per block; partial sums meet in LDS.
// FUSE = 0: y = A x.   FUSE = 1: x_out = x + omega * Dinv * (r - A x)  (one damped-Jacobi sweep, ping-pong buffers).
template <int FUSE>
__global__ void __launch_bounds__(320)
k_st_spmv5(MgGrid g, const double* __restrict__ A, const double* __restrict__ x, double* __restrict__ y, const double* __restrict__ Dinv,
           const double* __restrict__ r, const double* __restrict__ omega_dev) {
  __shared__ double red[5][3][64];
  const int n = (g.N + 1) * (g.M + 1);
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;  // q = dI + 2
  const int row = blockIdx.x * 64 + lane;
  double y0 = 0, y1 = 0, y2 = 0;
  if (row < n) {
    const int I = row / (g.M + 1), J = row % (g.M + 1);
    const int I2 = I + q - 2;
    if (I2 >= 0 && I2 <= g.N) {
#pragma unroll
      for (int dJ = -2; dJ <= 2; dJ++) {
        const int J2 = J + dJ;
        if (J2 < 0 || J2 > g.M) continue;
        const int s = q * 5 + (dJ + 2);
        const double* a = A + (size_t)s * 9 * n + row;
        const d3 xj = ld3(x, I2 * (g.M + 1) + J2);
        y0 += a[0] * xj.x + a[(size_t)n] * xj.y + a[2 * (size_t)n] * xj.z;
        y1 += a[3 * (size_t)n] * xj.x + a[4 * (size_t)n] * xj.y + a[5 * (size_t)n] * xj.z;
        y2 += a[6 * (size_t)n] * xj.x + a[7 * (size_t)n] * xj.y + a[8 * (size_t)n] * xj.z;
      }
    }
  }
  red[q][0][lane] = y0; red[q][1][lane] = y1; red[q][2][lane] = y2;
  __syncthreads();
  if (q == 0 && row < n) {
#pragma unroll
    for (int k = 1; k < 5; k++) { y0 += red[k][0][lane]; y1 += red[k][1][lane]; y2 += red[k][2][lane]; }
    if (FUSE) {
      m3 D;
#pragma unroll
      for (int e = 0; e < 9; e++) D.m[e] = Dinv[9 * (size_t)row + e];
      const d3 res = ld3(r, row) - d3(y0, y1, y2);
      st3(y, row, ld3(x, row) + (*omega_dev) * m3_mulv(D, res));
    } else {
      st3(y, row, d3(y0, y1, y2));
    }
  }
}

// ---- fused variants: a V-cycle is a chain of dependent launches of 4-5 us each on a few thousand nodes, so the number
// of launches, not the arithmetic, sets its cost.
// (1) first sweep from a zero guess + residual product: x = omega Dinv r, t = A x with the neighbours' x recomputed on the fly
__global__ void __launch_bounds__(320)
k_st_first_resid(MgGrid g, const double* __restrict__ A, const double* __restrict__ Dinv, const double* __restrict__ r, const double* __restrict__ omega_dev,
                 double* __restrict__ x, double* __restrict__ t) {
  __shared__ double red[5][3][64];
  const int n = (g.N + 1) * (g.M + 1);
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int row = blockIdx.x * 64 + lane;
  const double omega = *omega_dev;
  double y0 = 0, y1 = 0, y2 = 0;
  if (row < n) {
    const int I = row / (g.M + 1), J = row % (g.M + 1);
    const int I2 = I + q - 2;
    if (I2 >= 0 && I2 <= g.N) {
#pragma unroll
      for (int dJ = -2; dJ <= 2; dJ++) {
        const int J2 = J + dJ;
        if (J2 < 0 || J2 > g.M) continue;
        const int s = q * 5 + (dJ + 2);
        const double* a = A + (size_t)s * 9 * n + row;
        const int c = I2 * (g.M + 1) + J2;
        m3 D;
#pragma unroll
        for (int e = 0; e < 9; e++) D.m[e] = Dinv[9 * (size_t)c + e];
        const d3 xj = omega * m3_mulv(D, ld3(r, c));
        y0 += a[0] * xj.x + a[(size_t)n] * xj.y + a[2 * (size_t)n] * xj.z;
        y1 += a[3 * (size_t)n] * xj.x + a[4 * (size_t)n] * xj.y + a[5 * (size_t)n] * xj.z;
        y2 += a[6 * (size_t)n] * xj.x + a[7 * (size_t)n] * xj.y + a[8 * (size_t)n] * xj.z;
      }
    }
  }
  red[q][0][lane] = y0; red[q][1][lane] = y1; red[q][2][lane] = y2;
  __syncthreads();
  if (q == 0 && row < n) {
#pragma unroll
    for (int k = 1; k < 5; k++) { y0 += red[k][0][lane]; y1 += red[k][1][lane]; y2 += red[k][2][lane]; }
    st3(t, row, d3(y0, y1, y2));
    m3 D;
#pragma unroll
    for (int e = 0; e < 9; e++) D.m[e] = Dinv[9 * (size_t)row + e];
    st3(x, row, omega * m3_mulv(D, ld3(r, row)));
  }
}

// (1b) first sweep + residual + restriction in ONE launch.  With x = omega Dinv r the restricted residual is
//   rc = P^T (r - A x) = P^T r - omega (P^T A Dinv) r,
// so the operator S = P^T A Dinv is formed once per assembly (k_st_build_ra): 49 slots per COARSE node (restriction radius 1 +
// stencil radius 2 on the fine grid), i.e. 12.25 blocks per fine node instead of the 25 of A, and the fine-level residual is never
// formed (the post-sweep recomputes A x from scratch anyway).  One launch and half the bytes less per level and cycle.
// S[(s*9+e)*nc + crow], s = (dI+3)*7 + (dJ+3), column = fine node (2I+dI, 2J+dJ)
template <typename VT>
__global__ void k_st_build_ra(MgGrid gf, const VT* __restrict__ A, const double* __restrict__ Dinv, VT* __restrict__ S) {
  const int Nc = gf.N >> 1, Mc = gf.M >> 1;
  const int nc = (Nc + 1) * (Mc + 1), nf = (gf.N + 1) * (gf.M + 1);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int crow = t % nc, s = t / nc;
  if (s >= 49) return;
  const int I = crow / (Mc + 1), J = crow % (Mc + 1);
  const int dI = s / 7 - 3, dJ = s % 7 - 3;
  const int ci = 2 * I + dI, cj = 2 * J + dJ;
  double acc[9];
#pragma unroll
  for (int e = 0; e < 9; e++) acc[e] = 0.0;
  VT* dst = S + (size_t)s * 9 * nc + crow;
  if (ci < 0 || ci > gf.N || cj < 0 || cj > gf.M) {
#pragma unroll
    for (int e = 0; e < 9; e++) dst[(size_t)e * nc] = (VT)0;
    return;
  }
#pragma unroll
  for (int di = -1; di <= 1; di++) {
    const int i = 2 * I + di, a = dI - di;
    if (i < 0 || i > gf.N || a < -2 || a > 2) continue;
#pragma unroll
    for (int dj = -1; dj <= 1; dj++) {
      const int j = 2 * J + dj, b = dJ - dj;
      if (j < 0 || j > gf.M || b < -2 || b > 2) continue;
      const double w = (di == 0 ? 1.0 : 0.5) * (dj == 0 ? 1.0 : 0.5);
      const VT* src = A + (size_t)((a + 2) * 5 + (b + 2)) * 9 * nf + (i * (gf.M + 1) + j);
#pragma unroll
      for (int e = 0; e < 9; e++) acc[e] += w * (double)src[(size_t)e * nf];
    }
  }
  const double* D = Dinv + 9 * (size_t)(ci * (gf.M + 1) + cj);
  double d[9];
#pragma unroll
  for (int e = 0; e < 9; e++) d[e] = D[e];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int cc = 0; cc < 3; cc++) dst[(size_t)(3 * r + cc) * nc] = (VT)(acc[3 * r] * d[cc] + acc[3 * r + 1] * d[3 + cc] + acc[3 * r + 2] * d[6 + cc]);
}

// ROWS coarse nodes per workgroup, 7 threads per node (one per fine row offset dI); threads 1..4 of a node also write the
// smoothed iterate x = omega Dinv r of its four fine nodes (2I + {0,1}, 2J + {0,1}), thread 0 the coarse right-hand side.
template <int ROWS, typename VT>
__global__ void __launch_bounds__(7 * ROWS)
k_st_first_restrict(MgGrid g, const VT* __restrict__ S, const double* __restrict__ Dinv, const double* __restrict__ r, const double* __restrict__ omega_dev,
                    double* __restrict__ x, double* __restrict__ rc) {
  __shared__ double red[7][3][ROWS];
  const int Nc = g.N >> 1, Mc = g.M >> 1;
  const int nc = (Nc + 1) * (Mc + 1);
  const int lane = threadIdx.x % ROWS, q = threadIdx.x / ROWS;
  const int crow = blockIdx.x * ROWS + lane;
  const double omega = *omega_dev;
  double y0 = 0, y1 = 0, y2 = 0;
  int I = 0, J = 0;
  if (crow < nc) {
    I = crow / (Mc + 1); J = crow % (Mc + 1);
    const int i2 = 2 * I + q - 3;
    if (i2 >= 0 && i2 <= g.N) {
#pragma unroll
      for (int dJ = -3; dJ <= 3; dJ++) {
        const int j2 = 2 * J + dJ;
        if (j2 < 0 || j2 > g.M) continue;
        const VT* a = S + (size_t)(q * 7 + dJ + 3) * 9 * nc + crow;
        const d3 rj = ld3(r, i2 * (g.M + 1) + j2);
        y0 += a[0] * rj.x + a[(size_t)nc] * rj.y + a[2 * (size_t)nc] * rj.z;
        y1 += a[3 * (size_t)nc] * rj.x + a[4 * (size_t)nc] * rj.y + a[5 * (size_t)nc] * rj.z;
        y2 += a[6 * (size_t)nc] * rj.x + a[7 * (size_t)nc] * rj.y + a[8 * (size_t)nc] * rj.z;
      }
    }
  }
  // own work of the node's threads that does not depend on the partial sums: issued before the barrier
  d3 own = d3();
  int f_own = -1;
  if (crow < nc) {
    if (q == 0) {  // P^T r
#pragma unroll
      for (int di = -1; di <= 1; di++) {
        const int i = 2 * I + di;
        if (i < 0 || i > g.N) continue;
#pragma unroll
        for (int dj = -1; dj <= 1; dj++) {
          const int j = 2 * J + dj;
          if (j < 0 || j > g.M) continue;
          own = own + ((di == 0 ? 1.0 : 0.5) * (dj == 0 ? 1.0 : 0.5)) * ld3(r, i * (g.M + 1) + j);
        }
      }
    } else if (q <= 4) {
      const int i = 2 * I + ((q - 1) >> 1), j = 2 * J + ((q - 1) & 1);
      if (i <= g.N && j <= g.M) {
        f_own = i * (g.M + 1) + j;
        m3 D;
#pragma unroll
        for (int e = 0; e < 9; e++) D.m[e] = Dinv[9 * (size_t)f_own + e];
        own = omega * m3_mulv(D, ld3(r, f_own));
      }
    }
  }
  red[q][0][lane] = y0; red[q][1][lane] = y1; red[q][2][lane] = y2;
  __syncthreads();
  if (crow >= nc) return;
  if (q == 0) {
#pragma unroll
    for (int k = 1; k < 7; k++) { y0 += red[k][0][lane]; y1 += red[k][1][lane]; y2 += red[k][2][lane]; }
    st3(rc, crow, own - omega * d3(y0, y1, y2));
  } else if (f_own >= 0) {
    st3(x, f_own, own);
  }
}

// (x + P xc) at fine node (i, j) of grid gf
TSL_DEV d3 st_prolonged(MgGrid gf, const double* __restrict__ x, const double* __restrict__ xc, int i, int j) {
  const int Mc = gf.M >> 1;
  d3 acc = ld3(x, i * (gf.M + 1) + j);
#pragma unroll
  for (int a = 0; a < 2; a++) {
    double wa; const int I = mg_coarse(i, a, wa);
    if (wa == 0.0) continue;
#pragma unroll
    for (int b = 0; b < 2; b++) {
      double wb; const int J = mg_coarse(j, b, wb);
      if (wb == 0.0) continue;
      acc = acc + (wa * wb) * ld3(xc, I * (Mc + 1) + J);
    }
  }
  return acc;
}

// (2) prolongation + first post-smoothing sweep: xt = x + P xc (never stored), y = xt + omega Dinv (r - A xt).
// The 25 neighbours of the workgroup's 64 consecutive nodes lie in five runs of 68 consecutive linear indices (one per stencil row
// dI; a linear index that wraps into the adjacent grid row is never used: those slots fail the bounds test), so xt is formed ONCE
// per run entry and staged in LDS (340 prolongations per workgroup instead of 1600 gathers of up to five vectors each).
template <typename VT>
__global__ void __launch_bounds__(320)
k_st_prolong_sweep(MgGrid g, const VT* __restrict__ A, const double* __restrict__ x, const double* __restrict__ xc, double* __restrict__ y,
                   const double* __restrict__ Dinv, const double* __restrict__ r, const double* __restrict__ omega_dev) {
  __shared__ double red[5][3][64];
  __shared__ double xt[5][68][3];
  const int n = (g.N + 1) * (g.M + 1);
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int row0 = blockIdx.x * 64;
  const int row = row0 + lane;
  // operands of the final combine that do not depend on the staged vector: requested first
  m3 D;
  d3 rv = d3();
  if (q == 0 && row < n) {
#pragma unroll
    for (int e = 0; e < 9; e++) D.m[e] = Dinv[9 * (size_t)row + e];
    rv = ld3(r, row);
  }
  {
    const int base = row0 + (q - 2) * (g.M + 1) - 2;
    for (int k = lane; k < 68; k += 64) {
      const int f = base + k;
      d3 v = d3();
      if (f >= 0 && f < n) v = st_prolonged(g, x, xc, f / (g.M + 1), f % (g.M + 1));
      xt[q][k][0] = v.x; xt[q][k][1] = v.y; xt[q][k][2] = v.z;
    }
  }
  __syncthreads();
  double y0 = 0, y1 = 0, y2 = 0;
  if (row < n) {
    const int I = row / (g.M + 1), J = row % (g.M + 1);
    const int I2 = I + q - 2;
    if (I2 >= 0 && I2 <= g.N) {
#pragma unroll
      for (int dJ = -2; dJ <= 2; dJ++) {
        const int J2 = J + dJ;
        if (J2 < 0 || J2 > g.M) continue;
        const int s = q * 5 + (dJ + 2);
        const VT* a = A + (size_t)s * 9 * n + row;
        const double* xv = xt[q][lane + dJ + 2];
        const double x0 = xv[0], x1 = xv[1], x2 = xv[2];
        y0 += a[0] * x0 + a[(size_t)n] * x1 + a[2 * (size_t)n] * x2;
        y1 += a[3 * (size_t)n] * x0 + a[4 * (size_t)n] * x1 + a[5 * (size_t)n] * x2;
        y2 += a[6 * (size_t)n] * x0 + a[7 * (size_t)n] * x1 + a[8 * (size_t)n] * x2;
      }
    }
  }
  red[q][0][lane] = y0; red[q][1][lane] = y1; red[q][2][lane] = y2;
  __syncthreads();
  if (q == 0 && row < n) {
#pragma unroll
    for (int k = 1; k < 5; k++) { y0 += red[k][0][lane]; y1 += red[k][1][lane]; y2 += red[k][2][lane]; }
    const d3 res = rv - d3(y0, y1, y2);
    const double* xo = xt[2][lane + 2];
    st3(y, row, d3(xo[0], xo[1], xo[2]) + (*omega_dev) * m3_mulv(D, res));
  }
}

// Exact solve on the last level of the hierarchy.  Eight damped-Jacobi sweeps leave most of the smooth error there (measured on
// cfg4: 436 PCG iterations per solve with 8 sweeps on the 64-node level, 392 with 16, 322 with the exact solve) and cost 15 us per
// cycle; the dense inverse is rebuilt at every assembly by the blocked Gauss-Jordan below and applied as one dense product per cycle.
// (A single-workgroup Gauss-Jordan with the 192 x 192 matrix in registers was tried for the 64-node level: 1.1 ms per call, 192
// barrier-separated pivots, against 0.2 ms for the blocked version.)

// x = Cinv r: one wave per two rows (row-major rows are contiguous: coalesced), 8 rows per workgroup, every load independent.
// The inverse is stored symmetrised in single precision (it is a preconditioner component; the inversion itself runs in double).
__global__ void __launch_bounds__(256)
k_st_coarse_apply(int n3, const float* __restrict__ Cinv, const double* __restrict__ r, double* __restrict__ out) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i0 = blockIdx.x * 8 + 2 * w, i1 = i0 + 1;
  if (i0 >= n3) return;
  const float* c0 = Cinv + (size_t)i0 * n3;
  const float* c1 = Cinv + (size_t)(i1 < n3 ? i1 : i0) * n3;
  double s0 = 0, s1 = 0;
#pragma unroll 4
  for (int jj = lane; jj < n3; jj += 64) { const double rj = r[jj]; s0 += (double)c0[jj] * rj; s1 += (double)c1[jj] * rj; }
  s0 = wave_sum(s0); s1 = wave_sum(s1);
  if (lane == 0) { out[i0] = s0; if (i1 < n3) out[i1] = s1; }
}

// ---- dense level with more than 192 unknowns (e.g. 15 x 15 nodes: 675): blocked in-place Gauss-Jordan, two launches per block
// of GJ_B pivots instead of one per pivot (the pivot block of step k + 1 is inverted by one wave inside the update kernel of step k).  The matrix is n x n (ld = n, n a multiple of GJ_B; padding rows are identity rows).
// Block step k with K = [k B, k B + B):   P = inv(A_KK);  R' = P A_K,: ;  C = A_:,K (saved);
//   A_ij -= C_i R'_j (i, j outside K),  A_Kj = R'_j,  A_iK = -C_i P,  A_KK = P.
#define GJ_B 32
// expands the 25-slot operator into the dense matrix (zeroed by the caller): one thread per (node, slot, 3x3 entry)
__global__ void k_st_dense_build(MgGrid g, int ld, const double* __restrict__ A, double* __restrict__ D) {
  const int n = (g.N + 1) * (g.M + 1);
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)n * 225) return;
  const int row = (int)(t % n), se = (int)(t / n), s = se / 9, e = se % 9;
  const int I = row / (g.M + 1), J = row % (g.M + 1);
  const int I2 = I + s / 5 - 2, J2 = J + s % 5 - 2;
  if (I2 < 0 || I2 > g.N || J2 < 0 || J2 > g.M) return;
  const double v = A[(size_t)se * n + row];
  D[(size_t)(3 * row + e / 3) * ld + 3 * (I2 * (g.M + 1) + J2) + e % 3] = v;
}
// empty rows (frozen unknowns) and the padding become identity rows
__global__ void k_st_dense_diag(int n3, int ld, double* __restrict__ D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ld) return;
  if (i >= n3 || D[(size_t)i * ld + i] == 0.0) D[(size_t)i * ld + i] = 1.0;
}
// In-place Gauss-Jordan inversion of one GJ_B x GJ_B tile held in LDS by ONE wave (lane = (column, row half); a single wave runs in
// lockstep and its LDS accesses complete in order, so the 32 pivot steps need no workgroup barriers: ~5 us against 25 us for the
// 256-thread version with two barriers per pivot).  Called by wave 0 of a workgroup; the caller synchronises before and after.
TSL_DEV void gj_invert_tile_wave(double (*T)[GJ_B + 1], int* __restrict__ bad) {
  // lane = (column c, row half h) keeps its 16 elements in registers; per pivot only the pivot row and column go through LDS
  // (rowb / colb live in the last two rows' padding-free scratch below), the loop is fully unrolled so that register indices are static
  __shared__ double colb[GJ_B], rowb[GJ_B];
  const int lane = threadIdx.x & 63;
  const int c = lane & 31, h = lane >> 5;
  double a[GJ_B / 2];
#pragma unroll
  for (int m = 0; m < GJ_B / 2; m++) a[m] = T[h * (GJ_B / 2) + m][c];
  bool isbad = false;
#pragma unroll
  for (int p = 0; p < GJ_B; p++) {
    if (c == p) {
#pragma unroll
      for (int m = 0; m < GJ_B / 2; m++) colb[h * (GJ_B / 2) + m] = a[m];
    }
    if (h == p / (GJ_B / 2)) rowb[c] = a[p % (GJ_B / 2)];
    __builtin_amdgcn_wave_barrier();
    const double piv = rowb[p];
    const double rj = rowb[c];
    double ci[GJ_B / 2];
#pragma unroll
    for (int m = 0; m < GJ_B / 2; m++) ci[m] = colb[h * (GJ_B / 2) + m];
    __builtin_amdgcn_wave_barrier();
    isbad = isbad || !(piv > 0.0);
    const double ip = 1.0 / piv;
    const double rs = rj * ip;
#pragma unroll
    for (int m = 0; m < GJ_B / 2; m++) {
      const int i = h * (GJ_B / 2) + m;
      double v;
      if (i == p) v = (c == p) ? ip : rs;
      else if (c == p) v = -ci[m] * ip;
      else v = a[m] - ci[m] * rs;
      a[m] = v;
    }
  }
#pragma unroll
  for (int m = 0; m < GJ_B / 2; m++) T[h * (GJ_B / 2) + m][c] = a[m];
  if (lane == 0 && isbad) bad[0] = 1;
}
// inverse of the first pivot block -> Pout (the later ones come out of k_gj_update)
__global__ void __launch_bounds__(256)
k_gj_pivot0(int ld, const double* __restrict__ D, double* __restrict__ Pout, int* __restrict__ bad) {
  __shared__ double T[GJ_B][GJ_B + 1];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int q = 0; q < 4; q++) T[ty + 8 * q][tx] = D[(size_t)(ty + 8 * q) * ld + tx];
  __syncthreads();
  if (threadIdx.x < 64) gj_invert_tile_wave(T, bad);
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; q++) Pout[(ty + 8 * q) * GJ_B + tx] = T[ty + 8 * q][tx];
}
// Only the tiles on and above the block diagonal are stored and updated (half the traffic and flops): with P the set of pivot
// blocks already processed, the Gauss-Jordan iterate satisfies M_ji = s M_ij^T, s = -1 when exactly one of the two blocks is in P,
// +1 otherwise (scalar check: pivot k turns a_kj into a_kj / p and a_ik into -a_ik / p).
// panel kernel of block step k: workgroup b reads the one stored tile that couples chunk b with the pivot block and writes
// R'[:, chunk b] = P M[K, chunk b] and the saved column panel C[chunk b, :] = M[chunk b, K]
// (P = inverse of the pivot block, from k_gj_pivot0 / the previous k_gj_update)
__global__ void __launch_bounds__(256)
k_gj_panel(int ld, int k, const double* __restrict__ D, const double* __restrict__ Pin, double* __restrict__ Rn, double* __restrict__ Cs) {
  __shared__ double P[GJ_B][GJ_B + 1];
  __shared__ double T[GJ_B][GJ_B + 1];   // M[K, chunk b]
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int b = blockIdx.x;
  const int k0 = k * GJ_B, b0 = b * GJ_B;
  const int r0 = b < k ? b0 : k0, c0 = b < k ? k0 : b0;   // stored tile (min, max)
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int i = ty + 8 * q;
    P[i][tx] = Pin[i * GJ_B + tx];
    const double sv = D[(size_t)(r0 + i) * ld + c0 + tx];
    // b >= k: stored tile is M[K, b] itself and M[b, K] = +its transpose (both blocks unprocessed);
    // b <  k: stored tile is M[b, K] and M[K, b] = -its transpose (b processed, k not)
    if (b >= k) { T[i][tx] = sv; Cs[(size_t)(b0 + tx) * GJ_B + i] = sv; }
    else { T[tx][i] = -sv; Cs[(size_t)(b0 + i) * GJ_B + tx] = sv; }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int i = ty + 8 * q;
    double acc = 0;
#pragma unroll 8
    for (int m = 0; m < GJ_B; m++) acc += P[i][m] * T[m][tx];
    Rn[(size_t)i * ld + b0 + tx] = acc;
  }
}
// update kernel of block step k: one 32 x 32 tile of the matrix per workgroup; the workgroup of tile (k+1, k+1) also inverts its
// updated tile (the next pivot block) into Pnext
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8)))  // the rare inversion path may spill, the tile path must keep its occupancy
k_gj_update(int ld, int k, double* __restrict__ D, const double* __restrict__ Rn, const double* __restrict__ Cs, const double* __restrict__ Pin, double* __restrict__ Pnext,
            int* __restrict__ bad) {
  __shared__ double Ct[GJ_B][GJ_B + 1];
  __shared__ double Rt[GJ_B][GJ_B + 1];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  int bi = blockIdx.y, bj = blockIdx.x;
  // the tile of the next pivot block carries a ~30 us single-wave inversion: it swaps places with tile (0, 0) so that it is
  // dispatched first and the inversion overlaps with the other tiles
  if ((k + 1) * GJ_B < ld) {
    if (bi == 0 && bj == 0) bi = bj = k + 1;
    else if (bi == k + 1 && bj == k + 1) bi = bj = 0;
  }
  if (bi > bj) return;  // tiles below the block diagonal are implied by the ones above (see k_gj_panel)
  const int i0 = bi * GJ_B, j0 = bj * GJ_B;
  if (bi == k && bj == k) {
#pragma unroll
    for (int q = 0; q < 4; q++) D[(size_t)(i0 + ty + 8 * q) * ld + j0 + tx] = Pin[(ty + 8 * q) * GJ_B + tx];
    return;
  }
  if (bi == k) {  // row panel: R'
#pragma unroll
    for (int q = 0; q < 4; q++) D[(size_t)(i0 + ty + 8 * q) * ld + j0 + tx] = Rn[(size_t)(ty + 8 * q) * ld + j0 + tx];
    return;
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {
    Ct[ty + 8 * q][tx] = Cs[(size_t)(i0 + ty + 8 * q) * GJ_B + tx];
    Rt[ty + 8 * q][tx] = (bj == k) ? Pin[(ty + 8 * q) * GJ_B + tx] : Rn[(size_t)(ty + 8 * q) * ld + j0 + tx];
  }
  __syncthreads();
  // rank-32 update of the tile on the f64 matrix cores: wave (wi, wj) owns a 16 x 16 quadrant, eight v_mfma_f64_16x16x4_f64 per
  // quadrant (A: lane l holds C[l & 15][4 kk + (l >> 4)], B: R'[4 kk + (l >> 4)][l & 15]; result reg r of lane l is element
  // (row (l >> 4) + 4 r, column l & 15)).  The vector-FMA version of this loop was LDS-bound (two ds_read per FMA).
  typedef double d4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wi = w >> 1, wj = w & 1;
  const int lr = lane & 15, lk = lane >> 4;
  d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < GJ_B / 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ct[16 * wi + lr][4 * kk + lk], Rt[4 * kk + lk][16 * wj + lr], acc, 0, 0, 0);
  const bool next_pivot = (bi == k + 1 && bj == k + 1);
  if (next_pivot) __syncthreads();  // all quadrants done with Ct before it receives the updated tile
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = 16 * wi + lk + 4 * r, col = 16 * wj + lr;
    double* d = D + (size_t)(i0 + row) * ld + j0 + col;
    const double v = (bj == k) ? -acc[r] : *d - acc[r];
    *d = v;
    if (next_pivot) Ct[row][col] = v;
  }
  if (next_pivot) {  // next pivot block
    __syncthreads();
    if (threadIdx.x < 64) gj_invert_tile_wave(Ct, bad);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) Pnext[(ty + 8 * q) * GJ_B + tx] = Ct[ty + 8 * q][tx];
  }
}
// after the last block step: symmetrise into the inverse buffer (row stride n3), or the block-Jacobi fallback on a bad pivot
__global__ void k_gj_finish(int n3, int ld, const double* __restrict__ D, const int* __restrict__ bad, const double* __restrict__ Dinv, float* __restrict__ Cinv) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)n3 * n3) return;
  const int i = (int)(t / n3), j = (int)(t % n3);
  double v;
  if (!bad[0]) {
    const int ti = i / GJ_B, tj = j / GJ_B;  // only tiles on / above the block diagonal hold the result
    v = ti == tj ? 0.5 * (D[(size_t)i * ld + j] + D[(size_t)j * ld + i]) : ti < tj ? D[(size_t)i * ld + j] : D[(size_t)j * ld + i];
  }
  else v = (i / 3 == j / 3) ? 0.5 * Dinv[9 * (size_t)(i / 3) + 3 * (i % 3) + j % 3] : 0.0;
  Cinv[t] = (float)v;
}

// (3) the whole coarsest level (<= 64 nodes) in one workgroup: x = omega Dinv r, then sweeps - 1 damped-Jacobi sweeps with the
// iterate in LDS (ping-pong) and the 25-slot operator streamed from L2
__global__ void __launch_bounds__(320)
k_st_coarse(MgGrid g, const double* __restrict__ A, const double* __restrict__ Dinv, const double* __restrict__ r, const double* __restrict__ omega_dev, int sweeps,
            double* __restrict__ out) {
  __shared__ double red[5][3][64];
  __shared__ double xs[2][3][64];
  const int n = (g.N + 1) * (g.M + 1);
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int row = lane;
  const double omega = *omega_dev;
  m3 D;
  d3 rv = d3();
  const bool own = q == 0 && row < n;
  if (own) {
#pragma unroll
    for (int e = 0; e < 9; e++) D.m[e] = Dinv[9 * (size_t)row + e];
    rv = ld3(r, row);
    const d3 x0 = omega * m3_mulv(D, rv);
    xs[0][0][lane] = x0.x; xs[0][1][lane] = x0.y; xs[0][2][lane] = x0.z;
  }
  const int I = row / (g.M + 1), J = row % (g.M + 1);
  const int I2 = I + q - 2;
  int cur = 0;
  for (int it = 1; it < sweeps; it++) {
    __syncthreads();
    double y0 = 0, y1 = 0, y2 = 0;
    if (row < n && I2 >= 0 && I2 <= g.N) {
#pragma unroll
      for (int dJ = -2; dJ <= 2; dJ++) {
        const int J2 = J + dJ;
        if (J2 < 0 || J2 > g.M) continue;
        const int s = q * 5 + (dJ + 2);
        const double* a = A + (size_t)s * 9 * n + row;
        const int c = I2 * (g.M + 1) + J2;
        const double xx = xs[cur][0][c], xy = xs[cur][1][c], xz = xs[cur][2][c];
        y0 += a[0] * xx + a[(size_t)n] * xy + a[2 * (size_t)n] * xz;
        y1 += a[3 * (size_t)n] * xx + a[4 * (size_t)n] * xy + a[5 * (size_t)n] * xz;
        y2 += a[6 * (size_t)n] * xx + a[7 * (size_t)n] * xy + a[8 * (size_t)n] * xz;
      }
    }
    red[q][0][lane] = y0; red[q][1][lane] = y1; red[q][2][lane] = y2;
    __syncthreads();
    if (own) {
#pragma unroll
      for (int k = 1; k < 5; k++) { y0 += red[k][0][lane]; y1 += red[k][1][lane]; y2 += red[k][2][lane]; }
      const d3 xn = d3(xs[cur][0][lane], xs[cur][1][lane], xs[cur][2][lane]) + omega * m3_mulv(D, rv - d3(y0, y1, y2));
      xs[cur ^ 1][0][lane] = xn.x; xs[cur ^ 1][1][lane] = xn.y; xs[cur ^ 1][2][lane] = xn.z;
    }
    cur ^= 1;
  }
  if (own) st3(out, row, d3(xs[cur][0][lane], xs[cur][1][lane], xs[cur][2][lane]));
}

// Level-0 post-smoothing in one launch: workgroups [0, gb) do the damped block-Jacobi update of the rows that are not served by a
// dense body block (those have a zero Dinv block and are left alone), workgroups [gb, gb + bodies) the dense body update
// z_b += Binv (r - t)_b.  Partials of rdot . z_new: part[0, gb) and part[gb, ...).
__global__ void __launch_bounds__(256)
k_post_smooth(BodyDenseArgs B, const float* __restrict__ Binv, int gb, int n, const double* __restrict__ Dinv, const double* __restrict__ r, const double* __restrict__ t,
              const double* __restrict__ omega_dev, double* __restrict__ x, const double* __restrict__ rdot, double* __restrict__ part) {
  if ((int)blockIdx.x >= gb) {
    body_apply_block(B, Binv, 1, r, t, x, rdot, part ? part + gb : nullptr, (int)blockIdx.x - gb, true);
    return;
  }
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const double omega = *omega_dev;
  double acc = 0;
  if (p < n) {
    m3 D;
    bool any = false;
    // all operands are requested together with the block (the test on the block would put their round trip behind its own)
    const d3 xo = ld3(x, p), ro = ld3(r, p), to = ld3(t, p);
    const d3 rd = rdot ? ld3(rdot, p) : d3();
#pragma unroll
    for (int e = 0; e < 9; e++) { D.m[e] = Dinv[9 * (size_t)p + e]; any = any || D.m[e] != 0.0; }
    if (any) {
      const d3 xn = xo + omega * m3_mulv(D, ro - to);
      st3(x, p, xn);
      if (rdot) acc = dot(rd, xn);
    }
  }
  if (part) {
    __shared__ double s1[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s1[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = s1[0] + s1[1] + s1[2] + s1[3];
  }
}

// x = omega * Dinv * r   (first smoothing sweep from a zero guess)
__global__ void k_mg_jacobi_first(int n, const double* __restrict__ Dinv, const double* __restrict__ r, const double* __restrict__ omega_dev, double* __restrict__ x) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const double omega = *omega_dev;
  m3 D;
#pragma unroll
  for (int e = 0; e < 9; e++) D.m[e] = Dinv[9 * (size_t)p + e];
  st3(x, p, omega * m3_mulv(D, ld3(r, p)));
}

// x += omega * Dinv * (r - t), t = A x_old  (further sweeps); optional partial dot(rdot, x_new) per block
__global__ void __launch_bounds__(256)
k_mg_jacobi_next(int n, const double* __restrict__ Dinv, const double* __restrict__ r, const double* __restrict__ t, const double* __restrict__ omega_dev, double* __restrict__ x,
                 const double* __restrict__ rdot, double* __restrict__ part) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const double omega = *omega_dev;
  double acc = 0;
  if (p < n) {
    m3 D;
#pragma unroll
    for (int e = 0; e < 9; e++) D.m[e] = Dinv[9 * (size_t)p + e];
    const d3 xn = ld3(x, p) + omega * m3_mulv(D, ld3(r, p) - ld3(t, p));
    st3(x, p, xn);
    if (rdot) acc = dot(ld3(rdot, p), xn);
  }
  if (part) {
    __shared__ double s1[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s1[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = s1[0] + s1[1] + s1[2] + s1[3];
  }
}

// rc = P^T (r - t) from a fine STENCIL level (t may be null)
__global__ void k_st_restrict(MgGrid gf, const double* __restrict__ r, const double* __restrict__ t, double* __restrict__ rc) {
  const int Nc = gf.N >> 1, Mc = gf.M >> 1;
  const int nc = (Nc + 1) * (Mc + 1);
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= nc) return;
  const int I = row / (Mc + 1), J = row % (Mc + 1);
  d3 acc = d3();
#pragma unroll
  for (int di = -1; di <= 1; di++) {
    const int i = 2 * I + di;
    if (i < 0 || i > gf.N) continue;
#pragma unroll
    for (int dj = -1; dj <= 1; dj++) {
      const int j = 2 * J + dj;
      if (j < 0 || j > gf.M) continue;
      const double w = (di == 0 ? 1.0 : 0.5) * (dj == 0 ? 1.0 : 0.5);
      const int f = i * (gf.M + 1) + j;
      d3 v = ld3(r, f);
      if (t) v = v - ld3(t, f);
      acc = acc + w * v;
    }
  }
  st3(rc, row, acc);
}

// x_f += P x_c on a fine STENCIL level
__global__ void k_st_prolong_add(MgGrid gf, const double* __restrict__ xc, double* __restrict__ x) {
  const int n = (gf.N + 1) * (gf.M + 1);
  const int Mc = gf.M >> 1;
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  const int i = f / (gf.M + 1), j = f % (gf.M + 1);
  d3 acc = d3();
#pragma unroll
  for (int a = 0; a < 2; a++) {
    double wa; const int I = mg_coarse(i, a, wa);
    if (wa == 0.0) continue;
#pragma unroll
    for (int b = 0; b < 2; b++) {
      double wb; const int J = mg_coarse(j, b, wb);
      if (wb == 0.0) continue;
      acc = acc + (wa * wb) * ld3(xc, I * (Mc + 1) + J);
    }
  }
  st3(x, f, ld3(x, f) + acc);
}

// same two transfers between level 0 (global permuted vectors) and level 1 of one cloth
__global__ void k_mg_restrict0(MgGrid gf, int v_offset, const int* __restrict__ rowpos, const double* __restrict__ r, const double* __restrict__ t,
                               double* __restrict__ rc) {
  const int Nc = gf.N >> 1, Mc = gf.M >> 1;
  const int nc = (Nc + 1) * (Mc + 1);
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= nc) return;
  const int I = row / (Mc + 1), J = row % (Mc + 1);
  d3 acc = d3();
#pragma unroll
  for (int di = -1; di <= 1; di++) {
    const int i = 2 * I + di;
    if (i < 0 || i > gf.N) continue;
#pragma unroll
    for (int dj = -1; dj <= 1; dj++) {
      const int j = 2 * J + dj;
      if (j < 0 || j > gf.M) continue;
      const double w = (di == 0 ? 1.0 : 0.5) * (dj == 0 ? 1.0 : 0.5);
      const int p = rowpos[v_offset + i * (gf.M + 1) + j];
      acc = acc + w * (ld3(r, p) - ld3(t, p));
    }
  }
  st3(rc, row, acc);
}
__global__ void k_mg_prolong0_add(MgGrid gf, int v_offset, const int* __restrict__ rowpos, const double* __restrict__ xc, double* __restrict__ x) {
  const int n = (gf.N + 1) * (gf.M + 1);
  const int Mc = gf.M >> 1;
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  const int i = f / (gf.M + 1), j = f % (gf.M + 1);
  d3 acc = d3();
#pragma unroll
  for (int a = 0; a < 2; a++) {
    double wa; const int I = mg_coarse(i, a, wa);
    if (wa == 0.0) continue;
#pragma unroll
    for (int b = 0; b < 2; b++) {
      double wb; const int J = mg_coarse(j, b, wb);
      if (wb == 0.0) continue;
      acc = acc + (wa * wb) * ld3(xc, I * (Mc + 1) + J);
    }
  }
  const int p = rowpos[v_offset + f];
  st3(x, p, ld3(x, p) + acc);
}

// ---- Galerkin products
TSL_DEV void galerkin_scatter(MgGrid gf, int i, int j, int i2, int j2, const double* B, double* __restrict__ Ac, int nc) {
  const int Mc = gf.M >> 1;
#pragma unroll
  for (int a = 0; a < 2; a++) {
    double wa; const int I = mg_coarse(i, a, wa);
    if (wa == 0.0) continue;
#pragma unroll
    for (int b = 0; b < 2; b++) {
      double wb; const int J = mg_coarse(j, b, wb);
      if (wb == 0.0) continue;
      const int row = I * (Mc + 1) + J;
#pragma unroll
      for (int a2 = 0; a2 < 2; a2++) {
        double wa2; const int I2 = mg_coarse(i2, a2, wa2);
        if (wa2 == 0.0) continue;
#pragma unroll
        for (int b2 = 0; b2 < 2; b2++) {
          double wb2; const int J2 = mg_coarse(j2, b2, wb2);
          if (wb2 == 0.0) continue;
          const int s = (I2 - I + 2) * 5 + (J2 - J + 2);
          const double w = wa * wb * wa2 * wb2;
          double* dst = Ac + (size_t)s * 9 * nc + row;
#pragma unroll
          for (int e = 0; e < 9; e++) atomicAdd(dst + (size_t)e * nc, w * B[e]);
        }
      }
    }
  }
}

// level 0 (SELL-64, masked values) -> level 1 of the cloth occupying global vertices [v_offset, v_offset + (N+1)(M+1))
__global__ void k_galerkin0(MgGrid gf, int v_offset, int NV, int n_slices, const int* __restrict__ slice_off, const int* __restrict__ slice_len,
                            const int* __restrict__ colidx, const int* __restrict__ perm, const double* __restrict__ vals, double* __restrict__ Ac) {
  const int slice = blockIdx.x, lane = threadIdx.x & 63;
  if (slice >= n_slices) return;
  const int p = slice * 64 + lane;
  if (p >= NV) return;
  const int nf = (gf.N + 1) * (gf.M + 1);
  const int v = perm[p] - v_offset;
  if (v < 0 || v >= nf) return;
  const int i = v / (gf.M + 1), j = v % (gf.M + 1);
  const int nc = ((gf.N >> 1) + 1) * ((gf.M >> 1) + 1);
  const int off = slice_off[slice], len = slice_len[slice];
  for (int k = threadIdx.x >> 6; k < len; k += (blockDim.x >> 6)) {
    const int c = colidx[off + 64 * k + lane];
    const int u = perm[c] - v_offset;
    if (u < 0 || u >= nf) continue;
    const int i2 = u / (gf.M + 1), j2 = u % (gf.M + 1);
    if (abs(i2 - i) > 2 || abs(j2 - j) > 2) continue;  // padded slots point at row 0/1 with zero values
    const size_t base = ((size_t)off + 64 * (size_t)k) * 9 + lane;
    double B[9];
    bool nz = false;
#pragma unroll
    for (int e = 0; e < 9; e++) { B[e] = vals[base + 64 * e]; nz |= (B[e] != 0.0); }
    if (nz) galerkin_scatter(gf, i, j, i2, j2, B, Ac, nc);
  }
}

// extra diagonal blocks of the fine level (masked contact contribution, permuted row order) into level 1
__global__ void k_galerkin0_diag(MgGrid gf, int v_offset, const int* __restrict__ rowpos, const double* __restrict__ cdiag, double* __restrict__ Ac) {
  const int nf = (gf.N + 1) * (gf.M + 1);
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nf) return;
  const int i = f / (gf.M + 1), j = f % (gf.M + 1);
  const int nc = ((gf.N >> 1) + 1) * ((gf.M >> 1) + 1);
  double B[9];
  bool nz = false;
  const size_t p = (size_t)rowpos[v_offset + f];
#pragma unroll
  for (int e = 0; e < 9; e++) { B[e] = cdiag[9 * p + e]; nz |= (B[e] != 0.0); }
  if (nz) galerkin_scatter(gf, i, j, i, j, B, Ac, nc);
}

// stencil level l -> l+1
__global__ void k_galerkin_st(MgGrid gf, const double* __restrict__ Af, double* __restrict__ Ac) {
  const int nf = (gf.N + 1) * (gf.M + 1);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = t % nf, s = t / nf;
  if (s >= 25) return;
  const int i = row / (gf.M + 1), j = row % (gf.M + 1);
  const int i2 = i + s / 5 - 2, j2 = j + s % 5 - 2;
  if (i2 < 0 || i2 > gf.N || j2 < 0 || j2 > gf.M) return;
  const int nc = ((gf.N >> 1) + 1) * ((gf.M >> 1) + 1);
  double B[9];
  bool nz = false;
#pragma unroll
  for (int e = 0; e < 9; e++) { B[e] = Af[((size_t)s * 9 + e) * nf + row]; nz |= (B[e] != 0.0); }
  if (nz) galerkin_scatter(gf, i, j, i2, j2, B, Ac, nc);
}

// Dinv of a stencil level (centre slot 12)
__global__ void k_st_diag_inv(int n, const double* __restrict__ A, double* __restrict__ Dinv) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  m3 D;
#pragma unroll
  for (int e = 0; e < 9; e++) D.m[e] = A[((size_t)12 * 9 + e) * n + row];
  const m3 Di = m3_inv(D);
#pragma unroll
  for (int e = 0; e < 9; e++) Dinv[9 * (size_t)row + e] = Di.m[e];
}

// x (+)= alpha-free helpers on level-0 vectors
__global__ void k_part_dot(int n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ part) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  double acc = (p < n) ? dot(ld3(a, p), ld3(b, p)) : 0.0;
  __shared__ double s1[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s1[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = s1[0] + s1[1] + s1[2] + s1[3];
}

// ---- damping factor from a power iteration on D^-1 A (device resident, no host round trip)
// v0: deterministic pseudo-random start vector
__global__ void k_pi_init(int n, double* __restrict__ v) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const double a = sin(12.9898 * p + 1.0) * 43758.5453, b = sin(78.233 * p + 2.0) * 12345.678, c = sin(39.425 * p + 3.0) * 9876.543;
  st3(v, p, d3(a - floor(a) - 0.5, b - floor(b) - 0.5, c - floor(c) - 0.5));
}
// w = Dinv t ; part[block] = |w|^2
__global__ void __launch_bounds__(256)
k_pi_apply(int n, const double* __restrict__ Dinv, const double* __restrict__ t, double* __restrict__ w, double* __restrict__ part) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  double acc = 0;
  if (p < n) {
    m3 D;
#pragma unroll
    for (int e = 0; e < 9; e++) D.m[e] = Dinv[9 * (size_t)p + e];
    const d3 wv = m3_mulv(D, ld3(t, p));
    st3(w, p, wv);
    acc = dot(wv, wv);
  }
  __shared__ double s1[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s1[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = s1[0] + s1[1] + s1[2] + s1[3];
}
// norm2[k] = sum(part); on the last call omega = c_omega / sqrt(norm2[k] / norm2[k-1])
__global__ void k_pi_finish(const double* __restrict__ part, int nparts, double* __restrict__ norm2, int k, int last, double c_omega, double omega_max, double* __restrict__ omega_out) {
  __shared__ double sm[8];
  const double t = block_reduce_partials(part, nparts, sm);
  if (threadIdx.x == 0) {
    norm2[k] = t;
    if (last) {
      const double lam = sqrt(t / norm2[k - 1]);
      double om = (lam > 0.0 && isfinite(lam)) ? c_omega / lam : omega_max;
      omega_out[0] = fmin(om, omega_max);
      omega_out[1] = lam;
    }
  }
}
