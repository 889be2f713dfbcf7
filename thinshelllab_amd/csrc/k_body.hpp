// Exact diagonal blocks for the small FEM bodies (ball, tactile pads) inside the preconditioner.
//
// The volumetric bodies of the task scenes have a few hundred vertices each but are by far the stiffest part of the
// system (E = 3e5..5e5 against m/dt^2): with point-block Jacobi on their rows they dictate the PCG iteration count
// (measured: 85-110 iterations per solve against 25-30 with the bodies frozen).  Each body block
// H_bb = (static matrix + contact blocks restricted to the body) is inverted densely once per time step (in-place
// Gauss-Jordan, one launch per pivot over all bodies, fp64), stored symmetric in fp32, and applied as
//   z_b  = Binv r_b            (first smoothing sweep / plain preconditioner)
//   z_b += Binv (r - A z)_b    (later sweeps)
// while the point-Jacobi blocks of those rows are zeroed, i.e. the smoother is block Jacobi with one big block per body.
// The inverse is lagged over the Newton iterations of a step: a preconditioner only has to be fixed during one solve.
#pragma once
#include "tsl_device.hpp"

#define TSL_MAX_DENSE_BODIES 8
#define BODY_APPLY_ROWS 8

struct BodyDenseArgs {
  int nb;                              // dense bodies
  int n3[TSL_MAX_DENSE_BODIES];        // dofs (3 * vertices)
  int rows_off[TSL_MAX_DENSE_BODIES];  // offset into rows[] (permuted row of local vertex k)
  long w_off[TSL_MAX_DENSE_BODIES];    // offset into W / Binv (n3 * ld entries each, ld = n3 rounded up to 4)
  int scr_off[TSL_MAX_DENSE_BODIES];   // offset into the pivot scratch (n3 entries per array)
  int wg_off[TSL_MAX_DENSE_BODIES + 1];  // workgroup prefix of k_body_apply
  const int* rows;
  const int* body_of;   // per ORIGINAL vertex: dense body or -1
  const int* local_of;  // per ORIGINAL vertex: index inside its body
};

// W_b += static matrix entries whose row and column both belong to body b (one wave per body vertex)
__global__ void __launch_bounds__(64)
k_body_gather(BodyDenseArgs A, int b, const int* __restrict__ slice_off, const int* __restrict__ slice_len, const int* __restrict__ colidx, const int* __restrict__ perm,
              const double* __restrict__ vals, double* __restrict__ W) {
  const int k = blockIdx.x;
  const int n3 = A.n3[b];
  if (3 * k >= n3) return;
  const int p = A.rows[A.rows_off[b] + k];
  const int slice = p >> 6, lane = p & 63;
  const int off = slice_off[slice], len = slice_len[slice];
  double* Wb = W + A.w_off[b];
  for (int j = threadIdx.x; j < len; j += 64) {
    const int col = colidx[off + 64 * j + lane];
    const int orig = perm[col];
    if (A.body_of[orig] != b) continue;
    const int l = A.local_of[orig];
    const double* a = vals + ((size_t)off + 64 * (size_t)j) * 9 + lane;
#pragma unroll
    for (int e = 0; e < 9; e++) {
      const double v = a[64 * e];
      if (v != 0.0) atomicAdd(&Wb[(size_t)(3 * k + e / 3) * n3 + 3 * l + e % 3], v);  // padded slots carry zeros
    }
  }
}

// W_b += the parts of the (masked) 12x12 contact blocks whose two vertices lie in the same dense body
__global__ void __launch_bounds__(64)
k_body_contact(BodyDenseArgs A, int nc, const int* __restrict__ idx, const double* __restrict__ Hm, double* __restrict__ W) {
  const int ci = blockIdx.x * blockDim.x + threadIdx.x;
  if (ci >= nc) return;
  int bd[4], lc[4];
  for (int k = 0; k < 4; k++) { const int v = idx[4 * ci + k]; bd[k] = A.body_of[v]; lc[k] = A.local_of[v]; }
  const double* H = Hm + 144 * (size_t)ci;
  for (int a = 0; a < 4; a++) {
    if (bd[a] < 0) continue;
    const int n3 = A.n3[bd[a]];
    double* Wb = W + A.w_off[bd[a]];
    for (int c2 = 0; c2 < 4; c2++) {
      if (bd[c2] != bd[a]) continue;
      for (int e = 0; e < 9; e++) {
        const double v = H[(3 * a + e / 3) * 12 + 3 * c2 + e % 3];
        if (v != 0.0) atomicAdd(&Wb[(size_t)(3 * lc[a] + e / 3) * n3 + 3 * lc[c2] + e % 3], v);
      }
    }
  }
}

// pivot scratch for pivot 0: col[i] = W[i][0], row[j] = W[0][j]
__global__ void k_body_gj_init(BodyDenseArgs A, const double* __restrict__ W, double* __restrict__ scr_col, double* __restrict__ scr_row) {
  const int b = blockIdx.y;
  const int n3 = A.n3[b];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n3) return;
  const double* Wb = W + A.w_off[b];
  scr_col[A.scr_off[b] + i] = Wb[(size_t)i * n3];
  scr_row[A.scr_off[b] + i] = Wb[i];
}

// one in-place Gauss-Jordan inversion step (pivot k, no pivoting: the blocks are SPD) over all bodies; the column / row of
// the pivot come from the scratch written by the previous launch, the ones of pivot k+1 are saved for the next launch
__global__ void __launch_bounds__(256)
k_body_gj(BodyDenseArgs A, int k, double* __restrict__ W, const double* __restrict__ col_in, const double* __restrict__ row_in, double* __restrict__ col_out,
          double* __restrict__ row_out, int* __restrict__ bad) {
  const int b = blockIdx.y;
  const int n3 = A.n3[b];
  if (k >= n3) return;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)n3 * n3) return;
  const int i = (int)(t / n3), j = (int)(t % n3);
  const int so = A.scr_off[b];
  const double piv = col_in[so + k];
  const double ci = col_in[so + i], rj = row_in[so + j];
  double* a = W + A.w_off[b] + t;
  double v;
  if (i == k) v = (j == k) ? 1.0 / piv : rj / piv;
  else if (j == k) v = -ci / piv;
  else v = *a - ci * rj / piv;
  *a = v;
  if (j == k + 1) col_out[so + i] = v;
  if (i == k + 1) row_out[so + j] = v;
  if (i == k && j == k && !(piv > 0.0)) bad[b] = 1;
}

// Binv (fp32, symmetrised); a body whose block was not positive definite falls back to its point-Jacobi blocks
__global__ void __launch_bounds__(256)
k_body_finalize(BodyDenseArgs A, const double* __restrict__ W, const int* __restrict__ bad, const double* __restrict__ Dinv, float* __restrict__ Binv) {
  const int b = blockIdx.y;
  const int n3 = A.n3[b];
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)n3 * n3) return;
  const int i = (int)(t / n3), j = (int)(t % n3);
  const double* Wb = W + A.w_off[b];
  double v;
  if (!bad[b]) v = 0.5 * (Wb[(size_t)i * n3 + j] + Wb[(size_t)j * n3 + i]);
  else if (i / 3 == j / 3) v = Dinv[9 * (size_t)A.rows[A.rows_off[b] + i / 3] + 3 * (i % 3) + j % 3];
  else v = 0.0;
  Binv[A.w_off[b] + (size_t)i * ((n3 + 3) & ~3) + j] = (float)v;  // row stride padded to a multiple of 4 (16-byte loads)
}

// point-Jacobi blocks of the dense-body rows are switched off (their rows are served by k_body_apply)
__global__ void k_body_zero_dinv(BodyDenseArgs A, int n_rows, double* __restrict__ Dinv) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_rows) return;
  double* d = Dinv + 9 * (size_t)A.rows[q];
#pragma unroll
  for (int e = 0; e < 9; e++) d[e] = 0.0;
}

// mode 0: z_b = Binv r_b ; mode 1: z_b += Binv (r - t)_b ; optional per-workgroup partial of rdot . (Binv v).
// One wave per matrix row pair, 16-byte loads (the row length 3*n_verts is padded to a multiple of 4 in storage:
// ld = (n3 + 3) & ~3), BODY_APPLY_ROWS rows per 256-thread workgroup.
// bid: workgroup index within the body part of the launch; full_dot: the partial is rdot . z_new (the fused post-smoothing
// launch, where the Jacobi part leaves the body rows alone) instead of rdot . (z_new - z_old)
TSL_DEV void body_apply_block(const BodyDenseArgs& A, const float* __restrict__ Binv, int mode, const double* __restrict__ r, const double* __restrict__ t,
                              double* __restrict__ z, const double* __restrict__ rdot, double* __restrict__ part, int bid, bool full_dot) {
  __shared__ double v[3 * 512 + 4];
  __shared__ double s1[4];
  int b = 0;
  while (b + 1 < A.nb && bid >= A.wg_off[b + 1]) b++;
  const int n3 = A.n3[b];
  const int ld = (n3 + 3) & ~3;
  const int* rows = A.rows + A.rows_off[b];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row0 = (bid - A.wg_off[b]) * BODY_APPLY_ROWS;
  const float* Bb = Binv + A.w_off[b];
  const int nq = ld >> 2;
  // Everything that does not depend on the staged vector is requested first (matrix rows of this wave, the old z / rdot of its two
  // output rows): the kernel is a chain of dependent round trips otherwise (row ids -> r, t -> matrix -> row ids -> z), 6.8 us.
  const int i0 = row0 + w * (BODY_APPLY_ROWS / 4), i1 = i0 + 1;  // two rows per wave
  const bool live = i0 < n3, live1 = i1 < n3;
  constexpr int MAXQ = (3 * 512 / 4 + 63) / 64;
  float4 a0[MAXQ], a1[MAXQ];
  size_t g0 = 0, g1 = 0;
  double zo0 = 0, zo1 = 0, rd0 = 0, rd1 = 0;
  if (live) {
    const float4* b0 = (const float4*)(Bb + (size_t)i0 * ld);
    const float4* b1 = (const float4*)(Bb + (size_t)(live1 ? i1 : i0) * ld);
#pragma unroll
    for (int u = 0; u < MAXQ; u++) {
      const int j = lane + 64 * u;
      a0[u] = j < nq ? b0[j] : make_float4(0, 0, 0, 0);
      a1[u] = j < nq ? b1[j] : make_float4(0, 0, 0, 0);
    }
    if (lane == 0) {
      g0 = 3 * (size_t)rows[i0 / 3] + i0 % 3;
      if (mode) zo0 = z[g0];
      if (rdot) rd0 = rdot[g0];
      if (live1) {
        g1 = 3 * (size_t)rows[i1 / 3] + i1 % 3;
        if (mode) zo1 = z[g1];
        if (rdot) rd1 = rdot[g1];
      }
    }
  }
  for (int i = threadIdx.x; i < ld; i += 256) {
    double x = 0.0;
    if (i < n3) {
      const size_t g = 3 * (size_t)rows[i / 3] + i % 3;
      x = mode ? r[g] - t[g] : r[g];
    }
    v[i] = x;
  }
  __syncthreads();
  double acc = 0;
  if (live) {
    double sum0 = 0, sum1 = 0;
#pragma unroll
    for (int u = 0; u < MAXQ; u++) {
      const int j = lane + 64 * u;
      if (j < nq) {
        const double v0 = v[4 * j], v1 = v[4 * j + 1], v2 = v[4 * j + 2], v3 = v[4 * j + 3];
        sum0 += (double)a0[u].x * v0 + (double)a0[u].y * v1 + (double)a0[u].z * v2 + (double)a0[u].w * v3;
        sum1 += (double)a1[u].x * v0 + (double)a1[u].y * v1 + (double)a1[u].z * v2 + (double)a1[u].w * v3;
      }
    }
    sum0 = wave_sum(sum0); sum1 = wave_sum(sum1);
    if (lane == 0) {
      const double z0 = zo0 + sum0;   // zo = 0 in mode 0
      z[g0] = z0;
      acc += rd0 * (full_dot ? z0 : sum0);  // separate launches in mode 1: the Jacobi kernel already counted rdot . z_old
      if (live1) {
        const double z1 = zo1 + sum1;
        z[g1] = z1;
        acc += rd1 * (full_dot ? z1 : sum1);
      }
    }
  }
  if (part) {
    if (lane == 0) s1[w] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[bid] = s1[0] + s1[1] + s1[2] + s1[3];
  }
}

// Body part of the PCG update launch (k_pcg_update, workgroups behind the vertex part): the first smoothing sweep z_b = Binv r_b of
// the multigrid cycle that follows, without a launch of its own.  The vertex part updates r in place in the same launch, so the
// body workgroups never read r: they keep a compact ping-pong copy rb of the residual on the body rows, rebuilt here as
// rb_new = rb_old - alpha Ap (bitwise the values the vertex part writes), or as b - Ax when the solve (re)starts.
// start mode (b_init != null): only rb_new is written (the cycle of a start runs its own first sweep).
template <class AlphaFn>
TSL_DEV void body_update_block(const BodyDenseArgs& A, const float* __restrict__ Binv, AlphaFn alpha_fn, const double* __restrict__ Ap, const double* __restrict__ rb_old,
                               double* __restrict__ rb_new, const double* __restrict__ b_init, const double* __restrict__ Ax_init, double* __restrict__ z, int bid,
                               double* v /* LDS, 3 * 512 + 4 */) {
  int b = 0;
  while (b + 1 < A.nb && bid >= A.wg_off[b + 1]) b++;
  const int n3 = A.n3[b];
  const int ld = (n3 + 3) & ~3;
  const int* rows = A.rows + A.rows_off[b];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row0 = (bid - A.wg_off[b]) * BODY_APPLY_ROWS;
  const int so = A.scr_off[b];
  if (b_init) {  // compact residual of this workgroup's rows
    const int i = row0 + (int)threadIdx.x;
    if (threadIdx.x < BODY_APPLY_ROWS && i < n3) {
      const size_t g = 3 * (size_t)rows[i / 3] + i % 3;
      rb_new[so + i] = Ax_init ? b_init[g] - Ax_init[g] : b_init[g];
    }
    return;
  }
  const float* Bb = Binv + A.w_off[b];
  const int nq = ld >> 2;
  const int i0 = row0 + w * (BODY_APPLY_ROWS / 4), i1 = i0 + 1;  // two rows per wave
  const bool live = i0 < n3, live1 = i1 < n3;
  constexpr int MAXQ = (3 * 512 / 4 + 63) / 64;
  float4 a0[MAXQ], a1[MAXQ];
  size_t g0 = 0, g1 = 0;
  if (live) {
    const float4* b0 = (const float4*)(Bb + (size_t)i0 * ld);
    const float4* b1 = (const float4*)(Bb + (size_t)(live1 ? i1 : i0) * ld);
#pragma unroll
    for (int u = 0; u < MAXQ; u++) {
      const int j = lane + 64 * u;
      a0[u] = j < nq ? b0[j] : make_float4(0, 0, 0, 0);
      a1[u] = j < nq ? b1[j] : make_float4(0, 0, 0, 0);
    }
    if (lane == 0) {
      g0 = 3 * (size_t)rows[i0 / 3] + i0 % 3;
      if (live1) g1 = 3 * (size_t)rows[i1 / 3] + i1 % 3;
    }
  }
  // the old residual and A p of the body rows are requested before alpha is known (alpha_fn reduces the p.Ap partials: one more
  // memory round trip and two barriers that these loads and the matrix rows above now overlap)
  constexpr int MAXV = (3 * 512 + 4 + 255) / 256;
  double rbv[MAXV], apv[MAXV];
#pragma unroll
  for (int u = 0; u < MAXV; u++) {
    const int i = (int)threadIdx.x + 256 * u;
    rbv[u] = 0.0; apv[u] = 0.0;
    if (i < n3) {
      const size_t g = 3 * (size_t)rows[i / 3] + i % 3;
      rbv[u] = rb_old[so + i]; apv[u] = Ap[g];
    }
  }
  const double alpha = alpha_fn();  // NaN: the solve has stopped (uniform over the workgroup)
  if (alpha != alpha) return;
#pragma unroll
  for (int u = 0; u < MAXV; u++) {
    const int i = (int)threadIdx.x + 256 * u;
    if (i < ld) v[i] = rbv[u] - alpha * apv[u];
  }
  __syncthreads();
  if (live) {
    double sum0 = 0, sum1 = 0;
#pragma unroll
    for (int u = 0; u < MAXQ; u++) {
      const int j = lane + 64 * u;
      if (j < nq) {
        const double v0 = v[4 * j], v1 = v[4 * j + 1], v2 = v[4 * j + 2], v3 = v[4 * j + 3];
        sum0 += (double)a0[u].x * v0 + (double)a0[u].y * v1 + (double)a0[u].z * v2 + (double)a0[u].w * v3;
        sum1 += (double)a1[u].x * v0 + (double)a1[u].y * v1 + (double)a1[u].z * v2 + (double)a1[u].w * v3;
      }
    }
    sum0 = wave_sum(sum0); sum1 = wave_sum(sum1);
    if (lane == 0) {
      z[g0] = sum0; rb_new[so + i0] = v[i0];
      if (live1) { z[g1] = sum1; rb_new[so + i1] = v[i1]; }
    }
  }
}

__global__ void __launch_bounds__(256)
k_body_apply(BodyDenseArgs A, const float* __restrict__ Binv, int mode, const double* __restrict__ r, const double* __restrict__ t, double* __restrict__ z,
             const double* __restrict__ rdot, double* __restrict__ part) {
  body_apply_block(A, Binv, mode, r, t, z, rdot, part, (int)blockIdx.x, false);
}
