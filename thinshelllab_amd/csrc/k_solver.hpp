// Linear-solver kernels: SELL-64 block SpMV with fused dot products, block-Jacobi PCG vector updates,
// frozen-dof masking, permutations.  Replaces SparseMatrix.solve (cupyx spsolve,
// /root/reference/code/engine/sparse_solver.py:85-105) and the add_H frozen masking (BaseScene.py:399-405).
//
// Matrix layout (HBM): rows (= vertices, 3x3 blocks) are permuted by descending block count and cut into slices
// of 64 rows (one wavefront).  Slice s stores len_s block columns; block column k of the slice is 9 planes of 64
// doubles: element e of the block of lane L sits at  ((slice_off[s] + 64 k) * 9) + 64 e + L.  Every load of the
// SpMV is therefore one fully coalesced 512-B wave access; column ids are laid out the same way (64 ints).
// All solver vectors are AoS xyz in the permuted row order.
#pragma once
#include "tsl_ctx.hpp"
#include "tsl_device.hpp"

struct CgScal {
  double pAp[4], rzn[4], rr[4];
  double bb, thresh2, energy, pmax, aux[4];
  int flag, iters, pad0, pad1;
};
static_assert(sizeof(CgScal) <= sizeof(SolverScalars), "CgScal must fit the scalar record");

// dst = masked(src): rows/columns of frozen dofs are removed, frozen diagonal = m/dt^2
// (add_H drops entries touching a frozen dof; the mass diagonal is added unconditionally: BaseScene.py:399-405,
// model_fold_offset.py:468-470)
__global__ void k_mask_matrix(int n_slices, const int* __restrict__ slice_off, const int* __restrict__ slice_len, const int* __restrict__ colidx,
                              const unsigned char* __restrict__ fz, const double* __restrict__ mdt2, const double* __restrict__ src, double* __restrict__ dst, int NV) {
  const int slice = blockIdx.x, lane = threadIdx.x & 63;
  if (slice >= n_slices) return;
  const int off = slice_off[slice], len = slice_len[slice];
  const int p = slice * 64 + lane;
  const unsigned rm = (p < NV) ? fz[p] : 7u;
  const double md = (p < NV) ? mdt2[p] : 0.0;
  for (int k = threadIdx.x >> 6; k < len; k += (blockDim.x >> 6)) {
    const int c = colidx[off + 64 * k + lane];
    const unsigned cm = fz[c];
    const size_t base = ((size_t)off + 64 * (size_t)k) * 9 + lane;
#pragma unroll
    for (int e = 0; e < 9; e++) {
      const int r = e / 3, cc = e % 3;
      double v = src[base + 64 * e];
      if (((rm >> r) & 1u) || ((cm >> cc) & 1u)) v = 0.0;
      if (c == p && r == cc && ((rm >> r) & 1u)) v = md;
      dst[base + 64 * e] = v;
    }
  }
}

// Dinv[p] = inverse(diagonal block + extra diagonal (contact) contribution)
__global__ void k_block_jacobi(int NV, const int* __restrict__ diag_perm, const double* __restrict__ vals, const double* __restrict__ extra, double* __restrict__ Dinv) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= NV) return;
  const size_t base = (size_t)diag_perm[p];
  m3 D;
#pragma unroll
  for (int e = 0; e < 9; e++) D.m[e] = vals[base + 64 * e] + (extra ? extra[9 * (size_t)p + e] : 0.0);
  const m3 Di = m3_inv(D);
#pragma unroll
  for (int e = 0; e < 9; e++) Dinv[9 * (size_t)p + e] = Di.m[e];
}

// out_perm[p] = in_orig[perm[p]]  /  out_orig[perm[p]] = in_perm[p]
__global__ void k_gather_perm(int NV, const int* __restrict__ perm, const double* __restrict__ in, double* __restrict__ out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= NV) return;
  st3(out, p, ld3(in, perm[p]));
}
__global__ void k_scatter_perm(int NV, const int* __restrict__ perm, const double* __restrict__ in, double* __restrict__ out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= NV) return;
  st3(out, perm[p], ld3(in, p));
}

// y = A x ; optionally accumulates dot(x, y) into *dot_out.  One lane per row, one wave per slice.
// This is the dominant kernel of the engine (one launch per PCG iteration).
__global__ void __launch_bounds__(256)
k_spmv(int NV, int n_slices, const int* __restrict__ slice_off, const int* __restrict__ slice_len, const int* __restrict__ colidx,
       const double* __restrict__ vals, const double* __restrict__ x, double* __restrict__ y, CgScal* sc, int slot, int check_flag,
       unsigned long long* prof) {
  if (check_flag && sc->flag) return;
  // sampled launches record their own execution span with the constant-rate device clock (min start / max end over waves)
  unsigned long long t_start = 0;
  if (prof) t_start = wall_clock64();
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int slice = p >> 6, lane = p & 63;
  double acc = 0.0;
  if (slice < n_slices) {
    const int off = slice_off[slice], len = slice_len[slice];
    double y0 = 0, y1 = 0, y2 = 0;
    const int* cp = colidx + off + lane;
    const double* vp = vals + (size_t)off * 9 + lane;
#pragma unroll 2
    for (int k = 0; k < len; k++) {
      const int c = cp[64 * k];
      const double* a = vp + (size_t)k * 576;
      const double a0 = a[0], a1 = a[64], a2 = a[128], a3 = a[192], a4 = a[256], a5 = a[320], a6 = a[384], a7 = a[448], a8 = a[512];
      const d3 xj = ld3(x, c);
      y0 += a0 * xj.x + a1 * xj.y + a2 * xj.z;
      y1 += a3 * xj.x + a4 * xj.y + a5 * xj.z;
      y2 += a6 * xj.x + a7 * xj.y + a8 * xj.z;
    }
    if (p < NV) {
      st3(y, p, d3(y0, y1, y2));
      const d3 xi = ld3(x, p);
      acc = xi.x * y0 + xi.y * y1 + xi.z * y2;
    }
  }
  if (slot >= 0) {
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0 && acc != 0.0) atomicAdd(&sc->pAp[slot], acc);
    if (p == 0) { const int z = (slot + 2) & 3; sc->pAp[z] = 0; sc->rzn[z] = 0; sc->rr[z] = 0; }
  }
  if (prof && (threadIdx.x & 63) == 0) {  // one slot pair per wave: plain stores, reduced on the host
    const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    prof[2 * w] = t_start;
    prof[2 * w + 1] = (unsigned long long)wall_clock64();
  }
}

// r = b - Ax ; z = Dinv r ; p = z ; rzn[3] = r.z ; rr[3] = r.r   (start / restart of PCG)
__global__ void k_cg_init(int NV, const double* __restrict__ b, const double* __restrict__ Ax, const double* __restrict__ Dinv,
                          double* __restrict__ r, double* __restrict__ z, double* __restrict__ pv, CgScal* sc) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  double rz = 0, rr = 0;
  if (p < NV) {
    d3 rv = ld3(b, p);
    if (Ax) rv = rv - ld3(Ax, p);
    m3 D;
#pragma unroll
    for (int e = 0; e < 9; e++) D.m[e] = Dinv[9 * (size_t)p + e];
    const d3 zv = m3_mulv(D, rv);
    st3(r, p, rv); st3(z, p, zv); st3(pv, p, zv);
    rz = dot(rv, zv); rr = dot(rv, rv);
  }
  rz = wave_sum(rz); rr = wave_sum(rr);
  if ((threadIdx.x & 63) == 0) { atomicAdd(&sc->rzn[3], rz); atomicAdd(&sc->rr[3], rr); }
}

// alpha = rz/pAp ; x += alpha p ; r -= alpha Ap ; z = Dinv r ; rzn[slot] += r.z ; rr[slot] += r.r
__global__ void k_cg_update(int NV, const double* __restrict__ pv, const double* __restrict__ Ap, const double* __restrict__ Dinv,
                            double* __restrict__ x, double* __restrict__ r, double* __restrict__ z, CgScal* sc, int slot) {
  if (sc->flag) return;
  const double pAp = sc->pAp[slot], rz = sc->rzn[(slot + 3) & 3];
  if (!(pAp > 0.0) || !(rz > 0.0)) {  // breakdown: matrix not positive definite along p
    if (blockIdx.x == 0 && threadIdx.x == 0) sc->flag = 1;
    return;
  }
  const double alpha = rz / pAp;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  double rzn = 0, rr = 0;
  if (p < NV) {
    const d3 pp = ld3(pv, p);
    st3(x, p, ld3(x, p) + alpha * pp);
    const d3 rv = ld3(r, p) - alpha * ld3(Ap, p);
    m3 D;
#pragma unroll
    for (int e = 0; e < 9; e++) D.m[e] = Dinv[9 * (size_t)p + e];
    const d3 zv = m3_mulv(D, rv);
    st3(r, p, rv); st3(z, p, zv);
    rzn = dot(rv, zv); rr = dot(rv, rv);
  }
  rzn = wave_sum(rzn); rr = wave_sum(rr);
  if ((threadIdx.x & 63) == 0) { atomicAdd(&sc->rzn[slot], rzn); atomicAdd(&sc->rr[slot], rr); }
}

// beta = rzn/rz ; p = z + beta p ; convergence flag
__global__ void k_cg_p(int NV, const double* __restrict__ z, double* __restrict__ pv, CgScal* sc, int slot, int it) {
  if (sc->flag) return;
  const double rzn = sc->rzn[slot], rz = sc->rzn[(slot + 3) & 3], rr = sc->rr[slot];
  if (rr <= sc->thresh2) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc->flag = 2; sc->iters = it + 1; }
    return;
  }
  const double beta = rzn / rz;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < NV) st3(pv, p, ld3(z, p) + beta * ld3(pv, p));
}

// generic helpers ------------------------------------------------------------------------------
// sum of a*b over n doubles into *out
__global__ void k_dot(size_t n, const double* __restrict__ a, const double* __restrict__ b, double* out) {
  double s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i] * b[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}
// max |a_i| into *out (non-negative doubles compare like their bit patterns)
__global__ void k_absmax(size_t n, const double* __restrict__ a, double* out) {
  double s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s = fmax(s, fabs(a[i]));
  s = wave_max(s);
  if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long*)out, (unsigned long long)__double_as_longlong(s));
}
// y = a*x + b*y (b == 0 ignores old y)
__global__ void k_axpby(size_t n, double a, const double* __restrict__ x, double b, double* __restrict__ y) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = (b == 0.0) ? a * x[i] : a * x[i] + b * y[i];
}
// z = Dinv r
__global__ void k_precond(int NV, const double* __restrict__ Dinv, const double* __restrict__ r, double* __restrict__ z) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= NV) return;
  m3 D;
#pragma unroll
  for (int e = 0; e < 9; e++) D.m[e] = Dinv[9 * (size_t)p + e];
  st3(z, p, m3_mulv(D, ld3(r, p)));
}
// F[i] = 0 where frozen (BaseScene.apply_frozen :1071-1075)
__global__ void k_mask_vec(size_t n, const int* __restrict__ frozen, double* __restrict__ v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) if (frozen[i]) v[i] = 0.0;
}
// pos = x1 - alpha * p  (BaseScene.linesearch_step :1089-1094)
__global__ void k_linesearch(size_t n, const double* __restrict__ x1, const double* __restrict__ p, double alpha, double* __restrict__ pos) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) pos[i] = x1[i] - p[i] * alpha;
}
// vel = (pos - prev) * damping / dt  (BaseScene.update_vel :868-872)
__global__ void k_update_vel(size_t n, const double* __restrict__ pos, const double* __restrict__ prev, double s, double* __restrict__ vel) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) vel[i] = (pos[i] - prev[i]) * s;
}
