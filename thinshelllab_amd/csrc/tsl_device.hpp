// Device-side fp64 helpers for the gfx950 thin-shell kernels (no reference counterpart: Taichi
// supplies ti.Vector / ti.Matrix; here they are plain structs kept in VGPRs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TSL_DEV __device__ __forceinline__
#define TSL_HD __host__ __device__ __forceinline__

struct d3 {
  double x, y, z;
  TSL_HD d3() : x(0), y(0), z(0) {}
  TSL_HD d3(double a, double b, double c) : x(a), y(b), z(c) {}
  TSL_HD double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
TSL_HD d3 operator+(const d3& a, const d3& b) { return d3(a.x + b.x, a.y + b.y, a.z + b.z); }
TSL_HD d3 operator-(const d3& a, const d3& b) { return d3(a.x - b.x, a.y - b.y, a.z - b.z); }
TSL_HD d3 operator-(const d3& a) { return d3(-a.x, -a.y, -a.z); }
TSL_HD d3 operator*(const d3& a, double s) { return d3(a.x * s, a.y * s, a.z * s); }
TSL_HD d3 operator*(double s, const d3& a) { return d3(a.x * s, a.y * s, a.z * s); }
TSL_HD d3 operator/(const d3& a, double s) { return d3(a.x / s, a.y / s, a.z / s); }
TSL_HD double dot(const d3& a, const d3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
TSL_HD d3 cross(const d3& a, const d3& b) { return d3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
TSL_HD double norm(const d3& a) { return sqrt(dot(a, a)); }
TSL_HD d3 normalized(const d3& a) { return a / norm(a); }

TSL_DEV d3 ld3(const double* __restrict__ p, int i) { return d3(p[3 * (size_t)i], p[3 * (size_t)i + 1], p[3 * (size_t)i + 2]); }
TSL_DEV void st3(double* p, int i, const d3& v) { p[3 * (size_t)i] = v.x; p[3 * (size_t)i + 1] = v.y; p[3 * (size_t)i + 2] = v.z; }

// row-major 3x3
struct m3 {
  double m[9];
  TSL_HD double& operator()(int i, int j) { return m[i * 3 + j]; }
  TSL_HD double operator()(int i, int j) const { return m[i * 3 + j]; }
};
TSL_HD m3 m3_zero() { m3 r; for (int i = 0; i < 9; i++) r.m[i] = 0; return r; }
TSL_HD m3 m3_outer(const d3& a, const d3& b) {
  m3 r;
  r.m[0] = a.x * b.x; r.m[1] = a.x * b.y; r.m[2] = a.x * b.z;
  r.m[3] = a.y * b.x; r.m[4] = a.y * b.y; r.m[5] = a.y * b.z;
  r.m[6] = a.z * b.x; r.m[7] = a.z * b.y; r.m[8] = a.z * b.z;
  return r;
}
TSL_HD m3 m3_mul(const m3& a, const m3& b) {
  m3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i * 3 + j] = a.m[i * 3] * b.m[j] + a.m[i * 3 + 1] * b.m[3 + j] + a.m[i * 3 + 2] * b.m[6 + j];
  return r;
}
TSL_HD m3 m3_T(const m3& a) { m3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i * 3 + j] = a.m[j * 3 + i]; return r; }
TSL_HD double m3_det(const m3& a) {
  return a.m[0] * (a.m[4] * a.m[8] - a.m[5] * a.m[7]) - a.m[1] * (a.m[3] * a.m[8] - a.m[5] * a.m[6]) + a.m[2] * (a.m[3] * a.m[7] - a.m[4] * a.m[6]);
}
TSL_HD m3 m3_inv(const m3& a) {
  double d = m3_det(a);
  m3 r;
  r.m[0] = (a.m[4] * a.m[8] - a.m[5] * a.m[7]) / d;
  r.m[1] = (a.m[2] * a.m[7] - a.m[1] * a.m[8]) / d;
  r.m[2] = (a.m[1] * a.m[5] - a.m[2] * a.m[4]) / d;
  r.m[3] = (a.m[5] * a.m[6] - a.m[3] * a.m[8]) / d;
  r.m[4] = (a.m[0] * a.m[8] - a.m[2] * a.m[6]) / d;
  r.m[5] = (a.m[2] * a.m[3] - a.m[0] * a.m[5]) / d;
  r.m[6] = (a.m[3] * a.m[7] - a.m[4] * a.m[6]) / d;
  r.m[7] = (a.m[1] * a.m[6] - a.m[0] * a.m[7]) / d;
  r.m[8] = (a.m[0] * a.m[4] - a.m[1] * a.m[3]) / d;
  return r;
}
// cofactor products: m3_cof2(a, b)[i][j] = the 2 x 2 minor products of cof with the first factor taken from a and the second from b;
// cof(F) = det(F) F^-T = m3_cof2(F, F), and its differential in the direction dF is m3_cof2(F, dF) + m3_cof2(dF, F) (no division:
// exact where det F passes through zero, which F^-1 is not)
TSL_HD m3 m3_cof2(const m3& a, const m3& b) {
  m3 r;
  r.m[0] = a.m[4] * b.m[8] - a.m[5] * b.m[7];
  r.m[1] = a.m[5] * b.m[6] - a.m[3] * b.m[8];
  r.m[2] = a.m[3] * b.m[7] - a.m[4] * b.m[6];
  r.m[3] = a.m[2] * b.m[7] - a.m[1] * b.m[8];
  r.m[4] = a.m[0] * b.m[8] - a.m[2] * b.m[6];
  r.m[5] = a.m[1] * b.m[6] - a.m[0] * b.m[7];
  r.m[6] = a.m[1] * b.m[5] - a.m[2] * b.m[4];
  r.m[7] = a.m[2] * b.m[3] - a.m[0] * b.m[5];
  r.m[8] = a.m[0] * b.m[4] - a.m[1] * b.m[3];
  return r;
}
TSL_HD d3 m3_mulv(const m3& a, const d3& v) {
  return d3(a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z, a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z);
}

// Symmetric eigen-clamp A <- sum_{lambda>0} lambda q q^T by cyclic Jacobi, D x D in a private array
// with leading dimension D.  Replaces SPD_Projector.project (engine/linalg.py:132-148): the reference
// runs Householder + K shifted-QR sweeps (not always converged); a converged eigen-solve agrees with it
// wherever its QR converged.
template <int D>
TSL_DEV void spd_clamp(double* A) {
  double V[D * D];
#pragma unroll
  for (int i = 0; i < D; i++)
#pragma unroll
    for (int j = 0; j < D; j++) V[i * D + j] = (i == j) ? 1.0 : 0.0;
  // symmetrise (inputs are symmetric up to rounding)
  for (int i = 0; i < D; i++)
    for (int j = i + 1; j < D; j++) { double s = 0.5 * (A[i * D + j] + A[j * D + i]); A[i * D + j] = s; A[j * D + i] = s; }
  const int max_sweeps = (D <= 3) ? 12 : 30;
  for (int sweep = 0; sweep < max_sweeps; sweep++) {
    double off = 0, diag = 0;
    for (int i = 0; i < D; i++) {
      diag += A[i * D + i] * A[i * D + i];
      for (int j = i + 1; j < D; j++) off += A[i * D + j] * A[i * D + j];
    }
    if (off <= 1e-32 * (diag + off)) break;
    for (int p = 0; p < D - 1; p++)
      for (int q = p + 1; q < D; q++) {
        double apq = A[p * D + q];
        if (apq == 0.0) continue;
        double theta = (A[q * D + q] - A[p * D + p]) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < D; k++) {
          double akp = A[k * D + p], akq = A[k * D + q];
          A[k * D + p] = c * akp - s * akq;
          A[k * D + q] = s * akp + c * akq;
        }
        for (int k = 0; k < D; k++) {
          double apk = A[p * D + k], aqk = A[q * D + k];
          A[p * D + k] = c * apk - s * aqk;
          A[q * D + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < D; k++) {
          double vkp = V[k * D + p], vkq = V[k * D + q];
          V[k * D + p] = c * vkp - s * vkq;
          V[k * D + q] = s * vkp + c * vkq;
        }
      }
  }
  double lam[D];
  for (int e = 0; e < D; e++) lam[e] = A[e * D + e] > 0 ? A[e * D + e] : 0.0;
  for (int i = 0; i < D; i++)
    for (int j = 0; j < D; j++) {
      double s = 0;
      for (int e = 0; e < D; e++) s += lam[e] * V[i * D + e] * V[j * D + e];
      A[i * D + j] = s;
    }
}

// The same eigen-clamp started from the eigenvector basis of the PREVIOUS assembly of the same element (Vg: D x D doubles with
// element stride `vs`, i.e. entry e of this lane at Vg[e * vs]): A' = V^T A V is nearly diagonal when the element moved little (the
// Newton iterations of a time step), so the cyclic Jacobi converges in 1-2 sweeps instead of 6-8 (k_tet_hess: one lane per element,
// the longest kernel of an assembly).  Same convergence test, same reconstruction, i.e. the same clamped matrix to rounding; the new
// basis V R is stored for the next call.  `warm` false (first assembly, periodic refresh against the slow loss of orthogonality of an
// accumulated rotation product, non-finite stored basis) starts from the identity like spd_clamp.
template <int D>
TSL_DEV void spd_clamp_warm(double* A, double* __restrict__ Vg, size_t vs, bool warm) {
  double V[D * D];
  for (int i = 0; i < D; i++)
    for (int j = i + 1; j < D; j++) { double s = 0.5 * (A[i * D + j] + A[j * D + i]); A[i * D + j] = s; A[j * D + i] = s; }
  if (warm) {   // the stored basis must look like one: finite entries and squared Frobenius norm D (a slot never written -- an element
                // that was not clamped before -- or one poisoned by a non-finite block starts from the identity)
    double ss = 0.0;
#pragma unroll
    for (int e = 0; e < D * D; e++) { V[e] = Vg[e * vs]; ss += V[e] * V[e]; }
    warm = fabs(ss - (double)D) <= 1e-6 * D;
  }
  if (warm) {   // A <- V^T A V
    double T[D * D];
    for (int i = 0; i < D; i++)
      for (int j = 0; j < D; j++) { double s = 0; for (int k = 0; k < D; k++) s += A[i * D + k] * V[k * D + j]; T[i * D + j] = s; }
    for (int i = 0; i < D; i++)
      for (int j = i; j < D; j++) { double s = 0; for (int k = 0; k < D; k++) s += V[k * D + i] * T[k * D + j]; A[i * D + j] = s; A[j * D + i] = s; }
  } else {
#pragma unroll
    for (int i = 0; i < D; i++)
#pragma unroll
      for (int j = 0; j < D; j++) V[i * D + j] = (i == j) ? 1.0 : 0.0;
  }
  for (int sweep = 0; sweep < 30; sweep++) {
    double off = 0, diag = 0;
    for (int i = 0; i < D; i++) {
      diag += A[i * D + i] * A[i * D + i];
      for (int j = i + 1; j < D; j++) off += A[i * D + j] * A[i * D + j];
    }
    if (off <= 1e-32 * (diag + off)) break;
    for (int p = 0; p < D - 1; p++)
      for (int q = p + 1; q < D; q++) {
        double apq = A[p * D + q];
        if (apq == 0.0) continue;
        double theta = (A[q * D + q] - A[p * D + p]) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < D; k++) {
          double akp = A[k * D + p], akq = A[k * D + q];
          A[k * D + p] = c * akp - s * akq;
          A[k * D + q] = s * akp + c * akq;
        }
        for (int k = 0; k < D; k++) {
          double apk = A[p * D + k], aqk = A[q * D + k];
          A[p * D + k] = c * apk - s * aqk;
          A[q * D + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < D; k++) {
          double vkp = V[k * D + p], vkq = V[k * D + q];
          V[k * D + p] = c * vkp - s * vkq;
          V[k * D + q] = s * vkp + c * vkq;
        }
      }
  }
#pragma unroll
  for (int e = 0; e < D * D; e++) Vg[e * vs] = V[e];
  double lam[D];
  for (int e = 0; e < D; e++) lam[e] = A[e * D + e] > 0 ? A[e * D + e] : 0.0;
  for (int i = 0; i < D; i++)
    for (int j = 0; j < D; j++) {
      double s = 0;
      for (int e = 0; e < D; e++) s += lam[e] * V[i * D + e] * V[j * D + e];
      A[i * D + j] = s;
    }
}

// ---- cooperative 9 x 9 eigen-clamp (round 6): 16 lanes per matrix, matrix and eigenvector accumulator in LDS, PARALLEL rotation order -------------------
// The cyclic Jacobi of spd_clamp<9> walks 36 rotations per sweep one after the other; one lane per matrix keeps 2 x 81 doubles in private arrays with dynamic
// indices (scratch memory: k_tet_hess 1414 spilled registers, 130-170 us for 5.8k elements), 16 lanes per matrix in lock step pay three LDS round trips per
// rotation (k_contact_assemble_coop: 125-185 us for ~100 constraints).  Rotations of DISJOINT index pairs commute, so a sweep is nine rounds of four
// simultaneous rotations (round r pairs lane l with (r - l) mod 9; the lane with 2 l = r mod 9 sits out): lane l < 9 applies its pair's rotation to column l
// (matrix and eigenvectors), then to row l.  Same rotation formula, same convergence test (off^2 <= 1e-32 |A|^2), same sweep limit and the same reconstruction
// A <- sum_{lambda > 0} lambda v v^T as spd_clamp<9>: the clamped matrix agrees with the serial order to rounding (the eigen-decomposition is unique up to it).
// sa: the symmetric matrix (row-major 9 x 9), sv: the accumulator V (identity, or the warm-start basis with sa = V^T A V); the four groups of a wave run in lock
// step (LDS operations of a wave complete in order); `on` is uniform within a group; on return sa holds the clamped matrix (groups with `on`), sv the basis.
TSL_DEV void spd_grp_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
TSL_DEV void spd_clamp9_par(double* __restrict__ sa, double* __restrict__ sv, int l, bool on) {
  const bool row = l < 9;
  bool done = !on;
  double rel_prev = 1.0;
  for (int sweep = 0; sweep < 30; sweep++) {
    double off = 0.0, diag = 0.0;
    if (row) {
#pragma unroll
      for (int k = 0; k < 9; k++) { const double v = sa[l * 9 + k]; if (k == l) diag = v * v; else off += v * v; }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { off += __shfl_xor(off, o, 16); diag += __shfl_xor(diag, o, 16); }
    // converged: off^2 <= 1e-32 |A|^2 as in spd_clamp<9> -- or the sweep before did not halve off^2 any more below 1e-24 |A|^2: the quadratic convergence of the
    // Jacobi iteration ends on a floor of rounding noise (eigenvalues of 1e3 next to 1e-6 in a contact block: 3e-33 for the cyclic order, 1.4e-32 for this one --
    // the serial routine's bound sits between the two and this order ran its 30 sweeps on two blocks in three)
    const double rel = 0.5 * off / (diag + 0.5 * off);
    if (0.5 * off <= 1e-32 * (diag + 0.5 * off) || (rel <= 1e-24 && rel >= 0.5 * rel_prev)) done = true;
    rel_prev = rel;
    if (!__any(!done)) break;
    for (int r = 0; r < 9; r++) {
      int m = r - l; if (m < 0) m += 9;
      const bool act = row && m != l && !done;     // (lanes 9..15, the lane that sits out, groups that have converged: no rotation)
      const int mm = act ? m : 0, ll = row ? l : 0;
      const int p = min(ll, mm), q = max(ll, mm);
      const bool isp = ll < mm;
      double c = 1.0, s = 0.0;
      if (act) {
        const double apq = sa[p * 9 + q], app = sa[p * 9 + p], aqq = sa[q * 9 + q];
        if (apq != 0.0) {
          const double theta = (aqq - app) / (2.0 * apq);
          const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          c = 1.0 / sqrt(t * t + 1.0); s = t * c;
        }
      }
      // column l <- rotation of columns (p, q): new[p] = c old[p] - s old[q], new[q] = s old[p] + c old[q]; the same for the eigenvectors
      const double ss = isp ? -s : s;
      double ao[9], vo[9];
      if (act) {
#pragma unroll
        for (int i = 0; i < 9; i++) { ao[i] = c * sa[i * 9 + ll] + ss * sa[i * 9 + mm]; vo[i] = c * sv[i * 9 + ll] + ss * sv[i * 9 + mm]; }
      }
      spd_grp_sync();
      if (act) {
#pragma unroll
        for (int i = 0; i < 9; i++) { sa[i * 9 + ll] = ao[i]; sv[i * 9 + ll] = vo[i]; }
      }
      spd_grp_sync();
      // row l <- the same rotation of rows (p, q)
      if (act) {
#pragma unroll
        for (int j = 0; j < 9; j++) ao[j] = c * sa[ll * 9 + j] + ss * sa[mm * 9 + j];
      }
      spd_grp_sync();
      if (act) {
#pragma unroll
        for (int j = 0; j < 9; j++) sa[ll * 9 + j] = ao[j];
      }
      spd_grp_sync();
    }
  }
  // A = sum_e max(lambda_e, 0) v_e v_e^T, row l
  double outr[9];
  if (row) {
#pragma unroll
    for (int j = 0; j < 9; j++) outr[j] = 0.0;
    for (int e = 0; e < 9; e++) {
      const double d = sa[e * 9 + e];
      const double lam = d > 0.0 ? d : 0.0;
      const double vl = lam * sv[l * 9 + e];
#pragma unroll
      for (int j = 0; j < 9; j++) outr[j] += vl * sv[j * 9 + e];
    }
  }
  spd_grp_sync();
  if (row && on) {
#pragma unroll
    for (int j = 0; j < 9; j++) sa[l * 9 + j] = outr[j];
  }
  spd_grp_sync();
}

// the same from a cold start: V = identity, A symmetrised (inputs are symmetric up to rounding; a block that is not clamped keeps its non-symmetric part)
TSL_DEV void spd_clamp9_cold(double* __restrict__ sa, double* __restrict__ sv, int l, bool on) {
  if (l < 9) {
#pragma unroll
    for (int k = 0; k < 9; k++) sv[l * 9 + k] = (k == l) ? 1.0 : 0.0;
  }
  spd_grp_sync();
  if (l < 9 && on) {
    for (int k = l + 1; k < 9; k++) { const double t = 0.5 * (sa[l * 9 + k] + sa[k * 9 + l]); sa[l * 9 + k] = t; sa[k * 9 + l] = t; }
  }
  spd_grp_sync();
  spd_clamp9_par(sa, sv, l, on);
}

// 2x2 symmetric PSD projection (engine/linalg.py:5-12, closed form of the ti.svd based rule)
TSL_DEV void spd_clamp2(double& a, double& b, double& d) {
  double tr = a + d, df = a - d;
  double rad = sqrt(df * df * 0.25 + b * b);
  double l1 = tr * 0.5 + rad, l2 = tr * 0.5 - rad;
  double qx = 1, qy = 0;
  if (rad > 0) {
    double x0 = a - l2, y0 = b, x1 = b, y1 = d - l2;
    if (x0 * x0 + y0 * y0 >= x1 * x1 + y1 * y1) { qx = x0; qy = y0; } else { qx = x1; qy = y1; }
    double nn = sqrt(qx * qx + qy * qy);
    if (nn > 0) { qx /= nn; qy /= nn; } else { qx = 1; qy = 0; }
  }
  double p1 = l1 > 0 ? l1 : 0, p2 = l2 > 0 ? l2 : 0;
  a = p1 * qx * qx + p2 * qy * qy;
  b = p1 * qx * qy - p2 * qy * qx;
  d = p1 * qy * qy + p2 * qx * qx;
}

// 64-lane wave reduction, then one atomic per wave
TSL_DEV double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
TSL_DEV double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}
// block (<=1024 threads) sum via LDS; result valid in thread 0
TSL_DEV double block_sum(double v, double* sm) {
  v = wave_sum(v);
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sm[w] = v;
  __syncthreads();
  double r = 0;
  if (threadIdx.x < 64) {
    int nw = (blockDim.x + 63) >> 6;
    r = (threadIdx.x < nw) ? sm[threadIdx.x] : 0.0;
    r = wave_sum(r);
  }
  __syncthreads();
  return r;
}
