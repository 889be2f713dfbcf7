// Numeric kernels of the multifrontal LU that preconditions the Newton / adjoint solves (layout: direct_plan.hpp).
// The reference calls a direct sparse solver for every system (sparse_solver.py:85-105); at 100k triangles its dense-backed
// storage is impossible, and the iterative solvers of k_solver.hpp need 10^2..10^5 iterations on the wrinkled cfg4 states, so the
// same operator is factorised here: one dense front per supernode of the nested-dissection tree, fronts of one tree level
// batched into the same launches.  Per front:  [F11 F12; F21 F22] -> [W = F11^-1, G = W F12; F21, S = F22 - F21 G].
//   * W: blocked in-place Gauss-Jordan on F11 (DS_T pivots per step): fronts with at most DS_SMALL pivots by one workgroup with
//     the block in LDS, larger ones with one launch per block step (the pivot block of the next step is inverted by one wave inside
//     the step kernel).  No pivoting: measured on the cfg4 operators (forward and un-projected adjoint) a nested-dissection LU with
//     diagonal pivots reaches a 1e-10 relative residual.
//   * G = W F12 and S = F22 - F21 G: K = pp GEMMs on the f64 matrix cores (v_mfma_f64_16x16x4_f64), the bulk of the flops.
//   * extend-add as a GATHER on the parent's side (round 4): the Schur GEMM's epilogue forms S = sum_children ext(S_child) - F21 G
//     -- reading the children's stored S through their tables over this front's dofs -- and STORES it; the parent's panels take
//     their share in k_ds_extend_panels before the parent is factorised.  One writer per entry, fixed child order: no atomics, no
//     cleared F22, bit-reproducible factors.
// A solve is three matrix-vector passes per level (W, F21 upwards; G downwards), no triangular recurrences.
#pragma once
#include "direct_plan.hpp"
#include "tsl_device.hpp"

typedef double ds_d4 __attribute__((ext_vector_type(4)));

struct DsDev {                 // device views shared by the kernels
  const DsFrontDesc* fr;       // by supernode id
  const DsFrontDesc* frl;      // the same descriptors in level order: frl[i] = fr[level_sn[i]] (one load instead of two dependent ones)
  const int* level_sn;         // front ids, level after level
  double* A;                   // panel arena: top rows [F11 | F12] (row stride ld) and F21 (row stride pp) of every front
  double* S;                   // Schur arena: S of every front (bp x bp, row stride bp)
  double* G;                   // G = W F12 of every front (pp x bp, row stride bp)
  double* scr;                 // per-level scratch (pivot-block inverses, row panel, column panel per front)
  double* Y;                   // boundary updates of the upward sweep: y_f at DsFrontDesc.yoff (b entries per front)
  const DsChildRec* ch;        // children of every front (DsFrontDesc.ch_off / nchild), ascending supernode id
  const int* pmap;             // per child: table over the parent's local dofs -> the child's boundary dof, or -1
  const int* vtx;              // local vertex -> PERMUTED vertex position (rows of the solver vectors)
  int* bad;                    // [1..3]: perturbed pivots of the last factorisation by front size class, [4] + [8..]: log of the first ones
  double piv_tol;              // a pivot below piv_tol x its own scale is replaced by that bound
  int dbg;                     // diagnostics ("ds_dbg"): 21 forces the dataflow-abort branch of solve_perm (tests), 30 records the device-clock trace of a dataflow chain
  unsigned long long* tlog;    // "ds_dbg" 30: device-clock stamps of the dataflow launch's first front (pivot publications [0, 64), steps finished by two far workgroups [64, 192), start [192]); else null
};

// infinity norm of the SELL-64 matrix (largest absolute row sum), the yardstick of the solve's backward error; out must be zeroed
__global__ void k_ds_rownorm(int NV, const int* __restrict__ slice_off, const int* __restrict__ slice_len, const double* __restrict__ vals, double* out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  double m = 0.0;
  if (p < NV) {
    const int s = p >> 6, lane = p & 63;
    const size_t base = (size_t)slice_off[s] * 9 + lane;
    double r0 = 0, r1 = 0, r2 = 0;
    for (int k = 0; k < slice_len[s]; k++) {
      const double* b = vals + base + (size_t)k * 64 * 9;
      r0 += fabs(b[0]) + fabs(b[64]) + fabs(b[128]);
      r1 += fabs(b[192]) + fabs(b[256]) + fabs(b[320]);
      r2 += fabs(b[384]) + fabs(b[448]) + fabs(b[512]);
    }
    m = fmax(r0, fmax(r1, r2));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long*)out, (unsigned long long)__double_as_longlong(m));
}

// tsl_bench_direct, inversion classes: ones on the diagonal of every pivot block (the arena holds the pattern 4.8e-4)
__global__ void k_ds_bench_diag(DsDev D) {
  const DsFrontDesc f = D.fr[blockIdx.x];
  for (int i = threadIdx.x; i < f.pp; i += blockDim.x) D.A[f.off + (size_t)i * f.ld + i] = 1.0;
}

// ---- assembly -------------------------------------------------------------------------------------------------------------
// The matrix entries of ONE tree level go into the panels of its fronts when the level starts: the panels were just written -- cleared
// (leaf level: the contiguous head of the panel arena) or stored by the gather of the children's Schur complements (k_ds_extend_panels)
// --, so nothing above the leaves is ever cleared.  One launch per level, three thread ranges:
//   * static pattern: block i of the level-ordered list lives at vals[src[i] + 64 e] (SELL-64, element e of the 3 x 3 block);
//     every block has ONE destination: a plain add;
//   * identity on the padding of the pivot blocks;
//   * contact blocks (their own launch behind it, only on levels that have any): 16 vertex-pair sub-blocks per constraint, grouped by
//     destination on the host.
__global__ void k_ds_assemble_level(int i0, int nblk, const int* __restrict__ src, const double* __restrict__ vals, const long long* __restrict__ dst, const int* __restrict__ dld,
                                    int lv0, int nf, const DsFrontDesc* __restrict__ frl, double* __restrict__ A, int cleared) {
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < (long)nblk * 9) {   // (src / dst / dld: the level-ordered block list of the plan, three independent loads per entry)
    const long i = i0 + t / 9;
    const int e = (int)(t % 9);
    double* a = &A[dst[i] + (long long)(e / 3) * dld[i] + e % 3];
    const double v = vals[(size_t)src[i] + 64 * e];
    // one writer per entry; the panels of the leaf level were CLEARED, not gathered: 0 + v without reading the zero back (75 of the launch's 240 MB)
    *a = cleared ? 0.0 + v : *a + v;
    return;
  }
  t -= (long)nblk * 9;
  if (t < (long)nf * DS_T) {
    const DsFrontDesc f = frl[lv0 + t / DS_T];
    const int i = f.p + (int)(t % DS_T);
    if (i < f.pp) A[f.off + (long long)i * f.ld + i] = 1.0;
  }
}
// contact sub-blocks of one level, behind k_ds_assemble_level on the same stream: one thread per entry of a destination GROUP sums the
// group's sub-blocks (ascending constraint, slot) and adds the sum -- one writer per entry, a fixed order, no atomics
__global__ void __launch_bounds__(256) k_ds_assemble_contacts_level(int g0, int ng, const int* __restrict__ gptr, const int* __restrict__ gent, const long long* __restrict__ gdst,
                                                                    const int* __restrict__ gld, const double* __restrict__ H, double* __restrict__ A) {
  // one WAVE per destination group: lanes over the group's members (l, l + 64, ...: a pair of table vertices under a folded cloth is shared
  // by hundreds of constraints), the nine entries joined by the fixed tree of wave_sum, lane 0 adds them
  const int gi = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (gi >= ng) return;
  const int g = g0 + gi;
  double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int k = gptr[g] + lane; k < gptr[g + 1]; k += 64) {
    const int x = gent[k], c = x >> 4, sub = x & 15;
    const double* Hb = H + (size_t)c * 144 + (3 * (sub >> 2)) * 12 + 3 * (sub & 3);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int cc = 0; cc < 3; cc++) v[3 * r + cc] += Hb[12 * r + cc];
  }
#pragma unroll
  for (int e = 0; e < 9; e++) v[e] = wave_sum(v[e]);
  if (lane == 0) {
    double* dst = A + gdst[g];
    const int ld = gld[g];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int cc = 0; cc < 3; cc++) dst[(long long)r * ld + cc] += v[3 * r + cc];
  }
}

// ---- blocked Gauss-Jordan on the top block rows ---------------------------------------------------------------------------
// In-place Gauss-Jordan inversion of one DS_T x DS_T tile in LDS by the 256-thread workgroup with 4 x 4 BLOCK pivots on the matrix cores:
// eight dependent block steps instead of 32 scalar pivots.  The tile inversion sits on the critical path of every block step of the
// factorisation (it is half of a step of the dataflow chains), so its form was chosen with device-clock measurements inside a kernel
// (scripts/micro/inv_bench.hip, scripts/micro/lat_probe.hip; history in DESIGN 9c):
//   form 1  every wave a quadrant, every lane eliminates the 4 x 4 pivot block itself, two matrix-core products per step        5.6 us
//   form 2  the same with a per-lane cofactor inverse and one product per step                                                  4.9 us
//   form 4  wave-specialised: two workers own 16 rows each, a scout forms the next pivot block and its inverse one step ahead   4.4 us
//   form 5  ONE wave, whole tile in registers, no barrier: a v_mfma_f64 blocks its wave for 32 cycles and nothing of the same
//           wave overlaps with it -- six products + ~70 vector instructions per step in one instruction stream                 3.7 us
//   form 6  form 4 with the scout's 4 x 4 inverse spread over the lanes and no branch inside the steps (below)                  3.3 us
// Forms 6 (ds_invert_tile) and 4 (ds_invert_tile_guarded, the path of tiles that fail the static-pivot rule) are kept.
// Static pivoting: a scalar pivot that fell below tol (DsDev.piv_tol) x its own scale -- the diagonal entry the tile came in with -- is
// replaced by that bound and counted in bad[cls]; the refinement outside absorbs the perturbation.  (NOT measured against the largest entry
// of the tile: contact blocks on degenerate triangles put 1e13 next to the m/dt^2 = 0.3 of a frozen dof, both exact.)  All 256 threads must
// call it; the tile is complete in LDS on return (the function ends with a barrier).
#define DS_PB 4
#define DS_BADLOG 64
#define DS_REDO 6   // bad[DS_REDO]: tiles of the last factorisation that went through the guarded form
#define DS_CLS(f) ((f).pp > 512 ? 3 : ((f).pp > 128 ? 2 : 1))
TSL_DEV double ds_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);   // hardware estimate (~2^-27 relative) + one Newton step: 2^-53 without the IEEE division sequence
  return fma(fma(-x, r, 1.0), r, r);
}
// lane-dependent choice among four registers as a chain of conditional moves (a nested ?: is lowered to exec-mask branches)
TSL_DEV double ds_sel4(bool k1, bool k2, bool k3, double a0, double a1, double a2, double a3) {
  double r = a0;
  r = k1 ? a1 : r;
  r = k2 ? a2 : r;
  r = k3 ? a3 : r;
  return r;
}
TSL_DEV double ds_readlane_d(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
TSL_DEV double ds_dpp(double v) {   // v of another lane of the same row of 16 lanes (DPP control CTRL), both halves
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
TSL_DEV double ds_quad_sum(double v) {   // sum over the four lanes of a quad, the same bits in all four
  v += ds_dpp<0xB1>(v);   // quad_perm [1, 0, 3, 2]
  v += ds_dpp<0x4E>(v);   // quad_perm [2, 3, 0, 1]
  return v;
}
// LDS of one tile inversion (5.4 KB; both forms use the one copy: k_ds_inv_small keeps two workgroups with a 96-pivot block on a CU only
// while the static arrays stay below 7.4 KB)
struct DsInvLds {
  double rowp[2][DS_PB][DS_T], colp[2][DS_PB][DS_T + 1];   // pivot rows R~ / pivot columns -C' of a step, double-buffered
  double dnext[2][DS_PB][DS_PB], dinv[2][DS_PB][DS_PB];    // X[n, n] at the next pivot position; the inverse the workers apply
  double dd[DS_PB][2 * DS_PB];                              // the scout's next pivot block, its columns stored twice (entry (a, b) = D[a][b & 3]: a lane reads them rotated)
  double red[2], dg0[DS_T];
  int redo;
};

// Form 4 (round 3): wave-specialised.  A block step of the lock-step forms is one dependent chain that every wave walks -- panels to LDS,
// barrier, pivot block back, 4 x 4 inverse, B operand, matrix-core update, back to LDS.  Here the chain is cut in two that run on different
// SIMDs between the same pair of barriers:
//   * waves 0 / 1 (workers) own the upper / lower 16 rows of the tile (two accumulator quadrants each, lane l, register r = element
//     (16 w + (l >> 4) + 4 r, 16 q + (l & 15)): the accumulator layout of v_mfma_f64_16x16x4_f64); in step s they read row lk of Dinv(s) from
//     LDS, form the B operands, update, and write the panels of step s+1 (pivot rows R~ with the unit block in the pivot columns, pivot
//     columns -C' with the pivot rows zeroed -- every entry outside the pivot rows then comes out of X~ - C' (Dinv R~) by itself, the pivot
//     rows are overwritten with Dinv R~ --, and the 4 x 4 corner X[n, n] at the NEXT pivot position);
//   * wave 2 (scout) forms the pivot block of step s+1 from the panels of step s by a 16x16x4 product of its own -- A operand the
//     4 x 4 corner of the column panel, B operand Dinv(s) R at the next pivot columns, C operand X[n, n] --, passes it round its lanes
//     through a private LDS patch, computes Dinv(s+1) (cofactor form: a lane reads D with the columns rotated by lk and forms row 0 of THAT
//     inverse -- 6 shared 2 x 2 minors, four 3 x 3 minors, one determinant, one reciprocal) and leaves it in LDS for the workers; wave 3
//     only keeps the barriers.
// Static-pivot rule on the cofactor path: when the expansion of the determinant cancels (|det| < 1e-6 sum |terms|) OR an entry of the inverse
// exceeds 1 / (tol x the entry diagonal of its row) -- a pivot that lost its digits before it reached this block -- the scout falls back to a
// four-pivot elimination with per-pivot thresholds (uniform branch).  Blocks with a zero leading entry but a healthy determinant
// ([0 1; 1 0]) are inverted exactly instead of being perturbed.
TSL_DEV void ds_invert_tile_guarded(DsInvLds& L, double* __restrict__ T, int ldt, int* __restrict__ bad, int cls, int tag, double tol) {
  auto& rowp = L.rowp; auto& colp = L.colp; auto& dnext = L.dnext; auto& dinv = L.dinv; auto& dloc = L.dd; auto& red = L.red; auto& dg0 = L.dg0;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
  const bool worker = w < 2, scout = w == 2;
  ds_d4 acc[2];
  if (worker) {
    double amax = 0.0;
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        acc[q][r] = T[(16 * w + lk + 4 * r) * ldt + 16 * q + lr];
        amax = fmax(amax, fabs(acc[q][r]));
        if (q == w && lk + 4 * r == lr) dg0[16 * w + lr] = fabs(acc[q][r]);   // the diagonal on entry: the scale a pivot is measured against
      }
    amax = wave_max(amax);
    if (lane == 0) red[w] = amax;
    if (w == 0 && lr < DS_PB) dnext[1][lk][lr] = acc[0][0];   // the first pivot block (rows lk, columns lr < 4, register 0) where step "-1" would have left it
  }
  __syncthreads();
  const double tmax = fmax(red[0], red[1]);
  const double floor0 = fmax(tmax * 1e-20, 1e-300);
  const double mydg = dg0[lane & 31];
  unsigned badmask = 0;
  const int c0 = lk, c1 = (lk + 1) & 3, c2 = (lk + 2) & 3, c3 = (lk + 3) & 3;   // column rotation of this lane
  // row lk of the inverse of the 4 x 4 block D (pivot rows p0..p0+3) -> drow
  auto inv_row = [&](auto D, int p0, double* drow) {
    double a[4], b[4], c[4], e[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { a[i] = D[i][c0]; b[i] = D[i][c1]; c[i] = D[i][c2]; e[i] = D[i][c3]; }
    const double m01 = c[0] * e[1] - c[1] * e[0], m02 = c[0] * e[2] - c[2] * e[0], m03 = c[0] * e[3] - c[3] * e[0];
    const double m12 = c[1] * e[2] - c[2] * e[1], m13 = c[1] * e[3] - c[3] * e[1], m23 = c[2] * e[3] - c[3] * e[2];
    const double M0 = b[1] * m23 - b[2] * m13 + b[3] * m12;
    const double M1 = b[0] * m23 - b[2] * m03 + b[3] * m02;
    const double M2 = b[0] * m13 - b[1] * m03 + b[3] * m01;
    const double M3 = b[0] * m12 - b[1] * m02 + b[2] * m01;
    const double t0 = a[0] * M0, t1 = a[1] * M1, t2 = a[2] * M2, t3 = a[3] * M3;
    const double det = (t0 - t1) + (t2 - t3);
    const double dabs = (fabs(t0) + fabs(t1)) + (fabs(t2) + fabs(t3));
    const double idet = ds_rcp(det);
    drow[0] = M0 * idet; drow[1] = -M1 * idet; drow[2] = M2 * idet; drow[3] = -M3 * idet;
    // static-pivot rule on the cofactor path (see form 2): no cancellation in the determinant AND no entry of the lane's row of the inverse
    // above 1 / (tol x the entry diagonal of that row); the whole wave takes the same path
    const bool ok = fabs(det) >= 1e-6 * dabs && dabs < 1e300 && fmax(fmax(fabs(drow[0]), fabs(drow[1])), fmax(fabs(drow[2]), fabs(drow[3]))) * (tol * dg0[p0 + lk]) <= 1.0;
    if (__builtin_amdgcn_ballot_w64(!ok) != 0) {   // four-pivot elimination on the unrotated block with the per-pivot threshold (first form), then row lk
      double d[DS_PB][DS_PB];
#pragma unroll
      for (int i = 0; i < DS_PB; i++)
#pragma unroll
        for (int j = 0; j < DS_PB; j++) d[i][j] = D[i][j];
#pragma unroll
      for (int p = 0; p < DS_PB; p++) {
        const double piv0 = d[p][p];
        const double tiny = fmax(tol * ds_readlane_d(mydg, p0 + p), floor0);
        const bool small = !(fabs(piv0) >= tiny);
        const double piv = small ? copysign(tiny, piv0) : piv0;
        badmask |= small ? (1u << (p0 + p)) : 0u;
        const double ip = ds_rcp(piv);
#pragma unroll
        for (int j = 0; j < DS_PB; j++) d[p][j] = (j == p) ? ip : d[p][j] * ip;
#pragma unroll
        for (int i = 0; i < DS_PB; i++) {
          if (i == p) continue;
          const double f = d[i][p];
#pragma unroll
          for (int j = 0; j < DS_PB; j++) d[i][j] = (j == p) ? -f * ip : fma(-f, d[p][j], d[i][j]);
        }
      }
      const bool k1 = lk == 1, k2 = lk == 2, k3 = lk == 3;
#pragma unroll
      for (int j = 0; j < DS_PB; j++) drow[j] = ds_sel4(k1, k2, k3, d[0][j], d[1][j], d[2][j], d[3][j]);
    }
  };
  // panels of step s from the workers' accumulators (before the barrier that opens step s)
  auto write_panels = [&](int s) {
    const int buf = s & 1, p0 = DS_PB * s, wp = p0 >> 4, rp = (p0 & 15) >> 2, lc = p0 & 15;
    const int n0 = p0 + DS_PB, wn = n0 >> 4, rn = (n0 & 15) >> 2, ln = n0 & 15;
    if (s + 1 < DS_T / DS_PB && w == wn && lr >= ln && lr < ln + DS_PB) dnext[buf][lk][lr - ln] = acc[wn][rn];   // X[n, n] before step s
    const bool col_in = lr >= lc && lr < lc + DS_PB;   // (of column half wp)
    const int mc = (lr - lc) & 3;
    if (w == wp) {   // pivot rows: R~ carries the unit block in the pivot columns
#pragma unroll
      for (int q = 0; q < 2; q++) rowp[buf][lk][16 * q + lr] = (q == wp && col_in) ? (mc == lk ? 1.0 : 0.0) : acc[q][rp];
    }
    if (col_in) {    // pivot columns, negated, zero in the pivot rows; X~ has zero pivot columns
#pragma unroll
      for (int r = 0; r < 4; r++) {
        colp[buf][mc][16 * w + lk + 4 * r] = (w == wp && r == rp) ? 0.0 : -acc[wp][r];
        acc[wp][r] = 0.0;
      }
    }
  };
  double drow[DS_PB] = {0.0, 0.0, 0.0, 0.0};
  if (worker) write_panels(0);
  if (scout) {
    inv_row(dnext[1], 0, drow);
    if (lr == 0) {
#pragma unroll
      for (int j = 0; j < DS_PB; j++) dinv[0][lk][j] = drow[j];
    }
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < DS_T / DS_PB; s++) {
    const int buf = s & 1, p0 = DS_PB * s, wp = p0 >> 4, rp = (p0 & 15) >> 2;
    const bool has_next = s + 1 < DS_T / DS_PB;
    if (worker) {
      double dr[DS_PB];
#pragma unroll
      for (int j = 0; j < DS_PB; j++) dr[j] = dinv[buf][lk][j];
      const double aop = colp[buf][lk][16 * w + lr];
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int col = 16 * q + lr;
        const double bop = dr[0] * rowp[buf][0][col] + dr[1] * rowp[buf][1][col] + dr[2] * rowp[buf][2][col] + dr[3] * rowp[buf][3][col];
        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc[q], 0, 0, 0);   // X~ - C' (Dinv R~)
        if (w == wp) acc[q][rp] = bop;                                                // pivot rows = Dinv R~ (Dinv itself in the pivot columns)
      }
      if (has_next) write_panels(s + 1);
    } else if (scout && has_next) {
      const int n0 = p0 + DS_PB, cn = n0 + (lr & 3);
      const double bn = drow[0] * rowp[buf][0][cn] + drow[1] * rowp[buf][1][cn] + drow[2] * rowp[buf][2][cn] + drow[3] * rowp[buf][3][cn];
      const double an = lr < DS_PB ? colp[buf][lk][n0 + lr] : 0.0;
      const ds_d4 cin = {lr < DS_PB ? dnext[buf][lk][lr & 3] : 0.0, 0.0, 0.0, 0.0};
      const ds_d4 dn = __builtin_amdgcn_mfma_f64_16x16x4f64(an, bn, cin, 0, 0, 0);    // D(s+1) = X[n, n] - C[n, :] (Dinv(s) R[:, n]) in lanes (lk, lr < 4), register 0
      if (lr < DS_PB) dloc[lk][lr] = dn[0];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      inv_row(dloc, n0, drow);
      if (lr == 0) {
#pragma unroll
        for (int j = 0; j < DS_PB; j++) dinv[buf ^ 1][lk][j] = drow[j];
      }
    }
    __syncthreads();
  }
  if (worker) {
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
      for (int r = 0; r < 4; r++) T[(16 * w + lk + 4 * r) * ldt + 16 * q + lr] = acc[q][r];
  }
  if (scout && lane == 0 && badmask) {
    atomicAdd(bad + cls, __popc(badmask));
    const int slot = atomicAdd(bad + 4, 1);
    if (slot < DS_BADLOG) { int* L = bad + 8 + 4 * slot; L[0] = tag; L[1] = (int)badmask; L[2] = 0; L[3] = __float_as_int((float)tmax); }
  }
  __syncthreads();
}


// Form 6 (round 6): form 4's wave specialisation with
//   * the scout's 4 x 4 inverse spread over the lanes: lane (lk, lr) forms ONE entry (row lk, column lr & 3) -- a 3 x 3 minor of the block as it
//     lies in LDS with its columns stored twice (the lane reads them rotated by lk), the determinant by the expansion along column lk summed over
//     the quad with two DPP steps -- instead of a whole row per lane from ~55 dependent f64 instructions (a v_fma_f64 issues every 5.3 cycles,
//     dependent or not: scripts/micro/lat_probe.hip);
//   * NO branch inside the eight steps: the static-pivot rule (no term of the expansion above 2.5e5 |det|, no entry of the inverse above
//     1 / (tol x the entry diagonal of its row); NaN fails both) is evaluated lane by lane and and-ed up over the steps (a compare + ballot + branch
//     costs ~120 cycles per step, lat_probe); a tile that fails it anywhere is inverted AGAIN from its untouched LDS image by form 4, whose
//     fall-back path perturbs and counts the pivots.  Which tiles take that way depends on their entries only: all three factorisation paths
//     still give the same bits;
//   * the tile maximum (needed by the fall-back path only) is not formed here at all (six ds_bpermute rounds in front of the first step).
// EVERY path of the factorisation (k_ds_pivot0 / k_ds_gj_step, k_ds_gj_flow, k_ds_inv_small) inverts its pivot tiles with this one routine and
// forms the same products in the same order: the factors do not depend on which kernel a batch ran in (tests/test_gpu_direct.py asserts equal bits).
TSL_DEV void ds_invert_tile(double* __restrict__ T, int ldt, int* __restrict__ bad, int cls, int tag, double tol) {
  __shared__ DsInvLds L;
  auto& rowp = L.rowp; auto& colp = L.colp; auto& dnext = L.dnext; auto& dinv = L.dinv; auto& dd = L.dd; auto& dg0 = L.dg0;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
  const bool worker = w < 2, scout = w == 2;
  const int mj = lr & 3;
  ds_d4 acc[2];
  if (worker) {
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        acc[q][r] = T[(16 * w + lk + 4 * r) * ldt + 16 * q + lr];
        if (q == w && lk + 4 * r == lr) dg0[16 * w + lr] = fabs(acc[q][r]);   // the diagonal on entry: the scale a pivot is measured against
      }
    if (w == 0 && lr < DS_PB) {   // the first pivot block (rows lk, columns lr < 4, register 0), columns stored twice for the scout's rotated reads
      const double v = acc[0][0];
      dd[lk][lr] = v; dd[lk][lr + DS_PB] = v;
    }
  }
  const double sgn = (mj & 1) ? -1.0 : 1.0;
  const double *q0 = &dd[mj == 0 ? 1 : 0][lk], *q1 = &dd[mj <= 1 ? 2 : 1][lk], *q2 = &dd[mj <= 2 ? 3 : 2][lk];   // the three rows other than mj, columns from lk on
  bool okall = true;
  // entry (lk, mj) of the inverse of the 4 x 4 block in dd (scout)
  auto inv_entry = [&](int p0) -> double {
    // row lk of the inverse = row 0 of the inverse of D with its columns rotated by lk (a = column lk, then b, c, e); the lane's entry mj is
    // (-1)^mj minor(row mj deleted; columns b, c, e) / det with the three remaining rows r0 < r1 < r2 in natural order, the minor expanded along
    // column b over 2 x 2 minors of columns (c, e) -- the expressions form 4 evaluates per lane; det = the expansion along a, summed over the quad
    const double a_ = dd[mj][lk];
    const double b0 = q0[1], c0 = q0[2], e0 = q0[3], b1 = q1[1], c1 = q1[2], e1 = q1[3], b2 = q2[1], c2 = q2[2], e2 = q2[3];
    const double tdg = tol * dg0[p0 + lk];
    const double m12 = c1 * e2 - c2 * e1, m02 = c0 * e2 - c2 * e0, m01 = c0 * e1 - c1 * e0;
    const double M = b0 * m12 - b1 * m02 + b2 * m01;
    double t = (sgn * a_) * M;
    // the term must be ROUNDED before the quad sum.  Contracted into it -- fma(sgn a, M, the neighbour's term), what the compiler makes of
    // `t + dpp(t)` -- the four lanes of a quad get four different determinants, and a row of the inverse whose entries are divided by different
    // determinants leaves eps (|t| / |det|)^2 instead of eps |t| / |det| in Dinv D - I: measured 1e-5 against 9e-8 relative residual of the
    // first application of the factors on the cfg4 operator, 1.6 instead of 1.0 applications per solve
    asm volatile("" : "+v"(t));
    const double det = ds_quad_sum(t);
    const double idet = ds_rcp(det);
    const double e = (M * sgn) * idet;
    okall &= (fabs(det) >= 4e-6 * fabs(t)) & (fabs(e) * tdg <= 1.0);
    return e;
  };
  // panels of step s from the workers' accumulators (before the barrier that opens step s)
  auto write_panels = [&](int s) {
    const int buf = s & 1, p0 = DS_PB * s, wp = p0 >> 4, rp = (p0 & 15) >> 2, lc = p0 & 15;
    const int n0 = p0 + DS_PB, wn = n0 >> 4, rn = (n0 & 15) >> 2, ln = n0 & 15;
    if (s + 1 < DS_T / DS_PB && w == wn && lr >= ln && lr < ln + DS_PB) dnext[buf][lk][lr - ln] = acc[wn][rn];   // X[n, n] before step s
    const bool col_in = lr >= lc && lr < lc + DS_PB;   // (of column half wp)
    if (w == wp) {   // pivot rows: R~ carries the unit block in the pivot columns
#pragma unroll
      for (int q = 0; q < 2; q++) rowp[buf][lk][16 * q + lr] = (q == wp && col_in) ? (mj == lk ? 1.0 : 0.0) : acc[q][rp];
    }
    if (col_in) {    // pivot columns, negated, zero in the pivot rows; X~ has zero pivot columns
#pragma unroll
      for (int r = 0; r < 4; r++) {
        colp[buf][mj][16 * w + lk + 4 * r] = (w == wp && r == rp) ? 0.0 : -acc[wp][r];
        acc[wp][r] = 0.0;
      }
    }
  };
  double e = 0.0;   // the scout's entry (lk, mj) of the current inverse
  if (worker) write_panels(0);
  __syncthreads();
  if (scout) {
    e = inv_entry(0);
    if (lr < DS_PB) dinv[0][lk][lr] = e;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < DS_T / DS_PB; s++) {
    const int buf = s & 1, p0 = DS_PB * s, wp = p0 >> 4, rp = (p0 & 15) >> 2;
    const bool has_next = s + 1 < DS_T / DS_PB;
    if (worker) {
      double dr[DS_PB];
#pragma unroll
      for (int j = 0; j < DS_PB; j++) dr[j] = dinv[buf][lk][j];
      const double aop = colp[buf][lk][16 * w + lr];
      double bop[2];
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int col = 16 * q + lr;
        bop[q] = dr[0] * rowp[buf][0][col] + dr[1] * rowp[buf][1][col] + dr[2] * rowp[buf][2][col] + dr[3] * rowp[buf][3][col];
      }
#pragma unroll
      for (int q = 0; q < 2; q++) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop[q], acc[q], 0, 0, 0);   // X~ - C' (Dinv R~)
      if (w == wp) {                                                                                            // pivot rows = Dinv R~ (Dinv itself in the pivot columns)
#pragma unroll
        for (int q = 0; q < 2; q++) acc[q][rp] = bop[q];
      }
      if (has_next) write_panels(s + 1);
    } else if (scout) {
      if (has_next) {
        const int n0 = p0 + DS_PB, cn = n0 + mj;
        const double bn = ds_dpp<0x00>(e) * rowp[buf][0][cn] + ds_dpp<0x55>(e) * rowp[buf][1][cn] + ds_dpp<0xAA>(e) * rowp[buf][2][cn] + ds_dpp<0xFF>(e) * rowp[buf][3][cn];
        const double an = lr < DS_PB ? colp[buf][lk][n0 + lr] : 0.0;
        const ds_d4 cin = {lr < DS_PB ? dnext[buf][lk][mj] : 0.0, 0.0, 0.0, 0.0};
        const ds_d4 dn = __builtin_amdgcn_mfma_f64_16x16x4f64(an, bn, cin, 0, 0, 0);    // D(s+1) = X[n, n] - C[n, :] (Dinv(s) R[:, n]) in lanes (lk, lr < 4), register 0
        if (lr < DS_PB) { const double v = dn[0]; dd[lk][lr] = v; dd[lk][lr + DS_PB] = v; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        e = inv_entry(n0);
        if (lr < DS_PB) dinv[buf ^ 1][lk][lr] = e;
      } else {
        L.redo = __builtin_amdgcn_ballot_w64(!okall) != 0 ? 1 : 0;   // any lane, any step (every lane writes the same value)
      }
    }
    __syncthreads();
  }
  if (L.redo) {   // (uniform) a step met a pivot block the cofactor path must not invert: the whole tile again, from T, with the guarded form
    if (threadIdx.x == 0) atomicAdd(bad + DS_REDO, 1);
    __syncthreads();
    ds_invert_tile_guarded(L, T, ldt, bad, cls, tag, tol);
    return;
  }
  if (worker) {
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
      for (int r = 0; r < 4; r++) T[(16 * w + lk + 4 * r) * ldt + 16 * q + lr] = acc[q][r];
  }
  __syncthreads();
}

// scratch of a front inside the level scratch (fronts with more than DS_SMALL pivots): pivot-block inverses P[2] (ping-pong) and the
// side panels of the merged Gauss-Jordan step, row panel R[2] (DS_T x pp) and column panel C[2] (pp x DS_T)
TSL_DEV double* ds_scr_P(const DsDev& D, const DsFrontDesc& f, int which) { return D.scr + f.scr_off + which * DS_T * DS_T; }
TSL_DEV double* ds_scr_R(const DsDev& D, const DsFrontDesc& f, int which) { return D.scr + f.scr_off + 2 * DS_T * DS_T + (size_t)which * DS_T * f.pp; }
TSL_DEV double* ds_scr_C(const DsDev& D, const DsFrontDesc& f, int which) { return D.scr + f.scr_off + 2 * DS_T * DS_T + (size_t)(2 + which) * DS_T * f.pp; }

// inverse of the first pivot block of every front of the batch -> P[0]
__global__ void __launch_bounds__(256) k_ds_pivot0(DsDev D, int lv0) {
  __shared__ double T[DS_T][DS_T + 1];
  const DsFrontDesc f = D.frl[lv0 + blockIdx.x];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const double* A = D.A + f.off;
#pragma unroll
  for (int q = 0; q < 4; q++) T[ty + 8 * q][tx] = A[(size_t)(ty + 8 * q) * f.ld + tx];
  __syncthreads();
  ds_invert_tile(&T[0][0], DS_T + 1, D.bad, DS_CLS(f), D.level_sn[lv0 + blockIdx.x] << 6, D.piv_tol);
  double* P = ds_scr_P(D, f, 0);
#pragma unroll
  for (int q = 0; q < 4; q++) P[(ty + 8 * q) * DS_T + tx] = T[ty + 8 * q][tx];
}

// Block step k of the in-place Gauss-Jordan inversion W = F11^-1 as ONE launch (pivot rows K = [k T, k T + T)):
//   A_KK = P = inv(A_KK),  A_Kj = R'_j = P A_Kj,  A_iK = -A_iK P,  A_ij -= A_iK R'_j   (i, j outside K).
// Every workgroup owns one T x T tile and forms the R'_j it needs itself (one extra 32^3 product), so there is no panel launch.  The
// tiles of row K and column K are read by the other workgroups of the same launch, hence their new values go to side panels
// (R[k & 1], C[k & 1]) and the NEXT step reads row / column K from there; whoever owns such a tile in step k + 1 writes its updated
// value back into the front.  The workgroup of tile (k+1, k+1) also inverts its result (the next pivot block) into P[(k+1) & 1];
// it is dispatched first so that the single-wave inversion overlaps with the other tiles.  k_ds_gj_finish copies the side
// panels of the last step back.
TSL_DEV double ds_tile_elem(const double* __restrict__ A, int ld, const double* __restrict__ Rs, const double* __restrict__ Cs, int pp, int kp, int ti, int tj, int r, int c) {
  if (ti == kp) return Rs[(size_t)r * pp + tj * DS_T + c];
  if (tj == kp) return Cs[(size_t)(ti * DS_T + r) * DS_T + c];
  return A[(size_t)(ti * DS_T + r) * ld + tj * DS_T + c];
}
// Launch shape: 1-D grid of nf + nf tp^2 workgroups -- the first nf are the pivot workgroups (tile (k+1, k+1)) of the nf fronts, so
// that the longest workgroup of EVERY front is dispatched before any of the update tiles (with a (tp, tp, nf) grid the pivot
// workgroup of front z sat behind z tp^2 others, and the launch ended one dispatch round later), then the tiles front by front.
__global__ void __launch_bounds__(256) k_ds_gj_step(DsDev D, int lv0, int k, int tp, int nf) {
  __shared__ double Ps[DS_T][DS_T + 1];
  __shared__ double T1[DS_T][DS_T + 1];   // A[K, j], later R'_j
  __shared__ double T2[DS_T][DS_T + 1];   // A[i, K], later the next pivot block
  const int L = blockIdx.x;
  int fz = L, bi = k + 1, bj = k + 1;
  if (L >= nf) { const int q = L - nf, t2 = tp * tp; fz = q / t2; const int t = q - fz * t2; bi = t / tp; bj = t - bi * tp; }
  const DsFrontDesc f = D.frl[lv0 + fz];
  const int pp = f.pp, k0 = k * DS_T;
  if (k0 >= pp) return;
  const bool has_next = k0 + DS_T < pp;
  if (L < nf) { if (!has_next) return; }
  else if (has_next && bi == k + 1 && bj == k + 1) return;   // done by the pivot workgroup
  if (bi * DS_T >= pp || bj * DS_T >= pp) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wi = w >> 1, wj = w & 1, lr = lane & 15, lk = lane >> 4;
  double* A = D.A + f.off;
  const int ld = f.ld, kp = k - 1;
  const double* Pin = ds_scr_P(D, f, k & 1);
  const double* Rs = ds_scr_R(D, f, kp & 1);
  const double* Cs = ds_scr_C(D, f, kp & 1);
  double* Rn = ds_scr_R(D, f, k & 1);
  double* Cn = ds_scr_C(D, f, k & 1);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int r = ty + 8 * q;
    Ps[r][tx] = Pin[r * DS_T + tx];
    if (bj != k) T1[r][tx] = ds_tile_elem(A, ld, Rs, Cs, pp, kp, k, bj, r, tx);
    if (bi != k) T2[r][tx] = ds_tile_elem(A, ld, Rs, Cs, pp, kp, bi, k, r, tx);
  }
  __syncthreads();
  if (bi == k && bj == k) {  // A_KK = P
#pragma unroll
    for (int q = 0; q < 4; q++) Rn[(size_t)(ty + 8 * q) * pp + k0 + tx] = Ps[ty + 8 * q][tx];
    return;
  }
  // the tile's own entries (quadrant layout of the matrix-core result) are requested before the products
  ds_d4 old = {0.0, 0.0, 0.0, 0.0};
  if (bi != k && bj != k) {
#pragma unroll
    for (int r = 0; r < 4; r++) old[r] = ds_tile_elem(A, ld, Rs, Cs, pp, kp, bi, bj, 16 * wi + lk + 4 * r, 16 * wj + lr);
  }
  ds_d4 acc = {0.0, 0.0, 0.0, 0.0};
  if (bj == k) {  // A_iK = -A_iK P
#pragma unroll
    for (int kk = 0; kk < DS_T / 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(T2[16 * wi + lr][4 * kk + lk], Ps[4 * kk + lk][16 * wj + lr], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; r++) Cn[(size_t)(bi * DS_T + 16 * wi + lk + 4 * r) * DS_T + 16 * wj + lr] = -acc[r];
    return;
  }
  // R'_j = P A_Kj (quadrant (wi, wj))
#pragma unroll
  for (int kk = 0; kk < DS_T / 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ps[16 * wi + lr][4 * kk + lk], T1[4 * kk + lk][16 * wj + lr], acc, 0, 0, 0);
  if (bi == k) {
#pragma unroll
    for (int r = 0; r < 4; r++) Rn[(size_t)(16 * wi + lk + 4 * r) * pp + bj * DS_T + 16 * wj + lr] = acc[r];
    return;
  }
  __syncthreads();   // every quadrant has read T1
#pragma unroll
  for (int r = 0; r < 4; r++) T1[16 * wi + lk + 4 * r][16 * wj + lr] = acc[r];
  __syncthreads();
  acc = ds_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < DS_T / 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(T2[16 * wi + lr][4 * kk + lk], T1[4 * kk + lk][16 * wj + lr], acc, 0, 0, 0);
  const bool next_pivot = has_next && bi == k + 1 && bj == k + 1;
  if (next_pivot) __syncthreads();   // every quadrant has read T2
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = 16 * wi + lk + 4 * r, col = 16 * wj + lr;
    const double v = old[r] - acc[r];
    A[(size_t)(bi * DS_T + row) * ld + bj * DS_T + col] = v;
    if (next_pivot) T2[row][col] = v;
  }
  if (next_pivot) {
    __syncthreads();
    ds_invert_tile(&T2[0][0], DS_T + 1, D.bad, DS_CLS(f), (D.level_sn[lv0 + fz] << 6) | (k + 1), D.piv_tol);
    double* Pn = ds_scr_P(D, f, (k + 1) & 1);
#pragma unroll
    for (int q = 0; q < 4; q++) Pn[(ty + 8 * q) * DS_T + tx] = T2[ty + 8 * q][tx];
  }
}
// side panels of the last block step back into the front: workgroup (b, front) copies row-panel chunk b and column-panel chunk b
__global__ void __launch_bounds__(256) k_ds_gj_finish(DsDev D, int lv0) {
  const DsFrontDesc f = D.frl[lv0 + blockIdx.y];
  const int pp = f.pp, b0 = blockIdx.x * DS_T;
  if (b0 >= pp) return;
  const int kl = pp / DS_T - 1;
  const double* Rs = ds_scr_R(D, f, kl & 1);
  const double* Cs = ds_scr_C(D, f, kl & 1);
  double* A = D.A + f.off;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int r = ty + 8 * q;
    A[(size_t)(kl * DS_T + r) * f.ld + b0 + tx] = Rs[(size_t)r * pp + b0 + tx];
    if ((int)blockIdx.x != kl) A[(size_t)(b0 + r) * f.ld + kl * DS_T + tx] = Cs[(size_t)(b0 + r) * DS_T + tx];
  }
}

// ---- dataflow form of the same block Gauss-Jordan: ONE persistent launch per batch ("direct_flow") --------------------------------
// On the upper levels of the tree (1 - 16 fronts) a block step is a dependent launch of ~13.4 us of which the arithmetic is a fraction.
// Here the tiles stay in registers (matrix-core result layout) over all block steps of their front and the steps are ordered by
// point-to-point flags instead of kernel boundaries (`scripts/micro/flag_chain.hip`: a hop -- publish an 8 KB tile, raise a flag, see it
// from another workgroup, fetch the tile -- is 2.1-2.5 us for 4 to 1024 workgroups):
//   * what another workgroup needs of a tile is PUBLISHED into an exchange slot with agent-scope (write-through) stores, the publisher
//     waits for their completion and then raises the slot's flag to this launch's epoch; readers poll the flag and fetch the slot with
//     agent-scope loads (no L2 invalidation: an acquire fence per workgroup and step costs 30 ns x the number of workgroups per hop);
//   * tile (i, j) publishes twice at most: when step i is next (it lies in the row panel of that step) and when step j is next (column
//     panel); the owner of the next pivot tile inverts it and publishes the inverse P[k + 1];
//   * nobody but the owner reads the front itself, so the result goes back in place at the end.
// Round 5: a workgroup owns a SUPER-TILE of B x B tiles (B = 2: 64 x 64 entries).  The pivots still advance 32 at a time -- the arithmetic
// per entry is that of k_ds_gj_step, product for product, so both paths give the SAME BITS --, but the pivot tiles of B consecutive steps
// lie in one workgroup: the chain  fetch P[k] -> two 32^3 products -> inversion -> publication  of the one-tile-per-workgroup form (8-11 us
// per step of which 4.3 us the inversion) pays its hop once per B steps, the steps in between go  P[k] (LDS) -> two products -> inversion.
// The owner of the next pivot updates THAT tile first, inverts it, and only then touches its other tiles; a published inverse is flagged
// at once where the chain leaves the workgroup, together with the panel tiles where it stays.  A quarter of the workgroups (289 instead
// of 1089 for the 1056-pivot root of cfg4: two per CU, no spilled fifth one), 1.5 instead of 2 products per tile and step (R'_j = P A_Kj
// is shared by the B tiles of a column).
// The launch must be resident as a whole (the host checks workgroups <= CUs x occupancy and runs it only for a batch alone on its level);
// a flag that does not come within DS_FLOW_SPINS polls raises bad[DS_FLOW_ABORT] and lets every workgroup run out (the host reports it).
#define DS_FLOW_MAXF 128
#define DS_FLOW_ABORT 5
#define DS_FLOW_ARRIVE 7   // bad[DS_FLOW_ARRIVE]: workgroups of the dataflow launches of this factorisation that have STARTED (k_ds_flow_gate waits for a launch to be resident)
#define DS_FLOW_SPINS (1 << 19)   // (4096 polls back to back, then one per ~2 us: about a second)
#define DS_FLOW_B 2
struct DsFlowArgs {
  int nf, epoch;
  int tile0[DS_FLOW_MAXF + 1];     // first workgroup of front z of the batch (a front of nt x nt tiles has ceil(nt / B)^2 workgroups)
  int foff[DS_FLOW_MAXF];          // first flag of front z: nt pivot flags (one per 128 B), nt^2 row-panel flags, nt^2 column-panel flags
  long long xoff[DS_FLOW_MAXF];    // first exchange slot of front z (doubles): nt pivot inverses, nt^2 row-panel slots, nt^2 column-panel slots
};
// a flag is raised once per launch to the launch's epoch (a stale or foreign value never passes)
TSL_DEV void ds_flow_poll(const int* flag, int epoch, int* abort_w, int* s_dead) {
  int spins = 0;
  for (;;) {
    const int v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v == epoch) break;
    if (++spins >= DS_FLOW_SPINS) {
      if (atomicCAS(abort_w, 0, 1) == 0) {   // the first to give up leaves a note for the host's report: workgroups of this factorisation's dataflow launches that had started, this launch's grid, who waited for which flag
        int* note = abort_w - DS_FLOW_ABORT + 8 + 4 * DS_BADLOG;
        note[0] = __hip_atomic_load(abort_w - DS_FLOW_ABORT + DS_FLOW_ARRIVE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); note[1] = (int)gridDim.x; note[2] = (int)blockIdx.x; note[3] = v; note[4] = epoch;
      }
      *s_dead = 1; return;
    }
    if ((spins & 1023) == 0 && __hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { *s_dead = 1; return; }
    if (spins > 4096) __builtin_amdgcn_s_sleep(64);   // (a flag of the chain comes within microseconds: a wait this long is a launch that is not resident yet -- stop hammering the flag's memory channel)
  }
}
TSL_DEV void ds_flow_fetch(double (*T)[DS_T + 1], const double* __restrict__ slot, int tx, int ty) {
#pragma unroll
  for (int q = 0; q < 4; q++) T[ty + 8 * q][tx] = __hip_atomic_load(slot + (ty + 8 * q) * DS_T + tx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the stores of every wave are complete (write-through, waited for) before a flag is raised
// (a workgroup-scope release fence alone emits no wait on gfx950: the flag overtook the tiles -- 11 instead of 2 refinement iterations on cfg4)
TSL_DEV void ds_flow_commit() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0): the write-through stores of this wave are acknowledged
  __syncthreads();
}
TSL_DEV void ds_flow_raise(int* flag, int epoch) { __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one quadrant of the 32 x 32 x 32 product Am Bm (LDS tiles, row stride DS_T + 1): the operand order of k_ds_gj_step
TSL_DEV ds_d4 ds_prod32(const double (*Am)[DS_T + 1], const double (*Bm)[DS_T + 1], int wi, int wj, int lr, int lk) {
  ds_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < DS_T / 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Am[16 * wi + lr][4 * kk + lk], Bm[4 * kk + lk][16 * wj + lr], acc, 0, 0, 0);
  return acc;
}
// One wave that ends when `target` workgroups of this factorisation's dataflow launches have started (or after DS_FLOW_SPINS polls): launched on the look-ahead's
// side stream in front of the eager sweeps (thousands of small workgroups), so that the chain of the level is resident before they arrive.  (Insurance: the stalls
// that were measured -- see direct_factor -- were all launches of more workgroups than CUs, which now start with the side stream drained.)
__global__ void k_ds_flow_gate(const int* arrive, int target) {
  if (threadIdx.x != 0) return;
  for (int spins = 0; spins < DS_FLOW_SPINS; spins++) {
    if (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return;
    __builtin_amdgcn_s_sleep(32);
  }
}
template <int B>
__global__ void __launch_bounds__(256, 2) k_ds_gj_flow(DsDev D, int lv0, DsFlowArgs a, double* __restrict__ X, int* __restrict__ Fl) {
  __shared__ double Pb[2][DS_T][DS_T + 1];   // P[k] of the current step / the inverse being formed for the next one
  __shared__ double Rs[B][DS_T][DS_T + 1];   // per owned column: A[K, j] as it was before the step, then R'_j = P A[K, j]
  __shared__ double Cs[B][DS_T][DS_T + 1];   // per owned row: A[i, K] as it was before the step
  __shared__ int s_dead;
  const int L = blockIdx.x;
  int z = 0;
  for (int step = DS_FLOW_MAXF / 2; step > 0; step >>= 1) { const int q = z + step; if (q < a.nf && L >= a.tile0[q]) z = q; }   // last front with tile0 <= L
  const int sn = D.level_sn[lv0 + z];
  const DsFrontDesc f = D.frl[lv0 + z];
  const int nt = f.pp / DS_T, ns = (nt + B - 1) / B, t = L - a.tile0[z], I = t / ns, J = t - I * ns;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wi = w >> 1, wj = w & 1, lr = lane & 15, lk = lane >> 4;
  const int epoch = a.epoch, cls = DS_CLS(f);
  double* Xp = X + a.xoff[z];
  int* Fp = Fl + a.foff[z];
  int* abort_w = D.bad + DS_FLOW_ABORT;
#define DS_FLOW_PSLOT(k) (Xp + (size_t)(k) * (DS_T * DS_T))
#define DS_FLOW_RSLOT(i, j) (Xp + (size_t)(nt + (i) * nt + (j)) * (DS_T * DS_T))
#define DS_FLOW_CSLOT(i, j) (Xp + (size_t)(nt + nt * nt + (i) * nt + (j)) * (DS_T * DS_T))
#define DS_FLOW_PFLAG(k) (Fp + 32 * (k))
#define DS_FLOW_RFLAG(i, j) (Fp + 32 * nt + (i) * nt + (j))
#define DS_FLOW_CFLAG(i, j) (Fp + 32 * nt + nt * nt + (i) * nt + (j))
  if (threadIdx.x == 0) { s_dead = 0; atomicAdd(D.bad + DS_FLOW_ARRIVE, 1); }
  const bool tl = D.tlog != nullptr && z == 0 && threadIdx.x == 0;
  if (tl && t == 0) D.tlog[192] = wall_clock64();
  if (D.tlog != nullptr && threadIdx.x == 0) atomicMin(&D.tlog[194], wall_clock64());   // first workgroup of the launch to start
  bool va[B], vb[B];   // rows / columns of the super-tile inside the front
#pragma unroll
  for (int q = 0; q < B; q++) { va[q] = B * I + q < nt; vb[q] = B * J + q < nt; }
  ds_d4 own[B][B];
  const int qr = 16 * wi + lk, qc = 16 * wj + lr;   // this lane's quadrant entries: rows qr + 4 r, column qc
#pragma unroll
  for (int x = 0; x < B; x++)
#pragma unroll
    for (int y = 0; y < B; y++) {
      const double* At = D.A + f.off + (size_t)((B * I + x) * DS_T) * f.ld + (B * J + y) * DS_T;
#pragma unroll
      for (int r = 0; r < 4; r++) own[x][y][r] = (va[x] && vb[y]) ? At[(size_t)(qr + 4 * r) * f.ld + qc] : 0.0;
    }
  auto to_slot = [&](double* slot, const ds_d4& v) {
#pragma unroll
    for (int r = 0; r < 4; r++) __hip_atomic_store(slot + (qr + 4 * r) * DS_T + qc, v[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto to_lds = [&](double (*T)[DS_T + 1], const ds_d4& v) {
#pragma unroll
    for (int r = 0; r < 4; r++) T[qr + 4 * r][qc] = v[r];
  };
  __syncthreads();
  // k = -1 is the prologue (nothing to update, pivot 0 "is next"); step k >= 0 eliminates the pivots [32 k, 32 k + 32)
  for (int k = -1; k < nt; k++) {
    const int K = k >= 0 ? k / B : -1, ka = k - B * K;
    const bool rowK = k >= 0 && I == K, colK = k >= 0 && J == K;
    const int kn = k + 1, Kn = kn / B, an = kn - B * Kn;
    const bool more = kn < nt;
    const bool rowN = more && I == Kn, colN = more && J == Kn, pivn = rowN && colN;
    double (*Ps)[DS_T + 1] = Pb[k & 1];
    double (*Pn)[DS_T + 1] = Pb[kn & 1];
    if (k >= 0) {
      // ---- operands of the step.  Every workgroup of the front passes here once per step, so the memory round trips of this block ARE the
      // step rate of the launch: every flag is polled first (one thread per wave; where the chain has just changed workgroups the panel tiles
      // and P[k] arrive within a microsecond of each other), then ALL tiles -- up to 2 B panel tiles and P[k] -- are requested together, parked
      // in registers and written to LDS: ONE round trip.  (Tile by tile -- load, wait, store to LDS, next tile -- the five fetches of a step
      // were five dependent round trips: 1.07 instead of 0.97 ms per factorisation for the chains of cfg4; panels and P[k] in two trips: 10.3 us
      // per hop step of the root against 7.1 us for a step that stays in its workgroup.)
      bool need_r[B], need_c[B];   // tiles of the pivot row / column that live in another workgroup
#pragma unroll
      for (int q = 0; q < B; q++) { need_r[q] = vb[q] && !rowK && !(colK && q == ka); need_c[q] = va[q] && !colK && !(rowK && q == ka); }
      const bool fetch_p = !(rowK && colK);   // (the pivot's owner left P[k] in Ps when it inverted the tile)
#pragma unroll
      for (int y = 0; y < B; y++) if (need_r[y] && (int)threadIdx.x == 64 * (y & 3)) ds_flow_poll(DS_FLOW_RFLAG(k, B * J + y), epoch, abort_w, &s_dead);
#pragma unroll
      for (int x = 0; x < B; x++) if (need_c[x] && (int)threadIdx.x == 64 * ((B + x) & 3)) ds_flow_poll(DS_FLOW_CFLAG(B * I + x, k), epoch, abort_w, &s_dead);
      if (fetch_p && threadIdx.x == 32) ds_flow_poll(DS_FLOW_PFLAG(k), epoch, abort_w, &s_dead);
      __syncthreads();   // (also: every wave is done with Rs / Cs / Pn of the previous step)
      if (tl && pivn && kn < 64) D.tlog[256 + 8 * kn] = wall_clock64();
      double pr[B][4], pc[B][4], pp[4];
#pragma unroll
      for (int y = 0; y < B; y++) if (need_r[y]) {
        const double* slot = DS_FLOW_RSLOT(k, B * J + y);
#pragma unroll
        for (int q = 0; q < 4; q++) pr[y][q] = __hip_atomic_load(slot + (ty + 8 * q) * DS_T + tx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int x = 0; x < B; x++) if (need_c[x]) {
        const double* slot = DS_FLOW_CSLOT(B * I + x, k);
#pragma unroll
        for (int q = 0; q < 4; q++) pc[x][q] = __hip_atomic_load(slot + (ty + 8 * q) * DS_T + tx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (fetch_p) {
        const double* slot = DS_FLOW_PSLOT(k);
#pragma unroll
        for (int q = 0; q < 4; q++) pp[q] = __hip_atomic_load(slot + (ty + 8 * q) * DS_T + tx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // local panel tiles (this workgroup lies in the pivot's super-row / super-column): registers -> LDS
#pragma unroll
      for (int y = 0; y < B; y++) if (vb[y] && rowK && !(colK && y == ka)) {
#pragma unroll
        for (int x = 0; x < B; x++) if (x == ka) to_lds(Rs[y], own[x][y]);
      }
#pragma unroll
      for (int x = 0; x < B; x++) if (va[x] && colK && !(rowK && x == ka)) {
#pragma unroll
        for (int y = 0; y < B; y++) if (y == ka) to_lds(Cs[x], own[x][y]);
      }
#pragma unroll
      for (int y = 0; y < B; y++) if (need_r[y]) {
#pragma unroll
        for (int q = 0; q < 4; q++) Rs[y][ty + 8 * q][tx] = pr[y][q];
      }
#pragma unroll
      for (int x = 0; x < B; x++) if (need_c[x]) {
#pragma unroll
        for (int q = 0; q < 4; q++) Cs[x][ty + 8 * q][tx] = pc[x][q];
      }
      if (fetch_p) {
#pragma unroll
        for (int q = 0; q < 4; q++) Ps[ty + 8 * q][tx] = pp[q];
      }
      __syncthreads();
      if (tl && pivn && kn < 64) D.tlog[256 + 8 * kn + 1] = wall_clock64();
      if (s_dead) break;
      // ---- R'_j = P A[K, j] for the owned columns (the pivot row's tiles ARE R'_j afterwards).  Only the products a tile needs are formed: a
      // 32^3 product is 8 v_mfma_f64_16x16x4_f64 per wave at 64 cycles each -- 0.2 us at the matrix cores' rate, 0.5 us with its LDS operand
      // reads and barriers -- and every one in front of the inversion is on the chain (all B + B x B products of a step unconditionally, as one
      // straight line of matrix instructions: 2.2 instead of 1.2 us per step of the root's pivot owner).
      ds_d4 rp[B];
#pragma unroll
      for (int y = 0; y < B; y++) if (vb[y] && !(colK && y == ka)) rp[y] = ds_prod32(Ps, Rs[y], wi, wj, lr, lk);
      __syncthreads();   // every quadrant has read Rs
#pragma unroll
      for (int y = 0; y < B; y++) {
        if (!vb[y] || (colK && y == ka)) continue;
        to_lds(Rs[y], rp[y]);
        if (rowK) {
#pragma unroll
          for (int x = 0; x < B; x++) if (x == ka) own[x][y] = rp[y];
        }
      }
      __syncthreads();
    }
    // A[i, j] -= A[i, K] R'_j,  A[i, K] = -A[i, K] P   (i outside the pivot rows), in up to three passes: 0 the next pivot tile, 1 the other
    // tiles of the next pivot's row and column (published as its panels), 2 the rest
    auto update = [&](int pass_lo, int pass_hi) {
#pragma unroll
      for (int x = 0; x < B; x++) {
        if (!va[x] || (rowK && x == ka)) continue;
#pragma unroll
        for (int y = 0; y < B; y++) {
          if (!vb[y]) continue;
          const int pass = (pivn && x == an && y == an) ? 0 : (((rowN && x == an) || (colN && y == an)) ? 1 : 2);
          if (pass < pass_lo || pass > pass_hi) continue;
          if (colK && y == ka) own[x][y] = -ds_prod32(Cs[x], Ps, wi, wj, lr, lk);
          else own[x][y] -= ds_prod32(Cs[x], Rs[y], wi, wj, lr, lk);
        }
      }
    };
    // tiles of the next step's pivot row (read by the other super-rows) and pivot column -> their exchange slots (stores only)
    auto store_panels = [&]() {
      if (rowN) {
#pragma unroll
        for (int y = 0; y < B; y++) {
          if (!vb[y] || (colN && y == an)) continue;
#pragma unroll
          for (int x = 0; x < B; x++) if (x == an) to_slot(DS_FLOW_RSLOT(kn, B * J + y), own[x][y]);
        }
      }
      if (colN) {
#pragma unroll
        for (int x = 0; x < B; x++) {
          if (!va[x] || (rowN && x == an)) continue;
#pragma unroll
          for (int y = 0; y < B; y++) if (y == an) to_slot(DS_FLOW_CSLOT(B * I + x, kn), own[x][y]);
        }
      }
    };
    auto raise_panels = [&]() {   // (after ds_flow_commit)
      const int tq = threadIdx.x;
      if (rowN && tq >= 1 && tq <= B) { const int y = tq - 1; if (B * J + y < nt && !(colN && y == an)) ds_flow_raise(DS_FLOW_RFLAG(kn, B * J + y), epoch); }
      if (colN && tq >= 1 + B && tq <= 2 * B) { const int x = tq - 1 - B; if (B * I + x < nt && !(rowN && x == an)) ds_flow_raise(DS_FLOW_CFLAG(B * I + x, kn), epoch); }
    };
    const bool pub = ns > 1 && (rowN || colN);
    if (pivn) {
      // The owner of the next pivot updates THAT tile first, inverts it in Pn and publishes it as P[k + 1].  Where the chain LEAVES this
      // workgroup (the pivot after it lies in the next super-tile) the inverse is flagged at once: the next owner waits for nothing else of
      // this workgroup; its other tiles and their panel publications follow.  Where the chain STAYS, the workgroups of this super-row and
      // super-column -- who feed the owner after that -- need P[k + 1] AND this workgroup's panel tiles: those are updated and sent off in
      // front of the inversion (their stores complete behind it) and one completion wait flags everything.
      const bool leaves = an == B - 1 || kn + 1 >= nt;
      if (k >= 0) update(0, leaves ? 0 : 1);
      if (pub && !leaves) store_panels();
#pragma unroll
      for (int x = 0; x < B; x++)
#pragma unroll
        for (int y = 0; y < B; y++) if (x == an && y == an) to_lds(Pn, own[x][y]);
      __syncthreads();
      if (tl && kn < 64) D.tlog[256 + 8 * kn + 3] = wall_clock64();
      ds_invert_tile(&Pn[0][0], DS_T + 1, D.bad, cls, (sn << 6) | kn, D.piv_tol);   // (ends with a barrier)
      if (tl && kn < 64) D.tlog[256 + 8 * kn + 4] = wall_clock64();
      if (ns > 1) {
        double* slot = DS_FLOW_PSLOT(kn);
#pragma unroll
        for (int q = 0; q < 4; q++) __hip_atomic_store(slot + (ty + 8 * q) * DS_T + tx, Pn[ty + 8 * q][tx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int x = 0; x < B; x++)
#pragma unroll
        for (int y = 0; y < B; y++) if (x == an && y == an) {
#pragma unroll
          for (int r = 0; r < 4; r++) own[x][y][r] = Pn[qr + 4 * r][qc];
        }
      if (ns > 1) {
        ds_flow_commit();
        if (threadIdx.x == 0) { ds_flow_raise(DS_FLOW_PFLAG(kn), epoch); if (tl && kn < 64) D.tlog[kn] = wall_clock64(); }
        if (!leaves) raise_panels();
      }
      if (k >= 0) update(leaves ? 1 : 2, 2);
      if (pub && leaves) { store_panels(); ds_flow_commit(); raise_panels(); }
    } else {
      if (k >= 0) update(1, 1);
      if (pub) store_panels();
      if (k >= 0) update(2, 2);
      if (pub) { ds_flow_commit(); raise_panels(); }
    }
    if (tl && k >= 0 && k < 64 && I == ns - 1 && (J == ns - 1 || J == 0)) D.tlog[(J == 0 ? 128 : 64) + k] = wall_clock64();
    if (!more) break;
  }
#pragma unroll
  for (int x = 0; x < B; x++)
#pragma unroll
    for (int y = 0; y < B; y++) {
      if (!(va[x] && vb[y])) continue;
      double* At = D.A + f.off + (size_t)((B * I + x) * DS_T) * f.ld + (B * J + y) * DS_T;
#pragma unroll
      for (int r = 0; r < 4; r++) At[(size_t)(qr + 4 * r) * f.ld + qc] = own[x][y][r];
    }
  if (D.tlog != nullptr && threadIdx.x == 0) atomicMax(&D.tlog[193], wall_clock64());   // last workgroup of the launch to end
#undef DS_FLOW_PSLOT
#undef DS_FLOW_RSLOT
#undef DS_FLOW_CSLOT
#undef DS_FLOW_PFLAG
#undef DS_FLOW_RFLAG
#undef DS_FLOW_CFLAG
}

// W = F11^-1 of a front with at most DS_SMALL pivots by ONE workgroup with the block in LDS (row stride ls = batch maximum + 1):
// the same blocked Gauss-Jordan, all block steps inside the launch -- the workgroup inverts the pivot tile where it lies, every wave
// owns row chunks of the rank-T update on the matrix cores (its column-panel fragment lives in registers while the chunk is rewritten).
__global__ void __launch_bounds__(256) k_ds_inv_small(DsDev D, int lv0, int ls) {
  extern __shared__ double ds_sm[];
  double* M = ds_sm;   // the block, row stride ls; the pivot tile of a step is inverted IN PLACE (no copy: at 96 pivots the block alone is 74.5 KB and
                       // two workgroups share a CU only without a separate tile buffer)
  const DsFrontDesc f = D.frl[lv0 + blockIdx.x];
  const int pp = f.pp, nt = pp / DS_T;
  double* A = D.A + f.off;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
  // the block comes in with 16 loads of a lane in flight (four rows x four column groups per pass; pp is a multiple of 32, at most 128): one
  // load per loop iteration and lane made the copy 64 dependent round trips, as long as the inversion itself
  for (int i = ty; i < pp; i += 32) {
    double v[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++) v[a][b] = (b < nt) ? A[(size_t)(i + 8 * a) * f.ld + tx + 32 * b] : 0.0;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++) if (b < nt) M[(i + 8 * a) * ls + tx + 32 * b] = v[a][b];
  }
  __syncthreads();
  for (int k = 0; k < nt; k++) {
    const int k0 = k * DS_T;
    double* Tt = M + k0 * ls + k0;   // P after the call
    ds_invert_tile(Tt, ls, D.bad, 1, (D.level_sn[lv0 + blockIdx.x] << 6) | k, D.piv_tol);
    // R'_j = P A_Kj in place (a wave owns whole tiles: all its reads of a tile precede its writes)
    for (int j = w; j < nt; j += 4) {
      if (j == k) continue;
      const int j0 = j * DS_T;
      ds_d4 acc[2][2];
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = ds_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < DS_T / 4; kk++) {
        const double a0 = Tt[lr * ls + 4 * kk + lk], a1 = Tt[(16 + lr) * ls + 4 * kk + lk];
        const double b0 = M[(k0 + 4 * kk + lk) * ls + j0 + lr], b1 = M[(k0 + 4 * kk + lk) * ls + j0 + 16 + lr];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
          for (int r = 0; r < 4; r++) M[(k0 + 16 * a + lk + 4 * r) * ls + j0 + 16 * b + lr] = acc[a][b][r];
    }
    __syncthreads();
    // A_ij -= A_iK R'_j, A_iK = -A_iK P for the row chunks i != k.  The (nt - 1) nt tiles go round the four waves one by one (a wave per row chunk left
    // one or two waves idle at three or four chunks and gave the busy ones nt tiles each): every wave first takes the column-panel fragments of ITS
    // tiles into registers, then -- behind a barrier, tile (i, k) is overwritten by whoever owns it -- forms and stores them.
    const int nit = (nt - 1) * nt;
    double c0[3][DS_T / 4], c1[3][DS_T / 4];   // (nt <= 4: at most three tiles per wave)
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int t = w + 4 * q;
      if (t < nit) {
        const int ii = t / nt, i0 = (ii < k ? ii : ii + 1) * DS_T;
#pragma unroll
        for (int kk = 0; kk < DS_T / 4; kk++) { c0[q][kk] = M[(i0 + lr) * ls + k0 + 4 * kk + lk]; c1[q][kk] = M[(i0 + 16 + lr) * ls + k0 + 4 * kk + lk]; }
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int t = w + 4 * q;
      if (t < nit) {
        const int ii = t / nt, j = t - ii * nt, i0 = (ii < k ? ii : ii + 1) * DS_T, j0 = j * DS_T;
        // the product is formed from zero and subtracted afterwards -- A_ij - (A_iK R'_j), -(A_iK P) -- like k_ds_gj_step / k_ds_gj_flow form it:
        // the three inversion paths give the same bits (accumulating onto the old entry rounds differently)
        ds_d4 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
          for (int b = 0; b < 2; b++) acc[a][b] = ds_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < DS_T / 4; kk++) {   // row K of M holds R'_j for j != k and P itself at j == k
          const double b0 = M[(k0 + 4 * kk + lk) * ls + j0 + lr], b1 = M[(k0 + 4 * kk + lk) * ls + j0 + 16 + lr];
          acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(c0[q][kk], b0, acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(c0[q][kk], b1, acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(c1[q][kk], b0, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(c1[q][kk], b1, acc[1][1], 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
          for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
              double* m = &M[(i0 + 16 * a + lk + 4 * r) * ls + j0 + 16 * b + lr];
              *m = (j == k) ? -acc[a][b][r] : *m - acc[a][b][r];
            }
      }
    }
    __syncthreads();
  }
  for (int i = ty; i < pp; i += 32) {
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++) if (b < nt) A[(size_t)(i + 8 * a) * f.ld + tx + 32 * b] = M[(i + 8 * a) * ls + tx + 32 * b];
  }
}

// ---- the two GEMMs of a front on the f64 matrix cores -------------------------------------------------------------------------
//   mode 0:  G = W F12                               (pp x bp, K = pp)  into the G arena (row stride bp)
//   mode 1:  S = sum_children ext(S_child) - F21 G   (bp x bp, K = pp)  into the Schur arena (row stride bp)
// one 64 x 64 output tile per workgroup (four waves, each a 32 x 32 quadrant = 2 x 2 MFMA tiles), K in slabs of 32 through LDS.
// LDS strides 33 / 65 doubles: the compiler fetches the operands with ds_read2_b64 (two doubles of one lane per instruction), which
// is served in groups of 16 consecutive lanes against 32 four-byte banks: the 16 rows lr of the A operand must fall on 16 different
// bank pairs, i.e. row stride == 1 (mod 16) doubles; the B operand's 16 lanes are contiguous.  (Strides 34 / 80, conflict-free
// under the ds_read_b64 rule -- 32 lanes against 64 banks --, measured 25 % SQ_LDS_BANK_CONFLICT of SQ_LDS_IDX_ACTIVE.)
// WPC = workgroups per CU the register allocation aims at (one wave of each per SIMD): 4 (two LDS slab buffers with one barrier per slab at
// two workgroups per CU were measured slower and are gone: what four workgroups hide is each other's prologues and epilogues).
// Measured and dropped (round 2 / 3, cfg4 plan, per-batch replays; profiles/README.md): K slabs of 64, an XCD-aware workgroup -> tile
// map (round 4: kept for the batches of many fronts, k_ds_gemm_x below), a capped persistent grid walking the tiles, skipping the products of
// quadrants outside the front, 128 x 128 tiles; round 4: wave priorities (s_setprio) that differ between the workgroups sharing a CU, to
// stagger their K loops and epilogues -- no change (976-1004 us for the Schur launches of a factorisation under four priority patterns).
#define DS_SK 32
#define DS_GMC 8   // children of a front whose tables the Schur epilogue keeps in LDS per pass
// part (the look-ahead of the upper levels): 0 every tile; 1 the tiles of the leading block (Schur mode: rows and columns below DsFrontDesc.lead -- all of S that
// the parent's pivot block receives; G: the tile columns below lead, which those tiles read); 2 the others.  A tile is the same arithmetic whichever launch forms it.
template <int mode, int WPC>
TSL_DEV void ds_gemm_tile(const DsDev& D, int lv0, int bx, int by, int bz, int part = 0) {
  constexpr int SA = DS_SK + 1, SB = 64 + 1;
  __shared__ double As[64 * SA];
  __shared__ double Bs[DS_SK * SB];
  const DsFrontDesc f = D.frl[lv0 + bz];
  const int Mr = mode == 0 ? f.pp : f.bp, Nc = f.bp, K = f.pp;
  const int I0 = by * 64, J0 = bx * 64;
  if (I0 >= Mr || J0 >= Nc) return;
  if (mode == 1 && part != 0 && ((I0 < f.lead && J0 < f.lead) != (part == 1))) return;
  if (mode == 0 && part != 0 && ((J0 < f.lead) != (part == 1))) return;   // G: the columns the leading Schur tiles read (J0 is a multiple of 64) / the others
  const double* F = D.A + f.off;
  double* G = D.G + f.goff;
  const int ld = f.ld, pp = f.pp;
  const double* Am = mode == 0 ? F : D.A + f.off21;   // W rows (row stride ld) / F21 rows (row stride pp)
  const int lda = mode == 0 ? ld : pp;
  const double* Bm = mode == 0 ? F + pp : G;           // F12 (row stride ld) / G (row stride bp)
  const int ldb = mode == 0 ? ld : f.bp;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wi = w >> 1, wj = w & 1;
  const int lr = lane & 15, lk = lane >> 4;
  const bool rows_hi = I0 + 32 < Mr, cols_hi = J0 + 32 < Nc;   // Mr, Nc are multiples of 32: the second half of the tile may lie outside
  ds_d4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) acc[a][b] = ds_d4{0.0, 0.0, 0.0, 0.0};
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  // K loop, software-pipelined: the global loads of slab k + 1 are in flight (registers) while the matrix cores work on slab k in LDS
  double pa[8], pb0[4], pb1[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int r = ty + 8 * q;
      pa[q] = (r < 32 || rows_hi) ? Am[(size_t)(I0 + r) * lda + k0 + tx] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int r = ty + 8 * q;
      pb0[q] = Bm[(size_t)(k0 + r) * ldb + J0 + tx];
      pb1[q] = cols_hi ? Bm[(size_t)(k0 + r) * ldb + J0 + 32 + tx] : 0.0;
    }
  };
  gload(0);
  // Schur mode: the rows / columns of this tile in the tables of the first DS_GMC children are requested now and parked in registers over
  // the K loop (thread t: child t >> 7 and t >> 7 + 2 ..., table entry t & 127: 64 rows, then 64 columns); they go to LDS for the epilogue
  constexpr int GMC = DS_GMC;
  __shared__ int s_map[mode == 1 ? GMC : 1][128];
  __shared__ long long s_soff[mode == 1 ? GMC : 1];
  __shared__ int s_cbp[mode == 1 ? GMC : 1];
  int pre[GMC / 2];
  auto map_load = [&](int q) {   // this thread's entry of child q's table (q < nchild)
    const int k = threadIdx.x & 127;
    const int idx = k < 64 ? I0 + k : J0 + k - 64;
    return idx < f.bp ? D.pmap[D.ch[f.ch_off + q].pmap_off + pp + idx] : -1;
  };
  if (mode == 1) {
#pragma unroll
    for (int h = 0; h < GMC / 2; h++) { const int q = 2 * h + (threadIdx.x >> 7); pre[h] = q < f.nchild ? map_load(q) : -1; }
  }
  for (int k0 = 0; k0 < K; k0 += DS_SK) {
#pragma unroll
    for (int q = 0; q < 8; q++) As[(ty + 8 * q) * SA + tx] = pa[q];
#pragma unroll
    for (int q = 0; q < 4; q++) { Bs[(ty + 8 * q) * SB + tx] = pb0[q]; Bs[(ty + 8 * q) * SB + tx + 32] = pb1[q]; }
    __syncthreads();
    if (k0 + DS_SK < K) gload(k0 + DS_SK);
#pragma unroll
    for (int kk = 0; kk < DS_SK / 4; kk++) {
      const double a0 = As[(32 * wi + lr) * SA + 4 * kk + lk], a1 = As[(32 * wi + 16 + lr) * SA + 4 * kk + lk];
      const double b0 = Bs[(4 * kk + lk) * SB + 32 * wj + lr], b1 = Bs[(4 * kk + lk) * SB + 32 * wj + 16 + lr];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
  if (mode == 0) {
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 2; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = I0 + 32 * wi + 16 * a + lk + 4 * r, col = J0 + 32 * wj + 16 * b + lr;
          if (row < Mr && col < Nc) G[(size_t)row * f.bp + col] = acc[a][b][r];
        }
    return;
  }
  // Schur complement: S = F22 - F21 G with F22 = the children's Schur complements extended to this front, gathered here (F22 is never
  // materialised) child after child in the fixed order of the plan; S is stored once, coalesced, for the parent to gather in turn.
  if (f.parent < 0) return;
  // (the sums start from -F21 G and take the children in the plan's order: one set of 16 registers instead of two -- the accumulators next to the
  // children's sums and 16 gathered entries in flight spilled 12 registers at four workgroups per CU)
  double s22[2][2][4];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) s22[a][b][r] = -acc[a][b][r];
  const int row0 = I0 + 32 * wi + lk, col0 = J0 + 32 * wj + lr;   // + 16 a + 4 r / + 16 b
  for (int q0 = 0; q0 < f.nchild; q0 += GMC) {
    const int nq = min(GMC, f.nchild - q0);
    if (q0 > 0) __syncthreads();   // (more than DS_GMC children: further passes load their tables here)
#pragma unroll
    for (int h = 0; h < GMC / 2; h++) {
      const int q = 2 * h + (threadIdx.x >> 7);
      if (q < nq) s_map[q][threadIdx.x & 127] = q0 == 0 ? pre[h] : map_load(q0 + q);
    }
    if ((int)threadIdx.x < nq) { const DsChildRec c = D.ch[f.ch_off + q0 + threadIdx.x]; s_soff[threadIdx.x] = c.soff; s_cbp[threadIdx.x] = c.bp; }
    __syncthreads();
    for (int q = 0; q < nq; q++) {   // ascending child order: the fixed summation order
      // The 16 entries of a lane are REQUESTED TOGETHER (entries the child does not reach read its entry (0, 0) and are dropped by a select -- no
      // arithmetic on them, the sums keep their bits): behind `if (ci >= 0)` every load was a branch of its own and the additions waited for
      // them one by one -- sixteen dependent round trips per child and tile in the epilogue of every workgroup.
      const double* Sc = D.S + s_soff[q];
      const int cbp = s_cbp[q];
      const int cj0 = s_map[q][64 + 32 * wj + lr], cj1 = s_map[q][64 + 32 * wj + 16 + lr];
      const int dj0 = max(cj0, 0), dj1 = max(cj1, 0);
      int ci[2][4];
      double v0[2][4], v1[2][4];
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int r = 0; r < 4; r++) ci[a][r] = s_map[q][32 * wi + 16 * a + lk + 4 * r];
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const double* Srow = Sc + (size_t)max(ci[a][r], 0) * cbp;
          v0[a][r] = Srow[dj0]; v1[a][r] = Srow[dj1];
        }
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          s22[a][0][r] = (ci[a][r] >= 0 && cj0 >= 0) ? s22[a][0][r] + v0[a][r] : s22[a][0][r];
          s22[a][1][r] = (ci[a][r] >= 0 && cj1 >= 0) ? s22[a][1][r] + v1[a][r] : s22[a][1][r];
        }
    }
  }
  double* Sf = D.S + f.soff;
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = row0 + 16 * a + 4 * r;
      if (row >= f.b) continue;
#pragma unroll
      for (int b = 0; b < 2; b++) {
        const int col = col0 + 16 * b;
        if (col < f.b) Sf[(size_t)row * f.bp + col] = s22[a][b][r];
      }
    }
}

// The panels of the fronts of a level (F11, F12, F21) are WRITTEN here when the level starts: entry (i, j) = sum over the children, in
// the plan's fixed order, of S_child[pmap_child[i]][pmap_child[j]] -- the part of the extend-add that lands on the panels --, zero where
// no child reaches; the level's matrix entries are added right after (k_ds_assemble_level).  Every entry has one writer and is written
// once: no atomics, no cleared memory, a fixed summation order.
// Work item of a wave: the THREE rows of one local vertex of the front (own vertex: top rows 3 v .., columns 0 .. ld; boundary vertex:
// rows of F21, columns 0 .. pp) x a span of DS_XSPAN columns.  Lane q looks the vertex up in child q's table (the three dofs of a vertex
// are consecutive in every table), a ballot gives the children that reach it (usually one or two), then per child and pass
// DS_XU table loads and 3 DS_XU loads of S per lane are in flight.  The last two items of a front zero its padding rows.
#define DS_XU 4
#define DS_XSPAN 1024  // columns per work item (DS_XSPAN / (64 DS_XU) passes: the vertex look-up is paid once per span)
#define DS_XMAXC 64    // children whose look-up fits one ballot (fronts with more take the general loop)
// part (the look-ahead of the upper levels, direct_factor): 0 every panel; 1 the pivot block F11 only (top rows, columns 0 .. pp); 2 the rest (F12 = top rows,
// columns pp .. ld, and F21) -- every entry is the same sum in the same order whichever launch writes it.
__global__ void __launch_bounds__(256) k_ds_extend_panels(DsDev D, int lv0, int nsp, int part, int xspan) {   // nsp: column spans (of xspan columns, a multiple of 64 DS_XU) of the widest front
  const DsFrontDesc f = D.frl[lv0 + blockIdx.y];
  const int ib = blockIdx.x / nsp, sp = blockIdx.x - ib * nsp;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = 4 * ib + w, ng = f.nv_own + f.nv_bnd;
  if (g >= ng + 2) return;
  const bool top = g >= ng ? g == ng : g < f.nv_own;
  if (!top && part == 1) return;
  const int ncol = top ? f.ld : f.pp;                                               // row stride of the panel
  const int cA = (top && part == 2) ? f.pp : 0, cB = (top && part == 1) ? f.pp : ncol;   // columns of this launch
  const int c0 = cA + sp * xspan;
  if (c0 >= cB) return;
  const int c1 = min(cB, c0 + xspan);
  if (g >= ng) {   // padding rows: p .. pp of the top rows (the identity comes with k_ds_assemble_level) / b .. bp of F21
    const int nrow = top ? f.pp - f.p : f.bp - f.b;
    double* base = top ? D.A + f.off + (size_t)f.p * f.ld : D.A + f.off21 + (size_t)f.b * f.pp;
    for (int i = 0; i < nrow; i++)
      for (int j = c0 + lane; j < c1; j += 64) base[(size_t)i * ncol + j] = 0.0;
    return;
  }
  const int r = top ? 3 * g : f.pp + 3 * (g - f.nv_own);            // first of the three local dofs of the vertex
  double* row = top ? D.A + f.off + (size_t)r * f.ld : D.A + f.off21 + (size_t)(r - f.pp) * f.pp;
  const DsChildRec* ch = D.ch + f.ch_off;
  if (f.nchild <= DS_XMAXC) {   // (every front of the plans seen so far)
    int my_ci = -1, my_off = 0, my_bp = 0;
    long long my_soff = 0;
    if (lane < f.nchild) { const DsChildRec c = ch[lane]; my_off = c.pmap_off; my_bp = c.bp; my_soff = c.soff; my_ci = D.pmap[c.pmap_off + r]; }
    const unsigned long long mask = __ballot(my_ci >= 0);
    for (int j0 = c0; j0 < c1; j0 += 64 * DS_XU) {
      double v[3][DS_XU];
#pragma unroll
      for (int k = 0; k < 3; k++)
#pragma unroll
        for (int u = 0; u < DS_XU; u++) v[k][u] = 0.0;
      // ascending child order: the fixed summation order.  The children that reach the vertex are taken TWO at a time (a front of the upper levels has
      // two): the table look-ups of both, then the entries of both, then the sums in order -- two dependent round trips per pass instead of four
      auto child = [&](int q, int& ci, const int*& pm, const double*& S0, int& cbp) {
        ci = __builtin_amdgcn_readlane(my_ci, q);
        const int off = __builtin_amdgcn_readlane(my_off, q);
        cbp = __builtin_amdgcn_readlane(my_bp, q);
        const long long so = ((long long)__builtin_amdgcn_readlane((int)(my_soff >> 32), q) << 32) | (unsigned)__builtin_amdgcn_readlane((int)my_soff, q);
        pm = D.pmap + off;
        S0 = D.S + so + (size_t)ci * cbp;
      };
      unsigned long long m = mask;
      while (m != 0) {
        const int q0 = __builtin_ctzll(m);
        m &= m - 1;
        const bool two = m != 0;
        const int q1 = two ? __builtin_ctzll(m) : q0;
        if (two) m &= m - 1;
        int ci0, ci1, cbp0, cbp1;
        const int *pm0, *pm1;
        const double *Sa, *Sb;
        child(q0, ci0, pm0, Sa, cbp0);
        child(q1, ci1, pm1, Sb, cbp1);
        int cja[DS_XU], cjb[DS_XU];
#pragma unroll
        for (int u = 0; u < DS_XU; u++) { const int j = j0 + 64 * u + lane; cja[u] = j < c1 ? pm0[j] : -1; cjb[u] = (two && j < c1) ? pm1[j] : -1; }
        double sa[3][DS_XU], sb[3][DS_XU];
#pragma unroll
        for (int u = 0; u < DS_XU; u++) {
#pragma unroll
          for (int k = 0; k < 3; k++) { sa[k][u] = cja[u] >= 0 ? Sa[(size_t)k * cbp0 + cja[u]] : 0.0; sb[k][u] = cjb[u] >= 0 ? Sb[(size_t)k * cbp1 + cjb[u]] : 0.0; }
        }
#pragma unroll
        for (int u = 0; u < DS_XU; u++) {
          if (cja[u] >= 0) {
#pragma unroll
            for (int k = 0; k < 3; k++) v[k][u] += sa[k][u];
          }
          if (cjb[u] >= 0) {
#pragma unroll
            for (int k = 0; k < 3; k++) v[k][u] += sb[k][u];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < DS_XU; u++) {
        const int j = j0 + 64 * u + lane;
        if (j < c1) {
#pragma unroll
          for (int k = 0; k < 3; k++) row[(size_t)k * ncol + j] = v[k][u];
        }
      }
    }
    return;
  }
  for (int j = c0 + lane; j < c1; j += 64) {   // general form: any number of children, one column per lane and pass
    double v[3] = {0.0, 0.0, 0.0};
    for (int q = 0; q < f.nchild; q++) {
      const int* pm = D.pmap + ch[q].pmap_off;
      const int ci = pm[r], cj = pm[j];
      if (ci >= 0 && cj >= 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) v[k] += D.S[ch[q].soff + (size_t)(ci + k) * ch[q].bp + cj];
      }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) row[(size_t)k * ncol + j] = v[k];
  }
}

// G = W F12 with 32 x 32 output tiles (one 16 x 16 matrix-core tile per wave), for the batches of the upper levels: there the 64 x 64
// tiles of k_ds_gemm are 320 - 670 workgroups -- one to three per CU --, and with so few the global loads of the next slab (prefetched ONE
// slab = 0.5 us ahead, against ~2 us of latency) are what a slab waits for: the K loop runs at 26-42 instead of 63 TFLOP/s
// (scripts/exp_gemm_dbg.py).  Four times the workgroups, a quarter of the registers: the latency hides behind occupancy again.
__global__ void __launch_bounds__(256) k_ds_gemm_g32(DsDev D, int lv0, int part) {   // part 1 / 2: the columns below / from DsFrontDesc.lead rounded up to 64 (what the leading Schur tiles read)
  constexpr int SA = DS_SK + 1, SB = 32 + 1;
  __shared__ double As[32 * SA];
  __shared__ double Bs[DS_SK * SB];
  const DsFrontDesc f = D.frl[lv0 + blockIdx.z];
  const int Mr = f.pp, Nc = f.bp, K = f.pp;
  const int I0 = blockIdx.y * 32, J0 = blockIdx.x * 32;
  if (I0 >= Mr || J0 >= Nc) return;
  if (part != 0 && ((J0 < ((f.lead + 63) & ~63)) != (part == 1))) return;
  const double* F = D.A + f.off;
  double* G = D.G + f.goff;
  const int ld = f.ld;
  const double* Am = F;             // W rows, row stride ld
  const double* Bm = F + f.pp;      // F12, row stride ld
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wi = w >> 1, wj = w & 1, lr = lane & 15, lk = lane >> 4;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  ds_d4 acc = {0.0, 0.0, 0.0, 0.0};
  double pa[4], pb[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int r = ty + 8 * q;
      pa[q] = Am[(size_t)(I0 + r) * ld + k0 + tx];
      pb[q] = Bm[(size_t)(k0 + r) * ld + J0 + tx];
    }
  };
  gload(0);
  for (int k0 = 0; k0 < K; k0 += DS_SK) {
#pragma unroll
    for (int q = 0; q < 4; q++) { As[(ty + 8 * q) * SA + tx] = pa[q]; Bs[(ty + 8 * q) * SB + tx] = pb[q]; }
    __syncthreads();
    if (k0 + DS_SK < K) gload(k0 + DS_SK);
#pragma unroll
    for (int kk = 0; kk < DS_SK / 4; kk++)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(As[(16 * wi + lr) * SA + 4 * kk + lk], Bs[(4 * kk + lk) * SB + 16 * wj + lr], acc, 0, 0, 0);
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; r++) G[(size_t)(I0 + 16 * wi + lk + 4 * r) * f.bp + J0 + 16 * wj + lr] = acc[r];
}

template <int mode, int WPC>
__global__ void __launch_bounds__(256, WPC) k_ds_gemm(DsDev D, int lv0, int part) {
  ds_gemm_tile<mode, WPC>(D, lv0, blockIdx.x, blockIdx.y, blockIdx.z, part);
}

// The same tiles launched as a one-dimensional grid with an XCD-aware map ("direct_xcd", batches of at least that many fronts; default 64):
// the hardware hands consecutive workgroups to the eight XCDs in turn, each with an L2 of its own, so with blockIdx = (tile column, tile
// row, front) the tiles of one front -- which share its F21 / G (or W / F12) panels -- are spread over all eight L2s and every panel is
// fetched from memory up to eight times.  Here workgroup L is taken as the (L / 8)-th of XCD L % 8 and a whole front belongs to one XCD;
// fronts are dealt to the XCDs in turn (they are sorted by size: every XCD gets every eighth).  Affinity only -- nothing depends on where
// a workgroup runs.  Measured on cfg4 (scripts/pmc_gemm.sh, exp_batches.py): FETCH_SIZE of the batches it applies to falls 3x (Schur) and
// 2.3x (G) at equal time (+-3 %).  Batches of 2 .. 32 unequal fronts lose time under any such map -- a front per XCD is unbalanced, every
// front split 2 x 4 over the XCDs (an equal share of every front each) is 10-35 % slower than dealing single tiles -- and keep blockIdx order.
template <int mode, int WPC>
__global__ void __launch_bounds__(256, WPC) k_ds_gemm_x(DsDev D, int lv0, int gx, int gy, int nf) {
  const unsigned L = blockIdx.x, xcd = L & 7, k = L >> 3;
  const unsigned tpf = (unsigned)gx * gy;
  const unsigned bz = (k / tpf) * 8 + xcd, t = k % tpf;
  if (bz >= (unsigned)nf) return;
  ds_gemm_tile<mode, WPC>(D, lv0, t % gx, t / gx, bz);
}

// ---- solve ----------------------------------------------------------------------------------------------------------------
// One matrix-vector pass over the fronts of a level; a workgroup owns 16 rows of one front (4 per wave), the input vector of the
// front is staged in LDS in chunks.  The upward sweep carries the boundary updates from child to parent like the factorisation carries the
// Schur complements: every front STORES y_f (b entries: what its subtree subtracts from the right-hand side on its boundary dofs) and its
// parent gathers it through the same child tables (pmap) in the plan's fixed child order -- no atomics, the right-hand side is not
// modified (no copy), a fixed summation order:
//   mode 0 (up, W):    t[own i]  = sum_j W[i, j] (r[own j] - sum_children y_c[pmap_c[j]])
//   mode 1 (up, F21):  y_f[i]    = sum_children y_c[pmap_c[pp + i]] + sum_j F21[i, j] t[own j]
//   mode 2 (down, G):  x[own i]  = t[own i] - sum_j G[i, j] x[bnd j]  (x and t may alias)
#define DS_VCHUNK 2048
// what the children's subtrees subtract on local dof d (own: d < p, boundary: pp + i) of front f.  The table and y offsets of the first
// four children sit in the front's descriptor: four independent table loads, then four independent loads of y -- two round trips, the
// depth of the vertex-id -> vector gather they run next to
TSL_DEV double ds_child_sum(const DsDev& D, const DsFrontDesc& f, int d) {
  int ci[4];
#pragma unroll
  for (int q = 0; q < 4; q++) ci[q] = q < f.nchild ? D.pmap[f.cpm[q] + d] : -1;
  double y[4];
#pragma unroll
  for (int q = 0; q < 4; q++) y[q] = ci[q] >= 0 ? D.Y[f.cy[q] + ci[q]] : 0.0;
  double a = ((y[0] + y[1]) + y[2]) + y[3];   // (ascending child order)
  for (int q = 4; q < f.nchild; q++) {
    const DsChildRec c = D.ch[f.ch_off + q];
    const int cq = D.pmap[c.pmap_off + d];
    if (cq >= 0) a += D.Y[c.yoff + cq];
  }
  return a;
}
TSL_DEV void ds_gemv_stage(const DsDev& D, const DsFrontDesc& f, int mode, const int* vt, int in_v0, const double* vin, int c0, int cn, double* xs) {
  __syncthreads();
  for (int j = threadIdx.x; j < cn; j += 256) {
    const int jj = c0 + j;
    const double sub = (mode == 0 && f.nchild > 0) ? ds_child_sum(D, f, jj) : 0.0;
    xs[j] = vin[3 * (size_t)vt[in_v0 + jj / 3] + jj % 3] - sub;
  }
  __syncthreads();
}
__global__ void __launch_bounds__(256) k_ds_gemv(DsDev D, const int* __restrict__ wl_front, const int* __restrict__ wl_row, int wl0, int mode, const double* vin, double* vout) {
  __shared__ double xs[DS_VCHUNK];
  const DsFrontDesc f = D.fr[wl_front[wl0 + blockIdx.x]];
  const int r0 = wl_row[wl0 + blockIdx.x];
  const int nrows = mode == 1 ? f.b : f.p, ncols = mode == 2 ? f.b : f.p;
  const int* vt = D.vtx + f.vtx_off;
  const int in_v0 = mode == 2 ? f.nv_own : 0;
  const double* M = mode == 2 ? D.G + f.goff : D.A + (mode == 1 ? f.off21 : f.off);
  const int rs = mode == 2 ? f.bp : (mode == 1 ? f.pp : f.ld);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  // the four rows of a wave advance together and the column loop is unrolled: 16 independent loads in flight per wave (one row at
  // a time with a dependent accumulation ran at the memory latency: 30 us for a 832 x 832 block)
  const int ib = r0 + 4 * w;
  const double* row0 = M + (size_t)min(ib, nrows - 1) * rs;
  const double* row1 = M + (size_t)min(ib + 1, nrows - 1) * rs;
  const double* row2 = M + (size_t)min(ib + 2, nrows - 1) * rs;
  const double* row3 = M + (size_t)min(ib + 3, nrows - 1) * rs;
  for (int c0 = 0; c0 < ncols; c0 += DS_VCHUNK) {
    const int cn = min(DS_VCHUNK, ncols - c0);
    ds_gemv_stage(D, f, mode, vt, in_v0, vin, c0, cn, xs);
#pragma unroll 4
    for (int j = lane; j < cn; j += 64) {
      const double xv = xs[j];
      acc[0] += row0[c0 + j] * xv; acc[1] += row1[c0 + j] * xv; acc[2] += row2[c0 + j] * xv; acc[3] += row3[c0 + j] * xv;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int i = r0 + 4 * w + q;
    const double a = wave_sum(acc[q]);
    if (lane == 0 && i < nrows) {
      if (mode == 1) D.Y[f.yoff + i] = (f.nchild > 0 ? ds_child_sum(D, f, f.pp + i) : 0.0) + a;
      else {
        const size_t o = 3 * (size_t)vt[i / 3] + i % 3;
        vout[o] = mode == 0 ? a : vout[o] - a;
      }
    }
  }
}

// The same chunk for the launches of the upper levels, which have 66 - 600 workgroups of 16 rows: FOUR workgroups per chunk, each
// with 4 rows, its four waves a quarter of the columns each (partial sums joined through LDS) -- a quarter of the serial column loop
// per wave, four times the workgroups on a chip the launch did not fill ("direct_gemv_wide_below").
__global__ void __launch_bounds__(256) k_ds_gemv_wide(DsDev D, const int* __restrict__ wl_front, const int* __restrict__ wl_row, int wl0, int mode, const double* vin, double* vout) {
  __shared__ double xs[DS_VCHUNK];
  __shared__ double part[4][4];
  const int e = wl0 + (blockIdx.x >> 2);
  const DsFrontDesc f = D.fr[wl_front[e]];
  const int nrows = mode == 1 ? f.b : f.p, ncols = mode == 2 ? f.b : f.p;
  const int r0 = wl_row[e] + 4 * (blockIdx.x & 3);
  if (r0 >= nrows) return;
  const int* vt = D.vtx + f.vtx_off;
  const int in_v0 = mode == 2 ? f.nv_own : 0;
  const double* M = mode == 2 ? D.G + f.goff : D.A + (mode == 1 ? f.off21 : f.off);
  const int rs = mode == 2 ? f.bp : (mode == 1 ? f.pp : f.ld);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  const double* row0 = M + (size_t)min(r0, nrows - 1) * rs;
  const double* row1 = M + (size_t)min(r0 + 1, nrows - 1) * rs;
  const double* row2 = M + (size_t)min(r0 + 2, nrows - 1) * rs;
  const double* row3 = M + (size_t)min(r0 + 3, nrows - 1) * rs;
  for (int c0 = 0; c0 < ncols; c0 += DS_VCHUNK) {
    const int cn = min(DS_VCHUNK, ncols - c0);
    ds_gemv_stage(D, f, mode, vt, in_v0, vin, c0, cn, xs);
#pragma unroll 4
    for (int j = threadIdx.x; j < cn; j += 256) {
      const double xv = xs[j];
      acc[0] += row0[c0 + j] * xv; acc[1] += row1[c0 + j] * xv; acc[2] += row2[c0 + j] * xv; acc[3] += row3[c0 + j] * xv;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; q++) { const double a = wave_sum(acc[q]); if (lane == 0) part[w][q] = a; }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int q = threadIdx.x, i = r0 + q;
    if (i < nrows) {
      const double a = (part[0][q] + part[1][q]) + (part[2][q] + part[3][q]);
      if (mode == 1) D.Y[f.yoff + i] = (f.nchild > 0 ? ds_child_sum(D, f, f.pp + i) : 0.0) + a;
      else {
        const size_t o = 3 * (size_t)vt[i / 3] + i % 3;
        vout[o] = mode == 0 ? a : vout[o] - a;
      }
    }
  }
}
