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

// Matrix-free contact blocks folded into the SpMV kernels.  The constraint list is fixed within a time step, so contact detection
// builds a row -> (constraint, slot) CSR once per step (k_contact.hpp).  The entries of the 64 rows of a slice are contiguous in
// that CSR: ALL threads of the slice's workgroup share them (a pad vertex sits in ~100 constraints -- one lane per row serialises
// them, measured 68-164 us) and accumulate into per-row LDS sums with ds atomics.  Three launches per PCG iteration (product, its
// p.Ap partial, the two products of the V-cycle) disappear.  ptr == nullptr: no contacts.
struct ContactRows {
  const int* ptr;      // NV + 1 (permuted rows)
  const int* ent;      // (constraint << 2) | slot, grouped by row
  const int4* rows;    // permuted rows of the entry's four constraint vertices
  const double* H;     // masked 12x12 blocks
  int nv;              // rows
};
// adds sum_b H_c[a][b] x[row_c[b]] of entries [e0, e1) (the rows [row0, row0 + 64) of the slice) into acc[k][row - row0] (k = 0..2),
// optionally the same product with a second vector into acc[3 + k]; called by all threads of the workgroup, acc zeroed and
// synchronised by the caller.  e0 / e1 are fetched at kernel entry so that slices without contacts pay one barrier only.
template <int NVEC>
TSL_DEV void contact_slice_add(const ContactRows& C, int e0, int e1, int row0, const double* __restrict__ x, const double* __restrict__ x2, double (*acc)[64]) {
  // One WAVE per row of the slice (the workgroup's waves take the 64 rows in turn): lane l takes the row's entries l, l + 64, ... in their
  // stored order (ascending constraint, slot), the lanes' partial sums are joined by the fixed shuffle tree of wave_sum -- a fixed
  // summation order, no atomics, and a row with hundreds of entries (a table vertex under a folded cloth) is not walked by one thread.
  // (Round 3 summed a row's contributions with LDS atomics in arrival order.)
  (void)e0; (void)e1;
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  for (int l = threadIdx.x >> 6; l < 64; l += nw) {
    const bool in = row0 + l < C.nv;   // (the last slice is partial)
    const int r0 = in ? C.ptr[row0 + l] : 0, r1 = in ? C.ptr[row0 + l + 1] : 0;
    if (r1 <= r0) { if (lane == 0) { acc[0][l] = 0.0; acc[1][l] = 0.0; acc[2][l] = 0.0; if (NVEC == 2) { acc[3][l] = 0.0; acc[4][l] = 0.0; acc[5][l] = 0.0; } } continue; }
    double y0 = 0, y1 = 0, y2 = 0, q0 = 0, q1 = 0, q2 = 0;
    for (int e = r0 + lane; e < r1; e += 64) {
      const int q = C.ent[e];
      const int4 r4 = C.rows[e];
      const int c = q >> 2, a = q & 3;
      const double* H = C.H + 144 * (size_t)c + 36 * a;
      const int pb[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const d3 xb = ld3(x, pb[b]);
        y0 += H[3 * b] * xb.x + H[3 * b + 1] * xb.y + H[3 * b + 2] * xb.z;
        y1 += H[12 + 3 * b] * xb.x + H[12 + 3 * b + 1] * xb.y + H[12 + 3 * b + 2] * xb.z;
        y2 += H[24 + 3 * b] * xb.x + H[24 + 3 * b + 1] * xb.y + H[24 + 3 * b + 2] * xb.z;
        if (NVEC == 2) {
          const d3 wb = ld3(x2, pb[b]);
          q0 += H[3 * b] * wb.x + H[3 * b + 1] * wb.y + H[3 * b + 2] * wb.z;
          q1 += H[12 + 3 * b] * wb.x + H[12 + 3 * b + 1] * wb.y + H[12 + 3 * b + 2] * wb.z;
          q2 += H[24 + 3 * b] * wb.x + H[24 + 3 * b + 1] * wb.y + H[24 + 3 * b + 2] * wb.z;
        }
      }
    }
    y0 = wave_sum(y0); y1 = wave_sum(y1); y2 = wave_sum(y2);
    if (NVEC == 2) { q0 = wave_sum(q0); q1 = wave_sum(q1); q2 = wave_sum(q2); }
    if (lane == 0) { acc[0][l] = y0; acc[1][l] = y1; acc[2][l] = y2; if (NVEC == 2) { acc[3][l] = q0; acc[4][l] = q1; acc[5][l] = q2; } }
  }
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

// Product of one wave's share of a SELL slice (block columns k = w, w + WPS, ...) in explicit phases: the column ids of SELL_U
// columns, then their 9 * SELL_U value loads, then the gathers, then the arithmetic.  Written as a plain loop the compiler waits for
// the gathers of column k before it issues the loads of column k + WPS (vmcnt counts in order), which costs two memory latencies per
// column on rows that only have 9-13 blocks (3-4 columns per wave): measured 16 us for the fused PCG kernel against ~10 us phased.
// Columns past the end of the slice repeat a valid one with weight 0 (cached reloads, no branches).  NVEC = 2 also accumulates
// the product with a second vector (q).
#define SELL_U 4
// one phased trip over U block columns k0, k0 + WPS, ... of the wave
template <int U, int NVEC, int WPS, bool NT, typename VT>
TSL_DEV void sell_wave_trip(const int* __restrict__ cp, const VT* __restrict__ vp, int len, int k0, const double* __restrict__ x, const double* __restrict__ x2,
                            double& y0, double& y1, double& y2, double& q0, double& q1, double& q2) {
  int kk[U], c[U];
  double m[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int k = k0 + u * WPS;
    kk[u] = k < len ? k : k0;
    m[u] = k < len ? 1.0 : 0.0;
    c[u] = NT ? __builtin_nontemporal_load(cp + 64 * kk[u]) : cp[64 * kk[u]];
  }
  double a[U][9];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const VT* ap = vp + (size_t)kk[u] * 576;
#pragma unroll
    for (int e = 0; e < 9; e++) a[u][e] = (double)(NT ? __builtin_nontemporal_load(ap + 64 * e) : ap[64 * e]);
  }
  d3 xj[U], wj[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    xj[u] = ld3(x, c[u]);
    if (NVEC == 2) wj[u] = ld3(x2, c[u]);
  }
#pragma unroll
  for (int u = 0; u < U; u++) {
    y0 += m[u] * (a[u][0] * xj[u].x + a[u][1] * xj[u].y + a[u][2] * xj[u].z);
    y1 += m[u] * (a[u][3] * xj[u].x + a[u][4] * xj[u].y + a[u][5] * xj[u].z);
    y2 += m[u] * (a[u][6] * xj[u].x + a[u][7] * xj[u].y + a[u][8] * xj[u].z);
    if (NVEC == 2) {
      q0 += m[u] * (a[u][0] * wj[u].x + a[u][1] * wj[u].y + a[u][2] * wj[u].z);
      q1 += m[u] * (a[u][3] * wj[u].x + a[u][4] * wj[u].y + a[u][5] * wj[u].z);
      q2 += m[u] * (a[u][6] * wj[u].x + a[u][7] * wj[u].y + a[u][8] * wj[u].z);
    }
  }
}
// The interior cloth rows have 17 blocks: with four waves per slice wave 0 owns five of them, and a second trip for the fifth would
// put a whole extra index -> value -> gather chain behind the first (the other three waves idle meanwhile).  A wave with exactly
// five columns takes them in one trip; longer rows (FEM bodies) loop over trips of four.
template <int NVEC, int WPS, bool NT, typename VT = double>
TSL_DEV void sell_wave_product(const int* __restrict__ cp, const VT* __restrict__ vp, int len, int w, const double* __restrict__ x, const double* __restrict__ x2,
                               double& y0, double& y1, double& y2, double& q0, double& q1, double& q2) {
  const int nk = (len - w + WPS - 1) / WPS;   // columns of this wave (wave-uniform)
  if (nk == SELL_U + 1) {
    sell_wave_trip<SELL_U + 1, NVEC, WPS, NT, VT>(cp, vp, len, w, x, x2, y0, y1, y2, q0, q1, q2);
    return;
  }
  if (nk == 1) {
    sell_wave_trip<1, NVEC, WPS, NT, VT>(cp, vp, len, w, x, x2, y0, y1, y2, q0, q1, q2);
    return;
  }
  if (nk == 2) {  // the nine-block rows of the cloth: no masked repeats of the first column (cached, but they occupy the load path)
    sell_wave_trip<2, NVEC, WPS, NT, VT>(cp, vp, len, w, x, x2, y0, y1, y2, q0, q1, q2);
    return;
  }
  if (nk == 3) {
    sell_wave_trip<3, NVEC, WPS, NT, VT>(cp, vp, len, w, x, x2, y0, y1, y2, q0, q1, q2);
    return;
  }
  for (int k0 = w; k0 < len; k0 += SELL_U * WPS) sell_wave_trip<SELL_U, NVEC, WPS, NT, VT>(cp, vp, len, k0, x, x2, y0, y1, y2, q0, q1, q2);
}

__global__ void k_vals_to_f32(size_t n, const double* __restrict__ src, float* __restrict__ dst) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (float)src[i];
}

// Streaming-read yardstick for bench.py: reads n 16-byte words once (grid-stride, 4 independent loads in flight per lane) and leaves
// one partial per block -- what a kernel with NO index chain, gather or reduction needs for the bytes k_pcg_spmv reads.
__global__ void __launch_bounds__(256) k_stream_read(const double2* __restrict__ p, size_t n, double* __restrict__ part) {
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const double2 v0 = p[i], v1 = p[i + stride], v2 = p[i + 2 * stride], v3 = p[i + 3 * stride];
    a0 += v0.x + v0.y; a1 += v1.x + v1.y; a2 += v2.x + v2.y; a3 += v3.x + v3.y;
  }
  for (; i < n; i += stride) { const double2 v = p[i]; a0 += v.x + v.y; }
  double a = wave_sum((a0 + a1) + (a2 + a3));
  __shared__ double sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// Variant with WPS waves per slice: wave w of a slice handles block columns k = w, w+WPS, ... (more loads in flight per
// row: 784 single-wave slices cannot fill 1024 SIMDs at 100k triangles), partial row sums are combined through LDS.
// NT: matrix values / column ids are streamed with non-temporal loads so that they do not evict the x vector from L2.
// part != nullptr: the block's dot(x, y) partial is stored to part[blockIdx.x] (deterministic two-stage reduction, no
// same-address atomics).  blockDim.x = 64 * WPS * SPB (SPB slices per block).
// VT = float: the single-precision copy of the matrix that the multigrid smoother uses (k_vals_to_f32); products that define the
// solution (PCG operator, true residuals, MINRES / GMRES) always read the double-precision matrix.
template <int WPS, int SPB, bool NT, typename VT = double>
__global__ void __launch_bounds__(64 * WPS * SPB)
k_spmv_mw(int NV, int n_slices, const int* __restrict__ slice_off, const int* __restrict__ slice_len, const int* __restrict__ colidx,
          const VT* __restrict__ vals, const double* __restrict__ x, double* __restrict__ y, double* __restrict__ part, const int* __restrict__ flag,
          ContactRows CR) {
  __shared__ double red[SPB][WPS][3][64];
  __shared__ double dred[SPB];
  __shared__ double cacc[3][64];   // contact part (SPB == 1 only)
  if (flag && *flag) return;
  int ce0 = 0, ce1 = 0;
  if (SPB == 1 && CR.ptr) {
    ce0 = CR.ptr[blockIdx.x * 64]; ce1 = CR.ptr[min((int)blockIdx.x * 64 + 64, NV)];
    if (threadIdx.x < 192) cacc[threadIdx.x >> 6][threadIdx.x & 63] = 0.0;
  }
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int sl = wv / WPS, w = wv % WPS;
  const int slice = blockIdx.x * SPB + sl;
  double y0 = 0, y1 = 0, y2 = 0;
  if (slice < n_slices) {
    const int off = slice_off[slice], len = slice_len[slice];
    const int* cp = colidx + off + lane;
    const VT* vp = vals + (size_t)off * 9 + lane;
    double u0 = 0, u1 = 0, u2 = 0;
    sell_wave_product<1, WPS, NT, VT>(cp, vp, len, w, x, nullptr, y0, y1, y2, u0, u1, u2);
  }
  if (WPS > 1) {
    red[sl][w][0][lane] = y0; red[sl][w][1][lane] = y1; red[sl][w][2][lane] = y2;
    __syncthreads();
  }
  if (SPB == 1 && ce1 > ce0) {   // uniform over the workgroup
    if (WPS == 1) __syncthreads();
    contact_slice_add<1>(CR, ce0, ce1, blockIdx.x * 64, x, nullptr, cacc);
    __syncthreads();
  }
  double acc = 0.0;
  if (w == 0 && slice < n_slices) {
    if (WPS > 1) {
#pragma unroll
      for (int q = 1; q < WPS; q++) { y0 += red[sl][q][0][lane]; y1 += red[sl][q][1][lane]; y2 += red[sl][q][2][lane]; }
    }
    if (SPB == 1 && ce1 > ce0) { y0 += cacc[0][lane]; y1 += cacc[1][lane]; y2 += cacc[2][lane]; }
    const int p = slice * 64 + lane;
    if (p < NV) {
      st3(y, p, d3(y0, y1, y2));
      if (part) {  // the own-row gather is a dependent load at the tail of the kernel: only when the dot product is wanted
        const d3 xi = ld3(x, p);
        acc = xi.x * y0 + xi.y * y1 + xi.z * y2;
      }
    }
  }
  if (part) {
    acc = wave_sum(acc);
    if (w == 0 && lane == 0) dred[sl] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0;
#pragma unroll
      for (int q = 0; q < SPB; q++) t += dred[q];
      part[blockIdx.x] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Two-kernel PCG iteration with deterministic two-stage reductions (no same-address atomics: 784 waves adding into one
// double cost 7 us per kernel on MI355X).  Iteration k:
//   K1 = k_pcg_spmv  : rz_k = sum(part_rz), rr_k = sum(part_rr)  [every block re-reduces the <= PCG_MAXPART partials];
//                      stop if rr_k <= thresh2;  beta = rz_k / rz_{k-1};  p_k = z_k + beta p_{k-1} formed on the fly for the
//                      row and for every gathered neighbour (ping-pong p buffers);  Ap_k = H p_k;  part_pAp[block] = p_k.Ap_k
//   K2 = k_pcg_update: pAp = sum(part_pAp); alpha = rz_k / pAp (breakdown if pAp <= 0);  x += alpha p_k;  r -= alpha Ap_k;
//                      z_{k+1} = Dinv r;  part_rz[block] = r.z, part_rr[block] = r.r
#define PCG_MAXPART 1024
struct PcgScal {
  double rzh[2];      // rz history, slot k & 1 written by K1(k)
  double thresh2, bb;
  double rr_last, rz_last, pAp_last;
  int flag, iters;    // flag: 0 running, 1 breakdown, 2 converged
  int n_part1, n_part2;
};

TSL_DEV double block_reduce_partials(const double* __restrict__ part, int n, double* sm) {
  double v = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) v += part[i];
  // fixed-order tree: wave shuffle then LDS
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sm[w] = v;
  __syncthreads();
  double t = 0;
  const int nw = blockDim.x >> 6;
  for (int q = 0; q < nw; q++) t += sm[q];
  __syncthreads();
  return t;
}

// beta = rz / rz_old is only known after the partial sums of the previous update kernel have been reduced.  Instead of reducing
// first (1.5-2 us during which no matrix load is in flight) the kernel starts the loads of the partials before the matrix loop,
// forms A z, and finishes with the recurrence  A p_new = A z + beta A p_old  on its own rows (Ap still holds A p_old from the
// previous iteration; the first iteration of every (re)start writes A z itself, so the recurrence never runs longer than one
// restart interval and the true-residual test of the restart loop bounds its drift).  Gathering p_old a second time to form
// A p_old from scratch cost 3.5 us per launch (13.7 MB of L2 traffic).
template <int WPS, bool NT>
__global__ void __launch_bounds__(64 * WPS)
k_pcg_spmv(int NV, int n_slices, const int* __restrict__ slice_off, const int* __restrict__ slice_len, const int* __restrict__ colidx,
           const double* __restrict__ vals, const double* __restrict__ z, const double* __restrict__ p_old, double* __restrict__ p_new,
           double* __restrict__ Ap, const double* __restrict__ part_rz, const double* __restrict__ part_rr, double* __restrict__ part_pAp,
           PcgScal* sc, int parity, int first, unsigned long long* prof, ContactRows CR) {
  __shared__ double red[WPS][3][64];
  __shared__ double s2[2][WPS];
  __shared__ double cacc[3][64];
  // the stop flag is only acted on after the matrix loop: a branch on it here would put one more memory round trip in front of every load
  const int stop = sc->flag;
  int ce0 = 0, ce1 = 0;
  if (CR.ptr) {
    ce0 = CR.ptr[blockIdx.x * 64]; ce1 = CR.ptr[min((int)blockIdx.x * 64 + 64, NV)];
    if (threadIdx.x < 192) cacc[threadIdx.x >> 6][threadIdx.x & 63] = 0.0;
  }
  unsigned long long t_start = 0;
  if (prof) t_start = wall_clock64();
  // partial sums of r.z and r.r: loads issued now, consumed after the matrix loop
  // (held in registers un-added: an in-order wave would wait for them at the first addition, i.e. before the matrix loads go out)
  constexpr int MAXU = PCG_MAXPART / (64 * WPS);
  double pz[MAXU], pr[MAXU];
  {
    const int n = sc->n_part2;   // <= PCG_MAXPART (solve_perm)
#pragma unroll
    for (int u = 0; u < MAXU; u++) {
      const int i = (int)threadIdx.x + 64 * WPS * u;
      pz[u] = i < n ? part_rz[i] : 0.0;
      pr[u] = i < n ? part_rr[i] : 0.0;
    }
    for (int i = (int)threadIdx.x + 64 * WPS * MAXU; i < n; i += 64 * WPS) { pz[0] += part_rz[i]; pr[0] += part_rr[i]; }  // more than PCG_MAXPART partials (> 200k vertices)
  }
  const double rz_old = first ? 1.0 : sc->rzh[parity ^ 1];
  const double thresh2 = sc->thresh2;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int slice = blockIdx.x;
  const int prow = slice * 64 + lane;
  // own-row operands of the final combine (wave 0): in flight during the matrix loop
  d3 zi(0, 0, 0), poi(0, 0, 0), apo(0, 0, 0);
  if (w == 0 && prow < NV) {
    zi = ld3(z, prow);
    if (!first) { poi = ld3(p_old, prow); apo = ld3(Ap, prow); }
  }
  const int off = slice_off[slice], len = slice_len[slice];
  const int* cp = colidx + off + lane;
  const double* vp = vals + (size_t)off * 9 + lane;
  double y0 = 0, y1 = 0, y2 = 0, u0 = 0, u1 = 0, u2 = 0;
  sell_wave_product<1, WPS, NT>(cp, vp, len, w, z, nullptr, y0, y1, y2, u0, u1, u2);
  if (stop) return;  // uniform over the grid; nothing has been written yet
  // finish the two reductions (one LDS round trip shared with the wave partials of the product)
  double v0 = 0, v1 = 0;
#pragma unroll
  for (int u = 0; u < MAXU; u++) { v0 += pz[u]; v1 += pr[u]; }
  v0 = wave_sum(v0); v1 = wave_sum(v1);
  if (lane == 0) { s2[0][w] = v0; s2[1][w] = v1; }
  if (WPS > 1 && w > 0) { red[w][0][lane] = y0; red[w][1][lane] = y1; red[w][2][lane] = y2; }
  __syncthreads();
  double rz = 0, rr = 0;
#pragma unroll
  for (int q = 0; q < WPS; q++) { rz += s2[0][q]; rr += s2[1][q]; }
  if (rr <= thresh2) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc->flag = 2; sc->rr_last = rr; }
    return;
  }
  if (ce1 > ce0) {   // uniform over the workgroup
    contact_slice_add<1>(CR, ce0, ce1, slice * 64, z, nullptr, cacc);
    __syncthreads();
  }
  const double beta = first ? 0.0 : rz / rz_old;
  if (blockIdx.x == 0 && threadIdx.x == 0) { sc->rzh[parity] = rz; sc->rr_last = rr; sc->rz_last = rz; sc->iters = sc->iters + 1; }
  double acc = 0.0;
  if (w == 0) {
#pragma unroll
    for (int q = 1; q < WPS; q++) { y0 += red[q][0][lane]; y1 += red[q][1][lane]; y2 += red[q][2][lane]; }
    if (ce1 > ce0) { y0 += cacc[0][lane]; y1 += cacc[1][lane]; y2 += cacc[2][lane]; }
    if (prow < NV) {
      const d3 pi = zi + beta * poi;
      const d3 ap = d3(y0, y1, y2) + beta * apo;
      st3(p_new, prow, pi);
      st3(Ap, prow, ap);
      acc = dot(pi, ap);
    }
    acc = wave_sum(acc);
    if (lane == 0) part_pAp[blockIdx.x] = acc;
  }
  if (prof && lane == 0) {
    const size_t wi = (size_t)blockIdx.x * WPS + w;
    prof[2 * wi] = t_start;
    prof[2 * wi + 1] = (unsigned long long)wall_clock64();
  }
}

// end-of-chunk convergence test (one block): without it the converged state is only seen by K1 of the NEXT chunk,
// i.e. after a whole chunk of idle launches
__global__ void __launch_bounds__(256) k_pcg_check(const double* __restrict__ part_rr, PcgScal* sc) {
  __shared__ double sm[8];
  if (sc->flag) return;
  const double rr = block_reduce_partials(part_rr, sc->n_part2, sm);
  if (threadIdx.x == 0 && rr <= sc->thresh2) { sc->flag = 2; sc->rr_last = rr; }
}

// K2; also used (first = 1) to start / restart: r = b - Ax (Ax may be null), z = Dinv r, partials, no x update
__global__ void __launch_bounds__(256)
k_pcg_update(int NV, const double* __restrict__ pv, const double* __restrict__ Ap, const double* __restrict__ Dinv, double* __restrict__ x,
             double* __restrict__ r, double* __restrict__ z, const double* __restrict__ part_pAp, double* __restrict__ part_rz, double* __restrict__ part_rr,
             PcgScal* sc, int parity, const double* __restrict__ b_init, const double* __restrict__ Ax_init, int use_dinv,
             const double* __restrict__ omega_dev = nullptr, int gb = 0, BodyDenseArgs BD = BodyDenseArgs{}, const float* __restrict__ Binv = nullptr,
             double* __restrict__ rb = nullptr, int rb_n = 0) {
  __shared__ double sm[8];
  double alpha = 0;
  if (rb && (int)blockIdx.x >= gb) {  // dense-body workgroups (k_body.hpp): z_b = Binv r_b of the updated residual, compact residual ping-pong
    __shared__ double vb[3 * 512 + 4];
    double* rb_new = rb + (size_t)(b_init ? parity ^ 1 : parity) * rb_n;
    const double* rb_old = rb + (size_t)(parity ^ 1) * rb_n;
    auto alpha_fn = [&]() -> double {
      const double nan = __longlong_as_double(0x7ff8000000000000LL);
      if (sc->flag) return nan;
      const double pAp = block_reduce_partials(part_pAp, sc->n_part1, sm);
      const double rz = sc->rzh[parity];
      if (!(pAp > 0.0) || !(rz > 0.0)) return nan;
      return rz / pAp;
    };
    body_update_block(BD, Binv, alpha_fn, Ap, rb_old, rb_new, b_init, Ax_init, z, (int)blockIdx.x - gb, vb);
    return;
  }
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  // operands of the update: loaded before the reduction of the p.Ap partials so that the two memory round trips overlap
  d3 pp(0, 0, 0), xv(0, 0, 0), rv(0, 0, 0), apv(0, 0, 0);
  m3 D;
  if (p < NV) {
    if (b_init) {
      rv = ld3(b_init, p);
      if (Ax_init) rv = rv - ld3(Ax_init, p);
    } else {
      pp = ld3(pv, p); xv = ld3(x, p); rv = ld3(r, p); apv = ld3(Ap, p);
    }
    if (use_dinv) {
#pragma unroll
      for (int e = 0; e < 9; e++) D.m[e] = Dinv[9 * (size_t)p + e];
    }
  }
  if (!b_init) {
    if (sc->flag) return;
    const double pAp = block_reduce_partials(part_pAp, sc->n_part1, sm);
    const double rz = sc->rzh[parity];
    if (!(pAp > 0.0) || !(rz > 0.0)) {
      if (blockIdx.x == 0 && threadIdx.x == 0) { sc->flag = 1; sc->pAp_last = pAp; }
      return;
    }
    alpha = rz / pAp;
  }
  double rzn = 0, rr = 0;
  if (p < NV) {
    if (!b_init) {
      st3(x, p, xv + alpha * pp);
      rv = rv - alpha * apv;
    }
    st3(r, p, rv);
    rr = dot(rv, rv);
    if (use_dinv) {
      // use_dinv 2: z = omega Dinv r is the first smoothing sweep of the multigrid cycle that follows (it owns r.z)
      const d3 zv = (use_dinv == 2 ? *omega_dev : 1.0) * m3_mulv(D, rv);
      bool own = true;
      if (rb) {  // rows of a dense body (zero Dinv block) get their z from the body workgroups of this launch
        own = false;
#pragma unroll
        for (int e = 0; e < 9; e++) own = own || D.m[e] != 0.0;
      }
      if (own) st3(z, p, zv);
      rzn = dot(rv, zv);
    }
  }
  rzn = wave_sum(rzn); rr = wave_sum(rr);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __shared__ double s1[4], s2[4];
  if (lane == 0) { s1[w] = rzn; s2[w] = rr; }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (use_dinv == 1) part_rz[blockIdx.x] = s1[0] + s1[1] + s1[2] + s1[3];  // otherwise the preconditioner's last kernel writes r.z
    part_rr[blockIdx.x] = s2[0] + s2[1] + s2[2] + s2[3];
  }
}

// sum of a partial array into out[0] (host convergence probes)
__global__ void k_sum_partials(const double* __restrict__ part, int n, double* out) {
  __shared__ double sm[8];
  const double t = block_reduce_partials(part, n, sm);
  if (threadIdx.x == 0) out[0] = t;
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
// sum of a*b over n doubles ADDED to *out.  part / ticket (deterministic mode): every workgroup leaves its partial (waves in order), the
// last one to arrive adds the partials in workgroup order and makes the one addition to *out -- a fixed summation order; null: one f64
// atomic per wave
TSL_DEV void dot_finish(double s, double* out, double* __restrict__ part, int* __restrict__ ticket) {
  __shared__ double sw[4];
  __shared__ int s_last;
  s = wave_sum(s);
  if (!part) { if ((threadIdx.x & 63) == 0) atomicAdd(out, s); return; }
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x] = ((sw[0] + sw[1]) + sw[2]) + sw[3];
    __threadfence();
    s_last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (unsigned i = 0; i < gridDim.x; i++) t += __hip_atomic_load(&part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *out += t;
    *ticket = 0;
  }
}
__global__ void __launch_bounds__(256) k_dot(size_t n, const double* __restrict__ a, const double* __restrict__ b, double* out, double* __restrict__ part, int* __restrict__ ticket) {
  double s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i] * b[i];
  dot_finish(s, out, part, ticket);
}
// max |a_i| into *out (non-negative doubles compare like their bit patterns)
__global__ void k_absmax(size_t n, const double* __restrict__ a, double* out) {
  double s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s = fmax(s, fabs(a[i]));
  s = wave_max(s);
  if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long*)out, (unsigned long long)__double_as_longlong(s));
}
// ---- preconditioned MINRES with device-resident recurrence scalars (host reads one record per chunk of iterations)
struct MrScal {
  double gamma, gamma_prev, eta, s_prev, s_cur, c_prev, c_cur;  // recurrence state
  double delta, g2n;                                            // dot-product accumulators of the running iteration
  double a1, a2, a3, cx;                                        // coefficients of the w / x update
  double thresh_eta;                                            // stop when |eta| <= thresh_eta
  int flag, iters;                                              // 0 running, 2 converged, 1 breakdown
};
// The Lanczos vectors z_j = M^-1 v_j are kept UNSCALED (z~_j = gamma_j z_j): the scaling 1 / gamma_j is applied where they are consumed
// (operator product: here; direction update: k_mr_wx) instead of by a pass of its own, and delta_j = (H z~_j . z~_j) / gamma_j^2 comes
// from the per-slice partials that the product kernel writes (every block re-reduces them, as the PCG update kernel does).
// v_next = (H z~) / gamma - (delta / gamma) v_cur - (gamma / gamma_prev) v_prev
__global__ void __launch_bounds__(256)
k_mr_vnext(size_t n, double* __restrict__ v_next, const double* __restrict__ v_cur, const double* __restrict__ v_prev, MrScal* __restrict__ sc,
           const double* __restrict__ part, int n_part) {
  __shared__ double sm[8];
  if (sc->flag) return;
  const double draw = block_reduce_partials(part, n_part, sm);
  const double g = sc->gamma, ig = 1.0 / g;
  const double delta = draw * ig * ig;
  const double a = delta * ig, b = g / sc->gamma_prev;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) v_next[i] = v_next[i] * ig - a * v_cur[i] - b * v_prev[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) sc->delta = delta;
}
// Givens update of the Lanczos tridiagonal (one workgroup; g2n = v_next . M^-1 v_next from the per-workgroup partials of the
// preconditioner's last kernel when part != null, else from the accumulator a k_dot launch filled)
__global__ void __launch_bounds__(256) k_mr_scal(MrScal* sc, const double* __restrict__ part, int n_part) {
  __shared__ double sm[8];
  if (sc->flag) return;
  double g2n_part = 0.0;
  if (part) g2n_part = block_reduce_partials(part, n_part, sm);
  if (threadIdx.x != 0) return;
  const double delta = sc->delta, g2n = part ? g2n_part : sc->g2n, gamma = sc->gamma;
  sc->delta = 0.0; sc->g2n = 0.0;
  if (!(g2n >= 0.0) || !isfinite(g2n) || !isfinite(delta)) { sc->flag = 1; sc->delta = delta; sc->g2n = g2n; return; }
  const double gamma_next = sqrt(g2n);
  const double a0 = sc->c_cur * delta - sc->c_prev * sc->s_cur * gamma;
  const double a1 = sqrt(a0 * a0 + gamma_next * gamma_next);
  if (!(a1 > 0.0)) { sc->flag = 1; return; }
  sc->a1 = a1;
  sc->a2 = sc->s_cur * delta + sc->c_prev * sc->c_cur * gamma;
  sc->a3 = sc->s_prev * gamma;
  const double c_next = a0 / a1, s_next = gamma_next / a1;
  sc->cx = c_next * sc->eta;
  sc->eta = -s_next * sc->eta;
  sc->gamma_prev = gamma; sc->gamma = gamma_next;
  sc->s_prev = sc->s_cur; sc->s_cur = s_next; sc->c_prev = sc->c_cur; sc->c_cur = c_next;
  sc->iters = sc->iters + 1;
  // flag 3: the update of this iteration still has to run (k_mr_wx turns it into 2)
  if (fabs(sc->eta) <= sc->thresh_eta || gamma_next == 0.0) sc->flag = 3;
}
// w_next = (z~ / gamma - a3 w_prev - a2 w_cur) / a1 ; x += cx w_next
__global__ void k_mr_wx(size_t n, const double* __restrict__ z, const double* __restrict__ w_prev, const double* __restrict__ w_cur, double* __restrict__ w_next,
                        double* __restrict__ x, MrScal* sc) {
  const int flag = sc->flag;
  if (flag == 1 || flag == 2) return;
  const double i1 = 1.0 / sc->a1, a2 = sc->a2, a3 = sc->a3, cx = sc->cx;
  const double ig = 1.0 / sc->gamma_prev;  // z is unscaled; k_mr_scal has already advanced gamma -> gamma_prev
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double w = (z[i] * ig - a3 * w_prev[i] - a2 * w_cur[i]) * i1;
    w_next[i] = w;
    x[i] += cx * w;
  }
}
__global__ void k_mr_seal(MrScal* sc) { if (sc->flag == 3) sc->flag = 2; }

// out[j] += V_j . w for j < k (V: k vectors of stride ld); grid (chunks, k)
__global__ void __launch_bounds__(256) k_multi_dot(size_t n, const double* __restrict__ V, size_t ld, const double* __restrict__ w, double* __restrict__ out, double* __restrict__ part,
                                                   int* __restrict__ ticket) {
  const double* v = V + (size_t)blockIdx.y * ld;
  double s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += v[i] * w[i];
  dot_finish(s, &out[blockIdx.y], part ? part + (size_t)blockIdx.y * gridDim.x : (double*)nullptr, ticket ? ticket + blockIdx.y : (int*)nullptr);
}
// w += sign * sum_j h[j] V_j  (h on the device: no host round trip between the projection and the update)
__global__ void __launch_bounds__(256) k_multi_axpy(size_t n, const double* __restrict__ V, size_t ld, int k, const double* __restrict__ h, double sign, double* __restrict__ w) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    double acc = 0;
    for (int j = 0; j < k; j++) acc += h[j] * V[(size_t)j * ld + i];
    w[i] += sign * acc;
  }
}
// out += sum over free dofs of scale * a_i * b_i for i in [i0, i1)  (parameter gradients of the system-identification adjoint)
// (part / ticket: per-block partials joined in a fixed order by the last block, like k_dot; null: one atomic per wave)
__global__ void __launch_bounds__(256) k_dot_free(size_t i0, size_t i1, const double* __restrict__ a, const double* __restrict__ b, const int* __restrict__ frozen, double scale, double* out,
                                                  double* __restrict__ part, int* __restrict__ ticket) {
  double s = 0;
  for (size_t i = i0 + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < i1; i += (size_t)gridDim.x * blockDim.x)
    if (!frozen[i]) s += a[i] * b[i];
  if (part) { dot_finish(scale * s, out, part, ticket); return; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0 && s != 0.0) atomicAdd(out, scale * s);
}
// Iterative refinement of the factorised solve (tsl_hip.hip: direct_refine): r = b - w (w = H x) and
// out[0..3] = {r.r, x.x, b.b, max |x_i|} in ONE launch with a fixed summation order (per-block partials, the last block to finish
// adds them up: no f64 atomics, the same bits every run).  part: 4 x gridDim doubles, ticket: one int, zero on entry and left zero.
// max |x_i| is the Newton loop's |p|max (calc_p_norm, BaseScene.py:1096-1103): the stop rule reads it from the same host record.
__global__ void __launch_bounds__(256) k_ir_resid(size_t n, const double* __restrict__ b, const double* __restrict__ w, const double* __restrict__ x, double* __restrict__ r,
                                                  double* __restrict__ part, int* __restrict__ ticket, double* __restrict__ out) {
  __shared__ double sm[8];
  __shared__ int last;
  double rr = 0, xx = 0, bb = 0, xm = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double bi = b[i], ri = bi - w[i], xi = x[i];
    r[i] = ri;
    rr += ri * ri; xx += xi * xi; bb += bi * bi; xm = fmax(xm, fabs(xi));
  }
  rr = block_sum(rr, sm); xx = block_sum(xx, sm); bb = block_sum(bb, sm);
  xm = wave_max(xm);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = xm;
  __syncthreads();
  if (threadIdx.x == 0) {
    xm = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
    part[4 * blockIdx.x] = rr; part[4 * blockIdx.x + 1] = xx; part[4 * blockIdx.x + 2] = bb; part[4 * blockIdx.x + 3] = xm;
    __threadfence();
    last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  for (int k = threadIdx.x; k < (int)gridDim.x; k += blockDim.x) {
    a0 += __builtin_nontemporal_load(&part[4 * k]); a1 += __builtin_nontemporal_load(&part[4 * k + 1]); a2 += __builtin_nontemporal_load(&part[4 * k + 2]);
    a3 = fmax(a3, __builtin_nontemporal_load(&part[4 * k + 3]));
  }
  a0 = block_sum(a0, sm); a1 = block_sum(a1, sm); a2 = block_sum(a2, sm);
  a3 = wave_max(a3);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a3;
  __syncthreads();
  if (threadIdx.x == 0) { out[0] = a0; out[1] = a1; out[2] = a2; out[3] = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3])); *ticket = 0; }
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
