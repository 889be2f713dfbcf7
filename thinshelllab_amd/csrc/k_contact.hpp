// Vertex-triangle contact: broad phase (uniform grid, sorted cell keys), narrow phase (closest triangle),
// constraint build, penalty + lagged friction energy / gradient / 12x12 blocks, matrix-free product.
// Reference: /root/reference/code/engine/geometry.py:23-229, contact_diff.py:4-130,
//            BaseScene.py:453-598 (energy/gradient/Hessian), :778-850 (constraints, vertex normals).
// Native organisation: a body's triangles are bucketed by a radix sort of their cell ids (hipCUB) instead of
// the reference's three prefix passes over a dense 132^3 grid; a query lane binary-searches the <=27 cells it
// needs.  Contact couplings change every step, so they are NOT merged into the static SELL matrix: each
// constraint keeps its dense 12x12 block in HBM and the PCG operator adds  sum_c P_c^T H_c P_c x  by atomics.
#pragma once

#include "k_solver.hpp"
#include "tsl_ctx.hpp"
#include "tsl_device.hpp"

struct ContactArgs {
  const int* idx;
  const double *w, *n, *dx0, *k, *mu, *T;
  double k_contact, eps_contact, eps_vh;
};

// BaseScene.f0/f1/f2 (:453-478), eh = eps_v * h
TSL_DEV double fr_f0(double x, double eh) { return (x > eh) ? x : (-x * x * x / (3.0 * eh * eh) + x * x / eh + eh / 3.0); }
TSL_DEV double fr_f1(double x, double eh) { return (x > eh) ? 1.0 / x : (-x / (eh * eh) + 2.0 / eh); }
TSL_DEV double fr_f2(double x, double eh) { return (x > eh) ? -1.0 / (x * x) : -1.0 / (eh * eh); }

// ------------------------------------------------------------------------------------------------ vertex normals
// area-weighted normal per vertex (BaseScene.calc_vn): the sum over its incident surface triangles from static vertex -> triangle lists in
// ascending triangle order: no atomics, a fixed order (the projection query's side flag proj_dir is the sign of a dot product with these normals)
__global__ void k_vn_gather(int NV, const int* __restrict__ ptr, const int* __restrict__ lst, const int* __restrict__ faces, const double* __restrict__ pos, double* __restrict__ vn) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NV) return;
  d3 a = d3();
  for (int e = ptr[i]; e < ptr[i + 1]; e++) {
    const int f = lst[e];
    const d3 v1 = ld3(pos, faces[3 * f]), v2 = ld3(pos, faces[3 * f + 1]), v3 = ld3(pos, faces[3 * f + 2]);
    a = a + cross(v2 - v1, v3 - v1);
  }
  st3(vn, i, a);
}
__global__ void k_vn_normalize(int NV, double* __restrict__ vn) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NV) return;
  st3(vn, i, normalized(ld3(vn, i)));
}

// ------------------------------------------------------------------------------------------------ grid
struct GridArgs { double h, bound; int n; };
TSL_DEV void grid_idx3(const GridArgs& G, const d3& x, int o[3]) {
  o[0] = (int)floor(fmin(fmax(x.x, -G.bound), G.bound) / G.h) + G.n / 2;
  o[1] = (int)floor(fmin(fmax(x.y, -G.bound), G.bound) / G.h) + G.n / 2;
  o[2] = (int)floor(fmin(fmax(x.z, -G.bound), G.bound) / G.h) + G.n / 2;
}
// ---- broad phase: the reference's p2g is a counting sort of the triangle centroids into a dense 132^3 grid (geometry.py:96-163).  The
// refined scenes scale the cell with the mesh (up to 10^9 cells), so the counting sort runs over HASH BUCKETS of the cell id instead
// (table of >= 2 nf buckets): count per bucket, exclusive scan, scatter, and a rank pass that orders every bucket by (cell, triangle)
// -- the candidates of a cell are then contiguous inside their bucket in ascending triangle index, the order the sequential selection
// rule of project_pair is applied in (in the reference it is an atomic-append order).  All hand-written: k_grid_keys,
// k_scan_local / _top / _add, k_bucket_scatter, k_bucket_rank.
TSL_DEV int grid_bucket(int cell, int hshift) { return (int)(((unsigned)cell * 2654435761u) >> hshift); }
// geometry.p2g first pass (:108-113): cell of every triangle centroid + active box + bucket histogram
__global__ void k_grid_keys(GridArgs G, int f_start, int nf, const int* __restrict__ faces, const double* __restrict__ pos, int* __restrict__ key,
                            int* __restrict__ range, int hshift, int* __restrict__ cnt) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nf) return;
  const int f = f_start + t;
  const d3 mid = (ld3(pos, faces[3 * f]) + ld3(pos, faces[3 * f + 1]) + ld3(pos, faces[3 * f + 2])) / 3.0;
  int id[3];
  grid_idx3(G, mid, id);
  const int cell = (id[0] * G.n + id[1]) * G.n + id[2];
  key[t] = cell;
  atomicAdd(&cnt[grid_bucket(cell, hshift)], 1);
  for (int a = 0; a < 3; a++) { atomicMin(&range[a], id[a]); atomicMax(&range[3 + a], id[a]); }
}
__global__ void k_grid_range_init(int* range, int n) {
  if (threadIdx.x < 3) range[threadIdx.x] = n; else if (threadIdx.x < 6) range[threadIdx.x] = 0;
}
// exclusive prefix sum of n ints in three launches: 1024 elements per workgroup (4 per thread), the workgroup totals scanned by one
// workgroup, the offsets added back
#define SCAN_TILE 1024
TSL_DEV int wg_exclusive_scan256(int v, int* total) {   // 256 threads; returns the exclusive prefix of v, *total = sum (valid in every thread)
  __shared__ int s_w[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
  if (lane == 63) s_w[w] = incl;
  __syncthreads();
  int base = 0;
  for (int k = 0; k < w; k++) base += s_w[k];
  *total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  __syncthreads();
  return base + incl - v;
}
__global__ void __launch_bounds__(256) k_scan_local(int n, const int* __restrict__ in, int* __restrict__ out, int* __restrict__ bsum) {
  const int i0 = blockIdx.x * SCAN_TILE + 4 * threadIdx.x;
  int v[4];
#pragma unroll
  for (int k = 0; k < 4; k++) v[k] = i0 + k < n ? in[i0 + k] : 0;
  int tot;
  int ex = wg_exclusive_scan256(v[0] + v[1] + v[2] + v[3], &tot);
#pragma unroll
  for (int k = 0; k < 4; k++) { if (i0 + k < n) out[i0 + k] = ex; ex += v[k]; }
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(256) k_scan_top(int nb, int* __restrict__ bsum) {   // in place, nb workgroup totals (any count: chunks of 256)
  int carry = 0;
  for (int c0 = 0; c0 < nb; c0 += 256) {
    const int i = c0 + threadIdx.x;
    const int v = i < nb ? bsum[i] : 0;
    int tot;
    const int ex = wg_exclusive_scan256(v, &tot);
    if (i < nb) bsum[i] = carry + ex;
    carry += tot;
  }
}
__global__ void __launch_bounds__(256) k_scan_add(int n, int* __restrict__ out, const int* __restrict__ bsum) {
  const int i0 = blockIdx.x * SCAN_TILE + 4 * threadIdx.x;
  const int b = bsum[blockIdx.x];
#pragma unroll
  for (int k = 0; k < 4; k++) if (i0 + k < n) out[i0 + k] += b;
}
static void scan_exclusive(hipStream_t s, int n, const int* in, int* out, int* bsum) {   // bsum: (n + SCAN_TILE - 1) / SCAN_TILE ints of scratch
  const int nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(256), 0, s, n, in, out, bsum);
  if (nb > 1) {
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(256), 0, s, nb, bsum);
    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(256), 0, s, n, out, (const int*)bsum);
  }
}
// members of every bucket in arrival order (atomic cursor) ...
__global__ void k_bucket_scatter(int nf, const int* __restrict__ key, int hshift, const int* __restrict__ ptr, int* __restrict__ cur, int* __restrict__ tmp) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nf) return;
  const int b = grid_bucket(key[t], hshift);
  tmp[ptr[b] + atomicAdd(&cur[b], 1)] = t;
}
// ... then every member finds its rank by (cell, triangle) among the members of its bucket (a handful): the result does not depend on the
// arrival order.  The ranking is quadratic in the bucket size: a bucket of more than TSL_BUCKET_CAP triangles -- a body that lies outside
// grid_extent (every centroid is clamped into the boundary cells, grid_idx3) or a cell size far above the triangle size -- is reported
// through `err` instead of being ranked (tsl_contact_detect fails loudly; its first member copies the bucket in arrival order so that the
// kernels already queued behind read valid indices).
#define TSL_BUCKET_CAP 4096
__global__ void k_bucket_rank(int nf, int f_start, const int* __restrict__ key, int hshift, const int* __restrict__ ptr, const int* __restrict__ tmp,
                              int* __restrict__ skey, int* __restrict__ sval, int* __restrict__ err) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nf) return;
  const int k = key[t], b = grid_bucket(k, hshift);
  const int s0 = ptr[b], s1 = ptr[b + 1];
  if (s1 - s0 > TSL_BUCKET_CAP) {
    if (tmp[s0] == t) {
      atomicMax(err, s1 - s0);
      for (int e = s0; e < s1; e++) { skey[e] = key[tmp[e]]; sval[e] = f_start + tmp[e]; }
    }
    return;
  }
  int rank = 0;
  for (int e = s0; e < s1; e++) { const int u = tmp[e]; const int ku = key[u]; rank += (ku < k || (ku == k && u < t)) ? 1 : 0; }
  skey[s0 + rank] = k;
  sval[s0 + rank] = f_start + t;
}

// geometry.pt2tri (:23-87)
TSL_DEV void pt2tri(const d3& x, const d3& p1, const d3& p2, const d3& p3, int& c, double& d, d3& w) {
  const d3 e1 = normalized(p2 - p1), e2 = normalized(p3 - p2), e3 = normalized(p1 - p3);
  const d3 n = -normalized(cross(e1, e3));
  const d3 x1 = x - dot(x - p1, n) * n;
  c = 0; d = 0; w = d3();
  if (dot(cross(x1 - p1, e1), n) > 0) {
    if (dot(x1 - p1, e1) < 0) { c = 1; d = norm(x - p1); w = d3(1, 0, 0); }
    else if (dot(x1 - p2, e1) > 0) { c = 2; d = norm(x - p2); w = d3(0, 1, 0); }
    else { c = -3; const double al = dot(x1 - p1, e1) / dot(p2 - p1, e1); d = norm(x - (p1 + al * (p2 - p1))); w = d3(1 - al, al, 0); }
  } else if (dot(cross(x1 - p2, e2), n) > 0) {
    if (dot(x1 - p2, e2) < 0) { c = 2; d = norm(x - p2); w = d3(0, 1, 0); }
    else if (dot(x1 - p3, e2) > 0) { c = 3; d = norm(x - p3); w = d3(0, 0, 1); }
    else { c = -1; const double al = dot(x1 - p2, e2) / dot(p3 - p2, e2); d = norm(x - (p2 + al * (p3 - p2))); w = d3(0, 1 - al, al); }
  } else if (dot(cross(x1 - p3, e3), n) > 0) {
    if (dot(x1 - p3, e3) < 0) { c = 3; d = norm(x - p3); w = d3(0, 0, 1); }
    else if (dot(x1 - p1, e3) > 0) { c = 1; d = norm(x - p1); w = d3(1, 0, 0); }
    else { c = -2; const double al = dot(x1 - p3, e3) / dot(p1 - p3, e3); d = norm(x - (p3 + al * (p1 - p3))); w = d3(al, 0, 1 - al); }
  } else {
    d = norm(x - x1);
    const double S = norm(cross(p3 - p1, p2 - p1));
    w = d3(dot(cross(p3 - p2, x1 - p2), n) / S, dot(cross(p1 - p3, x1 - p3), n) / S, dot(cross(p2 - p1, x1 - p1), n) / S);
  }
}

// geometry.project_pair (:165-221).  G lanes (a power of two <= 64) share one query vertex: the candidates of each of the
// <= 27 cells are taken in chunks of G consecutive ones, one per lane, and the reference's running selection rule (closer by more
// than 1e-5, or within 1e-5 and larger cosine) is applied to the chunk EXACTLY as a sequential scan would: the state (d_min,
// cos_max) is uniform in the group; a ballot finds the first lane of the chunk whose candidate replaces it, that candidate becomes
// the state, and the lanes behind it are tested again against the new state -- replacements are rare, so a chunk costs one ballot.
// (The rule is not transitive: on the refined cloths, where 1e-5 m is 2 % of an edge, per-lane running bests merged at the end
// picked another triangle than the sequential scan for some vertices -- round 3 parity check at 200 x 100 and 224 x 224.)  The
// reference's own candidate order inside a cell is an atomic-append order; here and in the oracle it is ascending triangle index.
// One lane per query (G = 1 behaviour) left 19 waves on the GPU for the pad-vertices-against-cloth query of the 100k-triangle
// scene (0.9 ms per launch).
struct ProjBest { double d, cs; int pos; };
TSL_DEV bool proj_replaces(const ProjBest& cur, const ProjBest& cand) {  // cand comes later in scan order (geometry.py:190)
  return cand.d < cur.d - 1e-5 || (cand.d < cur.d + 1e-5 && cand.cs > cur.cs);
}
// SELF: geometry_self.project_pair_self (geometry_self.py:166-230) -- the query vertices are the body's own vertices: triangles that
// contain the vertex are skipped, only projections INSIDE a triangle (c == 0) are candidates, the flag is "a candidate exists"
template <int G, bool SELF>
__global__ void __launch_bounds__(256)
k_project_pair(GridArgs Gr, int v_start, int v_end, int ex_lo, int ex_hi, int body_idx, int NV, int hshift, const int* __restrict__ bptr, const int* __restrict__ skey, const int* __restrict__ sval,
               const int* __restrict__ range, const int* __restrict__ faces, const double* __restrict__ pos, const double* __restrict__ vn,
               const int* __restrict__ border, int* __restrict__ proj_flag, int* __restrict__ proj_dir, int* __restrict__ proj_idx,
               double* __restrict__ proj_w) {
  const int i = v_start + (blockIdx.x * blockDim.x + threadIdx.x) / G;
  const int g = threadIdx.x & (G - 1);
  const bool live = i < v_end && !(i >= ex_lo && i < ex_hi);   // [ex_lo, ex_hi): the target body's own vertices inside a launch over ALL other bodies
  const d3 xq = live ? ld3(pos, i) : d3();
  int q[3];
  grid_idx3(Gr, xq, q);
  int r0[3], r1[3];
  for (int a = 0; a < 3; a++) { r0[a] = max(q[a] - 1, range[a]); r1[a] = min(q[a] + 1, range[3 + a]) + 1; }
  ProjBest cur{1e6, -1e6, 0};
  int owner = -1;   // lane (inside the group) that holds the current winner's data
  int bc = 0, ba = 0, bb = 0, bc3 = 0;
  d3 pw = d3();
  const int gbase = (threadIdx.x & 63) & ~(G - 1);
  const unsigned long long gbits = G == 64 ? ~0ull : ((1ull << G) - 1);
  if (live)
    for (int gi = r0[0]; gi < r1[0]; gi++)
      for (int gj = r0[1]; gj < r1[1]; gj++)
        for (int gk = r0[2]; gk < r1[2]; gk++) {
          const int cell = (gi * Gr.n + gj) * Gr.n + gk;
          const int bk = grid_bucket(cell, hshift);
          const int s0 = bptr[bk], s1 = bptr[bk + 1];   // the cell's triangles lie in its hash bucket, contiguous, in ascending triangle index
          for (int sb = s0; sb < s1; sb += G) {
            const int sidx = sb + g;
            bool valid = sidx < s1 && skey[sidx] == cell;
            int a = 0, b = 0, c3 = 0, c = 0;
            double d = 0.0, cs = 0.0;
            d3 w = d3();
            if (valid) {
              const int f = sval[sidx];
              a = faces[3 * f]; b = faces[3 * f + 1]; c3 = faces[3 * f + 2];
              if (SELF && (i == a || i == b || i == c3)) valid = false;
              else {
                const d3 v1 = ld3(pos, a), v2 = ld3(pos, b), v3 = ld3(pos, c3);
                pt2tri(xq, v1, v2, v3, c, d, w);
                if (SELF && c != 0) valid = false;
                else {
                  const d3 vt = v1 * w.x + v2 * w.y + v3 * w.z;
                  const d3 nt = normalized(cross(v2 - v1, v3 - v1));
                  cs = dot(xq - vt, nt);
                }
              }
            }
            unsigned long long gm;
            do {   // sequential rule over the chunk: first replacing candidate, then the ones behind it against the new state
              const bool rep = valid && proj_replaces(cur, ProjBest{d, cs, 0});
              gm = (__ballot(rep) >> gbase) & gbits;
              if (gm) {
                const int j = __ffsll((long long)gm) - 1;
                cur.d = __shfl(d, j, G); cur.cs = __shfl(cs, j, G);
                owner = j;
                if (g == j) { bc = c; ba = a; bb = b; bc3 = c3; pw = w; }
                valid = valid && g > j;
              }
            } while (gm);
          }
        }
  if (!live) return;
  const bool none = owner < 0;
  if (none ? (g != 0) : (g != owner)) return;
  int pflag = 0, pi0 = 0, pi1 = 0, pi2 = 0;
  if (!none) {
    pi0 = ba; pi1 = bb; pi2 = bc3;
    if (SELF || bc == 0) pflag = 1;
    else if (bc > 0) { const int pv = (bc == 1) ? ba : ((bc == 2) ? bb : bc3); pflag = !border[pv]; }
    else {
      const int p1 = (bc != -3) ? bc3 : ba;
      const int p2 = (bc == -3) ? bb : ((bc == -1) ? bb : ba);  // particle_idx[pid, 2 + c]
      pflag = !(border[p1] && border[p2]);
    }
  }
  const d3 v = pw.x * ld3(pos, pi0) + pw.y * ld3(pos, pi1) + pw.z * ld3(pos, pi2);
  const d3 n = pw.x * ld3(vn, pi0) + pw.y * ld3(vn, pi1) + pw.z * ld3(vn, pi2);
  const size_t bi = (size_t)body_idx * NV + i;
  if (proj_flag[bi] == 0 && pflag == 1) proj_dir[bi] = dot(xq - v, n) > 0 ? 1 : 0;
  proj_flag[bi] = pflag;
  proj_idx[3 * bi] = pi0; proj_idx[3 * bi + 1] = pi1; proj_idx[3 * bi + 2] = pi2;
  proj_w[3 * bi] = pw.x; proj_w[3 * bi + 1] = pw.y; proj_w[3 * bi + 2] = pw.z;
}

// BaseScene.contact_pair_analysis (:778-816)
__global__ void k_contact_pair(int b_idx, int v_start, int v_end, double mu, int NV, int max_nc, double k_contact, double eps_contact,
                               const double* __restrict__ pos, const double* __restrict__ prev, const int* __restrict__ proj_flag, const int* __restrict__ proj_dir,
                               const int* __restrict__ proj_idx, const double* __restrict__ proj_w, int* nc, int* __restrict__ c_idx, double* __restrict__ c_w,
                               double* __restrict__ c_k, double* __restrict__ c_mu, double* __restrict__ c_dx0, double* __restrict__ c_T, double* __restrict__ c_n, int kind,
                               int* __restrict__ c_kind, int phase, int qoff, int* __restrict__ qflag, const int* __restrict__ qscan) {
  // Two phases per detection so that the constraint LIST has a fixed order -- pair after pair in the scene's call order, query vertices
  // ascending (the reference appends with an atomic counter: any order): phase 0 writes the activity flag of every query vertex,
  // an exclusive scan over all pairs gives the slots, phase 1 writes the constraints there.
  const int i = v_start + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= v_end) return;
  const int qi = qoff + (i - v_start);
  const size_t bi = (size_t)b_idx * NV + i;
  if (!proj_flag[bi]) { if (phase == 0) qflag[qi] = 0; return; }
  int i0 = proj_idx[3 * bi], i1 = proj_idx[3 * bi + 1], i2 = proj_idx[3 * bi + 2];
  double w0 = proj_w[3 * bi], w1 = proj_w[3 * bi + 1], w2 = proj_w[3 * bi + 2];
  const d3 x_c = ld3(pos, i0) * w0 + ld3(pos, i1) * w1 + ld3(pos, i2) * w2;
  const d3 x0_c = ld3(prev, i0) * w0 + ld3(prev, i1) * w1 + ld3(prev, i2) * w2;
  d3 n_c = normalized(cross(ld3(pos, i1) - ld3(pos, i0), ld3(pos, i2) - ld3(pos, i0)));
  if (proj_dir[bi] == 0) {
    n_c = -n_c;
    const int ti = i1; i1 = i2; i2 = ti;
    const double tw = w1; w1 = w2; w2 = tw;
  }
  const double gap = dot(ld3(pos, i) - x_c, n_c);
  if (phase == 0) { qflag[qi] = gap < eps_contact ? 1 : 0; return; }
  if (gap < eps_contact) {
    const int c = qscan[qi];
    if (c >= max_nc) return;
    const double cforce = k_contact * (gap - eps_contact);
    c_idx[4 * c] = i0; c_idx[4 * c + 1] = i1; c_idx[4 * c + 2] = i2; c_idx[4 * c + 3] = i;
    c_w[3 * c] = w0; c_w[3 * c + 1] = w1; c_w[3 * c + 2] = w2;
    c_k[c] = -mu * cforce;
    c_mu[c] = mu;
    c_kind[c] = kind;  // which friction parameter the pair uses (0 fixed, 1 mu_cloth_elastic, 2 mu_cloth_cloth)
    st3(c_dx0, c, ld3(prev, i) - x0_c);
    d3 t1 = (fabs(n_c.x) < 0.5) ? d3(n_c.x, n_c.z, -n_c.y) : d3(n_c.y, -n_c.x, n_c.z);
    const d3 t2 = cross(n_c, t1);
    t1 = cross(n_c, t2);
    c_T[6 * c] = t1.x; c_T[6 * c + 1] = t1.y; c_T[6 * c + 2] = t1.z; c_T[6 * c + 3] = t2.x; c_T[6 * c + 4] = t2.y; c_T[6 * c + 5] = t2.z;
    st3(c_n, c, n_c);
  }
}

// ------------------------------------------------------------------------------------------------ energy
// BaseScene.contact_energy(diff=False) (:490-543 normal, :548-595 friction)
__global__ void k_contact_energy(int nc, ContactArgs A, const double* __restrict__ pos, double* __restrict__ e_part) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double e = 0;
  if (i < nc) {
    const int i0 = A.idx[4 * i], i1 = A.idx[4 * i + 1], i2 = A.idx[4 * i + 2], i3 = A.idx[4 * i + 3];
    const d3 x0 = ld3(pos, i0), xa = ld3(pos, i1), xb = ld3(pos, i2), xp = ld3(pos, i3);
    const d3 p1 = xa - x0, p2 = xb - x0, p = xp - x0;
    const d3 cr = cross(p1, p2);
    const double D = dot(cr, p), C = norm(cr);
    const double d = D / C;
    if (d < A.eps_contact) e += 0.5 * A.k_contact * (d - A.eps_contact) * (d - A.eps_contact);
    const d3 x_c = x0 * A.w[3 * i] + xa * A.w[3 * i + 1] + xb * A.w[3 * i + 2];
    const d3 dx = xp - x_c - ld3(A.dx0, i);
    const double* T = A.T + 6 * (size_t)i;
    const double u0 = T[0] * dx.x + T[1] * dx.y + T[2] * dx.z, u1 = T[3] * dx.x + T[4] * dx.y + T[5] * dx.z;
    e += A.k[i] * fr_f0(sqrt(u0 * u0 + u1 * u1), A.eps_vh);
  }
  e = wave_sum(e);
  if ((threadIdx.x & 63) == 0) e_part[blockIdx.x] = e;   // (one wave per workgroup: a partial per workgroup, summed in order by k_energy_final)
}

// ------------------------------------------------------------------------------------------------ gradient + blocks
// Normal part: d = D / C with D = p . (p1 x p2), C = |p1 x p2| in the relative coordinates q = (p1, p2, p); quotient rule as BaseScene.py:502-521,
// 9 x 9 eigen-clamp (linalg.py:15-148), friction block (BaseScene.py:548-593).  16 lanes per constraint: lane l < 9 of a group owns ROW l of the
// 9 x 9 normal block (relative coordinate l = 3 (vertex - 1) + axis), lanes 9..11 the three rows of vertex 0; the Jacobi eigen-clamp runs on the
// group's matrix in LDS with the parallel rotation order of spd_clamp9_par (tsl_device.hpp).  (One lane per constraint with the block and the
// eigenvector matrix in private arrays ran ~10 Jacobi sweeps serially: 0.9 ms per launch at 200 constraints; the cyclic order with one rotation at
// a time and 16 lanes in lock step: 125-185 us; gone.)
TSL_DEV d3 c_unit(int a) { return d3(a == 0 ? 1.0 : 0.0, a == 1 ? 1.0 : 0.0, a == 2 ? 1.0 : 0.0); }
TSL_DEV double c_comp(const d3& v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }
__global__ void __launch_bounds__(256)
k_contact_assemble_coop(int nc, ContactArgs A, const double* __restrict__ pos, int spd, double* __restrict__ Hfull, double* __restrict__ cg) {
  const int l = threadIdx.x & 15;
  int ci = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 4);
  const bool valid = ci < nc;
  if (!valid) ci = nc - 1;   // whole groups beyond the list still take part in the wave-wide shuffles
  int id[4];
  for (int k = 0; k < 4; k++) id[k] = A.idx[4 * ci + k];
  const d3 x0 = ld3(pos, id[0]), xa = ld3(pos, id[1]), xb = ld3(pos, id[2]), xp = ld3(pos, id[3]);
  const d3 a = xa - x0, b = xb - x0, p = xp - x0;
  const d3 cr = cross(a, b);
  const double D = dot(cr, p), C = norm(cr);
  const bool active = D / C < A.eps_contact;
  double h[9], g9 = 0.0;   // row l of the normal block in relative coordinates, gradient entry l
#pragma unroll
  for (int k = 0; k < 9; k++) h[k] = 0.0;
  if (active && l < 9) {
    const d3 nh = cr / C;
    const int jb = l / 3, ja = l % 3;
    const d3 ej = c_unit(ja);
    const d3 gD0 = cross(b, p), gD1 = cross(p, a), gD2 = cr;
    const d3 gC0 = cross(b, nh), gC1 = cross(nh, a), gC2 = d3(0, 0, 0);
    const double gDj = c_comp(jb == 0 ? gD0 : (jb == 1 ? gD1 : gD2), ja), gCj = c_comp(jb == 0 ? gC0 : (jb == 1 ? gC1 : gC2), ja);
    const double pe_pd = A.k_contact * (D / C - A.eps_contact);
    const double G9j = gDj / C - D * gCj / (C * C);
    const d3 ejb = cross(ej, b), aej = cross(a, ej);
#pragma unroll
    for (int k = 0; k < 9; k++) {
      const int kb = k / 3, ka = k % 3;
      const d3 ek = c_unit(ka);
      const double gDk = c_comp(kb == 0 ? gD0 : (kb == 1 ? gD1 : gD2), ka), gCk = c_comp(kb == 0 ? gC0 : (kb == 1 ? gC1 : gC2), ka);
      const d3 ekb = cross(ek, b), aek = cross(a, ek), ejk = cross(ej, ek);
      double HCv = 0.0, HDv = 0.0;
      if (jb == 0 && kb == 0) HCv = (dot(ejb, ekb) - dot(nh, ejb) * dot(nh, ekb)) / C;
      else if (jb == 1 && kb == 1) HCv = (dot(aej, aek) - dot(nh, aej) * dot(nh, aek)) / C;
      else if (jb == 0 && kb == 1) HCv = (dot(ejb, aek) - dot(nh, ejb) * dot(nh, aek)) / C + dot(nh, ejk);
      else if (jb == 1 && kb == 0) HCv = (dot(ekb, aej) - dot(nh, ekb) * dot(nh, aej)) / C - dot(nh, ejk);   // hab(k, j): e_k x e_j = -e_j x e_k
      // D = a . (b x p): mixed second derivatives (e_j x e_k) . third vector, antisymmetric in (j, k)
      if (jb == 0 && kb == 1) HDv = dot(ejk, p);
      else if (jb == 1 && kb == 0) HDv = -dot(ejk, p);
      else if (jb == 1 && kb == 2) HDv = dot(ejk, a);
      else if (jb == 2 && kb == 1) HDv = -dot(ejk, a);
      else if (jb == 2 && kb == 0) HDv = dot(ejk, b);
      else if (jb == 0 && kb == 2) HDv = -dot(ejk, b);
      const double G9k = gDk / C - D * gCk / (C * C);
      const double H9 = HDv / C - gDj * gCk / (C * C) - gDk * gCj / (C * C) - D * HCv / (C * C) + 2 * D * gCj * gCk / (C * C * C);
      h[k] = A.k_contact * G9j * G9k + pe_pd * H9;
    }
    g9 = G9j * pe_pd;
  }
  if (spd) {
    __shared__ double sA[16][81], sV[16][81];
    const int g = threadIdx.x >> 4;
    if (l < 9) {
#pragma unroll
      for (int k = 0; k < 9; k++) sA[g][l * 9 + k] = h[k];
    }
    spd_clamp9_cold(sA[g], sV[g], l, active);   // (tsl_device.hpp: nine rounds of four simultaneous rotations per sweep)
    if (l < 9) {
#pragma unroll
      for (int k = 0; k < 9; k++) h[k] = sA[g][l * 9 + k];
    }
  }
  // 12 x 12 block: lane l < 9 -> row 3 + l; lanes 9..11 -> rows 0..2 of vertex 0 (minus the sums over the three other vertices)
  double out[12];
  double gout = 0.0;
  const int axis = l < 9 ? l % 3 : l - 9;
  {
    double s0[9], sg = 0.0;   // sums over the rows l', l' % 3 == axis (for the vertex-0 rows)
#pragma unroll
    for (int k = 0; k < 9; k++) {
      const int base = (l >= 9 && l < 12) ? l - 9 : 0;
      s0[k] = __shfl(h[k], base, 16) + __shfl(h[k], base + 3, 16) + __shfl(h[k], base + 6, 16);
    }
    {
      const int base = (l >= 9 && l < 12) ? l - 9 : 0;
      sg = __shfl(g9, base, 16) + __shfl(g9, base + 3, 16) + __shfl(g9, base + 6, 16);
    }
    if (l < 9) {
#pragma unroll
      for (int k = 0; k < 9; k++) out[3 + k] = h[k];
#pragma unroll
      for (int j2 = 0; j2 < 3; j2++) out[j2] = -(h[j2] + h[3 + j2] + h[6 + j2]);
      gout = g9;
    } else {
#pragma unroll
      for (int k = 0; k < 9; k++) out[3 + k] = -s0[k];
#pragma unroll
      for (int j2 = 0; j2 < 3; j2++) out[j2] = s0[j2] + s0[3 + j2] + s0[6 + j2];
      gout = -sg;
    }
  }
  // friction (BaseScene.py:548-593)
  {
    const double w[3] = {A.w[3 * ci], A.w[3 * ci + 1], A.w[3 * ci + 2]};
    const double kf = A.k[ci];
    const double* T = A.T + 6 * (size_t)ci;
    const d3 x_c = x0 * w[0] + xa * w[1] + xb * w[2];
    const d3 dx = xp - x_c - ld3(A.dx0, ci);
    const double u[2] = {T[0] * dx.x + T[1] * dx.y + T[2] * dx.z, T[3] * dx.x + T[4] * dx.y + T[5] * dx.z};
    const double r = sqrt(u[0] * u[0] + u[1] * u[1]);
    const double f1 = fr_f1(r, A.eps_vh), f2 = fr_f2(r, A.eps_vh);
    double ha = f1, hb = 0, hd = f1;
    if (r > 1e-9) { ha += f2 * u[0] * u[0] / r; hb += f2 * u[0] * u[1] / r; hd += f2 * u[1] * u[1] / r; }
    if (spd) spd_clamp2(ha, hb, hd);
    const double w1[4] = {-w[0], -w[1], -w[2], 1.0};
    const int i1 = l < 9 ? 1 + l / 3 : 0;
    const double wi = i1 == 0 ? w1[0] : (i1 == 1 ? w1[1] : (i1 == 2 ? w1[2] : w1[3]));
    const double Ta = T[axis], Tb = T[3 + axis];
    gout += wi * kf * f1 * (u[0] * Ta + u[1] * Tb);
#pragma unroll
    for (int i2 = 0; i2 < 4; i2++)
#pragma unroll
      for (int j2 = 0; j2 < 3; j2++)
        out[i2 * 3 + j2] += wi * w1[i2] * kf * (Ta * (ha * T[j2] + hb * T[3 + j2]) + Tb * (hb * T[j2] + hd * T[3 + j2]));
  }
  if (!valid || l >= 12) return;
  const int rowi = l < 9 ? 3 + l : l - 9;
  if (cg) cg[12 * (size_t)ci + rowi] = gout;   // (per-constraint gradients, summed per vertex by k_contact_row_gather)
  double* dst = Hfull + 144 * (size_t)ci + 12 * rowi;
#pragma unroll
  for (int k = 0; k < 12; k++) dst[k] = out[k];
}

// masked copy of the per-constraint blocks (add_H frozen rule, BaseScene.py:399-405)
// One thread per ENTRY (nc x 144): the one-thread-per-constraint version walked 144 dependent global accesses per lane and took
// 96 us at 135 constraints on the contact stream of an assembly.
__global__ void k_contact_mask(int nc, const int* __restrict__ idx, const int* __restrict__ frozen, const double* __restrict__ Hfull, double* __restrict__ Hm) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)nc * 144) return;
  const int ci = (int)(t / 144), e = (int)(t % 144), r = e / 12, c = e % 12;
  const int vr = idx[4 * ci + r / 3], vc = idx[4 * ci + c / 3];
  const double v = (frozen[3 * vr + r % 3] || frozen[3 * vc + c % 3]) ? 0.0 : Hfull[t];
  Hm[t] = v;
}
// their diagonal 3 x 3 blocks for the block-Jacobi preconditioner: one thread per (permuted) row sums the diagonal 3 x 3 sub-blocks of the row's entries in
// their stored order (ascending constraint, slot); every row is written (zero without entries): no clear needed
__global__ void __launch_bounds__(256) k_contact_diag(int NV, const int* __restrict__ ptr, const int* __restrict__ ent, const double* __restrict__ Hm, double* __restrict__ cdiag) {
  // one WAVE per row: lanes over the row's entries (l, l + 64, ...), joined by the fixed tree of wave_sum
  const int p = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (p >= NV) return;
  const int r0 = ptr[p], r1 = ptr[p + 1];
  double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int e = r0 + lane; e < r1; e += 64) {
    const int q = ent[e];
    const double* H = Hm + 144 * (size_t)(q >> 2) + 39 * (q & 3);   // element (3 a, 3 a) of the 12 x 12 block
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) a[3 * r + c] += H[12 * r + c];
  }
  if (r1 > r0) {
#pragma unroll
    for (int k = 0; k < 9; k++) a[k] = wave_sum(a[k]);
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 9; k++) cdiag[9 * (size_t)p + k] = a[k];
  }
}

// y += sum_c P_c^T H_c P_c x  (permuted vectors).  16 lanes per constraint (12 active, one per block row), 64 constraints
// per 1024-thread workgroup: one lane per constraint (12x12 serial product behind dependent loads) took 12 us for a
// hundred constraints.  Returns this lane's share of x . H_c x.
#define CONTACT_MV_THREADS 1024
TSL_DEV double contact_matvec_lane(int nc, const int* __restrict__ idx, const int* __restrict__ rowpos, const double* __restrict__ Hm, const double* __restrict__ x,
                                   double* __restrict__ y) {
  const int ci = blockIdx.x * 64 + (threadIdx.x >> 4);
  const int r = threadIdx.x & 15;
  if (ci >= nc || r >= 12) return 0.0;
  const double* H = Hm + 144 * (size_t)ci + 12 * r;
  double s = 0, xr = 0;
  int prr = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int pk = rowpos[idx[4 * ci + k]];
    const d3 v = ld3(x, pk);
    s += H[3 * k] * v.x + H[3 * k + 1] * v.y + H[3 * k + 2] * v.z;
    if (k == r / 3) { prr = pk; xr = (r % 3 == 0) ? v.x : (r % 3 == 1) ? v.y : v.z; }
  }
  if (s != 0.0) atomicAdd(&y[3 * (size_t)prr + (r % 3)], s);
  return s * xr;
}

// row -> (constraint, slot) lists for the SpMV kernels (ContactRows, k_solver.hpp): count, exclusive scan, fill
__global__ void k_cr_count(int nc, const int* __restrict__ idx, const int* __restrict__ rowpos, int* __restrict__ cnt) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= 4 * nc) return;
  atomicAdd(&cnt[rowpos[idx[q]]], 1);
}
// entries of every row in arrival order (atomic cursor) ...
__global__ void k_cr_fill(int nc, const int* __restrict__ idx, const int* __restrict__ rowpos, const int* __restrict__ ptr, int* __restrict__ fill, int* __restrict__ tmp) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= 4 * nc) return;
  const int p = rowpos[idx[q]];
  tmp[ptr[p] + atomicAdd(&fill[p], 1)] = q;
}
// ... then in ascending (constraint, slot) order: the sums over a row's entries (products, diagonal blocks, gradients) have a fixed order
__global__ void k_cr_rank(int nc, const int* __restrict__ idx, const int* __restrict__ rowpos, const int* __restrict__ ptr, const int* __restrict__ tmp, int* __restrict__ ent,
                          int4* __restrict__ rows) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= 4 * nc) return;
  const int c4 = q & ~3;
  const int4 r = make_int4(rowpos[idx[c4]], rowpos[idx[c4 + 1]], rowpos[idx[c4 + 2]], rowpos[idx[c4 + 3]]);
  const int p = (q & 3) == 0 ? r.x : (q & 3) == 1 ? r.y : (q & 3) == 2 ? r.z : r.w;
  const int s0 = ptr[p], s1 = ptr[p + 1];
  int rank = 0;
  for (int e = s0; e < s1; e++) rank += tmp[e] < q ? 1 : 0;
  ent[s0 + rank] = q;  // (constraint << 2) | slot
  rows[s0 + rank] = r;
}

// dot(x, H_c x) added to pAp[slot] (slot < 0: product only)
__global__ void __launch_bounds__(CONTACT_MV_THREADS)
k_contact_matvec(int nc, const int* __restrict__ idx, const int* __restrict__ rowpos, const double* __restrict__ Hm, const double* __restrict__ x,
                 double* __restrict__ y, CgScal* sc, int slot, int check_flag) {
  if (check_flag && sc->flag) return;
  double acc = contact_matvec_lane(nc, idx, rowpos, Hm, x, y);
  if (slot >= 0) {
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0 && acc != 0.0) atomicAdd(&sc->pAp[slot], acc);
  }
}



// BaseScene.contact_energy_backprop (:682-730): friction-lag adjoint into pos_grad[step-1] (pg points at that slice)
__global__ void k_contact_backprop(int nc, ContactArgs A, const double* __restrict__ pos, const double* __restrict__ z, double* __restrict__ cg) {
  const int ci = blockIdx.x * blockDim.x + threadIdx.x;
  if (ci >= nc) return;
  int id[4];
  for (int k = 0; k < 4; k++) id[k] = A.idx[4 * ci + k];
  const d3 x0 = ld3(pos, id[0]), xa = ld3(pos, id[1]), xb = ld3(pos, id[2]), xp = ld3(pos, id[3]);
  const double w[3] = {A.w[3 * ci], A.w[3 * ci + 1], A.w[3 * ci + 2]};
  const double kf = A.k[ci];
  const double* T = A.T + 6 * (size_t)ci;
  const d3 x_c = x0 * w[0] + xa * w[1] + xb * w[2];
  const d3 dx = xp - x_c - ld3(A.dx0, ci);
  const double u[2] = {T[0] * dx.x + T[1] * dx.y + T[2] * dx.z, T[3] * dx.x + T[4] * dx.y + T[5] * dx.z};
  const double r = sqrt(u[0] * u[0] + u[1] * u[1]);
  const double f1 = fr_f1(r, A.eps_vh), f2 = fr_f2(r, A.eps_vh);
  const double pressure = kf / A.mu[ci];
  double g1[3];
  for (int j = 0; j < 3; j++) g1[j] = kf * f1 * (u[0] * T[j] + u[1] * T[3 + j]);
  const d3 n_c = ld3(A.n, ci);
  double acc[12];
  for (int k = 0; k < 12; k++) acc[k] = 0;
  double zv[12];
  for (int k = 0; k < 4; k++) for (int j = 0; j < 3; j++) zv[3 * k + j] = z[3 * (size_t)id[k] + j];
  {
    const double wp[4] = {w[0], w[1], w[2], -1.0};
    double s = 0;
    for (int i1 = 0; i1 < 4; i1++)
      for (int j1 = 0; j1 < 3; j1++) s += zv[3 * i1 + j1] * (wp[i1] * g1[j1] / pressure);
    const double nn[3] = {n_c.x, n_c.y, n_c.z};
    for (int i2 = 0; i2 < 4; i2++)
      for (int j2 = 0; j2 < 3; j2++) acc[3 * i2 + j2] += s * wp[i2] * nn[j2] * A.k_contact;
  }
  {
    double ha = f1, hb = 0, hd = f1;
    if (r > 1e-9) { ha += f2 * u[0] * u[0] / r; hb += f2 * u[0] * u[1] / r; hd += f2 * u[1] * u[1] / r; }
    double h1[9];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) h1[a * 3 + b] = kf * (T[a] * (ha * T[b] + hb * T[3 + b]) + T[3 + a] * (hb * T[b] + hd * T[3 + b]));
    const double w1[4] = {-w[0], -w[1], -w[2], 1.0};
    for (int i1 = 0; i1 < 4; i1++)
      for (int i2 = 0; i2 < 4; i2++)
        for (int j1 = 0; j1 < 3; j1++)
          for (int j2 = 0; j2 < 3; j2++) acc[3 * i2 + j2] += zv[3 * i1 + j1] * w1[i1] * w1[i2] * h1[j1 * 3 + j2];
  }
  for (int k = 0; k < 12; k++) cg[12 * (size_t)ci + k] = acc[k];   // summed per vertex by k_contact_row_gather
}

// Scene_sliding.contact_energy_backprop_friction (Scene_sliding.py:139-176): d(loss)/d(mu_cloth_cloth) contribution of the
// constraints whose pair uses that parameter (the reference loops over the first nc1 constraints = the cloth-cloth pairs):
// sum over the free dofs of z * w1 * g1 / mu_cloth_cloth, g1 = T^T (k f1(r) u), w1 = (w0, w1, w2, -1).
// (part / ticket: one partial per workgroup = wave, joined in workgroup order by the last one to finish; null: one atomic per wave)
__global__ void k_contact_friction_grad(int nc, ContactArgs A, const int* __restrict__ kind, const int* __restrict__ frozen, const double* __restrict__ pos,
                                        const double* __restrict__ z, double mu_cc, double* out, double* __restrict__ part, int* __restrict__ ticket) {
  const int ci = blockIdx.x * blockDim.x + threadIdx.x;
  double s = 0;
  if (ci < nc && kind[ci] == 2) {
    int id[4];
    for (int k = 0; k < 4; k++) id[k] = A.idx[4 * ci + k];
    const d3 x0 = ld3(pos, id[0]), xa = ld3(pos, id[1]), xb = ld3(pos, id[2]), xp = ld3(pos, id[3]);
    const double w[3] = {A.w[3 * ci], A.w[3 * ci + 1], A.w[3 * ci + 2]};
    const double kf = A.k[ci];
    const double* T = A.T + 6 * (size_t)ci;
    const d3 dx = xp - (x0 * w[0] + xa * w[1] + xb * w[2]) - ld3(A.dx0, ci);
    const double u[2] = {T[0] * dx.x + T[1] * dx.y + T[2] * dx.z, T[3] * dx.x + T[4] * dx.y + T[5] * dx.z};
    const double r = sqrt(u[0] * u[0] + u[1] * u[1]);
    const double f1 = fr_f1(r, A.eps_vh);
    const double wp[4] = {w[0], w[1], w[2], -1.0};
    for (int i1 = 0; i1 < 4; i1++)
      for (int j1 = 0; j1 < 3; j1++) {
        if (frozen[3 * id[i1] + j1]) continue;
        const double g1 = kf * f1 * (u[0] * T[j1] + u[1] * T[3 + j1]);
        s += z[3 * (size_t)id[i1] + j1] * wp[i1] * g1 / mu_cc;
      }
  }
  s = wave_sum(s);
  if (part) {
    if (threadIdx.x == 0) {
      part[blockIdx.x] = s;
      __threadfence();
      if (atomicAdd(ticket, 1) == (int)gridDim.x - 1) {
        __threadfence();
        double t = 0.0;
        for (unsigned i = 0; i < gridDim.x; i++) t += __hip_atomic_load(&part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *out += t;
        *ticket = 0;
      }
    }
    return;
  }
  if ((threadIdx.x & 63) == 0 && s != 0.0) atomicAdd(out, s);
}

// batched projections for unit tests
__global__ void k_spd_batch(double* blocks, int n, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (D == 3) {
    double A[9];
    for (int k = 0; k < 9; k++) A[k] = blocks[9 * (size_t)i + k];
    spd_clamp<3>(A);
    for (int k = 0; k < 9; k++) blocks[9 * (size_t)i + k] = A[k];
  } else if (D == 9) {
    double A[81];
    for (int k = 0; k < 81; k++) A[k] = blocks[81 * (size_t)i + k];
    spd_clamp<9>(A);
    for (int k = 0; k < 81; k++) blocks[81 * (size_t)i + k] = A[k];
  } else {
    double a = blocks[4 * (size_t)i], b = 0.5 * (blocks[4 * (size_t)i + 1] + blocks[4 * (size_t)i + 2]), d = blocks[4 * (size_t)i + 3];
    spd_clamp2(a, b, d);
    blocks[4 * (size_t)i] = a; blocks[4 * (size_t)i + 1] = b; blocks[4 * (size_t)i + 2] = b; blocks[4 * (size_t)i + 3] = d;
  }
}

// ------------------------------------------------------------------------------------------------ host side
static inline int cnblk(long n, int b) { return (int)((n + b - 1) / b); }

static int contact_alloc(tsl_ctx* c, const tsl_scene_desc* d) {
  int rc = 0;
  const int NV = c->NV;
  std::vector<int> faces;
  if (d->tot_NF > 0 && d->faces_host) faces.assign(d->faces_host, d->faces_host + 3 * (size_t)d->tot_NF);
  rc |= c->faces.upload(faces);
  rc |= c->vn.alloc(3 * (size_t)NV);
  {   // vertex -> incident surface triangles (k_vn_gather)
    std::vector<int> ptr(NV + 1, 0), lst(faces.size());
    for (int v : faces) ptr[v + 1]++;
    for (int v = 0; v < NV; v++) ptr[v + 1] += ptr[v];
    std::vector<int> cur(ptr.begin(), ptr.end() - 1);
    for (size_t f = 0; f < faces.size() / 3; f++) for (int k = 0; k < 3; k++) lst[cur[faces[3 * f + k]]++] = (int)f;
    if (!faces.empty()) { rc |= c->vnf_ptr.upload(ptr); rc |= c->vnf_lst.upload(lst); }
  }
  const size_t nb = (size_t)std::max(c->n_body, 1);
  rc |= c->proj_flag.alloc(nb * NV); rc |= c->proj_dir.alloc(nb * NV); rc |= c->proj_idx.alloc(nb * NV * 3); rc |= c->proj_w.alloc(nb * NV * 3);
  rc |= c->nc_dev.alloc(2);   // [1]: largest oversized broad-phase bucket of the last detection (k_bucket_rank)
  const size_t mc = (size_t)c->max_n_constraints;
  rc |= c->c_idx.alloc(mc * 4); rc |= c->c_w.alloc(mc * 3); rc |= c->c_n.alloc(mc * 3); rc |= c->c_dx0.alloc(mc * 3);
  rc |= c->c_k.alloc(mc); rc |= c->c_mu.alloc(mc); rc |= c->c_T.alloc(mc * 6); rc |= c->c_kind.alloc(mc);
  rc |= c->c_H.alloc(mc * 144); rc |= c->c_Hfull.alloc(mc * 144); rc |= c->c_diag.alloc((size_t)NV * 9);
  int mbf = 1;
  for (auto& b : c->h_bodies) mbf = std::max(mbf, b.f_end - b.f_start);
  c->max_body_faces = mbf;
  int ts = 64;
  while (ts < 2 * mbf) ts <<= 1;   // hash buckets of the broad phase: a power of two >= 2 x the largest triangle set
  c->grid_buckets_max = ts;
  // every target body has its OWN broad-phase grid (keys, buckets, scan scratch): the grids of a detection are built and queried side by side on three streams
  {
    size_t tf = 0, tb = 0, tsn = 0;
    c->gb_f0.clear(); c->gb_t0.clear(); c->gb_s0.clear();
    for (auto& b : c->h_bodies) {
      const int nf = std::max(b.f_end - b.f_start, 0);
      int tsb = 64;
      while (tsb < 2 * nf) tsb <<= 1;
      c->gb_f0.push_back(tf); c->gb_t0.push_back(tb); c->gb_s0.push_back(tsn);
      tf += (size_t)nf + 1; tb += (size_t)tsb + 2; tsn += (size_t)(tsb + 1) / SCAN_TILE + 2;
    }
    rc |= c->grid_key.alloc(tf); rc |= c->grid_val.alloc(tf); rc |= c->grid_key2.alloc(tf); rc |= c->grid_val2.alloc(tf); rc |= c->grid_range.alloc(8 * std::max<size_t>(c->h_bodies.size(), 1));
    rc |= c->grid_cnt.alloc(tb); rc |= c->grid_ptr.alloc(tb); rc |= c->grid_cur.alloc(tb); rc |= c->grid_scan.alloc(tsn);
  }
  rc |= c->scan_tmp.alloc((size_t)std::max(ts, NV + 1) / SCAN_TILE + 2);
  // border_flag (BaseScene.py:82): all zero unless imported
  rc |= c->border.alloc(NV);
  if (rc) return -1;
  c->proj_flag.zero(); c->proj_dir.zero(); c->proj_idx.zero(); c->proj_w.zero(); c->border.zero(); c->nc_dev.zero();
  c->nc = 0;
  return 0;
}

extern "C" int tsl_contact_reset(tsl_ctx* c) { Scope scope(c); return c->proj_flag.zero(c->stream); }

extern "C" int tsl_contact_detect(tsl_ctx* c, const double* pos, const double* prev, int32_t* nc_host) {
  Scope scope(c);
  hipStream_t s = c->stream;
  const int NV = c->NV;
  c->nc = 0;
  c->bd_valid = false;
  c->ds.cons_checked = false;   // the factorisation plan compares the new constraint list with the one it was made for
  bool any_self = false;
  for (int v : c->self_contact) any_self |= v != 0;
  if ((c->n_body < 2 && !any_self) || c->NF == 0) { if (nc_host) *nc_host = 0; return 0; }   // a single body can still touch itself (geometry_self.py)
  HIP_OK(hipMemsetAsync(c->nc_dev.p + 1, 0, sizeof(int), s));
  // calc_vn
  hipLaunchKernelGGL(k_vn_gather, dim3(cnblk(NV, 256)), dim3(256), 0, s, NV, (const int*)c->vnf_ptr.p, (const int*)c->vnf_lst.p, c->faces.p, pos, c->vn.p);
  hipLaunchKernelGGL(k_vn_normalize, dim3(cnblk(NV, 256)), dim3(256), 0, s, NV, c->vn.p);
  // projection_query
  GridArgs G;
  G.h = c->grid_h; G.n = (int)floor(c->grid_extent / c->grid_h) * 2; G.bound = c->grid_h * (G.n - 1) / 2;
  // do the bodies' vertex ranges tile one interval (no gap, no overlap)?  Then a target body is queried by "everything else" in one launch
  int v_lo_all = 0, v_hi_all = 0;
  bool tiled = c->n_body > 0;
  {
    std::vector<std::pair<int, int>> rg;
    for (int b = 0; b < c->n_body; b++) if (c->h_bodies[b].v_end > c->h_bodies[b].v_start) rg.push_back({c->h_bodies[b].v_start, c->h_bodies[b].v_end});
    std::sort(rg.begin(), rg.end());
    for (size_t k = 0; k + 1 < rg.size(); k++) tiled &= rg[k].second == rg[k + 1].first;
    if (rg.empty()) tiled = false; else { v_lo_all = rg.front().first; v_hi_all = rg.back().second; }
  }
  // (Round 6) the target bodies' chains on the context's three streams side by side: 2 x 1.2 ms of dependent launches per bench step one after the other
  const bool multi = c->n_body > 2 && c->side != nullptr && c->side2 != nullptr;
  hipStream_t strm[3] = {c->stream, c->side, c->side2};
  int n_chain = 0;
  if (multi) {
    HIP_OK(hipEventRecord(c->ev_fork0, c->stream));
    HIP_OK(hipStreamWaitEvent(c->side, c->ev_fork0, 0));
    HIP_OK(hipStreamWaitEvent(c->side2, c->ev_fork0, 0));
  }
  for (int b = 0; b < c->n_body; b++) {
    const tsl_body& body = c->h_bodies[b];
    const int nf = body.f_end - body.f_start;
    if (nf <= 0) continue;
    int ts = 64, lg = 6;
    while (ts < 2 * nf) { ts <<= 1; lg++; }
    const int hshift = 32 - lg;
    // this body's grid buffers and its stream (the chains of the target bodies -- five small launches for the grid, one query launch -- are independent)
    hipStream_t s = multi ? strm[n_chain++ % 3] : c->stream;
    int *g_key = c->grid_key.p + c->gb_f0[b], *g_val = c->grid_val.p + c->gb_f0[b], *g_key2 = c->grid_key2.p + c->gb_f0[b], *g_val2 = c->grid_val2.p + c->gb_f0[b];
    int *g_cnt = c->grid_cnt.p + c->gb_t0[b], *g_ptr = c->grid_ptr.p + c->gb_t0[b], *g_cur = c->grid_cur.p + c->gb_t0[b], *g_range = c->grid_range.p + 8 * b, *g_scan = c->grid_scan.p + c->gb_s0[b];
    hipLaunchKernelGGL(k_grid_range_init, dim3(1), dim3(64), 0, s, g_range, G.n);
    HIP_OK(hipMemsetAsync(g_cnt, 0, ((size_t)ts + 1) * sizeof(int), s));
    HIP_OK(hipMemsetAsync(g_cur, 0, (size_t)ts * sizeof(int), s));
    hipLaunchKernelGGL(k_grid_keys, dim3(cnblk(nf, 256)), dim3(256), 0, s, G, body.f_start, nf, c->faces.p, pos, g_key, g_range, hshift, g_cnt);
    scan_exclusive(s, ts + 1, g_cnt, g_ptr, g_scan);
    hipLaunchKernelGGL(k_bucket_scatter, dim3(cnblk(nf, 256)), dim3(256), 0, s, nf, (const int*)g_key, hshift, (const int*)g_ptr, g_cur, g_val);
    hipLaunchKernelGGL(k_bucket_rank, dim3(cnblk(nf, 256)), dim3(256), 0, s, nf, body.f_start, (const int*)g_key, hshift, (const int*)g_ptr, (const int*)g_val,
                       g_key2, g_val2, c->nc_dev.p + 1);
    // Round 6: ONE query launch per target body over the vertices of ALL other bodies (the bodies tile the vertex array: [v_lo, v_hi) minus the target's own
    // range) instead of one launch per ordered pair -- 30 dependent launches of 60 us (the cloth's 52k vertices against a pad of a few hundred triangles: 200
    // workgroups, a latency-bound scan) were 1.9 ms per detection, 3.8 ms per bench step; the same arithmetic per query vertex, the same bits.
#define TSL_PROJ_LAUNCH(GW, SF, VLO, VHI, XLO, XHI)                                                                                                                  \
  hipLaunchKernelGGL((k_project_pair<GW, SF>), dim3(cnblk((long)((VHI) - (VLO)) * GW, 256)), dim3(256), 0, s, G, (VLO), (VHI), (XLO), (XHI), b, NV, hshift, (const int*)g_ptr, \
                     g_key2, g_val2, g_range, c->faces.p, pos, c->vn.p, c->border.p, c->proj_flag.p, c->proj_dir.p, c->proj_idx.p, c->proj_w.p)
#define TSL_PROJ_BY_SIZE(SF, VLO, VHI, XLO, XHI)                                                                                                                     \
  do { if (nf >= 8192) TSL_PROJ_LAUNCH(64, SF, VLO, VHI, XLO, XHI); else if (nf >= 512) TSL_PROJ_LAUNCH(8, SF, VLO, VHI, XLO, XHI); else TSL_PROJ_LAUNCH(1, SF, VLO, VHI, XLO, XHI); } while (0)
    // (lanes per query vertex by the size of the triangle set it scans: many triangles per cell on refined cloths)
    if (tiled) {
      if (v_hi_all - v_lo_all > body.v_end - body.v_start) TSL_PROJ_BY_SIZE(false, v_lo_all, v_hi_all, body.v_start, body.v_end);
    } else {
      for (int b2 = 0; b2 < c->n_body; b2++) {
        if (b2 == b) continue;
        const tsl_body& q = c->h_bodies[b2];
        if (q.v_end - q.v_start <= 0) continue;
        TSL_PROJ_BY_SIZE(false, q.v_start, q.v_end, 0, 0);
      }
    }
    if (b < (int)c->self_contact.size() && c->self_contact[b]) {   // geometry_self.projection_query (geometry_self.py:290-297)
      if (body.v_end - body.v_start > 0) TSL_PROJ_BY_SIZE(true, body.v_start, body.v_end, 0, 0);
    }
#undef TSL_PROJ_BY_SIZE
#undef TSL_PROJ_LAUNCH
  }
  if (multi) {
    HIP_OK(hipEventRecord(c->ev_join, c->side)); HIP_OK(hipEventRecord(c->ev_join2, c->side2));
    HIP_OK(hipStreamWaitEvent(c->stream, c->ev_join, 0)); HIP_OK(hipStreamWaitEvent(c->stream, c->ev_join2, 0));
  }
  // contact_analysis: flags of every pair's query vertices, one exclusive scan, then the constraints at their slots (fixed list order)
  long Q = 0;
  for (const auto& pr : c->h_pairs) Q += std::max(0, pr.v_end - pr.v_start);
  if (c->cq_flag.n < (size_t)Q + 1) { if (c->cq_flag.alloc((size_t)Q + 1) | c->cq_scan.alloc((size_t)Q + 1)) return -1; }
  if (c->scan_tmp.n < (size_t)(Q + 1) / SCAN_TILE + 2) { if (c->scan_tmp.alloc((size_t)std::max<long>(std::max<long>(c->grid_buckets_max, NV + 1), Q + 1) / SCAN_TILE + 2)) return -1; }
  HIP_OK(hipMemsetAsync(c->cq_flag.p + Q, 0, sizeof(int), s));
  for (int phase = 0; phase < 2; phase++) {
    long qoff = 0;
    for (const auto& pr : c->h_pairs) {
      const int nq = pr.v_end - pr.v_start;
      if (nq <= 0) continue;
      // parameter-driven pairs may carry a factor in mu (Scene_card.py:122-126: mu_cloth_elastic * 10 for the upper cards)
      const double live = pr.mu_is_param == 2 ? c->mu_cloth_cloth : c->mu_cloth_elastic;  // Scene_sliding.py:80 has a second live parameter
      const double mu = pr.mu_is_param ? live * (pr.mu > 0 ? pr.mu : 1.0) : pr.mu;
      hipLaunchKernelGGL(k_contact_pair, dim3(cnblk(nq, 128)), dim3(128), 0, s, pr.b_idx, pr.v_start, pr.v_end, mu, NV, c->max_n_constraints, c->k_contact, c->eps_contact, pos,
                         prev, c->proj_flag.p, c->proj_dir.p, c->proj_idx.p, c->proj_w.p, c->nc_dev.p, c->c_idx.p, c->c_w.p, c->c_k.p, c->c_mu.p, c->c_dx0.p, c->c_T.p, c->c_n.p, pr.mu_is_param,
                         c->c_kind.p, phase, (int)qoff, c->cq_flag.p, (const int*)c->cq_scan.p);
      qoff += nq;
    }
    if (phase == 0 && Q > 0) scan_exclusive(s, (int)Q + 1, c->cq_flag.p, c->cq_scan.p, c->scan_tmp.p);
  }
  int nc = 0, big_bucket = 0;
  if (Q > 0) HIP_OK(hipMemcpyAsync(&nc, c->cq_scan.p + Q, sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_OK(hipMemcpyAsync(&big_bucket, c->nc_dev.p + 1, sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  HIP_OK(hipGetLastError());
  if (big_bucket > 0) {
    c->nc = 0;
    return tsl_fail("contact detection: %d triangles of one body fall into one broad-phase cell (more than %d are not ranked): the body lies outside grid_extent = %g m "
                    "(centroids are clamped into the boundary cells) or grid_h = %g m is far larger than its triangles", big_bucket, TSL_BUCKET_CAP, c->grid_extent, c->grid_h);
  }
  if (nc > c->max_n_constraints) {
    // more constraints than the scene's cap (the reference would drop the surplus in atomic-append order): an error
    c->nc = 0;
    return tsl_fail("contact detection: %d active constraints exceed max_n_constraints = %d (raise the scene's max_n_constraints)", nc, c->max_n_constraints);
  }
  c->nc = nc;
  if (nc_host) *nc_host = c->nc;
  if (c->nc > 0) {
    const int n1 = NV + 1;
    if (c->cr_ptr.n == 0) {
      if (c->cr_ptr.alloc(n1) | c->cr_cnt.alloc(n1) | c->cr_fill.alloc(n1) | c->cr_ent.alloc(4 * (size_t)c->max_n_constraints) | c->cr_rows.alloc(4 * (size_t)c->max_n_constraints) | c->cr_tmp.alloc(4 * (size_t)c->max_n_constraints)) return -1;
    }
    HIP_OK(hipMemsetAsync(c->cr_cnt.p, 0, n1 * sizeof(int), s));
    HIP_OK(hipMemsetAsync(c->cr_fill.p, 0, n1 * sizeof(int), s));
    hipLaunchKernelGGL(k_cr_count, dim3(cnblk(4 * (long)c->nc, 256)), dim3(256), 0, s, c->nc, c->c_idx.p, c->rowpos.p, c->cr_cnt.p);
    scan_exclusive(s, n1, c->cr_cnt.p, c->cr_ptr.p, c->scan_tmp.p);
    hipLaunchKernelGGL(k_cr_fill, dim3(cnblk(4 * (long)c->nc, 256)), dim3(256), 0, s, c->nc, c->c_idx.p, c->rowpos.p, (const int*)c->cr_ptr.p, c->cr_fill.p, c->cr_tmp.p);
    hipLaunchKernelGGL(k_cr_rank, dim3(cnblk(4 * (long)c->nc, 256)), dim3(256), 0, s, c->nc, c->c_idx.p, c->rowpos.p, (const int*)c->cr_ptr.p, (const int*)c->cr_tmp.p, c->cr_ent.p, c->cr_rows.p);
    HIP_OK(hipGetLastError());
  }
  return 0;
}

static bool direct_takes_solve(tsl_ctx* c);
static void contact_diag_refresh(tsl_ctx* c, hipStream_t s) {
  hipLaunchKernelGGL(k_contact_diag, dim3(cnblk((long)c->NV * 64, 256)), dim3(256), 0, s, c->NV, (const int*)c->cr_ptr.p, (const int*)c->cr_ent.p, (const double*)c->c_H.p, c->c_diag.p);
  c->cdiag_valid = true;
}
static int contact_assemble(tsl_ctx* c, const double* pos, int spd, double* grad, hipStream_t s) {
  if (c->nc <= 0) return 0;
  ContactArgs A;
  A.idx = c->c_idx.p; A.w = c->c_w.p; A.n = c->c_n.p; A.dx0 = c->c_dx0.p; A.k = c->c_k.p; A.mu = c->c_mu.p; A.T = c->c_T.p;
  A.k_contact = c->k_contact; A.eps_contact = c->eps_contact; A.eps_vh = c->eps_v * c->dt;
  if (grad && c->c_G.n < 12 * (size_t)c->max_n_constraints) { if (c->c_G.alloc(12 * (size_t)c->max_n_constraints)) return -1; }
  double* cg = grad ? c->c_G.p : (double*)nullptr;   // per-constraint gradients, summed per vertex by k_contact_row_gather
  hipLaunchKernelGGL(k_contact_assemble_coop, dim3(cnblk((long)c->nc * 16, 256)), dim3(256), 0, s, c->nc, A, pos, spd, c->c_Hfull.p, cg);
  hipLaunchKernelGGL(k_contact_mask, dim3(cnblk((long)c->nc * 144, 256)), dim3(256), 0, s, c->nc, c->c_idx.p, c->frozen.p, c->c_Hfull.p, c->c_H.p);
  // the diagonal 3 x 3 blocks of the contact terms feed the block-Jacobi inverse and the hierarchy's Galerkin diagonal only: a solve that goes to the
  // factorisation never reads them (45 us on the longest chain of an assembly); block_jacobi_refresh forms them when the hierarchy runs after all
  if (direct_takes_solve(c)) c->cdiag_valid = false;
  else contact_diag_refresh(c, s);
  return 0;
}

extern "C" int tsl_constraints_export(tsl_ctx* c, int32_t* idx, double* w, double* k, double* dx0, double* T, double* n, double* mu, int32_t max_n) {
  Scope scope(c);
  HIP_OK(hipStreamSynchronize(c->stream));
  const int m = std::min(c->nc, (int)max_n);
  if (m <= 0) return 0;
  HIP_OK(hipMemcpy(idx, c->c_idx.p, (size_t)m * 4 * sizeof(int), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(w, c->c_w.p, (size_t)m * 3 * sizeof(double), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(k, c->c_k.p, (size_t)m * sizeof(double), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(dx0, c->c_dx0.p, (size_t)m * 3 * sizeof(double), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(T, c->c_T.p, (size_t)m * 6 * sizeof(double), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(n, c->c_n.p, (size_t)m * 3 * sizeof(double), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(mu, c->c_mu.p, (size_t)m * sizeof(double), hipMemcpyDeviceToHost));
  return m;
}

extern "C" int tsl_contact_blocks_export(tsl_ctx* c, double* blocks_host, int32_t max_n, int32_t masked) {
  Scope scope(c);
  HIP_OK(hipStreamSynchronize(c->stream));
  const int m = std::min(c->nc, (int)max_n);
  if (m <= 0) return 0;
  HIP_OK(hipMemcpy(blocks_host, masked ? c->c_H.p : c->c_Hfull.p, (size_t)m * 144 * sizeof(double), hipMemcpyDeviceToHost));
  return m;
}

extern "C" int tsl_proj_export(tsl_ctx* c, int32_t* flag, int32_t* dir, int32_t* pidx, double* pw) {
  Scope scope(c);
  HIP_OK(hipStreamSynchronize(c->stream));
  const size_t n = (size_t)std::max(c->n_body, 1) * c->NV;
  HIP_OK(hipMemcpy(flag, c->proj_flag.p, n * sizeof(int), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(dir, c->proj_dir.p, n * sizeof(int), hipMemcpyDeviceToHost));
  if (pidx) HIP_OK(hipMemcpy(pidx, c->proj_idx.p, n * 3 * sizeof(int), hipMemcpyDeviceToHost));
  if (pw) HIP_OK(hipMemcpy(pw, c->proj_w.p, n * 3 * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

// BaseScene.border_flag (BaseScene.py:104; written only by Scene_balancing.load_all :213-222, read by project_pair geometry.py:194-201)
extern "C" int tsl_set_border(tsl_ctx* c, const int32_t* border_host) {
  Scope scope(c);
  (void)hipStreamSynchronize(c->stream);
  HIP_OK(hipMemcpy(c->border.p, border_host, (size_t)c->NV * sizeof(int), hipMemcpyHostToDevice));
  return 0;
}

extern "C" int tsl_proj_import(tsl_ctx* c, const int32_t* flag, const int32_t* dir) {
  Scope scope(c);
  (void)hipStreamSynchronize(c->stream);
  const size_t n = (size_t)std::max(c->n_body, 1) * c->NV;
  HIP_OK(hipMemcpy(c->proj_flag.p, flag, n * sizeof(int), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(c->proj_dir.p, dir, n * sizeof(int), hipMemcpyHostToDevice));
  return 0;
}
