// Test infrastructure (tests/ only): plain-loop CPU execution of the multifrontal plan of thinshelllab_amd/csrc/direct_plan.hpp.
// It checks the HOST logic of the product (ordering, elimination tree, front layout, index maps) without a GPU by running the
// same algorithm the kernels of k_direct.hpp run -- Gauss-Jordan on [F11 | F12] without pivoting, Schur complement, extend-add,
// forward / backward sweeps -- and comparing with scipy.  Never linked into libtsl_hip.so.
#include <cmath>
#include <cstring>
#include "../../thinshelllab_amd/csrc/direct_plan.hpp"

extern "C" int dsref_solve(int NV, const int* row_ptr, const int* col, const double* vals, int n_grids, const int* grids, int n_blocks, const int* blocks,
                           int n_cons, const int* cons, const double* conH, int leaf, const double* rhs, double* x, double* stats) {
  std::vector<std::vector<int>> adj(NV);
  for (int r = 0; r < NV; r++) adj[r].assign(col + row_ptr[r], col + row_ptr[r + 1]);
  std::vector<int> rp(row_ptr, row_ptr + NV + 1);
  std::vector<DsGrid> G; std::vector<DsBlock> B;
  for (int i = 0; i < n_grids; i++) G.push_back({grids[3 * i], grids[3 * i + 1], grids[3 * i + 2]});
  for (int i = 0; i < n_blocks; i++) B.push_back({blocks[2 * i], blocks[2 * i + 1]});
  DirectPlan P;
  P.sym.build_partition(NV, adj, G, B, leaf);
  const int rc = P.build(adj, rp, cons, n_cons);
  if (rc) return rc;
  // Panel arena (top rows + F21 of every front) and Schur arena, as the GPU path treats them: only the panels of the LEAF level (the
  // head of the panel arena) are cleared; everything else starts as NaN -- the panels of a higher level must be WRITTEN by the gather
  // of the children's Schur complements before the level's matrix entries are added, and every Schur entry the gather reads must
  // have been stored by the child before.
  std::vector<double> A((size_t)P.arena, std::nan("")), SA((size_t)P.sarena, std::nan(""));
  std::fill(A.begin(), A.begin() + P.arena_leaf, 0.0);
  const int S = P.sym.n_sn;
  double min_piv = 1e300;
  long n_gather = 0, n_level_viol = 0;
  // sum over the children (fixed order) of the entries of their Schur complements that land on (i, j) of front f, local dofs
  auto gather = [&](const DsFrontDesc& f, int i, int j) {
    double v = 0.0;
    for (int q = f.ch_off; q < f.ch_off + f.nchild; q++) {
      const DsChildRec& c = P.ch_rec[q];
      const int ci = P.pmap[c.pmap_off + i], cj = P.pmap[c.pmap_off + j];
      if (ci >= 0 && cj >= 0) { v += SA[c.soff + (long long)ci * c.bp + cj]; n_gather++; }
    }
    return v;
  };
  for (int l = 0; l < P.n_levels; l++) {
    // start of the level: panels written from the children (k_ds_extend_panels; level 0 was cleared), then the level's matrix entries,
    // contact blocks and the identity on the padding of the pivot blocks (k_ds_assemble_level)
    for (int q = P.level_ptr[l]; q < P.level_ptr[l + 1]; q++) {
      const DsFrontDesc& f = P.fr[P.level_sn[q]];
      if (l == 0) { if (f.off21 + (long long)f.bp * f.pp > P.arena_leaf) n_level_viol++; continue; }
      if (f.off < P.arena_leaf) n_level_viol++;
      for (int i = 0; i < f.pp; i++) for (int j = 0; j < f.ld; j++) A[f.off + (size_t)i * f.ld + j] = gather(f, i, j);
      for (int i = 0; i < f.bp; i++) for (int j = 0; j < f.pp; j++) A[f.off21 + (size_t)i * f.pp + j] = gather(f, f.pp + i, j);
    }
    for (int i = P.blk_lptr[l]; i < P.blk_lptr[l + 1]; i++) {
      const int q = P.blk_q[i];
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) A[P.blk_dst[q] + (long long)r * P.blk_ld[q] + c] += vals[(size_t)q * 9 + 3 * r + c];
    }
    for (int e = 0; e < n_cons; e++)
      for (int a = 0; a < 4; a++)
        for (int b = 0; b < 4; b++) {
          if ((P.con_lvl[(size_t)e * 16 + 4 * a + b] >> 1) != l) continue;
          for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) A[P.con_dst[(size_t)e * 16 + 4 * a + b] + (long long)r * P.con_ld[(size_t)e * 16 + 4 * a + b] + c] += conH[(size_t)e * 144 + (3 * a + r) * 12 + 3 * b + c];
        }
    for (int q = P.level_ptr[l]; q < P.level_ptr[l + 1]; q++) { const DsFrontDesc& f = P.fr[P.level_sn[q]]; for (int i = f.p; i < f.pp; i++) A[f.off + (long long)i * f.ld + i] = 1.0; }
    for (int q = P.level_ptr[l]; q < P.level_ptr[l + 1]; q++) {
      const int s = P.level_sn[q];
      const DsFrontDesc& f = P.fr[s];
      for (int c = f.ch_off; c < f.ch_off + f.nchild; c++) if (P.sym.level[P.ch_rec[c].sn] >= l) n_level_viol++;
      double* F = A.data() + f.off;     // top rows, stride ld
      double* F21 = A.data() + f.off21; // stride pp
      double* Sf = SA.data() + f.soff;  // stride bp
      const int ld = f.ld, pp = f.pp;
      // in-place Gauss-Jordan on rows 0..pp of [F11 | F12]
      for (int k = 0; k < pp; k++) {
        const double piv = F[(size_t)k * ld + k];
        min_piv = std::min(min_piv, std::fabs(piv));
        const double ip = 1.0 / piv;
        for (int j = 0; j < ld; j++) F[(size_t)k * ld + j] *= ip;
        F[(size_t)k * ld + k] = ip;
        for (int i = 0; i < pp; i++) {
          if (i == k) continue;
          const double c = F[(size_t)i * ld + k];
          if (c == 0.0) continue;
          F[(size_t)i * ld + k] = 0.0;
          for (int j = 0; j < ld; j++) F[(size_t)i * ld + j] -= c * F[(size_t)k * ld + j];
        }
      }
      // S = sum_children ext(S_child) - F21 G, stored (the Schur GEMM's epilogue)
      for (int i = 0; i < f.b; i++)
        for (int j = 0; j < f.b; j++) {
          double acc = 0.0;
          for (int k = 0; k < f.p; k++) acc += F21[(size_t)i * pp + k] * F[(size_t)k * ld + pp + j];
          Sf[(size_t)i * f.bp + j] = gather(f, pp + i, pp + j) - acc;
        }
    }
  }
  // solve
  std::vector<double> w(rhs, rhs + 3 * (size_t)NV), t(3 * (size_t)NV, 0.0);
  for (int l = 0; l < P.n_levels; l++)
    for (int q = P.level_ptr[l]; q < P.level_ptr[l + 1]; q++) {
      const int s = P.level_sn[q];
      const DsFrontDesc& f = P.fr[s];
      const double* F = A.data() + f.off;
      const int* vt = P.vtx.data() + f.vtx_off;
      for (int i = 0; i < f.p; i++) {
        double acc = 0;
        for (int j = 0; j < f.p; j++) acc += F[(size_t)i * f.ld + j] * w[3 * (size_t)vt[j / 3] + j % 3];
        t[3 * (size_t)vt[i / 3] + i % 3] = acc;
      }
      for (int i = 0; i < f.b; i++) {
        double acc = 0;
        for (int j = 0; j < f.p; j++) acc += A[f.off21 + (size_t)i * f.pp + j] * t[3 * (size_t)vt[j / 3] + j % 3];
        w[3 * (size_t)vt[f.nv_own + i / 3] + i % 3] -= acc;
      }
    }
  for (int l = P.n_levels - 1; l >= 0; l--)
    for (int q = P.level_ptr[l]; q < P.level_ptr[l + 1]; q++) {
      const int s = P.level_sn[q];
      const DsFrontDesc& f = P.fr[s];
      const double* F = A.data() + f.off;
      const int* vt = P.vtx.data() + f.vtx_off;
      for (int i = 0; i < f.p; i++) {
        double acc = t[3 * (size_t)vt[i / 3] + i % 3];
        for (int j = 0; j < f.b; j++) acc -= F[(size_t)i * f.ld + f.pp + j] * x[3 * (size_t)vt[f.nv_own + j / 3] + j % 3];
        x[3 * (size_t)vt[i / 3] + i % 3] = acc;
      }
    }
  if (stats) { stats[0] = S; stats[1] = P.n_levels; stats[2] = (double)P.arena; stats[3] = P.flops; stats[4] = min_piv; stats[5] = (double)n_level_viol; stats[6] = (double)n_gather; stats[7] = (double)P.sarena; }
  return 0;
}

// plan statistics for a pattern + constraint set (scripts / tests): prints the batches of the factorisation
extern "C" int dsref_plan_stats(int NV, const int* row_ptr, const int* col, int n_grids, const int* grids, int n_blocks, const int* blocks, int n_cons, const int* cons, int leaf,
                                int verbose, double* out) {
  std::vector<std::vector<int>> adj(NV);
  for (int r = 0; r < NV; r++) adj[r].assign(col + row_ptr[r], col + row_ptr[r + 1]);
  std::vector<int> rp(row_ptr, row_ptr + NV + 1);
  std::vector<DsGrid> G; std::vector<DsBlock> B;
  for (int i = 0; i < n_grids; i++) G.push_back({grids[3 * i], grids[3 * i + 1], grids[3 * i + 2]});
  for (int i = 0; i < n_blocks; i++) B.push_back({blocks[2 * i], blocks[2 * i + 1]});
  DirectPlan P;
  P.sym.build_partition(NV, adj, G, B, leaf);
  const int rc = P.build(adj, rp, cons, n_cons);
  if (rc) return rc;
  int steps = 0;
  double solve_bytes = 0;
  for (const DsFrontDesc& f : P.fr) solve_bytes += 8.0 * ((double)f.p * f.p + 2.0 * f.p * f.b);
  for (const DsBatch& b : P.batches) {
    steps += b.max_pp / DS_T;
    if (verbose) {
      printf("level %2d: %5d fronts  max pp %4d  max ld %4d  max bp %4d", b.level, b.count, b.max_pp, b.max_ld, b.max_bp);
      if (b.count <= 8) for (int q = 0; q < b.count; q++) { const DsFrontDesc& f = P.fr[P.level_sn[b.first + q]]; printf("  [p %d b %d lead %d]", f.p, f.b, f.lead); }
      printf("\n");
    }
  }
  double n_ent = 0;   // Schur-complement entries stored per factorisation
  for (const DsFrontDesc& f : P.fr) n_ent += (double)f.b * f.b;
  out[7] = n_ent;
  out[0] = P.sym.n_sn; out[1] = P.n_levels; out[2] = (double)P.batches.size(); out[3] = steps; out[4] = P.flops; out[5] = (double)P.arena * 8; out[6] = solve_bytes;
  return 0;
}

// The tables of the look-ahead (DirectPlan: DsFrontDesc.lead, blk_lmid, cgr_lmid, la_from), checked against their definitions with plain loops.
// out: {fronts with a parent, violations of the lead rule, blocks listed, blocks on the wrong side of blk_lmid, contact groups, groups on the wrong side of cgr_lmid,
//       la_from, levels, levels >= la_from that have more than one batch}
extern "C" int dsref_check_lookahead(int NV, const int* row_ptr, const int* col, int n_grids, const int* grids, int n_blocks, const int* blocks, int n_cons, const int* cons, int leaf,
                                     double* out) {
  std::vector<std::vector<int>> adj(NV);
  for (int r = 0; r < NV; r++) adj[r].assign(col + row_ptr[r], col + row_ptr[r + 1]);
  std::vector<int> rp(row_ptr, row_ptr + NV + 1);
  std::vector<DsGrid> G; std::vector<DsBlock> B;
  for (int i = 0; i < n_grids; i++) G.push_back({grids[3 * i], grids[3 * i + 1], grids[3 * i + 2]});
  for (int i = 0; i < n_blocks; i++) B.push_back({blocks[2 * i], blocks[2 * i + 1]});
  DirectPlan P;
  P.sym.build_partition(NV, adj, G, B, leaf);
  const int rc = P.build(adj, rp, cons, n_cons);
  if (rc) return rc;
  long with_parent = 0, lead_viol = 0, nblk = 0, blk_viol = 0, ngrp = 0, grp_viol = 0, multi = 0;
  // lead: the child's boundary dofs that are own dofs of the parent are exactly its FIRST `lead` boundary dofs (the table over the parent's dofs is monotone)
  for (int s = 0; s < P.sym.n_sn; s++) {
    const DsFrontDesc& f = P.fr[s];
    if (f.parent < 0) continue;
    with_parent++;
    const DsFrontDesc& pf = P.fr[f.parent];
    std::vector<int> where(f.b, -1);   // boundary dof of the child -> local dof of the parent
    for (int d = 0; d < pf.ld; d++) { const int cb = P.pmap[f.pmap_off + d]; if (cb >= 0) { if (cb >= f.b || where[cb] >= 0) lead_viol++; else where[cb] = d; } }
    int n_own = 0;
    for (int i = 0; i < f.b; i++) {
      if (where[i] < 0) { lead_viol++; continue; }
      const bool own = where[i] < pf.pp;
      if (own) n_own++;
      if (own != (i < f.lead)) lead_viol++;
    }
    if (n_own != f.lead) lead_viol++;
  }
  // block lists: [blk_lptr[l], blk_lmid[l]) lie inside the pivot block F11 of a front of level l, [blk_lmid[l], blk_lptr[l + 1]) in F12 or F21 of one
  auto region_of = [&](long long dst, int ld, int l) -> int {   // 0 F11, 1 F12 / F21, -1 nowhere on level l
    for (int q = P.level_ptr[l]; q < P.level_ptr[l + 1]; q++) {
      const DsFrontDesc& f = P.fr[P.level_sn[q]];
      if (dst >= f.off && dst < f.off + (long long)f.pp * f.ld) { if (ld != f.ld) return -1; return (dst - f.off) % f.ld < f.pp ? 0 : 1; }
      if (dst >= f.off21 && dst < f.off21 + (long long)f.bp * f.pp) return ld == f.pp ? 1 : -1;
    }
    return -1;
  };
  for (int l = 0; l < P.n_levels; l++) {
    if (P.blk_lmid[l] < P.blk_lptr[l] || P.blk_lmid[l] > P.blk_lptr[l + 1]) blk_viol++;
    for (int k = P.blk_lptr[l]; k < P.blk_lptr[l + 1]; k++) {
      const int q = P.blk_q[k];
      nblk++;
      if (region_of(P.blk_dst[q], P.blk_ld[q], l) != (k < P.blk_lmid[l] ? 0 : 1)) blk_viol++;
    }
    if (n_cons > 0) {
      if (P.cgr_lmid[l] < P.cgr_lptr[l] || P.cgr_lmid[l] > P.cgr_lptr[l + 1]) grp_viol++;
      for (int g = P.cgr_lptr[l]; g < P.cgr_lptr[l + 1]; g++) { ngrp++; if (region_of(P.cgr_dst[g], P.cgr_ld[g], l) != (g < P.cgr_lmid[l] ? 0 : 1)) grp_viol++; }
    }
  }
  std::vector<int> nb(P.n_levels, 0);
  for (const DsBatch& b : P.batches) nb[b.level]++;
  for (int l = 0; l < P.n_levels; l++) if (l >= P.la_from && nb[l] != 1) multi++;
  out[0] = (double)with_parent; out[1] = (double)lead_viol; out[2] = (double)nblk; out[3] = (double)blk_viol; out[4] = (double)ngrp; out[5] = (double)grp_viol;
  out[6] = P.la_from < P.n_levels ? (double)P.la_from : -1.0; out[7] = (double)P.n_levels; out[8] = (double)multi;
  return 0;
}
