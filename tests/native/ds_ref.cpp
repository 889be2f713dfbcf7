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
  std::vector<double> A((size_t)P.arena, 0.0);
  for (int q = 0; q < row_ptr[NV]; q++)
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) A[P.blk_dst[q] + (long long)r * P.blk_ld[q] + c] += vals[(size_t)q * 9 + 3 * r + c];
  for (int e = 0; e < n_cons; e++)
    for (int a = 0; a < 4; a++)
      for (int b = 0; b < 4; b++)
        for (int r = 0; r < 3; r++)
          for (int c = 0; c < 3; c++) A[P.con_dst[(size_t)e * 16 + 4 * a + b] + (long long)r * P.con_ld[(size_t)e * 16 + 4 * a + b] + c] += conH[(size_t)e * 144 + (3 * a + r) * 12 + 3 * b + c];
  const int S = P.sym.n_sn;
  for (int s = 0; s < S; s++) { const DsFrontDesc& f = P.fr[s]; for (int i = f.p; i < f.pp; i++) A[f.off + (long long)i * f.ld + i] = 1.0; }
  double min_piv = 1e300;
  long n_viol = 0, n_store = 0, n_add = 0;
  for (int l = 0; l < P.n_levels; l++)
    for (int q = P.level_ptr[l]; q < P.level_ptr[l + 1]; q++) {
      const int s = P.level_sn[q];
      const DsFrontDesc& f = P.fr[s];
      double* F = A.data() + f.off;
      const int ld = f.ld;
      // in-place Gauss-Jordan on rows 0..pp of [F11 | F12]
      for (int k = 0; k < f.pp; k++) {
        const double piv = F[(size_t)k * ld + k];
        min_piv = std::min(min_piv, std::fabs(piv));
        const double ip = 1.0 / piv;
        for (int j = 0; j < ld; j++) F[(size_t)k * ld + j] *= ip;
        F[(size_t)k * ld + k] = ip;
        for (int i = 0; i < f.pp; i++) {
          if (i == k) continue;
          const double c = F[(size_t)i * ld + k];
          if (c == 0.0) continue;
          F[(size_t)i * ld + k] = 0.0;
          for (int j = 0; j < ld; j++) F[(size_t)i * ld + j] -= c * F[(size_t)k * ld + j];
        }
      }
      // S = F22 - F21 G
      for (int i = f.pp; i < f.pp + f.b; i++)
        for (int k = 0; k < f.p; k++) {
          const double c = F[(size_t)i * ld + k];
          if (c == 0.0) continue;
          for (int j = f.pp; j < f.pp + f.b; j++) F[(size_t)i * ld + j] -= c * F[(size_t)k * ld + j];
        }
      if (f.parent >= 0) {
        const DsFrontDesc& pf = P.fr[f.parent];
        double* PF = A.data() + pf.off;
        // the extend-add with the semantics of the kernel's epilogue: entries marked single-writer are STORED (a wrong mark loses a
        // contribution and the solve below goes wrong); a non-zero value underneath such an entry is counted as a violation
        for (int iv = 0; iv < f.nv_bnd; iv++)
          for (int jv = 0; jv < f.nv_bnd; jv++) {
            const int ri = P.rel[f.rel_off + iv], rj = P.rel[f.rel_off + jv];
            const int pi = ri & DS_REL_MASK, pj = rj & DS_REL_MASK;
            const bool store = pi >= pf.pp && pj >= pf.pp && ((ri | rj) & DS_REL_EXCL);
            for (int r = 0; r < 3; r++)
              for (int c = 0; c < 3; c++) {
                double& dst = PF[(size_t)(pi + r) * pf.ld + pj + c];
                const double v = F[(size_t)(f.pp + 3 * iv + r) * ld + f.pp + 3 * jv + c];
                if (store) { if (dst != 0.0) n_viol++; dst = v; n_store++; } else { dst += v; n_add++; }
              }
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
        for (int j = 0; j < f.p; j++) acc += F[(size_t)(f.pp + i) * f.ld + j] * t[3 * (size_t)vt[j / 3] + j % 3];
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
  if (stats) { stats[0] = S; stats[1] = P.n_levels; stats[2] = (double)P.arena; stats[3] = P.flops; stats[4] = min_piv; stats[5] = (double)n_viol; stats[6] = (double)n_store; stats[7] = (double)n_add; }
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
    if (verbose) printf("level %2d: %5d fronts  max pp %4d  max ld %4d  max bp %4d\n", b.level, b.count, b.max_pp, b.max_ld, b.max_bp);
  }
  double n_store = 0, n_add = 0;   // Schur-complement entries that reach their parent by a plain store / by an atomic add
  for (const DsFrontDesc& f : P.fr) {
    if (f.parent < 0) continue;
    const DsFrontDesc& pf = P.fr[f.parent];
    long ex = 0, bnd = 0;
    for (int iv = 0; iv < f.nv_bnd; iv++) { const int r = P.rel[f.rel_off + iv]; if ((r & DS_REL_MASK) >= pf.pp) { bnd++; if (r & DS_REL_EXCL) ex++; } }
    const double st = 9.0 * ((double)bnd * bnd - (double)(bnd - ex) * (bnd - ex));
    n_store += st; n_add += 9.0 * (double)f.nv_bnd * f.nv_bnd - st;
  }
  out[7] = n_store / std::max(n_store + n_add, 1.0);
  out[0] = P.sym.n_sn; out[1] = P.n_levels; out[2] = (double)P.batches.size(); out[3] = steps; out[4] = P.flops; out[5] = (double)P.arena * 8; out[6] = solve_bytes;
  return 0;
}
