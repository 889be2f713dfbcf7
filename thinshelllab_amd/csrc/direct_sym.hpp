// Symbolic part of the sparse direct factorisation that preconditions the Newton / adjoint solves (host only, no HIP).
//
// The reference solves every system with a direct sparse solver (cupyx spsolve, /root/reference/code/engine/sparse_solver.py:85-105).
// Here the Krylov solvers of tsl_solve are preconditioned by a multifrontal LU of the same operator:
//   * cloth vertices are ordered by geometric nested dissection of their (N+1) x (M+1) grid (separators two grid lines wide, then
//     trimmed against the real adjacency: the hinge stencil reaches two lines only from every second vertex),
//   * every FEM body is one dense supernode ordered before the cloth (its contacts make the touched cloth vertices its boundary),
//   * the contact constraints of the current step add cliques to the adjacency, so boundaries, the elimination tree and the
//     front layout are recomputed whenever the constraint set changes (a few ms of host work per time step).
// Granularity: vertices (3 x 3 blocks).  A front of supernode s holds its own vertices followed by its boundary vertices
// (later-eliminated vertices coupled to it directly or through descendants), sorted by elimination position.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

struct DsGrid { int v_offset, N, M; };      // vertex id = v_offset + i * (M + 1) + j, 0 <= i <= N, 0 <= j <= M
struct DsBlock { int v_offset, n_verts; };  // a dense body

struct DirectSym {
  int NV = 0;
  // ---- static partition
  std::vector<int> order, epos, sn_of, sn_ptr;  // position -> vertex, vertex -> position, vertex -> supernode, supernode -> [first, last) position
  int n_sn = 0;
  // ---- per constraint set
  std::vector<std::vector<int>> bnd;  // boundary vertices per supernode, sorted by elimination position
  std::vector<int> parent, level;
  std::vector<std::vector<int>> by_level;
  int n_levels = 0;

  int own(int s) const { return sn_ptr[s + 1] - sn_ptr[s]; }

  // position of vertex v inside the front of supernode s in vertices (own vertices first, then the boundary); -1 if absent
  int local(int s, int v) const {
    if (sn_of[v] == s) return epos[v] - sn_ptr[s];
    const auto& b = bnd[s];
    const int e = epos[v];
    int lo = 0, hi = (int)b.size();
    while (lo < hi) { const int m = (lo + hi) >> 1; if (epos[b[m]] < e) lo = m + 1; else hi = m; }
    if (lo < (int)b.size() && b[lo] == v) return own(s) + lo;
    return -1;
  }

  // ------------------------------------------------------------------------------------------ static: ordering
  // adj: sorted adjacency of every vertex (may include the vertex itself)
  void build_partition(int nv, const std::vector<std::vector<int>>& adj, const std::vector<DsGrid>& grids, const std::vector<DsBlock>& blocks, int leaf_verts) {
    NV = nv;
    order.clear(); sn_ptr.assign(1, 0);
    std::vector<char> placed(nv, 0);
    auto close_sn = [&]() { if ((int)order.size() > sn_ptr.back()) sn_ptr.push_back((int)order.size()); };
    for (const DsBlock& b : blocks) {
      for (int v = b.v_offset; v < b.v_offset + b.n_verts; v++) { order.push_back(v); placed[v] = 1; }
      close_sn();
    }
    std::vector<int> side(nv, 0);  // scratch of the bisection: 1 left, 2 right, 3 separator
    for (const DsGrid& g : grids) {
      std::vector<int> region((size_t)(g.N + 1) * (g.M + 1));
      for (size_t k = 0; k < region.size(); k++) { region[k] = g.v_offset + (int)k; placed[region[k]] = 1; }
      dissect(g, adj, region, leaf_verts, side, close_sn);
    }
    for (int v = 0; v < nv; v++)
      if (!placed[v]) { order.push_back(v); close_sn(); }
    n_sn = (int)sn_ptr.size() - 1;
    epos.assign(nv, 0); sn_of.assign(nv, 0);
    for (int p = 0; p < nv; p++) epos[order[p]] = p;
    for (int s = 0; s < n_sn; s++)
      for (int p = sn_ptr[s]; p < sn_ptr[s + 1]; p++) sn_of[order[p]] = s;
  }

  template <class F>
  void dissect(const DsGrid& g, const std::vector<std::vector<int>>& adj, std::vector<int>& region, int leaf_verts, std::vector<int>& side, F&& close_sn) {
    if (region.empty()) return;
    const int W = g.M + 1;
    int i0 = 1 << 30, i1 = -1, j0 = 1 << 30, j1 = -1;
    for (int v : region) { const int i = (v - g.v_offset) / W, j = (v - g.v_offset) % W; i0 = std::min(i0, i); i1 = std::max(i1, i); j0 = std::min(j0, j); j1 = std::max(j1, j); }
    const int ni = i1 - i0 + 1, nj = j1 - j0 + 1;
    if ((int)region.size() <= leaf_verts || (ni <= 3 && nj <= 3)) {
      std::sort(region.begin(), region.end());
      for (int v : region) order.push_back(v);
      close_sn();
      return;
    }
    const bool along_i = ni >= nj;
    const int m = along_i ? (i0 + i1) / 2 : (j0 + j1) / 2;  // separator lines m, m + 1
    auto coord = [&](int v) { const int k = v - g.v_offset; return along_i ? k / W : k % W; };
    auto other = [&](int v) { const int k = v - g.v_offset; return along_i ? k % W : k / W; };
    std::vector<int> L, R, S;
    for (int v : region) { const int cv = coord(v); side[v] = cv < m ? 1 : cv > m + 1 ? 2 : 3; }
    // trim: a separator vertex without neighbour on one side joins the other side (second line first, so that it can leave)
    std::vector<int> sep;
    for (int v : region) if (side[v] == 3) sep.push_back(v);
    std::stable_sort(sep.begin(), sep.end(), [&](int a, int b) { return coord(a) > coord(b); });
    for (int v : sep) {
      bool hasL = false, hasR = false;
      for (int u : adj[v]) { if (side[u] == 1) hasL = true; else if (side[u] == 2) hasR = true; }
      if (!hasL) side[v] = 2; else if (!hasR) side[v] = 1;
    }
    for (int v : region) { if (side[v] == 1) L.push_back(v); else if (side[v] == 2) R.push_back(v); else S.push_back(v); }
    for (int v : region) side[v] = 0;
    std::vector<int>().swap(region);
    dissect(g, adj, L, leaf_verts, side, close_sn);
    dissect(g, adj, R, leaf_verts, side, close_sn);
    std::sort(S.begin(), S.end(), [&](int a, int b) { return other(a) != other(b) ? other(a) < other(b) : coord(a) < coord(b); });
    for (int v : S) order.push_back(v);
    close_sn();
  }

  // ------------------------------------------------------------------------------------------ per constraint set: boundaries and tree
  // extra: additional cliques (the 4 vertices of every contact constraint), flattened, `clique` vertices each
  void build_tree(const std::vector<std::vector<int>>& adj, const int* extra, int n_extra, int clique) {
    std::vector<std::vector<int>> xadj;  // contact adjacency, only for touched vertices
    std::vector<int> xidx(NV, -1);
    for (int e = 0; e < n_extra; e++)
      for (int a = 0; a < clique; a++) {
        const int va = extra[e * clique + a];
        if (xidx[va] < 0) { xidx[va] = (int)xadj.size(); xadj.emplace_back(); }
        for (int b = 0; b < clique; b++) if (b != a) xadj[xidx[va]].push_back(extra[e * clique + b]);
      }
    bnd.assign(n_sn, {});
    parent.assign(n_sn, -1);
    level.assign(n_sn, 0);
    std::vector<std::vector<int>> children(n_sn);
    std::vector<int> mark(NV, -1);
    for (int s = 0; s < n_sn; s++) {
      std::vector<int>& b = bnd[s];
      auto visit = [&](int u) { if (sn_of[u] > s && mark[u] != s) { mark[u] = s; b.push_back(u); } };
      for (int p = sn_ptr[s]; p < sn_ptr[s + 1]; p++) {
        const int v = order[p];
        for (int u : adj[v]) visit(u);
        if (xidx[v] >= 0) for (int u : xadj[xidx[v]]) visit(u);
      }
      for (int c : children[s]) for (int u : bnd[c]) visit(u);
      std::sort(b.begin(), b.end(), [&](int x, int y) { return epos[x] < epos[y]; });
      if (!b.empty()) { parent[s] = sn_of[b[0]]; children[parent[s]].push_back(s); }
    }
    n_levels = 0;
    for (int s = 0; s < n_sn; s++) {
      int lv = 0;
      for (int c : children[s]) lv = std::max(lv, level[c] + 1);
      level[s] = lv;
      n_levels = std::max(n_levels, lv + 1);
    }
    by_level.assign(n_levels, {});
    for (int s = 0; s < n_sn; s++) by_level[level[s]].push_back(s);
  }

  // factorisation flops (inverse of the pivot block, G = W F12, Schur complement) and front storage in doubles
  void stats(double* flops, double* front_doubles, int* max_front, int* max_p) const {
    double fl = 0, mem = 0;
    int mf = 0, mp = 0;
    for (int s = 0; s < n_sn; s++) {
      const double p = 3.0 * own(s), b = 3.0 * bnd[s].size();
      fl += 2.0 * p * p * p + 2.0 * p * p * b + 2.0 * p * b * b;
      mem += (p + b) * (p + b);
      mf = std::max(mf, (int)(p + b)); mp = std::max(mp, (int)p);
    }
    *flops = fl; *front_doubles = mem; *max_front = mf; *max_p = mp;
  }
};
