// Symbolic part of the sparse direct factorisation that preconditions the Newton / adjoint solves (host only, no HIP).
//
// The reference solves every system with a direct sparse solver (cupyx spsolve, /root/reference/code/engine/sparse_solver.py:85-105).
// Here the Krylov solvers of tsl_solve are preconditioned by a multifrontal LU of the same operator:
//   * cloth vertices are ordered by geometric nested dissection of their (N+1) x (M+1) grid (separators two grid lines wide, then
//     trimmed against the real adjacency: the hinge stencil reaches two lines only from every second vertex),
//   * every FEM body gets its own dissection tree (breadth-first level structures of the tet mesh); its vertices without contact
//     are eliminated first in that order, its constrained vertices in front of the cloth subtree that holds the vertices they touch,
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
  // ---- static partition of the cloth grids (build_partition): supernodes in postorder of the dissection trees
  std::vector<int> c_order;   // cloth vertices in static elimination order
  std::vector<int> c_ptr;     // static supernode -> [first, last) in c_order
  std::vector<int> c_lo;      // static supernode -> first position of its whole subtree (the subtree ends with the supernode itself)
  std::vector<int> c_grid;    // static supernode -> grid
  std::vector<int> c_pos;     // vertex -> position in c_order, -1 for vertices outside the grids
  std::vector<int> c_sn;      // vertex -> static supernode, -1 outside the grids
  std::vector<DsBlock> blocks;
  std::vector<int> body_of;   // vertex -> FEM body, -1 if none
  std::vector<std::vector<std::vector<int>>> b_sn;  // per body: its supernodes (graph dissection of the tet mesh), postorder
  std::vector<int> loose;     // vertices in no grid and no body
  // ---- per constraint set (build_tree)
  std::vector<int> order, epos, sn_of, sn_ptr;  // position -> vertex, vertex -> position, vertex -> supernode, supernode -> [first, last) position
  int n_sn = 0;
  int merge_sep = 16;    // "direct_merge_sep": separators of at most this many vertices join the supernode of the enclosing separator (0: off).
                         // cfg4: one level and 500 fronts less, 6 % more flops, 25 M extend-add atomics less per factorisation.  Measured when
                         // the Schur launches ran at two workgroups per CU: the same 383 ms per step (12), 388 (20); with the final GEMMs,
                         // three runs each: 316.2 (0) / 313.0 (12) / 310.4 (16) ms per step, 311.5 (24), 320.5 (48)
  bool merge_k = true;   // constrained body vertices share the supernode of the separator they are placed at ("direct_merge_k"; a supernode of their own
                         // adds two elimination levels with one or two small fronts each: cfg4 391 -> 377 ms per step)
  std::vector<std::vector<int>> bnd;  // boundary vertices per supernode, sorted by elimination position
  std::vector<int> parent, level;
  int n_levels = 0;

  // the static part (build_partition) of another instance: what build_tree starts from
  void copy_partition(const DirectSym& o) {
    NV = o.NV; c_order = o.c_order; c_ptr = o.c_ptr; c_lo = o.c_lo; c_grid = o.c_grid; c_pos = o.c_pos; c_sn = o.c_sn;
    blocks = o.blocks; body_of = o.body_of; b_sn = o.b_sn; loose = o.loose; merge_sep = o.merge_sep; merge_k = o.merge_k;
  }

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

  // ------------------------------------------------------------------------------------------ static: dissection of the grids
  // adj: sorted adjacency of every vertex (may include the vertex itself)
  void build_partition(int nv, const std::vector<std::vector<int>>& adj, const std::vector<DsGrid>& grids, const std::vector<DsBlock>& blocks_in, int leaf_verts) {
    NV = nv;
    blocks = blocks_in;
    c_order.clear(); c_ptr.assign(1, 0); c_lo.clear(); c_grid.clear();
    c_pos.assign(nv, -1); c_sn.assign(nv, -1); body_of.assign(nv, -1); loose.clear();
    for (size_t b = 0; b < blocks.size(); b++)
      for (int v = blocks[b].v_offset; v < blocks[b].v_offset + blocks[b].n_verts; v++) body_of[v] = (int)b;
    b_sn.assign(blocks.size(), {});
    {
      std::vector<int> lab(nv, -1);
      for (size_t b = 0; b < blocks.size(); b++) {
        std::vector<int> region(blocks[b].n_verts);
        for (int k = 0; k < blocks[b].n_verts; k++) region[k] = blocks[b].v_offset + k;
        dissect_graph(adj, region, std::max(leaf_verts, 40), lab, b_sn[b]);
      }
    }
    std::vector<int> side(nv, 0);  // scratch of the bisection: 1 left, 2 right, 3 separator
    for (size_t gi = 0; gi < grids.size(); gi++) {
      const DsGrid& g = grids[gi];
      std::vector<int> region((size_t)(g.N + 1) * (g.M + 1));
      for (size_t k = 0; k < region.size(); k++) region[k] = g.v_offset + (int)k;
      dissect(g, (int)gi, adj, region, leaf_verts, side);
    }
    for (size_t q = 0; q < c_order.size(); q++) c_pos[c_order[q]] = (int)q;
    for (size_t s = 0; s + 1 < c_ptr.size(); s++)
      for (int q = c_ptr[s]; q < c_ptr[s + 1]; q++) c_sn[c_order[q]] = (int)s;
    for (int v = 0; v < nv; v++) if (c_pos[v] < 0 && body_of[v] < 0) loose.push_back(v);
  }

  // nested dissection of a general vertex set (the tet mesh of a FEM body): the separator is the middle level of a breadth-first
  // level structure rooted at a pseudo-peripheral vertex.  lab: scratch, -1 outside the current region.
  void dissect_graph(const std::vector<std::vector<int>>& adj, std::vector<int>& region, int leaf, std::vector<int>& lab, std::vector<std::vector<int>>& out) {
    if (region.empty()) return;
    if ((int)region.size() <= leaf) { std::sort(region.begin(), region.end()); out.push_back(region); return; }
    std::vector<int> q;
    auto bfs = [&](int start) {
      for (int v : region) lab[v] = 0;
      q.clear(); q.push_back(start); lab[start] = 1;
      for (size_t h = 0; h < q.size(); h++)
        for (int u : adj[q[h]]) if (lab[u] == 0) { lab[u] = lab[q[h]] + 1; q.push_back(u); }
    };
    bfs(region[0]);
    if (q.size() < region.size()) {  // disconnected: the reached component and the rest are independent
      std::vector<int> comp(q), rest;
      for (int v : region) if (lab[v] == 0) rest.push_back(v);
      for (int v : region) lab[v] = -1;
      dissect_graph(adj, comp, leaf, lab, out);
      dissect_graph(adj, rest, leaf, lab, out);
      return;
    }
    bfs(q.back());
    const int nl = lab[q.back()];
    if (nl < 3) { for (int v : region) lab[v] = -1; std::sort(region.begin(), region.end()); out.push_back(region); return; }
    std::vector<int> cnt(nl + 2, 0);
    for (int v : region) cnt[lab[v]]++;
    int m = 2, acc = cnt[1];
    while (m < nl - 1 && 2 * (acc + cnt[m]) < (int)region.size()) { acc += cnt[m]; m++; }
    std::vector<int> A, B, S;
    for (int v : region) { if (lab[v] < m) A.push_back(v); else if (lab[v] > m) B.push_back(v); else S.push_back(v); }
    for (int v : region) lab[v] = -1;
    std::vector<int>().swap(region);
    dissect_graph(adj, A, leaf, lab, out);
    dissect_graph(adj, B, leaf, lab, out);
    std::sort(S.begin(), S.end());
    out.push_back(S);
  }

  void close_static(int lo, int grid) {
    if ((int)c_order.size() > c_ptr.back()) { c_ptr.push_back((int)c_order.size()); c_lo.push_back(lo); c_grid.push_back(grid); }
  }

  void dissect(const DsGrid& g, int gi, const std::vector<std::vector<int>>& adj, std::vector<int>& region, int leaf_verts, std::vector<int>& side, std::vector<int>* up = nullptr) {
    if (region.empty()) return;
    const int lo = (int)c_order.size();
    const int W = g.M + 1;
    int i0 = 1 << 30, i1 = -1, j0 = 1 << 30, j1 = -1;
    for (int v : region) { const int i = (v - g.v_offset) / W, j = (v - g.v_offset) % W; i0 = std::min(i0, i); i1 = std::max(i1, i); j0 = std::min(j0, j); j1 = std::max(j1, j); }
    const int ni = i1 - i0 + 1, nj = j1 - j0 + 1;
    if ((int)region.size() <= leaf_verts || (ni <= 3 && nj <= 3)) {
      std::sort(region.begin(), region.end());
      for (int v : region) c_order.push_back(v);
      close_static(lo, gi);
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
    // Optional (merge_sep > 0): a small separator (the last split above the leaves) gets no supernode of its own, it is handed to the
    // enclosing region and opens the supernode of that region's separator (a front of 32 pivots with a 224-wide boundary is all
    // extend-add traffic and costs an elimination level in every sweep; the larger parent fronts cost the same again).
    std::vector<int> handed;
    dissect(g, gi, adj, L, leaf_verts, side, &handed);
    dissect(g, gi, adj, R, leaf_verts, side, &handed);
    std::sort(S.begin(), S.end(), [&](int a, int b) { return other(a) != other(b) ? other(a) < other(b) : coord(a) < coord(b); });
    if (up != nullptr && handed.empty() && (int)S.size() <= merge_sep) { up->insert(up->end(), S.begin(), S.end()); return; }
    for (int v : handed) c_order.push_back(v);
    for (int v : S) c_order.push_back(v);
    close_static(lo, gi);
  }

  // ------------------------------------------------------------------------------------------ per constraint set: order, boundaries, tree
  // extra: additional cliques (the 4 vertices of every contact constraint), flattened, `clique` vertices each.
  // FEM bodies: the vertices of a body that sit in no constraint keep the body's own dissection tree and are eliminated first
  // (their boundaries stay inside the body), the constrained ones are eliminated right before the smallest dissection subtree that holds
  // every grid vertex they are coupled to (they ride in the fronts between those leaves and that separator), merged per such
  // place; coupled to anything else, they go to the end of the order.
  void build_tree(const std::vector<std::vector<int>>& adj, const int* extra, int n_extra, int clique) {
    std::vector<std::vector<int>> xadj;  // contact adjacency, only for touched vertices
    std::vector<int> xidx(NV, -1);
    for (int e = 0; e < n_extra; e++)
      for (int a = 0; a < clique; a++) {
        const int va = extra[e * clique + a];
        if (xidx[va] < 0) { xidx[va] = (int)xadj.size(); xadj.emplace_back(); }
        for (int b = 0; b < clique; b++) if (b != a) xadj[xidx[va]].push_back(extra[e * clique + b]);
      }
    // ---- order
    const int n_static = (int)c_ptr.size() - 1, END = n_static;
    const int nb = (int)blocks.size();
    std::vector<int> ins(nb, -1);                 // static supernode the constrained vertices of the body are placed in front of
    std::vector<std::vector<int>> K(nb);
    for (int b = 0; b < nb; b++) {
      int pmin = 1 << 30, pmax = -1, grid = -1;
      bool to_end = false;
      for (int v = blocks[b].v_offset; v < blocks[b].v_offset + blocks[b].n_verts; v++) {
        if (xidx[v] < 0) continue;
        K[b].push_back(v);
        for (int u : xadj[xidx[v]]) {
          if (body_of[u] == b) continue;
          if (c_pos[u] < 0) { to_end = true; continue; }
          const int gu = c_grid[c_sn[u]];
          if (grid >= 0 && gu != grid) to_end = true;
          grid = gu;
          pmin = std::min(pmin, c_pos[u]); pmax = std::max(pmax, c_pos[u]);
        }
      }
      if (K[b].empty()) continue;
      if (to_end || pmax < 0) { ins[b] = END; continue; }
      int s = c_sn[c_order[pmax]];
      while (s < n_static && !(c_lo[s] <= pmin && c_ptr[s + 1] > pmax && c_grid[s] == grid)) s++;
      ins[b] = s;  // == END if nothing qualifies
    }
    order.clear(); sn_ptr.assign(1, 0);
    auto close_sn = [&]() { if ((int)order.size() > sn_ptr.back()) sn_ptr.push_back((int)order.size()); };
    for (int b = 0; b < nb; b++)
      for (const auto& sn : b_sn[b]) { for (int v : sn) if (xidx[v] < 0) order.push_back(v); close_sn(); }
    for (int v : loose) { order.push_back(v); close_sn(); }
    std::vector<std::vector<int>> at(n_static + 1);
    for (int b = 0; b < nb; b++) if (ins[b] >= 0) at[ins[b]].push_back(b);
    for (int s = 0; s <= n_static; s++) {
      for (int b : at[s]) for (int v : K[b]) order.push_back(v);
      if (!merge_k || s == n_static) close_sn();   // merge_k: the constrained body vertices open the supernode of the separator they sit in front of
      if (s < n_static) { for (int q = c_ptr[s]; q < c_ptr[s + 1]; q++) order.push_back(c_order[q]); close_sn(); }
    }
    n_sn = (int)sn_ptr.size() - 1;
    epos.assign(NV, 0); sn_of.assign(NV, 0);
    for (int p = 0; p < NV; p++) epos[order[p]] = p;
    for (int s = 0; s < n_sn; s++)
      for (int p = sn_ptr[s]; p < sn_ptr[s + 1]; p++) sn_of[order[p]] = s;
    // ---- boundaries and elimination tree
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
