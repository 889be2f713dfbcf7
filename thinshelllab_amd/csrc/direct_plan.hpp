// Front layout and index maps of the multifrontal factorisation (host only, no HIP): turns the symbolic result of direct_sym.hpp
// into the arrays the kernels of k_direct.hpp consume.  Recomputed whenever the contact constraint set changes.
//
// Front of supernode s (row-major, leading dimension ld = pp + bp, all multiples of DS_T):
//      [ F11 (pp x pp) | F12 (pp x bp) ]      own dofs 0..p-1 (padding p..pp-1 carries an identity diagonal)
//      [ F21 (bp x pp) | F22 (bp x bp) ]      boundary dofs pp..pp+b-1 (padding is zero)
// after the factorisation:  [ W = F11^-1 | G = W F12 ] / [ F21 | S = F22 - F21 G ];  S is added into the parent's front.
#pragma once
#include "direct_sym.hpp"

#define DS_T 32  // tile edge of the dense kernels; p and b are padded to multiples of it

struct DsFrontDesc {
  long long off;             // first element of the front in the arena (doubles)
  long long goff;            // first element of G = W F12 (pp x bp, row stride bp) in the G arena
  int p, pp, b, bp, ld;      // own dofs, padded; boundary dofs, padded; leading dimension
  int parent;                // supernode of the parent front, -1 at a root
  int rel_off;               // rel[rel_off + iv]: local dof index (in the parent front) of the first dof of boundary vertex iv
  int vtx_off;               // vtx[vtx_off + iv]: vertex id of local vertex iv (own vertices, then boundary vertices)
  int nv_own, nv_bnd;
  int scr_off;               // offset of this front's scratch inside the per-level scratch (doubles)
};

// fronts of one tree level with pivot blocks of similar size: one set of launches (level_sn[first .. first + count), sorted by pp descending)
struct DsBatch {
  int first, count, level;
  int max_pp, max_ld, max_bp;
  int act_off;   // act_n / act_ld [act_off + k]: fronts still active at block step k (a prefix) and their largest ld
};

struct DirectPlan {
  DirectSym sym;
  std::vector<DsFrontDesc> fr;
  std::vector<int> rel, vtx;
  std::vector<long long> blk_dst;  // per block of the static pattern (CSR order of `adj`): top-left element in the arena
  std::vector<int> blk_ld;
  std::vector<long long> con_dst;  // per constraint x 16 (vertex pair a, b)
  std::vector<int> con_ld;
  std::vector<int> level_ptr, level_sn;          // fronts per level (level 0 = leaves)
  std::vector<DsBatch> batches;                  // in level order
  std::vector<int> act_n, act_ld;
  // solve work lists: (front, first row) of every 16-row chunk of the own rows / boundary rows, level after level
  std::vector<int> wl_front, wl_row, wl_own_ptr, wl_bnd_ptr;   // own chunks of level l: [wl_own_ptr[l], wl_bnd_ptr[l]); boundary chunks: [wl_bnd_ptr[l], wl_own_ptr[l + 1])
  int n_levels = 0;
  long long arena = 0;      // doubles
  long long garena = 0;     // doubles
  long long scratch = 0;    // doubles, max over the levels
  double flops = 0;

  static int pad(int n) { return (n + DS_T - 1) / DS_T * DS_T; }

  // local dof index of vertex v in front s (own: 3 idx; boundary: pp + 3 idx); -1 if absent
  int local_dof(int s, int v) const {
    const int l = sym.local(s, v);
    if (l < 0) return -1;
    const int no = sym.own(s);
    return l < no ? 3 * l : fr[s].pp + 3 * (l - no);
  }

  // adj: sorted adjacency (with or without self); row_ptr: CSR offsets of adj (blocks are numbered row by row);
  // cons: n_cons x 4 vertex ids of the contact constraints
  int build(const std::vector<std::vector<int>>& adj, const std::vector<int>& row_ptr, const int* cons, int n_cons) {
    sym.build_tree(adj, cons, n_cons, 4);
    const int S = sym.n_sn;
    fr.assign(S, DsFrontDesc{});
    rel.clear(); vtx.clear();
    arena = 0; garena = 0; flops = 0;
    for (int s = 0; s < S; s++) {
      DsFrontDesc& f = fr[s];
      f.nv_own = sym.own(s); f.nv_bnd = (int)sym.bnd[s].size();
      f.p = 3 * f.nv_own; f.b = 3 * f.nv_bnd;
      f.pp = pad(f.p); f.bp = pad(f.b); f.ld = f.pp + f.bp;
      f.off = arena;
      arena += (long long)f.ld * f.ld;
      f.goff = garena;
      garena += (long long)f.pp * f.bp;
      f.parent = sym.parent[s];
      f.vtx_off = (int)vtx.size();
      for (int q = sym.sn_ptr[s]; q < sym.sn_ptr[s + 1]; q++) vtx.push_back(sym.order[q]);
      for (int v : sym.bnd[s]) vtx.push_back(v);
      const double p = f.pp, b = f.bp;
      flops += 2.0 * p * p * p + 2.0 * p * p * b + 2.0 * p * b * b;
    }
    for (int s = 0; s < S; s++) {
      DsFrontDesc& f = fr[s];
      f.rel_off = (int)rel.size();
      for (int v : sym.bnd[s]) {
        const int l = local_dof(f.parent, v);
        if (l < 0) return -1;  // the boundary of a child is contained in the front of its parent
        rel.push_back(l);
      }
    }
    // levels: as soon as possible (a front sits one level above its deepest child): the small fronts of the FEM bodies' own
    // dissection trees and of shallow subtrees then share the batches of the ~10^3 cloth leaves instead of adding batches of their
    // own next to the few large fronts near the root, where every batch costs its block steps in sequence
    const int L = sym.n_levels;
    n_levels = L;
    const std::vector<int>& alap = sym.level;
    std::vector<std::vector<int>> by_level(L);
    for (int s = 0; s < S; s++) by_level[alap[s]].push_back(s);
    level_ptr.assign(L + 1, 0); level_sn.clear();
    batches.clear(); act_n.clear(); act_ld.clear();
    wl_front.clear(); wl_row.clear(); wl_own_ptr.assign(L + 1, 0); wl_bnd_ptr.assign(L, 0);
    scratch = 0;
    // a new batch starts where the pivot block falls below a quarter of the batch's largest (empty workgroups of the smaller
    // fronts are cheap, an extra batch costs its block steps in sequence); on a level with hundreds of fronts a batch of 32 or
    // more also ends where the pivot block shrinks at all (a handful of larger fronts must not size the grid of a thousand leaves)
    for (int l = 0; l < L; l++) {
      std::vector<int>& fl = by_level[l];
      std::stable_sort(fl.begin(), fl.end(), [&](int a, int b) { return fr[a].pp > fr[b].pp; });
      long long scr = 0;
      for (size_t i = 0; i < fl.size(); i++) {
        const int s = fl[i];
        DsFrontDesc& f = fr[s];
        if (i == 0 || 4 * f.pp <= batches.back().max_pp || (fl.size() > 256 && batches.back().count >= 32 && f.pp < fr[fl[i - 1]].pp)) {
          DsBatch b{};
          b.first = (int)level_sn.size(); b.count = 0; b.level = l;
          batches.push_back(b);
        }
        DsBatch& b = batches.back();
        b.count++;
        b.max_pp = std::max(b.max_pp, f.pp); b.max_ld = std::max(b.max_ld, f.ld); b.max_bp = std::max(b.max_bp, f.bp);
        level_sn.push_back(s);
        f.scr_off = (int)scr;
        scr += 2LL * DS_T * DS_T + 4LL * DS_T * f.pp;  // pivot-block inverses, row and column side panels (ping-pong each)
      }
      level_ptr[l + 1] = (int)level_sn.size();
      scratch = std::max(scratch, scr);
      wl_own_ptr[l] = (int)wl_front.size();
      for (int s : fl) for (int r = 0; r < fr[s].p; r += 16) { wl_front.push_back(s); wl_row.push_back(r); }
      wl_bnd_ptr[l] = (int)wl_front.size();
      for (int s : fl) for (int r = 0; r < fr[s].b; r += 16) { wl_front.push_back(s); wl_row.push_back(r); }
    }
    wl_own_ptr[L] = (int)wl_front.size();
    for (DsBatch& b : batches) {
      b.act_off = (int)act_n.size();
      for (int k = 0; k * DS_T < b.max_pp; k++) {
        int n = 0, mld = 0;
        while (n < b.count && fr[level_sn[b.first + n]].pp > k * DS_T) { mld = std::max(mld, fr[level_sn[b.first + n]].ld); n++; }
        act_n.push_back(n); act_ld.push_back(mld);
      }
    }
    // static blocks
    const int NV = sym.NV;
    blk_dst.assign(row_ptr[NV], -1); blk_ld.assign(row_ptr[NV], 0);
    for (int r = 0; r < NV; r++) {
      const auto& row = adj[r];
      for (int k = 0; k < (int)row.size(); k++) {
        const int c = row[k];
        const int s = std::min(sym.sn_of[r], sym.sn_of[c]);
        const int lr = local_dof(s, r), lc = local_dof(s, c);
        if (lr < 0 || lc < 0) return -2;
        blk_dst[row_ptr[r] + k] = fr[s].off + (long long)lr * fr[s].ld + lc;
        blk_ld[row_ptr[r] + k] = fr[s].ld;
      }
    }
    con_dst.assign((size_t)n_cons * 16, -1); con_ld.assign((size_t)n_cons * 16, 0);
    for (int e = 0; e < n_cons; e++)
      for (int a = 0; a < 4; a++)
        for (int b = 0; b < 4; b++) {
          const int va = cons[4 * e + a], vb = cons[4 * e + b];
          const int s = std::min(sym.sn_of[va], sym.sn_of[vb]);
          const int lr = local_dof(s, va), lc = local_dof(s, vb);
          if (lr < 0 || lc < 0) return -3;
          con_dst[(size_t)e * 16 + a * 4 + b] = fr[s].off + (long long)lr * fr[s].ld + lc;
          con_ld[(size_t)e * 16 + a * 4 + b] = fr[s].ld;
        }
    return 0;
  }
};
