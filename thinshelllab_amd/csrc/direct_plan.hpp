// Front layout and index maps of the multifrontal factorisation (host only, no HIP): turns the symbolic result of direct_sym.hpp
// into the arrays the kernels of k_direct.hpp consume.  Recomputed whenever the contact constraint set changes.
//
// Front of supernode s (p own dofs padded to pp, b boundary dofs padded to bp, all multiples of DS_T; ld = pp + bp):
//      [ F11 (pp x pp) | F12 (pp x bp) ]      "top rows", row stride ld, in the panel arena at `off` (padding p..pp-1: identity diagonal)
//      [ F21 (bp x pp) ]                      row stride pp, in the panel arena at `off21` = off + pp ld (padding rows are zero)
//      [ S   (bp x bp) ]                      row stride bp, in the Schur arena at `soff`
// F22 is never materialised: no matrix entry lies between two boundary vertices (a block lives in the front of the earlier-eliminated
// of its two vertices), F22 is the sum of the children's Schur complements only.  After the factorisation:
//      [ W = F11^-1 | G = W F12 (G arena) ] / [ F21 ] / [ S = sum_children ext(S_child) - F21 G ].
// The extend-add is a GATHER on the parent's side (round 4): a child STORES its S (one writer, coalesced); the parent's panels
// (F11, F12, F21) and its own Schur complement read the children's S through `pmap` -- for every child a table over the parent's local
// dofs that gives the child's boundary dof or -1 -- in the fixed child order of `ch_list`.  No atomics, no cleared F22, every sum has
// a fixed order (bit-reproducible factors).
#pragma once
#include <atomic>
#include <chrono>
#include <thread>
#include "direct_sym.hpp"

#define DS_T 32  // tile edge of the dense kernels; p and b are padded to multiples of it
#define DS_SMALL 128   // fronts of at most this many (padded) pivots can be inverted by ONE workgroup with the pivot block in LDS (k_ds_inv_small)
// workgroups of k_ds_inv_small a CU holds: the block (row stride max_pp + 1) + 4.5 KB of static arrays, 160 KB of LDS -- two at 96 pivots, one at 128
static inline size_t ds_small_lds(int max_pp) { return (size_t)max_pp * (max_pp + 1) * sizeof(double); }   // the block alone: the pivot tile is inverted in place
static inline int ds_small_per_cu(int max_pp) { return (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / (ds_small_lds(max_pp) + 5 * 1024))); }

struct DsFrontDesc {
  long long off;             // top rows [F11 | F12] in the panel arena (doubles)
  long long off21;           // F21 (bp x pp, row stride pp) in the panel arena
  long long soff;            // S (bp x bp, row stride bp) in the Schur arena
  long long goff;            // first element of G = W F12 (pp x bp, row stride bp) in the G arena
  int p, pp, b, bp, ld;      // own dofs, padded; boundary dofs, padded; pp + bp
  int parent;                // supernode of the parent front, -1 at a root
  int pmap_off;              // pmap[pmap_off + d], d a local dof of the PARENT (0 .. parent ld): this front's boundary dof that lands there, or -1
  int vtx_off;               // vtx[vtx_off + iv]: vertex id of local vertex iv (own vertices, then boundary vertices)
  int nv_own, nv_bnd;
  int scr_off;               // offset of this front's scratch inside the per-level scratch (doubles)
  int yoff;                  // y_f (the boundary update of the upward solve sweep, b entries) in the Y buffer
  int cpm[4], cy[4];         // pmap_off / yoff of the first four children, inline: the sweeps' gather then needs no look-up of the child records
  int ch_off, nchild;        // child fronts: ch_rec[ch_off .. ch_off + nchild), ascending supernode id (the summation order of the gather)
  int lead;                  // boundary dofs that are OWN dofs of the parent: the first `lead` of the boundary (it is sorted by elimination position), i.e. the
                             // leading lead x lead block of S is all of S the parent's pivot block F11 receives (the look-ahead split of the Schur launches)
};
// what the gather kernels need of a child, in the parent's child order
struct DsChildRec {
  long long soff;            // the child's S
  int bp, b;                 // its row stride and size
  int pmap_off;              // its table over the parent's dofs
  int sn;
  int nb;                    // boundary dofs that land in the parent's BOUNDARY part (the others are own dofs of the parent)
  int yoff;                  // the child's y_f in the Y buffer
};

// fronts of one tree level with pivot blocks of similar size: one set of launches (level_sn[first .. first + count), sorted by pp descending)
struct DsBatch {
  int first, count, level;
  int max_pp, max_ld, max_bp;
  int act_off;   // act_n / act_ld [act_off + k]: fronts still active at block step k (a prefix) and their largest ld
};

// dynamic parallel loop over [0, n) on up to `nt` host threads (chunks handed out by an atomic counter); body(thread, index)
template <class F>
static void ds_parallel_for(int n, int nt, int chunk, F body) {
  nt = std::max(1, std::min(nt, (n + chunk - 1) / chunk));
  if (nt == 1) { for (int i = 0; i < n; i++) body(0, i); return; }
  std::atomic<int> next{0};
  auto work = [&](int t) {
    for (;;) {
      const int i0 = next.fetch_add(chunk);
      if (i0 >= n) break;
      for (int i = i0; i < std::min(n, i0 + chunk); i++) body(t, i);
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nt; t++) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
}

struct DirectPlan {
  DirectSym sym;
  std::vector<DsFrontDesc> fr;
  std::vector<int> pmap, vtx;
  std::vector<DsChildRec> ch_rec;
  std::vector<long long> blk_dst;  // per block of the static pattern (CSR order of `adj`): top-left element in the panel arena
  std::vector<int> blk_ld;
  std::vector<int> blk_q, blk_lptr;   // the same blocks level by level: blk_q[blk_lptr[l] .. blk_lptr[l + 1]) land in fronts of level l
  // look-ahead (round 6): within a level the blocks that land in a pivot block F11 come first, [blk_lptr[l], blk_lmid[l]); the same for the contact groups
  // (cgr_lmid).  Levels >= la_from have ONE batch each: there the leading block of every Schur complement (all the parent's F11 receives) is formed first,
  // the parent's pivot block is gathered and inverted while the rest of the Schur launch and the gather of the parent's F12 / F21 run on a second stream.
  // Empty tables (merged plans of a scene group) / la_from >= n_levels: no look-ahead.
  std::vector<int> blk_lmid, cgr_lmid;
  int la_from = 1 << 30;
  std::vector<long long> con_dst;  // per constraint x 16 (vertex pair a, b)
  std::vector<int> con_ld, con_lvl;   // row stride and tree level of the destination (con_lvl: 2 level + 1 outside F11 -- the groups of a level that land in F11 come first)
  // the same sub-blocks grouped by DESTINATION, level by level (groups cgr_lptr[l] .. cgr_lptr[l + 1] land on level l): group g adds the
  // sub-blocks cgr_ent[cgr_ptr[g] .. cgr_ptr[g + 1]) (16 c + sub, ascending) to the 3 x 3 block at cgr_dst[g] -- one writer per entry, a
  // fixed order (several constraints share vertex pairs)
  std::vector<int> cgr_lptr, cgr_ptr, cgr_ent, cgr_ld;
  std::vector<long long> cgr_dst;
  std::vector<int> level_maxld;
  long long arena_leaf = 0;        // extent of the leaf level's panels at the head of the panel arena (doubles)
  std::vector<std::pair<long long, long long>> leaf_ranges;   // merged plan of a scene group: (offset, extent) of every member's leaf panels; empty: [0, arena_leaf)
  // merged plan: first chunk of the WIDE part of the own / boundary chunk lists of every level (chunks of members whose own launch would be
  // k_ds_gemv_wide come last: the two sweep kernels round differently, every member keeps the kernel of its single-scene run); empty: the rule of ds_launch_gemv
  std::vector<int> wl_own_wide, wl_bnd_wide;
  std::vector<std::vector<int>> lists;   // scratch of the level-by-level block list (per host thread and level)
  std::vector<int> level_ptr, level_sn;          // fronts per level (level 0 = leaves)
  std::vector<DsBatch> batches;                  // in level order
  std::vector<int> act_n, act_ld;
  // solve work lists: (front, first row) of every 16-row chunk of the own rows / boundary rows, level after level
  std::vector<int> sweep_cache; int sweep_cache_L0 = -1;   // phase table of the one-launch sweeps of levels >= sweep_cache_L0 (filled by direct_apply, dropped by build)
  std::vector<int> wl_front, wl_row, wl_own_ptr, wl_bnd_ptr;   // own chunks of level l: [wl_own_ptr[l], wl_bnd_ptr[l]); boundary chunks: [wl_bnd_ptr[l], wl_own_ptr[l + 1])
  int n_levels = 0;
  long long arena = 0;      // panel arena (top rows + F21 of every front), doubles; the fronts of level 0 come first
  long long sarena = 0;     // Schur arena, doubles
  long long garena = 0;     // doubles
  long long ylen = 0;       // doubles: boundary updates of the solve's upward sweep, bp per front
  long long scratch = 0;    // doubles, max over the levels
  double phase_ms[6] = {0, 0, 0, 0, 0, 0};   // host time of the last build: tree, descriptors + parent maps, levels / batches / work lists, static block map, contact map
  std::vector<int> tpos;                 // static: index of the mirrored block of every pattern block
  std::vector<std::vector<int>> locs;    // per host thread: vertex -> local dof of the front being scattered (-1 outside)
  int threads = 0;                       // host threads of the map construction (0: min(8, hardware))
  int n_cu = 256;                        // compute units of the device (rounds of the LDS kernel)
  bool split_small = true, split_rem = true;   // batch rules of round 4 ("direct_split": bit 0 / bit 1)
  int n_threads() const { return threads > 0 ? threads : (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency())); }
  double flops = 0;

  static int pad(int n) { return (n + DS_T - 1) / DS_T * DS_T; }

  // local dof index of vertex v in front s (own: 3 idx; boundary: pp + 3 idx); -1 if absent
  int local_dof(int s, int v) const {
    const int l = sym.local(s, v);
    if (l < 0) return -1;
    const int no = sym.own(s);
    return l < no ? 3 * l : fr[s].pp + 3 * (l - no);
  }
  // panel-arena address and row stride of the entry (lr, lc) of front f, local dofs; -1 inside F22 (no matrix entry lives there)
  static long long panel_addr(const DsFrontDesc& f, int lr, int lc, int* ld_out) {
    if (lr < f.pp) { *ld_out = f.ld; return f.off + (long long)lr * f.ld + lc; }
    if (lc >= f.pp) { *ld_out = 0; return -1; }
    *ld_out = f.pp;
    return f.off21 + (long long)(lr - f.pp) * f.pp + lc;
  }

  // fronts of every level (by_level[l]: ids into `fr`) -> level_sn / level_ptr, the batches of every level, the active-front tables of the
  // block steps, the per-level scratch offsets of the fronts, the work lists of the solve sweeps.  Used by build() for one scene and by
  // merge() for the fronts of several scenes that share the launches of a level (direct_group.hpp).
  void build_levels(std::vector<std::vector<int>>& by_level) {
    const int L = (int)by_level.size();
    n_levels = L;
    level_ptr.assign(L + 1, 0); level_sn.clear();
    batches.clear(); act_n.clear(); act_ld.clear();
    wl_front.clear(); wl_row.clear(); wl_own_ptr.assign(L + 1, 0); wl_bnd_ptr.assign(L, 0);
    sweep_cache_L0 = -1;
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
        // (round 4) ... and where fronts that fit the LDS kernel, of at most half the batch's largest pivot block, follow fronts that do not
        // (cfg4 level 2: 2 x 224, 2 x 192 and 128 x 64 pivots -- the 128 small ones went through the seven block steps of the four large ones)
        if (i == 0 || 4 * f.pp <= batches.back().max_pp || (fl.size() > 256 && batches.back().count >= 32 && f.pp < fr[fl[i - 1]].pp) ||
            (split_small && batches.back().max_pp > DS_SMALL && f.pp <= DS_SMALL && 2 * f.pp <= batches.back().max_pp)) {
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
      // (round 4) a batch of the LDS kernel that needs one more round of the chip for a few fronts hands them to a batch of their own, which
      // takes another path next to it (cfg4 level 1: 261 fronts of 128 pivots, one workgroup per CU -- 256 in the first round, 5 in a second
      // one that took as long)
      if (split_rem)
        for (size_t q = 0; q < batches.size(); q++) {
          DsBatch& b = batches[q];
          if (b.level != l || b.max_pp > DS_SMALL) continue;
          const int cap = n_cu * ds_small_per_cu(b.max_pp), rem = b.count % cap;
          if (b.count <= cap || rem == 0 || rem > cap / 8) continue;
          DsBatch r{};
          r.first = b.first + b.count - rem; r.count = rem; r.level = l;
          b.count -= rem;
          for (int z = 0; z < r.count; z++) { const DsFrontDesc& f = fr[level_sn[r.first + z]]; r.max_pp = std::max(r.max_pp, f.pp); r.max_ld = std::max(r.max_ld, f.ld); r.max_bp = std::max(r.max_bp, f.bp); }
          b.max_ld = 0; b.max_bp = 0;
          for (int z = 0; z < b.count; z++) { const DsFrontDesc& f = fr[level_sn[b.first + z]]; b.max_ld = std::max(b.max_ld, f.ld); b.max_bp = std::max(b.max_bp, f.bp); }
          batches.insert(batches.begin() + q + 1, r);
          q++;
        }
      level_ptr[l + 1] = (int)level_sn.size();
      scratch = std::max(scratch, scr);
      wl_own_ptr[l] = (int)wl_front.size();
      for (int s : fl) for (int r = 0; r < fr[s].p; r += 16) { wl_front.push_back(s); wl_row.push_back(r); }
      wl_bnd_ptr[l] = (int)wl_front.size();
      for (int s : fl) for (int r = 0; r < fr[s].b; r += 16) { wl_front.push_back(s); wl_row.push_back(r); }
    }
    wl_own_ptr[L] = (int)wl_front.size();
    {   // lowest level from which every level is ONE batch (the chain of single launches towards the root); at least two such levels or none
      std::vector<int> nb(L, 0);
      for (const DsBatch& b : batches) nb[b.level]++;
      int l0 = L;
      while (l0 > 0 && nb[l0 - 1] == 1) l0--;
      la_from = (L - l0 >= 2) ? std::max(l0, 1) : (1 << 30);
    }
    for (DsBatch& b : batches) {
      b.act_off = (int)act_n.size();
      for (int k = 0; k * DS_T < b.max_pp; k++) {
        int n = 0, mld = 0;
        while (n < b.count && fr[level_sn[b.first + n]].pp > k * DS_T) { mld = std::max(mld, fr[level_sn[b.first + n]].ld); n++; }
        act_n.push_back(n); act_ld.push_back(mld);
      }
    }
  }

  // adj: sorted adjacency (with or without self); row_ptr: CSR offsets of adj (blocks are numbered row by row);
  // cons: n_cons x 4 vertex ids of the contact constraints
  int build(const std::vector<std::vector<int>>& adj, const std::vector<int>& row_ptr, const int* cons, int n_cons) {
    auto tick = std::chrono::steady_clock::now();
    auto lap = [&](int k) { const auto n = std::chrono::steady_clock::now(); phase_ms[k] = 1e3 * std::chrono::duration<double>(n - tick).count(); tick = n; };
    sym.build_tree(adj, cons, n_cons, 4);
    lap(0);
    const int S = sym.n_sn;
    fr.assign(S, DsFrontDesc{});
    pmap.clear(); vtx.clear(); ch_rec.clear();
    arena = 0; sarena = 0; garena = 0; flops = 0; ylen = 0;
    for (int s = 0; s < S; s++) {
      DsFrontDesc& f = fr[s];
      f.nv_own = sym.own(s); f.nv_bnd = (int)sym.bnd[s].size();
      f.p = 3 * f.nv_own; f.b = 3 * f.nv_bnd;
      f.pp = pad(f.p); f.bp = pad(f.b); f.ld = f.pp + f.bp;
      f.parent = sym.parent[s];
      f.vtx_off = (int)vtx.size();
      for (int q = sym.sn_ptr[s]; q < sym.sn_ptr[s + 1]; q++) vtx.push_back(sym.order[q]);
      for (int v : sym.bnd[s]) vtx.push_back(v);
      const double p = f.pp, b = f.bp;
      flops += 2.0 * p * p * p + 2.0 * p * p * b + 2.0 * p * b * b;
    }
    // arena addresses level by level (the panels of the leaf level are one contiguous range at the start of the panel arena)
    {
      std::vector<int> cnt(sym.n_levels + 1, 0), ord(S);
      for (int s = 0; s < S; s++) cnt[sym.level[s] + 1]++;
      for (int l = 0; l < sym.n_levels; l++) cnt[l + 1] += cnt[l];
      for (int s = 0; s < S; s++) ord[cnt[sym.level[s]]++] = s;
      for (int s : ord) {
        DsFrontDesc& f = fr[s];
        f.off = arena; f.off21 = arena + (long long)f.pp * f.ld;
        arena += (long long)f.pp * f.ld + (long long)f.bp * f.pp;
        f.soff = sarena; sarena += (long long)f.bp * f.bp;
        f.goff = garena; garena += (long long)f.pp * f.bp;
        f.yoff = (int)ylen; ylen += f.bp;
      }
    }
    // children of every front in ascending supernode order, and for every child its table over the PARENT's local dofs: the
    // child's boundary dof that lands there, or -1 (the boundary of a child is contained in the front of its parent).  The parent's
    // front is scattered into a vertex table once and every child reads its boundary from it.
    {
      std::vector<int> cptr(S + 1, 0);
      for (int s = 0; s < S; s++) if (fr[s].parent >= 0) { cptr[fr[s].parent + 1]++; fr[fr[s].parent].nchild++; }
      for (int s = 0; s < S; s++) cptr[s + 1] += cptr[s];
      ch_rec.assign(cptr[S], DsChildRec{});
      long long total = 0;
      { std::vector<int> fill(cptr.begin(), cptr.end() - 1);
        for (int s = 0; s < S; s++) { fr[s].ch_off = cptr[s]; if (fr[s].parent >= 0) { fr[s].pmap_off = (int)total; total += fr[fr[s].parent].ld; ch_rec[fill[fr[s].parent]++].sn = s; } } }
      if (total > 0x7fffffffLL) return -1;
      pmap.assign((size_t)total, -1);
      if (locs.empty()) locs.resize(1);
      std::vector<int>& lc = locs[0];
      if ((int)lc.size() != sym.NV) lc.assign(sym.NV, -1);
      bool bad = false;
      for (int p = 0; p < S; p++) {   // 0.3 ms on one thread (MI355X host); starting threads costs more than they save here
        if (cptr[p] == cptr[p + 1]) continue;
        const DsFrontDesc& f = fr[p];
        const int* fv = &vtx[f.vtx_off];
        for (int i = 0; i < f.nv_own; i++) lc[fv[i]] = 3 * i;
        for (int i = 0; i < f.nv_bnd; i++) lc[fv[f.nv_own + i]] = f.pp + 3 * i;
        for (int q = cptr[p]; q < cptr[p + 1]; q++) {
          DsChildRec& cr = ch_rec[q];
          const DsFrontDesc& ch = fr[cr.sn];
          cr.soff = ch.soff; cr.bp = ch.bp; cr.b = ch.b; cr.pmap_off = ch.pmap_off; cr.yoff = ch.yoff;
          if (q - cptr[p] < 4) { fr[p].cpm[q - cptr[p]] = ch.pmap_off; fr[p].cy[q - cptr[p]] = ch.yoff; }
          const int* cv = &vtx[ch.vtx_off + ch.nv_own];
          int* pm = &pmap[ch.pmap_off];
          int prev = -1;
          for (int i = 0; i < ch.nv_bnd; i++) {
            const int l = lc[cv[i]];
            if (l < 0 || l <= prev) { bad = true; continue; }   // (the boundary is sorted by elimination position: the table is monotone)
            prev = l;
            pm[l] = 3 * i; pm[l + 1] = 3 * i + 1; pm[l + 2] = 3 * i + 2;
            if (l >= f.pp) cr.nb += 3;
          }
          fr[cr.sn].lead = ch.b - cr.nb;
        }
        for (int i = 0; i < f.nv_own + f.nv_bnd; i++) lc[fv[i]] = -1;
      }
      if (bad) return -1;
      for (int s = 0; s < S; s++) if (fr[s].parent < 0 && fr[s].nv_bnd > 0) return -1;
    }
    lap(1);
    // levels: as soon as possible (a front sits one level above its deepest child): the small fronts of the FEM bodies' own
    // dissection trees and of shallow subtrees then share the batches of the ~10^3 cloth leaves instead of adding batches of their
    // own next to the few large fronts near the root, where every batch costs its block steps in sequence
    {
      std::vector<std::vector<int>> by_level(sym.n_levels);
      for (int s = 0; s < S; s++) by_level[sym.level[s]].push_back(s);
      build_levels(by_level);
    }
    const int L = sym.n_levels;
    const std::vector<int>& alap = sym.level;
    lap(2);
    // static blocks: block (r, c) of the pattern lives in the front of the earlier-eliminated of its two vertices.  One pass over
    // the supernodes with a scattered vertex -> local index table (own vertices and boundary of the current front) and the static
    // index of the mirrored block (c, r): no searches (the first version looked both vertices up by bisection per block: 4.5 of
    // the 6.5 ms a plan cost on cfg4, twice per time step)
    const int NV = sym.NV;
    if ((int)tpos.size() != row_ptr[NV]) {   // pattern is static: position of r in the row of c, once
      tpos.assign(row_ptr[NV], -1);
      for (int r = 0; r < NV; r++)
        for (int k = 0; k < (int)adj[r].size(); k++) {
          const int c = adj[r][k];
          const auto& rc = adj[c];
          const auto it = std::lower_bound(rc.begin(), rc.end(), r);
          if (it != rc.end() && *it == r) tpos[row_ptr[r] + k] = row_ptr[c] + (int)(it - rc.begin());
        }
    }
    blk_dst.resize(row_ptr[NV]); blk_ld.resize(row_ptr[NV]);   // every entry is written below (counted), no refill
    {
      const int nt = n_threads();
      if ((int)locs.size() < nt) locs.resize(nt);
      // the blocks are also listed LEVEL BY LEVEL (blk_q): the panels of a level are written by the gather of the children's Schur
      // complements when the level starts (a store, nothing is cleared), its matrix entries are added right after
      if ((int)lists.size() < 2 * nt * L) lists.resize((size_t)2 * nt * L);   // per thread and level: blocks inside F11, then the others
      for (auto& v : lists) v.clear();
      std::atomic<int> bad{0};
      std::atomic<long long> written{0};
      ds_parallel_for(S, nt, 16, [&](int t, int s) {
        std::vector<int>& loc = locs[t];
        if ((int)loc.size() != NV) loc.assign(NV, -1);
        std::vector<int>& mine11 = lists[2 * ((size_t)t * L + alap[s])];
        std::vector<int>& mine = lists[2 * ((size_t)t * L + alap[s]) + 1];
        const DsFrontDesc& f = fr[s];
        const int no = f.nv_own;
        long long nw = 0;
        const int* fv = &vtx[f.vtx_off];
        for (int i = 0; i < no; i++) loc[fv[i]] = 3 * i;
        for (int i = 0; i < f.nv_bnd; i++) loc[fv[no + i]] = f.pp + 3 * i;
        for (int i = 0; i < no; i++) {
          const int r = fv[i];
          const auto& row = adj[r];
          for (int k = 0; k < (int)row.size(); k++) {
            const int c = row[k];
            if (sym.sn_of[c] < s) continue;   // placed from the front of c
            const int lc = loc[c];
            if (lc < 0) { bad = 1; continue; }
            const int q = row_ptr[r] + k;
            blk_dst[q] = f.off + (long long)(3 * i) * f.ld + lc; blk_ld[q] = f.ld; nw++; (lc < f.pp ? mine11 : mine).push_back(q);
            if (sym.sn_of[c] > s) {   // the mirrored block (c, r) lies in F21 (c is a boundary vertex of this front)
              const int qt = tpos[q];
              if (qt >= 0) { blk_dst[qt] = f.off21 + (long long)(lc - f.pp) * f.pp + 3 * i; blk_ld[qt] = f.pp; nw++; mine.push_back(qt); }
            }
          }
        }
        for (int i = 0; i < no + f.nv_bnd; i++) loc[fv[i]] = -1;
        written += nw;
      });
      if (bad || written != row_ptr[NV]) return -2;
      blk_q.resize(row_ptr[NV]); blk_lptr.assign(L + 1, 0); blk_lmid.assign(L, 0);
      size_t o = 0;
      for (int l = 0; l < L; l++) {
        for (int part = 0; part < 2; part++) {
          for (int t = 0; t < nt; t++) { const std::vector<int>& v = lists[2 * ((size_t)t * L + l) + part]; std::copy(v.begin(), v.end(), blk_q.begin() + o); o += v.size(); }
          if (part == 0) blk_lmid[l] = (int)o;
        }
        blk_lptr[l + 1] = (int)o;
      }
    }
    // panels of the leaf level: the contiguous head of the panel arena (the only part that is cleared)
    arena_leaf = 0;
    for (int q = level_ptr[0]; q < level_ptr[1]; q++) { const DsFrontDesc& f = fr[level_sn[q]]; arena_leaf = std::max(arena_leaf, f.off21 + (long long)f.bp * f.pp); }
    level_maxld.assign(L, 0);
    for (int l = 0; l < L; l++) for (int q = level_ptr[l]; q < level_ptr[l + 1]; q++) level_maxld[l] = std::max(level_maxld[l], fr[level_sn[q]].ld);
    lap(3);
    const int rc = build_con(cons, n_cons);
    lap(4);
    return rc;
  }

  // destination of the 16 vertex-pair sub-blocks of every contact block, in the ORDER of `cons` (the engine appends constraints in
  // no fixed order: a plan found again for the same constraint SET only needs this map redone)
  int build_con(const int* cons, int n_cons) {
    con_dst.assign((size_t)n_cons * 16, -1); con_ld.assign((size_t)n_cons * 16, 0); con_lvl.assign((size_t)n_cons * 16, 0);
    for (int e = 0; e < n_cons; e++)
      for (int a = 0; a < 4; a++)
        for (int b = 0; b < 4; b++) {
          const int va = cons[4 * e + a], vb = cons[4 * e + b];
          const int s = std::min(sym.sn_of[va], sym.sn_of[vb]);
          const int lr = local_dof(s, va), lc = local_dof(s, vb);
          if (lr < 0 || lc < 0) return -3;
          int ldq = 0;
          const long long dst = panel_addr(fr[s], lr, lc, &ldq);   // one of the two vertices is an own vertex of front s
          if (dst < 0) return -3;
          con_dst[(size_t)e * 16 + a * 4 + b] = dst;
          con_ld[(size_t)e * 16 + a * 4 + b] = ldq;
          con_lvl[(size_t)e * 16 + a * 4 + b] = 2 * sym.level[s] + ((lr < fr[s].pp && lc < fr[s].pp) ? 0 : 1);
        }
    {
      const size_t n = (size_t)n_cons * 16;
      std::vector<int> ord(n);
      for (size_t i = 0; i < n; i++) ord[i] = (int)i;
      std::sort(ord.begin(), ord.end(), [&](int x, int y) {
        if (con_lvl[x] != con_lvl[y]) return con_lvl[x] < con_lvl[y];
        if (con_dst[x] != con_dst[y]) return con_dst[x] < con_dst[y];
        return x < y;
      });
      cgr_lptr.assign(n_levels + 1, 0); cgr_lmid.assign(n_levels, 0); cgr_ptr.clear(); cgr_ent.assign(ord.begin(), ord.end()); cgr_ld.clear(); cgr_dst.clear();
      for (size_t i = 0; i < n; i++) {
        const int x = ord[i];
        if (i == 0 || con_dst[x] != con_dst[ord[i - 1]] || con_lvl[x] != con_lvl[ord[i - 1]]) {
          cgr_ptr.push_back((int)i); cgr_dst.push_back(con_dst[x]); cgr_ld.push_back(con_ld[x]);
          cgr_lptr[(con_lvl[x] >> 1) + 1]++;
          if ((con_lvl[x] & 1) == 0) cgr_lmid[con_lvl[x] >> 1]++;
        }
      }
      cgr_ptr.push_back((int)n);
      for (int l = 0; l < n_levels; l++) { cgr_lmid[l] += cgr_lptr[l]; cgr_lptr[l + 1] += cgr_lptr[l]; }
    }
    return 0;
  }
};
