// Scene group: several scenes of one GPU whose sparse direct solves share their launches (included by tsl_hip.hip after direct_host.hpp).
//
// One Newton iteration of a single 100k-triangle scene fills the chip for a fraction of its 4.6 ms: the block-step chains of the upper tree
// levels, the ~75 dependent launches of the sweeps and the level starts are latency, not work.  The trajectory-optimisation batch of the reference
// (training/trajopt_*.py: independent rollouts of one scene topology) offers the work to fill it: the fronts of S scenes are independent, so the
// fronts of level l of EVERY scene can go through the same launches.  This file makes that a property of the memory layout instead of the kernels:
//   * the members' matrices (SELL values), masked contact blocks, right-hand side and solution vectors, panel / Schur / G arenas and sweep
//     vectors are VIEWS into buffers of the group (DevBuf::view), member s at offset off[s];
//   * the members keep their own plans (built by direct_plan from their own constraint sets, as in a single-scene run);  the group MERGES them:
//     front descriptors, child records and maps concatenated with the offsets shifted by the member's bases, the fronts of a level of all
//     members sorted and cut into batches by the same rules (DirectPlan::build_levels), the level-ordered matrix / contact block lists
//     concatenated level by level;
//   * the merged plan belongs to a pseudo-context `g` whose arenas are the whole group buffers: direct_factor(g) / direct_apply(g, ...) --
//     the unchanged host code and the unchanged kernels -- factorise and apply ALL members at once.
// Every front is computed by the same arithmetic in the same order whatever batch it rides in (k_direct.hpp: all inversion paths and GEMM tilings
// give the same bits; the two sweep kernels do not, so each member's chunks stay with the kernel its own launch would have used), hence a
// member's factors, solutions and tape are BIT-IDENTICAL to its single-scene run.  A member's own plan addresses the same memory: a solve that
// needs more than the merged first pass (refinement, GMRES, the adjoint step) runs on the member's own path without copying anything.
#pragma once
#include <condition_variable>
#include <functional>
#include <thread>
#include "direct_host.hpp"

// Host threads of a group: member i's ~35 assembly launches of a Newton iteration are issued by thread i (the caller's thread takes member 0).
// One thread issuing them member after member paces the GPU -- a launch costs the host 8-10 us, the kernels are shorter: measured 0.87 ms from
// the first to the last assembly kernel of two cfg4 scenes, of which scene 2 waited 0.29 ms for its first launch.
struct GroupPool {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv, cv_done;
  std::function<int(int)> job;
  std::vector<int> rc;
  std::vector<std::string> err;
  int gen = 0, pending = 0, dev = 0;
  bool stop = false;
  void start(int n, int device) {
    dev = device; rc.assign(n, 0); err.assign(n, "");
    for (int i = 1; i < n; i++) th.emplace_back([this, i] {
      (void)hipSetDevice(dev);
      int seen = 0;
      for (;;) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || gen != seen; });
        if (stop) return;
        seen = gen;
        lk.unlock();
        const int r = job(i);
        lk.lock();
        rc[i] = r; if (r) err[i] = g_tsl_err;
        if (--pending == 0) cv_done.notify_one();
      }
    });
  }
  // fn(i) for i in idx, in parallel; the first failure is reported through tsl_fail on the caller's thread
  int run(const std::vector<int>& idx, const std::function<int(int)>& fn) {
    if (idx.empty()) return 0;
    std::vector<int> mine, theirs;
    for (int i : idx) (i == 0 || th.empty() ? mine : theirs).push_back(i);
    if (!theirs.empty()) {
      std::unique_lock<std::mutex> lk(mu);
      std::vector<char> sel(rc.size(), 0);
      for (int i : theirs) sel[i] = 1;
      job = [fn, sel](int i) { return sel[i] ? fn(i) : 0; };
      for (size_t i = 0; i < rc.size(); i++) rc[i] = 0;
      pending = (int)th.size(); gen++;
      lk.unlock();
      cv.notify_all();
    }
    int r0 = 0;
    for (int i : mine) { r0 = fn(i); if (r0) break; }
    if (!theirs.empty()) {
      std::unique_lock<std::mutex> lk(mu);
      cv_done.wait(lk, [&] { return pending == 0; });
      for (size_t i = 1; i < rc.size(); i++) if (rc[i]) return tsl_fail("%s", err[i].c_str());
    }
    return r0;
  }
  ~GroupPool() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv.notify_all();
    for (auto& t : th) t.join();
  }
};

struct tsl_group {
  std::mutex mu_layout;          // group_ensure_arenas (members build their plans side by side)
  std::vector<tsl_ctx*> m;       // members (not owned)
  tsl_ctx* g = nullptr;          // pseudo-context of the merged solver (owned)
  DevBuf<double> vals, cH, vb, vx, arena, sarena, garena, w;
  std::vector<size_t> vals_off, cH_off, vec_off;                 // doubles
  std::vector<int> nv_off, nc_off;                               // rows / constraints in front of member s
  std::vector<size_t> cap_a, cap_s, cap_g, cap_w, off_a, off_s, off_g, off_w;   // capacity and base of every member inside the arenas (doubles)
  std::vector<long> seen_gen;    // plan_gen of every member at the last merge
  bool merged_valid = false;
  bool flow_lost = false;        // a dataflow launch of the merged factorisation lost a flag once: the group stays on the launch-per-block-step path
  std::vector<hipEvent_t> ev_m;  // member stream -> group stream
  hipEvent_t ev_g = nullptr;     // group stream -> member streams
  long n_merge = 0, n_relayout = 0, n_own_path = 0;
  double t_merge = 0;
  std::unique_ptr<GroupPool> pool;
  ~tsl_group();                  // frees the pseudo-context, its streams and the events (defined behind group_destroy)
};

static inline size_t grp_align(size_t n) { return (n + 31) & ~(size_t)31; }   // 256-byte granules

// the arenas of member c must hold its current plan: nothing to do while its capacity suffices, else EVERY member gets a new base
// (25 % headroom for the one that grew), the shared buffers grow if they must, all factors are declared invalid
static int group_ensure_arenas(tsl_ctx* c) {
  tsl_group* G = c->group;
  std::lock_guard<std::mutex> lk(G->mu_layout);
  int me = -1;
  for (size_t i = 0; i < G->m.size(); i++) if (G->m[i] == c) me = (int)i;
  if (me < 0) return tsl_fail("scene group: context is not a member");
  const DirectPlan& P = c->ds.plan;
  if ((size_t)P.arena <= G->cap_a[me] && (size_t)P.sarena <= G->cap_s[me] && (size_t)P.garena <= G->cap_g[me] && (size_t)P.ylen <= G->cap_w[me]) return 0;
  HIP_OK(hipDeviceSynchronize());   // (every member, the group's own streams and the side-stream clears)
  auto grow = [&](std::vector<size_t>& cap, size_t need) { if (need > cap[me]) cap[me] = grp_align(need + need / 4 + 32); };
  grow(G->cap_a, (size_t)P.arena); grow(G->cap_s, (size_t)P.sarena); grow(G->cap_g, (size_t)P.garena); grow(G->cap_w, (size_t)P.ylen);
  const size_t n = G->m.size();
  auto lay = [&](const std::vector<size_t>& cap, std::vector<size_t>& off, DevBuf<double>& buf, const char* what) -> int {
    size_t tot = 0;
    for (size_t i = 0; i < n; i++) { off[i] = tot; tot += cap[i]; }
    if (buf.n < tot) { if (buf.alloc(tot + tot / 16 + 32)) return tsl_fail("scene group: out of device memory (%.2f GB of %s for %zu scenes)", tot * 8e-9, what, n); }
    return 0;
  };
  TSL_TRY(lay(G->cap_a, G->off_a, G->arena, "front panels")); TSL_TRY(lay(G->cap_s, G->off_s, G->sarena, "Schur complements"));
  TSL_TRY(lay(G->cap_g, G->off_g, G->garena, "G panels")); TSL_TRY(lay(G->cap_w, G->off_w, G->w, "sweep vectors"));
  for (size_t i = 0; i < n; i++) {
    DirectSolver& d = G->m[i]->ds;
    d.arena.view(G->arena.p + G->off_a[i], G->cap_a[i]); d.sarena.view(G->sarena.p + G->off_s[i], G->cap_s[i]);
    d.garena.view(G->garena.p + G->off_g[i], G->cap_g[i]); d.w.view(G->w.p + G->off_w[i], G->cap_w[i]);
    d.numeric_valid = false; d.have_factor = false; d.prezero_pending = false;
  }
  DirectSolver& gd = G->g->ds;
  gd.arena.view(G->arena.p, G->arena.n); gd.sarena.view(G->sarena.p, G->sarena.n); gd.garena.view(G->garena.p, G->garena.n); gd.w.view(G->w.p, G->w.n);
  gd.numeric_valid = false; gd.have_factor = false; gd.prezero_pending = false;
  G->merged_valid = false;
  G->n_relayout++;
  return 0;
}

// the merged plan of the members' current plans -> G->g->ds (host tables + device arrays)
static int group_merge(tsl_group* G) {
  const auto t0 = std::chrono::steady_clock::now();
  tsl_ctx* g = G->g;
  DirectSolver& gd = g->ds;
  DirectPlan& M = gd.plan;
  hipStream_t s = g->stream;
  const int n = (int)G->m.size();
  HIP_OK(hipStreamSynchronize(s));
  if (gd.zstream) HIP_OK(hipStreamSynchronize(gd.zstream));
  gd.prezero_pending = false;
  std::vector<int> sn_base(n + 1, 0), pmap_base(n + 1, 0), vtx_base(n + 1, 0), ch_base(n + 1, 0);
  int L = 0, nc_tot = 0;
  for (int i = 0; i < n; i++) {
    const DirectSolver& d = G->m[i]->ds;
    if (!d.plan_valid) return tsl_fail("scene group: member %d has no plan", i);
    const DirectPlan& P = d.plan;
    sn_base[i + 1] = sn_base[i] + (int)P.fr.size(); pmap_base[i + 1] = pmap_base[i] + (int)P.pmap.size();
    vtx_base[i + 1] = vtx_base[i] + (int)P.vtx.size(); ch_base[i + 1] = ch_base[i] + (int)P.ch_rec.size();
    L = std::max(L, P.n_levels);
    nc_tot += G->m[i]->nc;
    if ((long long)pmap_base[i] + (long long)P.pmap.size() > 0x7fffffffLL) return tsl_fail("scene group: child tables exceed 2^31 entries");
  }
  g->nc = nc_tot;
  M.n_cu = G->m[0]->ds.plan.n_cu; M.split_small = G->m[0]->ds.plan.split_small; M.split_rem = G->m[0]->ds.plan.split_rem;
  M.fr.resize(sn_base[n]); M.pmap.resize(pmap_base[n]); M.ch_rec.resize(ch_base[n]);
  std::vector<int> vtxp(vtx_base[n]), owner(sn_base[n]);
  M.arena = (long long)G->arena.n; M.sarena = (long long)G->sarena.n; M.garena = (long long)G->garena.n; M.ylen = (long long)G->w.n;
  M.flops = 0; M.leaf_ranges.clear(); M.arena_leaf = 0;
  std::vector<std::vector<int>> by_level(L);
  std::vector<int> all(n);
  for (int i = 0; i < n; i++) all[i] = i;
  // the members' tables copied side by side (one host thread per member: disjoint ranges of the merged tables)
  TSL_TRY(G->pool->run(all, [&](int i) -> int {
    tsl_ctx* c = G->m[i];
    const DirectPlan& P = c->ds.plan;
    const long long oa = (long long)G->off_a[i], os = (long long)G->off_s[i], og = (long long)G->off_g[i];
    const int ow = (int)G->off_w[i];
    for (size_t q = 0; q < P.fr.size(); q++) {
      DsFrontDesc f = P.fr[q];
      f.off += oa; f.off21 += oa; f.soff += os; f.goff += og;
      if (f.parent >= 0) f.parent += sn_base[i];
      f.pmap_off += pmap_base[i]; f.vtx_off += vtx_base[i]; f.yoff += ow; f.ch_off += ch_base[i];
      for (int k = 0; k < std::min(4, f.nchild); k++) { f.cpm[k] += pmap_base[i]; f.cy[k] += ow; }
      M.fr[sn_base[i] + q] = f;
      owner[sn_base[i] + q] = i;
    }
    std::copy(P.pmap.begin(), P.pmap.end(), M.pmap.begin() + pmap_base[i]);
    for (size_t q = 0; q < P.vtx.size(); q++) vtxp[vtx_base[i] + q] = c->h_rowpos[P.vtx[q]] + G->nv_off[i];
    for (size_t q = 0; q < P.ch_rec.size(); q++) {
      DsChildRec r = P.ch_rec[q];
      r.soff += os; r.pmap_off += pmap_base[i]; r.sn += sn_base[i]; r.yoff += ow;
      M.ch_rec[ch_base[i] + q] = r;
    }
    return 0;
  }));
  for (int i = 0; i < n; i++) {
    const DirectPlan& P = G->m[i]->ds.plan;
    for (int l = 0; l < P.n_levels; l++)
      for (int q = P.level_ptr[l]; q < P.level_ptr[l + 1]; q++) by_level[l].push_back(sn_base[i] + P.level_sn[q]);
    M.flops += P.flops;
    M.leaf_ranges.push_back({(long long)G->off_a[i], P.arena_leaf});
    M.arena_leaf += P.arena_leaf;
  }
  M.build_levels(by_level);
  // sweep work lists again, every level's chunks in two parts: members whose own launch of that level is the plain kernel, then those of the wide
  // kernel (ds_launch_gemv's rule applied to the MEMBER's chunk count: the two kernels sum in different orders)
  {
    const int wb = gd.gemv_wide_below;
    M.wl_front.clear(); M.wl_row.clear(); M.wl_own_ptr.assign(L + 1, 0); M.wl_bnd_ptr.assign(L, 0); M.wl_own_wide.assign(L, 0); M.wl_bnd_wide.assign(L, 0);
    for (int l = 0; l < L; l++) {
      auto wide = [&](int i, bool bnd) {
        const DirectPlan& P = G->m[i]->ds.plan;
        if (l >= P.n_levels) return false;
        const int cnt = bnd ? P.wl_own_ptr[l + 1] - P.wl_bnd_ptr[l] : P.wl_bnd_ptr[l] - P.wl_own_ptr[l];
        return wb > 0 && cnt < wb;
      };
      M.wl_own_ptr[l] = (int)M.wl_front.size();
      for (int part = 0; part < 2; part++) {
        if (part == 1) M.wl_own_wide[l] = (int)M.wl_front.size();
        for (int q = M.level_ptr[l]; q < M.level_ptr[l + 1]; q++) {
          const int sfr = M.level_sn[q];
          if ((int)wide(owner[sfr], false) != part) continue;
          for (int r = 0; r < M.fr[sfr].p; r += 16) { M.wl_front.push_back(sfr); M.wl_row.push_back(r); }
        }
      }
      M.wl_bnd_ptr[l] = (int)M.wl_front.size();
      for (int part = 0; part < 2; part++) {
        if (part == 1) M.wl_bnd_wide[l] = (int)M.wl_front.size();
        for (int q = M.level_ptr[l]; q < M.level_ptr[l + 1]; q++) {
          const int sfr = M.level_sn[q];
          if ((int)wide(owner[sfr], true) != part) continue;
          for (int r = 0; r < M.fr[sfr].b; r += 16) { M.wl_front.push_back(sfr); M.wl_row.push_back(r); }
        }
      }
    }
    M.wl_own_ptr[L] = (int)M.wl_front.size();
  }
  M.level_maxld.assign(L, 0);
  for (int l = 0; l < L; l++) for (int q = M.level_ptr[l]; q < M.level_ptr[l + 1]; q++) M.level_maxld[l] = std::max(M.level_maxld[l], M.fr[M.level_sn[q]].ld);
  // level-ordered matrix blocks (source address in the group's value buffer, destination in the group's panel arena, row stride) and contact groups
  size_t nbk = 0, ngr = 0, nge = 0;
  for (int i = 0; i < n; i++) { const DirectPlan& P = G->m[i]->ds.plan; nbk += P.blk_q.size(); ngr += P.cgr_dst.size(); nge += P.cgr_ent.size(); }
  std::vector<int> src(nbk), ld(nbk), cg_ptr, cg_ent, cg_ld;
  std::vector<long long> dst(nbk), cg_dst;
  cg_ptr.reserve(ngr + 1); cg_ent.reserve(nge); cg_ld.reserve(ngr); cg_dst.reserve(ngr);
  M.blk_lptr.assign(L + 1, 0); M.cgr_lptr.assign(L + 1, 0);
  {
    std::vector<size_t> at((size_t)L * n, 0);   // where member i's blocks of level l start in the merged, level-ordered list
    size_t o = 0;
    for (int l = 0; l < L; l++) {
      for (int i = 0; i < n; i++) {
        tsl_ctx* c = G->m[i];
        const DirectPlan& P = c->ds.plan;
        if (l >= P.n_levels) continue;
        if (G->vals_off[i] + c->vals.n > 0x7fffffffULL) return tsl_fail("scene group: matrix values exceed 2^31 doubles");
        at[(size_t)l * n + i] = o;
        o += (size_t)(P.blk_lptr[l + 1] - P.blk_lptr[l]);
        if (c->nc > 0 && !P.cgr_lptr.empty()) {
          const long long oa = (long long)G->off_a[i];
          const int eo = 16 * G->nc_off[i];
          for (int gq = P.cgr_lptr[l]; gq < P.cgr_lptr[l + 1]; gq++) {
            cg_ptr.push_back((int)cg_ent.size());
            for (int e = P.cgr_ptr[gq]; e < P.cgr_ptr[gq + 1]; e++) cg_ent.push_back(P.cgr_ent[e] + eo);
            cg_dst.push_back(P.cgr_dst[gq] + oa); cg_ld.push_back(P.cgr_ld[gq]);
          }
        }
      }
      M.blk_lptr[l + 1] = (int)o;
      M.cgr_lptr[l + 1] = (int)cg_dst.size();
    }
    cg_ptr.push_back((int)cg_ent.size());
    TSL_TRY(G->pool->run(all, [&](int i) -> int {
      tsl_ctx* c = G->m[i];
      const DirectPlan& P = c->ds.plan;
      const int vo = (int)G->vals_off[i];
      const long long oa = (long long)G->off_a[i];
      for (int l = 0; l < P.n_levels; l++) {
        size_t w = at[(size_t)l * n + i];
        for (int k = P.blk_lptr[l]; k < P.blk_lptr[l + 1]; k++, w++) { const int q = P.blk_q[k]; src[w] = c->ds.h_c2s[q] + vo; dst[w] = P.blk_dst[q] + oa; ld[w] = P.blk_ld[q]; }
      }
      return 0;
    }));
  }
  // uploads (pinned staging arena of the group's solver)
  std::vector<DsFrontDesc> frl(M.level_sn.size());
  for (size_t i = 0; i < frl.size(); i++) frl[i] = M.fr[M.level_sn[i]];
  ds_pin_reset(gd, 16 * nbk + 2 * sizeof(DsFrontDesc) * M.fr.size() + sizeof(DsChildRec) * M.ch_rec.size() +
                       4 * (M.level_sn.size() + M.pmap.size() + vtxp.size() + M.wl_front.size() + M.wl_row.size()) + 20 * (cg_ent.size() + cg_ptr.size()) + 64 * 256);
  TSL_TRY(ds_upload_grow(gd.fr, M.fr, s, &gd)); TSL_TRY(ds_upload_grow(gd.frl, frl, s, &gd)); TSL_TRY(ds_upload_grow(gd.level_sn, M.level_sn, s, &gd));
  TSL_TRY(ds_upload_grow(gd.pmap, M.pmap, s, &gd)); TSL_TRY(ds_upload_grow(gd.ch_rec, M.ch_rec, s, &gd)); TSL_TRY(ds_upload_grow(gd.vtx, vtxp, s, &gd));
  TSL_TRY(ds_upload_grow(gd.blk_q, src, s, &gd)); TSL_TRY(ds_upload_grow(gd.blk_dst, dst, s, &gd)); TSL_TRY(ds_upload_grow(gd.blk_ld, ld, s, &gd));
  TSL_TRY(ds_upload_grow(gd.cgr_ptr, cg_ptr, s, &gd)); TSL_TRY(ds_upload_grow(gd.cgr_ent, cg_ent, s, &gd)); TSL_TRY(ds_upload_grow(gd.cgr_ld, cg_ld, s, &gd)); TSL_TRY(ds_upload_grow(gd.cgr_dst, cg_dst, s, &gd));
  TSL_TRY(ds_upload_grow(gd.wl_front, M.wl_front, s, &gd)); TSL_TRY(ds_upload_grow(gd.wl_row, M.wl_row, s, &gd));
  if (gd.scr.n < (size_t)M.scratch) { if (gd.scr.alloc((size_t)M.scratch + (size_t)M.scratch / 8)) return -1; }
  HIP_OK(hipStreamSynchronize(s));   // (host vectors of this function go out of scope)
  gd.plan_valid = true; gd.numeric_valid = false; gd.have_factor = false;
  for (int i = 0; i < n; i++) G->seen_gen[i] = G->m[i]->ds.plan_gen;
  G->merged_valid = true;
  G->n_merge++;
  G->t_merge += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (g->verbose >= 3)
    for (const DsBatch& b : M.batches) {
      fprintf(stderr, "[tsl]   merged level %2d: %5d fronts, pivots <= %4d, boundary <= %4d, dataflow workgroups %ld (cap %d)%s; fronts by padded pivot count:", b.level, b.count, b.max_pp, b.max_bp,
              ds_flow_wgs(M, b), gd.flow_cap, ds_use_small(gd, b) ? " (LDS kernel)" : "");
      for (int q = 0, run = 0; q < b.count; q++) {
        run++;
        if (q + 1 == b.count || M.fr[M.level_sn[b.first + q + 1]].pp != M.fr[M.level_sn[b.first + q]].pp) { fprintf(stderr, " %d x %d", run, M.fr[M.level_sn[b.first + q]].pp); run = 0; }
      }
      fprintf(stderr, "\n");
    }
  if (g->verbose >= 2)
    fprintf(stderr, "[tsl] scene group: merged plan of %d scenes: %zu fronts, %d levels, %zu batches, %.1f GFLOP per factorisation, %d constraints; host %.2f ms\n", n, M.fr.size(), L, M.batches.size(),
            M.flops * 1e-9, nc_tot, 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  return 0;
}

// Takes the members' matrix / contact-block / vector buffers into the group's memory (contents preserved) and creates the pseudo-context.
static int group_create(tsl_ctx* const* ctxs, int n, tsl_group** out) {
  if (n < 1 || n > 64) return tsl_fail("tsl_group_create: 1..64 scenes");
  int dev = ctxs[0]->ds.device;
  for (int i = 0; i < n; i++) {
    if (ctxs[i]->group) return tsl_fail("tsl_group_create: scene %d is a member of another group", i);
    if (ctxs[i]->ds.device != dev) return tsl_fail("tsl_group_create: the scenes of a group live on one device");
    for (int j = 0; j < i; j++) if (ctxs[j] == ctxs[i]) return tsl_fail("tsl_group_create: scene %d listed twice", i);
  }
  HIP_OK(hipDeviceSynchronize());
  std::unique_ptr<tsl_group> G(new tsl_group());
  G->m.assign(ctxs, ctxs + n);
  G->vals_off.resize(n); G->cH_off.resize(n); G->vec_off.resize(n); G->nv_off.resize(n); G->nc_off.resize(n);
  for (auto* v : {&G->cap_a, &G->cap_s, &G->cap_g, &G->cap_w, &G->off_a, &G->off_s, &G->off_g, &G->off_w}) v->assign(n, 0);
  G->seen_gen.assign(n, -1);
  size_t tv = 0, th = 0, tx = 0;
  int tnv = 0, tnc = 0;
  for (int i = 0; i < n; i++) {
    tsl_ctx* c = ctxs[i];
    G->vals_off[i] = tv; tv += grp_align(c->vals.n);
    G->cH_off[i] = th; G->nc_off[i] = tnc; th += c->c_H.n; tnc += (int)(c->c_H.n / 144);
    G->vec_off[i] = tx; G->nv_off[i] = tnv; tx += 3 * (size_t)c->NV; tnv += c->NV;
    if (c->c_H.n % 144) return tsl_fail("tsl_group_create: contact block buffer of scene %d is not a multiple of 144 doubles", i);
  }
  TSL_TRY(G->vals.alloc(tv + 32)); TSL_TRY(G->cH.alloc(th + 144)); TSL_TRY(G->vb.alloc(tx + 32)); TSL_TRY(G->vx.alloc(tx + 32));
  HIP_OK(hipMemset(G->vals.p, 0, G->vals.n * sizeof(double)));
  // Everything that can fail comes BEFORE a member's buffer is touched (a failure behind the hand-over left the members with views into freed
  // memory): the pseudo-context, its stream, counters and events -- owned by G, whose destructor frees them -- and the copies of the members' contents.
  // pseudo-context: only what direct_factor / direct_apply / ds_dev read
  tsl_ctx* g = new tsl_ctx();
  G->g = g;
  g->NV = tnv; g->nc = 0; g->verbose = ctxs[0]->verbose;
  HIP_OK(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
  g->vals.view(G->vals.p, G->vals.n); g->c_H.view(G->cH.p, G->cH.n);
  DirectSolver& gd = g->ds;
  const DirectSolver& d0 = ctxs[0]->ds;
  // (the LDS kernel takes a batch that fits the chip in `small_rounds` rounds: n scenes bring n times the leaf fronts, and the alternative -- one launch per
  // block step over all of them -- costs the same per front: measured 1.1 ms of k_ds_gj_step for the 1690 leaf fronts of two cfg4 scenes against 0.14 ms per scene in the LDS kernel)
  // (the tuning parameters are read from the members again at every merged solve: group_solve)
  gd.merged = true; gd.enable = 1; gd.device = dev; gd.flow = d0.flow; gd.small_rounds = d0.small_rounds * n; gd.xcd_map = d0.xcd_map; gd.gemv_wide_below = d0.gemv_wide_below;
  gd.g32_below = d0.g32_below; gd.piv_tol = d0.piv_tol; gd.prezero = d0.prezero;
  gd.static_ready = true;
  if (gd.bad.alloc(8 + 4 * DS_BADLOG)) return -1;
  HIP_OK(hipFuncSetAttribute((const void*)k_ds_inv_small, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ds_small_lds(DS_SMALL)));
  HIP_OK(hipEventCreateWithFlags(&G->ev_g, hipEventDisableTiming));
  G->ev_m.assign(n, nullptr);
  for (int i = 0; i < n; i++) HIP_OK(hipEventCreateWithFlags(&G->ev_m[i], hipEventDisableTiming));
  for (int i = 0; i < n; i++) {   // copies (the members keep their buffers until every copy has succeeded)
    tsl_ctx* c = ctxs[i];
    auto copy = [&](const DevBuf<double>& b, double* dstp, size_t cnt) -> int {
      if (b.n && b.p) HIP_OK(hipMemcpy(dstp, b.p, std::min(b.n, cnt) * sizeof(double), hipMemcpyDeviceToDevice));
      return 0;
    };
    TSL_TRY(copy(c->vals, G->vals.p + G->vals_off[i], c->vals.n)); TSL_TRY(copy(c->c_H, G->cH.p + G->cH_off[i], c->c_H.n));
    TSL_TRY(copy(c->v_b, G->vb.p + G->vec_off[i], 3 * (size_t)c->NV)); TSL_TRY(copy(c->v_x, G->vx.p + G->vec_off[i], 3 * (size_t)c->NV));
    DirectSolver& d = c->ds;
    if (d.prezero_pending) { HIP_OK(hipEventSynchronize(d.ev_zero)); d.prezero_pending = false; }
  }
  HIP_OK(hipDeviceSynchronize());
  for (int i = 0; i < n; i++) {   // hand-over: nothing below can fail
    tsl_ctx* c = ctxs[i];
    const size_t nvals = c->vals.n, nch = c->c_H.n;
    c->vals.view(G->vals.p + G->vals_off[i], nvals); c->c_H.view(G->cH.p + G->cH_off[i], nch);
    c->v_b.view(G->vb.p + G->vec_off[i], 3 * (size_t)c->NV); c->v_x.view(G->vx.p + G->vec_off[i], 3 * (size_t)c->NV);
    // the arenas follow at the member's next plan (group_ensure_arenas); what it holds now is dropped
    DirectSolver& d = c->ds;
    d.arena.release(); d.sarena.release(); d.garena.release(); d.w.release();
    d.arena.view(nullptr, 0); d.sarena.view(nullptr, 0); d.garena.view(nullptr, 0); d.w.view(nullptr, 0);
    d.plan_valid = false; d.numeric_valid = false; d.have_factor = false; d.cons_checked = false;
    for (auto& sl : d.cache) sl->used = false;   // (parked plans were sized against the old arenas: rebuilt on demand)
  }
  for (int i = 0; i < n; i++) { ctxs[i]->group = G.get(); ctxs[i]->ds.token_lender = &gd; ctxs[i]->ds.keep_host_maps = true; ds_flow_token_release(ctxs[i]->ds); }
  G->pool.reset(new GroupPool());
  if (n > 1 && !getenv("TSL_GROUP_NO_THREADS")) G->pool->start(n, dev);
  *out = G.release();
  return 0;
}

// gives the members buffers of their own again (contents preserved) and frees the group
static void group_destroy(tsl_group* G) {
  if (!G) return;
  G->pool.reset();
  (void)hipDeviceSynchronize();
  for (tsl_ctx* c : G->m) {
    auto give = [&](DevBuf<double>& b) {
      const double* src = b.p; const size_t cnt = b.n;
      b.view(nullptr, 0); b.release();
      if (cnt && b.alloc(cnt) == 0) (void)hipMemcpy(b.p, src, cnt * sizeof(double), hipMemcpyDeviceToDevice);
      else if (cnt) fprintf(stderr, "[tsl] scene group: out of device memory while a member took back a buffer of %zu doubles: the member is left with an EMPTY buffer (its next call fails loudly)\n", cnt);
    };
    give(c->vals); give(c->c_H); give(c->v_b); give(c->v_x);
    DirectSolver& d = c->ds;
    d.arena.release(); d.sarena.release(); d.garena.release(); d.w.release();
    d.plan_valid = false; d.numeric_valid = false; d.have_factor = false; d.cons_checked = false; d.prezero_pending = false;
    for (auto& sl : d.cache) sl->used = false;
    d.token_lender = nullptr; d.keep_host_maps = false;
    c->group = nullptr;
  }
  delete G;
}

tsl_group::~tsl_group() {
  pool.reset();
  if (g) {
    ds_flow_token_release(g->ds);
    DirectSolver& gd = g->ds;
    if (gd.zstream) (void)hipStreamDestroy(gd.zstream);
    for (int k = 0; k < DS_NSIDE; k++) if (gd.fstream[k]) (void)hipStreamDestroy(gd.fstream[k]);
    if (gd.pin) (void)hipHostFree(gd.pin);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
    g = nullptr;
  }
  for (hipEvent_t e : ev_m) if (e) (void)hipEventDestroy(e);
  if (ev_g) (void)hipEventDestroy(ev_g);
}
