// Host orchestration of the sparse direct preconditioner (included by tsl_hip.hip after k_direct.hpp):
//   direct_factor(c)       plan (if the constraint set changed) + numeric factorisation of the current operator (vals + c_H)
//   direct_apply(c, r, z)  z = (LU)^-1 r on permuted solver vectors
#pragma once
#include <array>
#include <chrono>

#include "k_direct.hpp"
#include "tsl_ctx.hpp"

// The dataflow token of a device: an advisory lock (flock on /dev/shm/tsl_flow_<PCI bus id>; a lock belongs to an open file description, so it
// excludes other contexts of this process and other processes alike).  The persistent launch k_ds_gj_flow needs every one of its workgroups
// resident; two such grids on one device -- two contexts of a process, two ranks on one GPU -- can each fit and together not, and then both
// wait for flags that never come.  So a context launches the kernel only while it holds the token; it asks for it at its first eligible
// factorisation and keeps it until it is destroyed, a context that was refused asks again every 256 factorisations (the holder may be gone)
// and runs the same block steps as one launch each meanwhile.  Every inversion path produces the same bits (k_direct.hpp), so which
// context holds the token changes times, not answers.
#include <cerrno>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>
static void ds_flow_token_acquire(DirectSolver& d) {
  char bus[64] = "dev";
  if (hipDeviceGetPCIBusId(bus, sizeof(bus), d.device) != hipSuccess) { (void)hipGetLastError(); snprintf(bus, sizeof(bus), "dev%d", d.device); }
  for (char* q = bus; *q; q++) if (*q == ':' || *q == '/') *q = '_';
  d.flow_token = -1; d.flow_token_asked = d.n_factor;
  for (const char* dir : {"/dev/shm", "/tmp"}) {
    const std::string path = std::string(dir) + "/tsl_flow_" + bus;
    // an existing file first, WITHOUT O_CREAT: with fs.protected_regular set, O_CREAT on another user's file in a sticky directory fails with EACCES although the
    // file can be opened; only a missing DIRECTORY (ENOENT / ENOTDIR from the create) sends the process to the next directory -- any other failure is a refusal,
    // or two processes would hold "the" token in two directories
    int fd = open(path.c_str(), O_RDWR | O_CLOEXEC);
    if (fd < 0 && errno == ENOENT) {
      fd = open(path.c_str(), O_CREAT | O_RDWR | O_CLOEXEC, 0666);
      if (fd < 0 && (errno == ENOENT || errno == ENOTDIR)) continue;   // no such directory on this box
      if (fd < 0 && errno == EEXIST) fd = open(path.c_str(), O_RDWR | O_CLOEXEC);
    } else if (fd < 0 && errno == ENOTDIR) continue;
    if (fd < 0) return;   // refused (EACCES, EPERM, ...): this context stays on the launch-per-block-step path and asks again later
    (void)fchmod(fd, 0666);   // (another user's process must be able to open it)
    if (flock(fd, LOCK_EX | LOCK_NB) == 0) { d.flow_token = 1; d.flow_token_fd = fd; }
    else close(fd);
    return;   // (the first directory that can hold the file decides: every process looks there first)
  }
}
static void ds_flow_token_release(DirectSolver& d) {
  if (d.flow_token_fd >= 0) { (void)flock(d.flow_token_fd, LOCK_UN); close(d.flow_token_fd); }
  d.flow_token_fd = -1; d.flow_token = 0;
}
static bool ds_flow_token_held(DirectSolver& d) {
  if (d.token_lender) return ds_flow_token_held(*d.token_lender);   // member of a scene group: the group's token (its host thread runs group and members one after the other)
  if (d.flow_token == 0 || (d.flow_token < 0 && d.n_factor - d.flow_token_asked >= 256)) ds_flow_token_acquire(d);
  return d.flow_token > 0;
}

static inline int ds_nblk(long n, int b) { return (int)((n + b - 1) / b); }

// a batch goes to the LDS kernel (all block steps in one launch, one workgroup per front) when its fronts fit the chip in
// `ds_small_rounds` rounds (workgroups per CU by LDS: 4.5 KB of static arrays of the tile inversion on top of the block -- two at 96
// pivots, one at 128); a larger batch has enough tile parallelism for the block-step kernel (round 2: 1024 fronts of 96 pivots 322 us
// at one workgroup per CU against ~190 us tile-parallel)
// ("direct_small_rounds", default 2: 847 leaf fronts of 96 pivots 139.5 us in two rounds of the LDS kernel against 204 us on the block-step path, 5 launches)
static inline bool ds_use_small(const DirectSolver& d, const DsBatch& b) {
  if (b.max_pp > DS_SMALL) return false;
  return (size_t)b.count <= (size_t)d.plan.n_cu * ds_small_per_cu(b.max_pp) * (size_t)d.small_rounds;
}

// G = W F12 (mode 0) / S = sum_children ext(S_child) - F21 G, stored (mode 1) of a batch: 64 x 64 output tiles; "direct_g32_below": G of a
// batch with few 64 x 64 tiles (upper levels) in 32 x 32 tiles, four times the workgroups (the same for the Schur complements was measured
// slower on the leaf levels and no faster above: gone)
// "direct_xcd" (DirectSolver::xcd_map): batches of at least this many fronts launch their GEMM tiles with the XCD-aware map (k_ds_gemm_x: a front per XCD); 0: never
// part (mode 1, the look-ahead of direct_factor): 1 the tiles of the leading lead x lead blocks only (max_lead: the largest of the batch), 2 all others, 0 every tile
static void ds_launch_gemm(hipStream_t s, const DsDev& D, const DsBatch& b, int mode, int ds_xcd_map, int ds_g32_below = 0, int part = 0, int max_lead = 0) {
  const int rows = mode == 0 ? b.max_pp : b.max_bp, cols = b.max_bp;
  const dim3 grid((cols + 63) / 64, (rows + 63) / 64, b.count);
  const long tiles = (long)grid.x * grid.y * grid.z;
  const int ta = (max_lead + 63) / 64;   // (part 1: tile columns [0, ta); the kernels drop the tiles of the other part)
  if (part == 1 && ta == 0) return;
  if (mode == 1 && part != 0) {
    hipLaunchKernelGGL((k_ds_gemm<1, 4>), part == 1 ? dim3(ta, ta, b.count) : grid, dim3(256), 0, s, D, b.first, part);
    return;
  }
  if (mode == 0 && ds_g32_below > 0 && tiles < ds_g32_below) {   // (the kernel is chosen by the size of the whole G: both parts in the same one)
    hipLaunchKernelGGL(k_ds_gemm_g32, dim3(part == 1 ? 2 * ta : (cols + 31) / 32, (rows + 31) / 32, b.count), dim3(256), 0, s, D, b.first, part);
    return;
  }
  if (mode == 0 && part != 0) {
    hipLaunchKernelGGL((k_ds_gemm<0, 4>), part == 1 ? dim3(ta, grid.y, b.count) : grid, dim3(256), 0, s, D, b.first, part);
    return;
  }
  if (ds_xcd_map > 0 && b.count >= ds_xcd_map) {
    const int gx = grid.x, gy = grid.y, nf = b.count;
    const long ng = ((long)nf + 7) / 8 * 8 * gx * gy;
    if (mode == 0) hipLaunchKernelGGL((k_ds_gemm_x<0, 4>), dim3((unsigned)ng), dim3(256), 0, s, D, b.first, gx, gy, nf);
    else hipLaunchKernelGGL((k_ds_gemm_x<1, 4>), dim3((unsigned)ng), dim3(256), 0, s, D, b.first, gx, gy, nf);
    return;
  }
  if (mode == 0) hipLaunchKernelGGL((k_ds_gemm<0, 4>), grid, dim3(256), 0, s, D, b.first, 0);
  else hipLaunchKernelGGL((k_ds_gemm<1, 4>), grid, dim3(256), 0, s, D, b.first, 0);
}
// the panels of the fronts level_sn[lv0 .. lv0 + nf) (one level, or one batch of it) are written from their children's Schur complements
// (part 1 / 2: the pivot blocks only / everything else; max_span: the widest column range of the launch -- ld, pp, or max(ld - pp, pp))
static void ds_launch_extend(hipStream_t s, const DsDev& D, int lv0, int nf, int max_ld, int part = 0, int max_span = 0) {
  const int xspan = part == 1 ? 64 * DS_XU : DS_XSPAN;   // (the pivot blocks alone are few rows x few columns, on the critical path of the look-ahead: one pass per work item)
  const int nsp = ((part == 0 ? max_ld : max_span) + xspan - 1) / xspan;
  if (nsp <= 0) return;
  hipLaunchKernelGGL(k_ds_extend_panels, dim3(((max_ld / 3 + 2 + 3) / 4) * nsp, nf), dim3(256), 0, s, D, lv0, nsp, part, xspan);   // items: local vertices (<= ld / 3) + 2 for the padding rows, four per workgroup
}
// start of level l: its panels are written (level 0: cleared before, see direct_prezero), then its matrix entries are added
// part 1 / 2 (look-ahead; l > 0, a level of one batch b): the pivot blocks F11 -- gather, matrix entries inside F11, identity on the padding, contact groups inside F11 --
// / the panels F12 and F21 with their entries; the block and group lists of a level hold the F11 part first (blk_lmid, cgr_lmid)
static void ds_launch_level_start(tsl_ctx* c, hipStream_t s, const DsDev& D, int l, int part = 0, const DsBatch* b = nullptr) {
  DirectSolver& d = c->ds;
  const DirectPlan& P = d.plan;
  const int lv0 = P.level_ptr[l], nf = P.level_ptr[l + 1] - lv0;
  if (l > 0) ds_launch_extend(s, D, lv0, nf, P.level_maxld[l], part, part == 0 ? 0 : part == 1 ? b->max_pp : std::max(b->max_bp, b->max_pp));
  const int i0 = part == 2 ? P.blk_lmid[l] : P.blk_lptr[l], i1 = part == 1 ? P.blk_lmid[l] : P.blk_lptr[l + 1];
  const int nb = i1 - i0, npad = part == 2 ? 0 : nf;
  const long nt = (long)nb * 9 + (long)npad * DS_T;
  if (nt > 0)
    hipLaunchKernelGGL(k_ds_assemble_level, dim3(ds_nblk(nt, 256)), dim3(256), 0, s, i0, nb, (const int*)d.blk_q.p, (const double*)c->vals.p, (const long long*)d.blk_dst.p,
                       (const int*)d.blk_ld.p, lv0, npad, d.frl.p, d.arena.p, l == 0 ? 1 : 0);
  if (c->nc > 0) {
    const int g0 = part == 2 ? P.cgr_lmid[l] : P.cgr_lptr[l], g1 = part == 1 ? P.cgr_lmid[l] : P.cgr_lptr[l + 1];
    const int ng = g1 - g0;
    if (ng > 0) hipLaunchKernelGGL(k_ds_assemble_contacts_level, dim3(ds_nblk((long)ng * 64, 256)), dim3(256), 0, s, g0, ng, (const int*)d.cgr_ptr.p, (const int*)d.cgr_ent.p,
                                   (const long long*)d.cgr_dst.p, (const int*)d.cgr_ld.p, (const double*)c->c_H.p, d.arena.p);
  }
}


// "direct_flow": the block steps of a batch as ONE persistent dataflow launch (k_ds_gj_flow) -- for ONE batch of a level (no second
// persistent grid next to it; ordinary launches of sibling batches end by themselves), of at most DS_FLOW_MAXF fronts, whose tiles are all resident at once.  Fills the launch
// arguments and grows the exchange buffers; false = the batch stays on the launch-per-block-step path.
// could this batch take the dataflow path (no side effects: direct_bench asks for the earlier batches of a level)?
static long ds_flow_wgs(const DirectPlan& P, const DsBatch& b) {   // workgroups of the batch's dataflow launch: a super-tile of DS_FLOW_B x DS_FLOW_B tiles each
  long n = 0;
  for (int z = 0; z < b.count; z++) { const long ns = (P.fr[P.level_sn[b.first + z]].pp / DS_T + DS_FLOW_B - 1) / DS_FLOW_B; n += ns * ns; }
  return n;
}
static bool ds_flow_eligible(DirectSolver& d, const DirectPlan& P, const DsBatch& b, hipStream_t s) {
  if (!d.flow || b.count > DS_FLOW_MAXF || b.max_pp < 2 * DS_T) return false;
  { hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;   // a captured launch would be replayed with ONE epoch: flags of the previous replay would pass
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false; }
  if (!ds_flow_token_held(d)) return false;   // another context / process launches the persistent kernel on this device
  if (ds_use_small(d, b) && !(d.flow & 2)) return false;   // bit 1: also the batches the LDS kernel would take (64 / 32 fronts of <= 128 pivots on levels 3 and 4 of cfg4)
  if (d.flow_cap == 0) {
    int occ = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_ds_gj_flow<DS_FLOW_B>, 256, 0) != hipSuccess) { d.flow_cap = -1; return false; }
    d.flow_cap = std::max(1, occ * prop.multiProcessorCount);
    if (getenv("TSL_FLOW_DEBUG")) fprintf(stderr, "[tsl] k_ds_gj_flow: %d workgroups per CU x %d CUs resident\n", occ, prop.multiProcessorCount);
  }
  return ds_flow_wgs(P, b) <= d.flow_cap;
}
static bool ds_flow_prepare(DirectSolver& d, const DirectPlan& P, const DsBatch& b, bool flow_free, DsFlowArgs& a, hipStream_t s) {   // false = not on this path
  if (!flow_free || !ds_flow_eligible(d, P, b, s)) return false;
  long wgs = 0, x = 0, fl = 0;
  for (int z = 0; z < b.count; z++) {
    const long nt = P.fr[P.level_sn[b.first + z]].pp / DS_T, ns = (nt + DS_FLOW_B - 1) / DS_FLOW_B;
    a.tile0[z] = (int)wgs; a.xoff[z] = x; a.foff[z] = (int)fl;
    wgs += ns * ns; x += (nt + 2 * nt * nt) * (DS_T * DS_T); fl += 32 * nt + 2 * nt * nt;
  }
  a.tile0[b.count] = (int)wgs; a.nf = b.count;
  if (d.flow_x.n < (size_t)x) { if (d.flow_x.alloc((size_t)x)) return false; }
  if (d.flow_f.n < (size_t)fl) {   // flags start below every epoch
    if (d.flow_f.alloc((size_t)fl + 1024)) return false;
    // on the stream of the launch: a hipMemset on the null stream is not ordered against a non-blocking stream (seen once: the
    // first launch of a fresh context started before the clear had run and lost its flags)
    if (hipMemsetAsync(d.flow_f.p, 0, d.flow_f.n * sizeof(int), s) != hipSuccess) return false;   // (the epoch keeps counting: zero is below every epoch)
  }
  a.epoch = ++d.flow_epoch;
  return true;
}
// The dataflow launches of batch b (empty: not on this path).  A batch that passes every test but the residency -- more fronts than DS_FLOW_MAXF or
// more workgroups than the device holds -- is cut into up to DS_FLOW_PIECES consecutive pieces that run ONE AFTER THE OTHER on the batch's stream
// (round 5, cfg4 level 0: 6 fronts of 160 pivots and 188 of 128 took five launches of k_ds_gj_step at 22-73 us each, every one reading and
// writing all pivot blocks: 290 us; as two dataflow launches the blocks are read and written once).
#define DS_FLOW_PIECES 4
static std::vector<DsBatch> ds_flow_pieces(DirectSolver& d, const DirectPlan& P, const DsBatch& b, bool flow_free, hipStream_t s) {
  std::vector<DsBatch> out;
  if (!flow_free) return out;
  if (ds_flow_eligible(d, P, b, s)) { out.push_back(b); return out; }
  if (ds_use_small(d, b) || b.count < 2) return out;   // (the LDS kernel takes it / a single front that is too large)
  DsBatch one = b;
  one.count = 1;
  if (!ds_flow_eligible(d, P, one, s) || d.flow_cap <= 0) return out;   // (switch, capture, token, a first front that fits: everything but the size of the batch)
  int start = 0;
  while (start < b.count) {
    DsBatch pc{};
    pc.first = b.first + start; pc.level = b.level; pc.act_off = b.act_off;
    long wgs = 0;
    while (start + pc.count < b.count && pc.count < DS_FLOW_MAXF) {
      const DsFrontDesc& f = P.fr[P.level_sn[pc.first + pc.count]];
      const long ns = (f.pp / DS_T + DS_FLOW_B - 1) / DS_FLOW_B;
      if (pc.count > 0 && wgs + ns * ns > d.flow_cap) break;
      wgs += ns * ns; pc.count++;
      pc.max_pp = std::max(pc.max_pp, f.pp); pc.max_bp = std::max(pc.max_bp, f.bp); pc.max_ld = std::max(pc.max_ld, f.ld);
    }
    if (pc.max_pp < 2 * DS_T || !ds_flow_eligible(d, P, pc, s) || (int)out.size() == DS_FLOW_PIECES) { out.clear(); return out; }
    out.push_back(pc);
    start += pc.count;
  }
  return out;
}
// The launch must be resident as a whole (its workgroups wait for each other's flags).  Inside ONE context the host guarantees that by
// running one such launch per level.  ACROSS contexts and processes the device's dataflow token does (ds_flow_token_*, tsl_ctx_create): the
// context that holds it is the only one on the device that launches this kernel, for as long as it lives; every other context runs the same
// block steps as one launch each (k_ds_gj_step) and gets the same bits.  A launch that still cannot become resident (a foreign process that
// does not take part in the protocol) runs into DS_FLOW_SPINS, raises bad[DS_FLOW_ABORT], and the solve refactorises on the
// launch-per-block-step path (solve_perm).
static void ds_flow_launch(hipStream_t s, const DsDev& D, int lv0, const DsFlowArgs& fa, DirectSolver& d) {
  hipLaunchKernelGGL(k_ds_gj_flow<DS_FLOW_B>, dim3(fa.tile0[fa.nf]), dim3(256), 0, s, D, lv0, fa, d.flow_x.p, d.flow_f.p);
}

static bool direct_enabled(tsl_ctx* c) {
  DirectSolver& d = c->ds;
  if (d.enable == 0) return false;
  if (d.enable == 1) return true;
  for (const DsGrid& g : d.grids) if ((long)g.N * g.M >= 1024) return true;
  return false;
}

static DsDev ds_dev(tsl_ctx* c) {
  DirectSolver& d = c->ds;
  DsDev D;
  D.fr = d.fr.p; D.frl = d.frl.p; D.level_sn = d.level_sn.p; D.A = d.arena.p; D.S = d.sarena.p; D.Y = d.w.p; D.G = d.garena.p; D.scr = d.scr.p; D.ch = d.ch_rec.p; D.pmap = d.pmap.p; D.vtx = d.vtx.p; D.bad = d.bad.p; D.dbg = d.dbg; D.piv_tol = d.piv_tol;
  D.tlog = (d.dbg == 30 && d.tlog.n >= 1024) ? d.tlog.p : nullptr;
  return D;
}

// Pinned staging arena of the plan uploads: the maps of a plan (~14 MB on cfg4, of which 11 MB the level-ordered block lists) go through it
// as true asynchronous copies.  ds_pin_reset only after the stream was synchronised (the arena is reused); ds_pin_take returns null when
// the arena has no room (the copy then goes the pageable way).
static void ds_pin_reset(DirectSolver& d, size_t need) {
  if (d.pin_cap < need) {
    if (d.pin) (void)hipHostFree(d.pin);
    d.pin = nullptr; d.pin_cap = 0;
    void* p = nullptr;
    if (hipHostMalloc(&p, need + need / 4) == hipSuccess) { d.pin = (char*)p; d.pin_cap = need + need / 4; }
    else (void)hipGetLastError();
  }
  d.pin_off = 0;
}
static void* ds_pin_take(DirectSolver& d, size_t bytes) {
  const size_t o = (d.pin_off + 255) & ~(size_t)255;
  if (!d.pin || o + bytes > d.pin_cap) return nullptr;
  d.pin_off = o + bytes;
  return d.pin + o;
}
template <class T>
static int ds_upload_grow(DevBuf<T>& buf, const std::vector<T>& h, hipStream_t s, DirectSolver* d = nullptr) {
  if (buf.n < h.size()) { if (buf.alloc(h.size() + h.size() / 4 + 16)) return -1; }
  if (h.empty()) return 0;
  const void* src = h.data();
  if (d) if (void* st = ds_pin_take(*d, h.size() * sizeof(T))) { memcpy(st, h.data(), h.size() * sizeof(T)); src = st; }
  HIP_OK(hipMemcpyAsync(buf.p, src, h.size() * sizeof(T), hipMemcpyHostToDevice, s));
  return 0;
}

// contact maps of the active plan (redone whenever the constraint ORDER changes: build_con)
static int ds_upload_con(DirectSolver& d, hipStream_t s) {
  const DirectPlan& P = d.plan;
  TSL_TRY(ds_upload_grow(d.cgr_ptr, P.cgr_ptr, s, &d)); TSL_TRY(ds_upload_grow(d.cgr_ent, P.cgr_ent, s, &d)); TSL_TRY(ds_upload_grow(d.cgr_ld, P.cgr_ld, s, &d));
  TSL_TRY(ds_upload_grow(d.cgr_dst, P.cgr_dst, s, &d));
  return 0;
}

// once per context: CSR numbering of the static pattern, its SELL addresses, the nested-dissection partition
static int direct_static(tsl_ctx* c) {
  DirectSolver& d = c->ds;
  if (d.static_ready) return 0;
  const int NV = c->NV;
  d.row_ptr.assign(NV + 1, 0);
  for (int v = 0; v < NV; v++) d.row_ptr[v + 1] = d.row_ptr[v] + (int)c->h_rows[v].size();
  std::vector<int> c2s(d.row_ptr[NV]);
  for (int v = 0; v < NV; v++) {
    const int p = c->h_rowpos[v], s = p >> 6, lane = p & 63;
    for (int k = 0; k < (int)c->h_rows[v].size(); k++) c2s[d.row_ptr[v] + k] = (int)(((long)c->h_slice_off[s] + 64L * k) * 9 + lane);
  }
  d.h_c2s = c2s;   // (SELL address of every CSR block: the plan uploads bake it into their level-ordered lists)
  if (d.bad.alloc(8 + 4 * DS_BADLOG + 8)) return -1;   // (+ 8: the note of a dataflow launch that gave up, ds_flow_poll)
  HIP_OK(hipFuncSetAttribute((const void*)k_ds_inv_small, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ds_small_lds(DS_SMALL)));
  { hipDeviceProp_t prop;   // compute units of THIS device: rounds of the LDS kernel, remainder rule of the batches (was a constant 256)
    if (hipGetDeviceProperties(&prop, d.device) == hipSuccess && prop.multiProcessorCount > 0) d.plan.n_cu = prop.multiProcessorCount; else (void)hipGetLastError(); }
  d.plan.sym.build_partition(NV, c->h_rows, d.grids, d.blocks, d.leaf);
  d.cache.clear();   // parked plans belong to the previous partition
  d.static_ready = true;
  d.plan_valid = false;
  return 0;
}

static uint64_t ds_cons_key(const std::vector<int>& cons) {   // FNV-1a over the constraint vertex list
  uint64_t h = 1469598103934665603ull;
  for (int v : cons) { h ^= (uint64_t)(uint32_t)v; h *= 1099511628211ull; }
  return h ^ (uint64_t)cons.size();
}
// active plan <-> cache slot (host plan, constraint list, every device array that belongs to a plan)
static void ds_swap_slot(DirectSolver& d, DsPlanSlot& sl) {
  std::swap(d.plan, sl.plan); d.h_cons.swap(sl.h_cons); d.h_cset.swap(sl.h_cset);
  // the build's scratch stays with the ACTIVE plan (host block maps, child tables, per-thread lists, the static mirror table: 15 MB that only
  // the build and the upload right after it read -- a parked plan does not carry them, a new build does not allocate and fault them in again)
  // (a member of a scene group keeps the maps WITH their plan: the group's merge reads the block lists and child tables of the active plan on the host --
  // a plan that came back from the cache without them merged garbage: NaN factors for every member in the one step of a rollout whose constraint set repeated)
  if (!d.keep_host_maps) { d.plan.blk_dst.swap(sl.plan.blk_dst); d.plan.blk_ld.swap(sl.plan.blk_ld); d.plan.blk_q.swap(sl.plan.blk_q); d.plan.pmap.swap(sl.plan.pmap); }
  d.plan.lists.swap(sl.plan.lists); d.plan.locs.swap(sl.plan.locs); d.plan.tpos.swap(sl.plan.tpos);
  d.level_sn.swap(sl.level_sn); d.pmap.swap(sl.pmap); d.ch_rec.swap(sl.ch_rec); d.vtx.swap(sl.vtx); d.blk_ld.swap(sl.blk_ld); 
  d.wl_front.swap(sl.wl_front); d.wl_row.swap(sl.wl_row); d.blk_dst.swap(sl.blk_dst); d.fr.swap(sl.fr); d.frl.swap(sl.frl); d.blk_q.swap(sl.blk_q); d.cgr_ptr.swap(sl.cgr_ptr); d.cgr_ent.swap(sl.cgr_ent); d.cgr_ld.swap(sl.cgr_ld); d.cgr_dst.swap(sl.cgr_dst);
}

// device + host bytes a parked plan holds (index maps, descriptors, host tree)
static size_t ds_slot_bytes(const DsPlanSlot& sl) {
  const DirectPlan& P = sl.plan;
  size_t b = 4 * (sl.level_sn.n + sl.pmap.n + sl.vtx.n + sl.blk_ld.n + sl.cgr_ptr.n + sl.cgr_ent.n + sl.cgr_ld.n + sl.wl_front.n + sl.wl_row.n + sl.blk_q.n) + 8 * (sl.blk_dst.n + sl.cgr_dst.n) +
             sizeof(DsFrontDesc) * (sl.fr.n + sl.frl.n) + sizeof(DsChildRec) * sl.ch_rec.n;
  b += 4 * (P.pmap.size() + P.vtx.size() + P.blk_ld.size() + P.blk_q.size() + P.wl_front.size() + P.wl_row.size() + P.level_sn.size()) + 8 * P.blk_dst.size() + sizeof(DsFrontDesc) * P.fr.size();
  return b;
}
// the cache is bounded by slots ("direct_plan_cache") AND by bytes ("direct_plan_cache_mb"): least recently used plans are dropped, their
// device arrays released (a plan is ~25 MB on cfg4; a long tape would otherwise pin 64 of them per context)
static void ds_cache_trim(DirectSolver& d) {
  for (;;) {
    size_t total = 0;
    DsPlanSlot* lru = nullptr;
    for (auto& sl : d.cache) if (sl->used) { total += ds_slot_bytes(*sl); if (!lru || sl->stamp < lru->stamp) lru = sl.get(); }
    if (!lru || total <= (size_t)d.cache_mb << 20) return;
    lru->used = false;
    lru->level_sn.release(); lru->pmap.release(); lru->vtx.release(); lru->blk_ld.release(); lru->cgr_ptr.release(); lru->cgr_ent.release(); lru->cgr_ld.release(); lru->cgr_dst.release(); lru->wl_front.release(); lru->wl_row.release();
    lru->blk_q.release(); lru->blk_dst.release(); lru->fr.release(); lru->frl.release(); lru->ch_rec.release();
    d.n_plan_evicted++;
  }
}

// plan for the current constraint set (rebuilt only when the set differs from the one the plan was made for)
static int group_ensure_arenas(tsl_ctx* c);   // direct_group.hpp
static int direct_plan(tsl_ctx* c) {
  DirectSolver& d = c->ds;
  if (d.merged) return 0;   // the merged plan of a scene group is managed by the group
  hipStream_t s = c->stream;
  TSL_TRY(direct_static(c));
  if (d.plan_valid && d.cons_checked) return 0;   // every Newton iteration of a step factorises on the step's constraint set
  std::vector<int> cons((size_t)c->nc * 4);
  if (c->nc > 0) {
    HIP_OK(hipMemcpyAsync(cons.data(), c->c_idx.p, cons.size() * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
  }
  d.cons_checked = true;
  if (d.plan_valid && cons == d.h_cons) return 0;
  // the plan depends on the constraint SET (tree, fronts, maps of the static blocks); the order in which the engine appended the
  // constraints (atomic append: different between the forward detection of a step and the adjoint's re-detection) only enters the
  // map of the contact blocks
  std::vector<std::array<int, 4>> canon((size_t)c->nc);
  for (int e = 0; e < c->nc; e++) canon[e] = {cons[4 * e], cons[4 * e + 1], cons[4 * e + 2], cons[4 * e + 3]};
  std::sort(canon.begin(), canon.end());
  std::vector<int> cset((size_t)c->nc * 4);
  for (int e = 0; e < c->nc; e++) for (int k = 0; k < 4; k++) cset[4 * e + k] = canon[e][k];
  auto same_set_new_order = [&]() -> int {   // active plan is for this set: contact map for the current order
    if (cons != d.h_cons) {
      HIP_OK(hipStreamSynchronize(s));
      if (d.plan.build_con(cons.data(), c->nc)) return tsl_fail("direct solver: constraint vertex outside its front");
      ds_pin_reset(d, 20 * (d.plan.cgr_ent.size() + d.plan.cgr_ptr.size()) + 4096);
      TSL_TRY(ds_upload_con(d, s));
      HIP_OK(hipStreamSynchronize(s));
      d.h_cons = cons;
      d.plan_gen++;
    }
    d.numeric_valid = false; d.have_factor = false;
    return 0;
  };
  if (d.plan_valid && cset == d.h_cset) return same_set_new_order();
  if (d.cache_cap > 0) {
    const uint64_t key = ds_cons_key(cset);
    HIP_OK(hipStreamSynchronize(s));   // the active plan's arrays may still be in use by the previous solve
    for (auto& sl : d.cache)
      if (sl->used && sl->key == key && sl->h_cset == cset) {   // a constraint set seen before: its plan comes back, the active one takes the slot
        ds_swap_slot(d, *sl);
        sl->used = d.plan_valid; sl->key = ds_cons_key(sl->h_cset); sl->stamp = ++d.cache_clock;
        d.plan_valid = true;
        d.n_plan_hits++; d.plan_gen++;
        if (c->group) TSL_TRY(group_ensure_arenas(c));
        if (c->verbose >= 2) fprintf(stderr, "[tsl] direct plan: cached plan reused (nc %d, %d supernodes)\n", c->nc, d.plan.sym.n_sn);
        return same_set_new_order();
      }
    if (d.plan_valid) {   // park the active plan before building the new one in its place
      DsPlanSlot* dst = nullptr;
      for (auto& sl : d.cache) if (!sl->used) { dst = sl.get(); break; }
      if (!dst && (int)d.cache.size() < d.cache_cap) {
        d.cache.emplace_back(new DsPlanSlot());
        dst = d.cache.back().get();
        dst->plan.sym.copy_partition(d.plan.sym); dst->plan.threads = d.plan.threads; dst->plan.n_cu = d.plan.n_cu; dst->plan.split_small = d.plan.split_small; dst->plan.split_rem = d.plan.split_rem;   // the static part a build starts from
      }
      if (!dst) { for (auto& sl : d.cache) if (!dst || sl->stamp < dst->stamp) dst = sl.get(); }
      ds_swap_slot(d, *dst);
      dst->used = true; dst->key = ds_cons_key(dst->h_cset); dst->stamp = ++d.cache_clock;
      d.plan_valid = false;
      ds_cache_trim(d);
    }
  }
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = d.plan.build(c->h_rows, d.row_ptr, cons.data(), c->nc);
  const auto t_build = std::chrono::steady_clock::now();
  if (rc) return tsl_fail("direct solver: inconsistent elimination tree (code %d)", rc);
  DirectPlan& P = d.plan;
  // local vertex -> permuted row of the solver vectors
  std::vector<int> vtxp(P.vtx.size());
  for (size_t i = 0; i < vtxp.size(); i++) vtxp[i] = c->h_rowpos[P.vtx[i]];
  HIP_OK(hipStreamSynchronize(s));  // the previous plan's arrays may still be in use (and the staging arena of the previous upload)
  const size_t nbk = P.blk_q.size();
  ds_pin_reset(d, 16 * nbk + 2 * sizeof(DsFrontDesc) * P.fr.size() + sizeof(DsChildRec) * P.ch_rec.size() +
                      4 * (P.level_sn.size() + P.pmap.size() + vtxp.size() + P.wl_front.size() + P.wl_row.size()) + 20 * (P.cgr_ent.size() + P.cgr_ptr.size()) + 64 * 256);
  TSL_TRY(ds_upload_grow(d.fr, P.fr, s, &d)); TSL_TRY(ds_upload_grow(d.level_sn, P.level_sn, s, &d)); TSL_TRY(ds_upload_grow(d.pmap, P.pmap, s, &d)); TSL_TRY(ds_upload_grow(d.ch_rec, P.ch_rec, s, &d));
  TSL_TRY(ds_upload_grow(d.vtx, vtxp, s, &d));
  // the static blocks in the plan's level order: SELL source address, destination, row stride (the device arrays blk_q / blk_dst / blk_ld),
  // written straight into the staging arena
  {
    std::vector<int> src_v, ld_v;
    std::vector<long long> dst_v;
    int* src_l = (int*)ds_pin_take(d, 4 * nbk); int* ld_l = (int*)ds_pin_take(d, 4 * nbk); long long* dst_l = (long long*)ds_pin_take(d, 8 * nbk);
    if (!src_l || !ld_l || !dst_l) { src_v.resize(nbk); ld_v.resize(nbk); dst_v.resize(nbk); src_l = src_v.data(); ld_l = ld_v.data(); dst_l = dst_v.data(); }
    ds_parallel_for((int)((nbk + 4095) / 4096), P.n_threads(), 1, [&](int, int ch) {
      for (size_t i = (size_t)ch * 4096; i < std::min(nbk, (size_t)(ch + 1) * 4096); i++) { const int q = P.blk_q[i]; src_l[i] = d.h_c2s[q]; dst_l[i] = P.blk_dst[q]; ld_l[i] = P.blk_ld[q]; }
    });
    if (d.blk_dst.n < nbk) { if (d.blk_dst.alloc(nbk + nbk / 4 + 16)) return -1; }
    if (d.blk_ld.n < nbk) { if (d.blk_ld.alloc(nbk + nbk / 4 + 16)) return -1; }
    if (d.blk_q.n < nbk) { if (d.blk_q.alloc(nbk + nbk / 4 + 16)) return -1; }
    if (nbk > 0) {
      HIP_OK(hipMemcpyAsync(d.blk_dst.p, dst_l, 8 * nbk, hipMemcpyHostToDevice, s));
      HIP_OK(hipMemcpyAsync(d.blk_ld.p, ld_l, 4 * nbk, hipMemcpyHostToDevice, s));
      HIP_OK(hipMemcpyAsync(d.blk_q.p, src_l, 4 * nbk, hipMemcpyHostToDevice, s));
      if (!src_v.empty()) HIP_OK(hipStreamSynchronize(s));   // (pageable fallback: the vectors end here)
    }
  }
  TSL_TRY(ds_upload_con(d, s));
  std::vector<DsFrontDesc> frl(P.level_sn.size());
  for (size_t i = 0; i < frl.size(); i++) frl[i] = P.fr[P.level_sn[i]];
  TSL_TRY(ds_upload_grow(d.frl, frl, s, &d));
  TSL_TRY(ds_upload_grow(d.wl_front, P.wl_front, s, &d)); TSL_TRY(ds_upload_grow(d.wl_row, P.wl_row, s, &d));
  if (c->group) TSL_TRY(group_ensure_arenas(c));   // panels, Schur complements, G and the sweeps' boundary vector live in the group's memory
  else {
    if (d.prezero_pending && d.arena.n < (size_t)P.arena) { HIP_OK(hipEventSynchronize(d.ev_zero)); d.prezero_pending = false; }   // the clear runs on the buffer about to be replaced
    if (d.arena.n < (size_t)P.arena) { if (d.arena.alloc((size_t)P.arena + (size_t)P.arena / 8)) return tsl_fail("direct solver: out of device memory (%.2f GB of front panels)", P.arena * 8e-9); }
    if (d.sarena.n < (size_t)P.sarena) { if (d.sarena.alloc((size_t)P.sarena + (size_t)P.sarena / 8 + 16)) return tsl_fail("direct solver: out of device memory (%.2f GB of Schur complements)", P.sarena * 8e-9); }
    if (d.garena.n < (size_t)P.garena) { if (d.garena.alloc((size_t)P.garena + (size_t)P.garena / 8 + 16)) return tsl_fail("direct solver: out of device memory (G arena)"); }
    if (d.w.n < (size_t)P.ylen) { if (d.w.alloc((size_t)P.ylen + (size_t)P.ylen / 4 + 16)) return -1; }
  }
  if (d.scr.n < (size_t)P.scratch) { if (d.scr.alloc((size_t)P.scratch + (size_t)P.scratch / 8)) return -1; }
  HIP_OK(hipStreamSynchronize(s));  // host vectors of this function go out of scope
  d.h_cons.swap(cons); d.h_cset.swap(cset);
  d.plan_valid = true;
  d.numeric_valid = false;
  d.have_factor = false;
  d.n_plans++; d.plan_gen++;
  d.t_plan += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (c->verbose >= 2)
    fprintf(stderr, "[tsl] direct plan: %d supernodes, %d levels, %zu batches, %.2f + %.2f GB of panels + Schur complements, %.1f GFLOP per factorisation, nc %d; host %.2f ms (tree + maps %.2f)\n", P.sym.n_sn, P.n_levels, P.batches.size(), P.arena * 8e-9, P.sarena * 8e-9,
            P.flops * 1e-9, c->nc, 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), 1e3 * std::chrono::duration<double>(t_build - t0).count());
  if (c->verbose >= 2) fprintf(stderr, "[tsl]   plan phases (ms): tree %.2f, descriptors + parent maps %.2f, levels %.2f, static block map %.2f, contact map %.2f\n", P.phase_ms[0], P.phase_ms[1], P.phase_ms[2], P.phase_ms[3], P.phase_ms[4]);
  if (c->verbose >= 3)
    for (const DsBatch& b : P.batches) {
      fprintf(stderr, "[tsl]   level %2d: %5d fronts, pivots <= %4d, boundary <= %4d%s; fronts by padded pivot count:", b.level, b.count, b.max_pp, b.max_bp, ds_use_small(d, b) ? " (LDS kernel)" : "");
      for (int q = 0, run = 0; q < b.count; q++) {   // (sorted by pp, descending)
        run++;
        if (q + 1 == b.count || P.fr[P.level_sn[b.first + q + 1]].pp != P.fr[P.level_sn[b.first + q]].pp) { fprintf(stderr, " %d x %d", run, P.fr[P.level_sn[b.first + q]].pp); run = 0; }
      }
      fprintf(stderr, "\n");
    }
  return 0;
}

// Inside a time step the factors die with the solve of their Newton iteration: the clear of the LEAF level's panels for the next
// factorisation (the contiguous head of the panel arena, ~0.3 GB on cfg4; the panels of every other level are written by the gather of
// their children's Schur complements, the Schur arena is never cleared) starts on a side stream as soon as that solve is done and runs
// next to the line search, the energy evaluations and the next assembly.  The factors are marked invalid here.
static int direct_prezero(tsl_ctx* c) {
  DirectSolver& d = c->ds;
  if (!d.prezero || !d.plan_valid || d.arena.n == 0 || d.prezero_pending) return 0;
  if (d.zstream == nullptr) {
    HIP_OK(hipStreamCreateWithFlags(&d.zstream, hipStreamNonBlocking));   // (a lowest-priority stream made the step 3 % slower: the next factorisation waits for the clear)
    HIP_OK(hipEventCreateWithFlags(&d.ev_zfork, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&d.ev_zero, hipEventDisableTiming));
  }
  HIP_OK(hipEventRecord(d.ev_zfork, c->stream));
  HIP_OK(hipStreamWaitEvent(d.zstream, d.ev_zfork, 0));
  d.prezero_n = (size_t)d.plan.arena_leaf;
  if (d.plan.leaf_ranges.empty()) HIP_OK(hipMemsetAsync(d.arena.p, 0, d.prezero_n * sizeof(double), d.zstream));
  else for (const auto& r : d.plan.leaf_ranges) HIP_OK(hipMemsetAsync(d.arena.p + r.first, 0, (size_t)r.second * sizeof(double), d.zstream));   // (merged plan: every member's leaf panels)
  HIP_OK(hipEventRecord(d.ev_zero, d.zstream));
  d.prezero_pending = true;
  d.numeric_valid = false; d.have_factor = false;
  return 0;
}

// z = (LU)^-1 r, permuted solver vectors (r is not modified; z may not alias r)
// one launch of the level sweeps: chunks wl[o .. o + n) in `mode`; few chunks (upper levels) -> four narrow workgroups per chunk
static void ds_launch_gemv(hipStream_t s, const DsDev& D, DirectSolver& d, int o, int n, int mode, const double* vin, double* vout, int wide_from = -1) {
  if (n <= 0) return;
  if (wide_from >= 0) {   // merged plan of a scene group: chunks [o, wide_from) in the plain kernel, [wide_from, o + n) in the wide one (every member keeps the kernel of its own launch)
    const int n0 = wide_from - o, n1 = o + n - wide_from;
    if (n0 > 0) hipLaunchKernelGGL(k_ds_gemv, dim3(n0), dim3(256), 0, s, D, d.wl_front.p, d.wl_row.p, o, mode, vin, vout);
    if (n1 > 0) hipLaunchKernelGGL(k_ds_gemv_wide, dim3(4 * n1), dim3(256), 0, s, D, d.wl_front.p, d.wl_row.p, wide_from, mode, vin, vout);
    return;
  }
  if (d.gemv_wide_below > 0 && n < d.gemv_wide_below) hipLaunchKernelGGL(k_ds_gemv_wide, dim3(4 * n), dim3(256), 0, s, D, d.wl_front.p, d.wl_row.p, o, mode, vin, vout);
  else hipLaunchKernelGGL(k_ds_gemv, dim3(n), dim3(256), 0, s, D, d.wl_front.p, d.wl_row.p, o, mode, vin, vout);
}
// upward sweep of level l: t = W (r - children) on the own dofs, then y_f on the boundary dofs
static void ds_sweep_up_level(hipStream_t s, const DsDev& D, DirectSolver& d, int l, const double* r, double* z) {
  const DirectPlan& P = d.plan;
  const int o0 = P.wl_own_ptr[l], b0 = P.wl_bnd_ptr[l], o1 = P.wl_own_ptr[l + 1];
  ds_launch_gemv(s, D, d, o0, b0 - o0, 0, r, z, P.wl_own_wide.empty() ? -1 : P.wl_own_wide[l]);
  ds_launch_gemv(s, D, d, b0, o1 - b0, 1, (const double*)z, nullptr, P.wl_bnd_wide.empty() ? -1 : P.wl_bnd_wide[l]);
}
// numeric factorisation of the operator of the last assemble (c->vals + masked contact blocks c->c_H)
// eager_r / eager_z (round 6, with the look-ahead; "direct_lookahead" bit 1): the right-hand side and the result vector of the application that FOLLOWS this factorisation
// (the first pass of the solve the factors are made for).  The upward sweep of every level below the last two runs on the look-ahead's side stream next to the chains of the
// two largest pivot blocks -- ~0.4 ms in which a few workgroups at a time work and the chip is otherwise idle; direct_apply(eager_r, eager_z) then starts at the level they stopped at.
static int direct_factor(tsl_ctx* c, int stop_sn = -1, const char* dump_path = nullptr, const double* eager_r = nullptr, double* eager_z = nullptr) {
  DirectSolver& d = c->ds;
  hipStream_t s = c->stream;
  TSL_TRY(direct_plan(c));
  if (d.numeric_valid) return 0;
  d.eager_n = 0;
  const DirectPlan& P = d.plan;
  const DsDev D = ds_dev(c);
  if (d.prezero_pending) {   // cleared on the side stream since the last solve (direct_prezero)
    HIP_OK(hipStreamWaitEvent(s, d.ev_zero, 0));
    d.prezero_pending = false;
    if (P.leaf_ranges.empty() && d.prezero_n < (size_t)P.arena_leaf) HIP_OK(hipMemsetAsync(d.arena.p + d.prezero_n, 0, ((size_t)P.arena_leaf - d.prezero_n) * sizeof(double), s));   // a new plan with more leaf panels
  } else if (P.leaf_ranges.empty()) HIP_OK(hipMemsetAsync(d.arena.p, 0, (size_t)P.arena_leaf * sizeof(double), s));
  else for (const auto& r : P.leaf_ranges) HIP_OK(hipMemsetAsync(d.arena.p + r.first, 0, (size_t)r.second * sizeof(double), s));
  HIP_OK(hipMemsetAsync(d.bad.p, 0, 8 * sizeof(int), s));
  // (a persistent dataflow launch runs next to ordinary launches of sibling batches -- those end by themselves --, never next to a second one)
  bool swept_at_root = false;
  int flow_wgs_sent = 0;   // workgroups of the dataflow launches of this factorisation so far (bad[DS_FLOW_ARRIVE] counts those that started)
  auto invert_batch = [&](const DsBatch& b, hipStream_t bs, bool& flow_free) {
    const int lv0 = b.first, nf = b.count;
    const int tp = b.max_pp / DS_T;
    bool flowed = false;
    {
      const std::vector<DsBatch> pieces = ds_flow_pieces(d, P, b, flow_free, bs);
      std::vector<DsFlowArgs> fas(pieces.size());
      bool ready = !pieces.empty();
      for (size_t i = 0; ready && i < pieces.size(); i++) ready = ds_flow_prepare(d, P, pieces[i], true, fas[i], bs);   // (every piece's buffers before any piece goes out:
                                                                                                                           // a batch is inverted on ONE path)
      if (ready) {
        for (size_t i = 0; i < pieces.size(); i++) { ds_flow_launch(bs, D, pieces[i].first, fas[i], d); d.n_flow++; flow_wgs_sent += fas[i].tile0[fas[i].nf]; }
        flowed = true; flow_free = false;
      }
    }
    if (flowed) {}
    else if (ds_use_small(d, b)) hipLaunchKernelGGL(k_ds_inv_small, dim3(nf), dim3(256), ds_small_lds(b.max_pp), bs, D, lv0, b.max_pp + 1);
    else {
      hipLaunchKernelGGL(k_ds_pivot0, dim3(nf), dim3(256), 0, bs, D, lv0);
      for (int k = 0; k < tp; k++) { const int na = P.act_n[b.act_off + k]; hipLaunchKernelGGL(k_ds_gj_step, dim3(na + na * tp * tp), dim3(256), 0, bs, D, lv0, k, tp, na); }   // fronts are sorted by pp: the active ones are a prefix
      hipLaunchKernelGGL(k_ds_gj_finish, dim3(tp, nf), dim3(256), 0, bs, D, lv0);
    }
  };
  auto run_batch = [&](const DsBatch& b, hipStream_t bs, bool& flow_free) {
    invert_batch(b, bs, flow_free);
    if (b.max_bp > 0) {
      ds_launch_gemm(bs, D, b, 0, d.xcd_map, d.g32_below);
      ds_launch_gemm(bs, D, b, 1, d.xcd_map);
    }
  };
  auto side_streams = [&]() -> int {
    if (d.fstream[0] == nullptr)
      for (int k = 0; k < DS_NSIDE; k++) { HIP_OK(hipStreamCreateWithFlags(&d.fstream[k], hipStreamNonBlocking)); HIP_OK(hipEventCreateWithFlags(&d.ev_fjoin[k], hipEventDisableTiming)); }
    if (d.ev_ffork == nullptr) HIP_OK(hipEventCreateWithFlags(&d.ev_ffork, hipEventDisableTiming));
    for (int k = 0; k < 5; k++) if (d.ev_la[k] == nullptr) HIP_OK(hipEventCreateWithFlags(&d.ev_la[k], hipEventDisableTiming));
    if (d.lastream == nullptr) {   // the side stream of the look-ahead at the LOWEST priority: what it carries is filler next to the chain on the engine stream
      int lo = 0, hi = 0;
      HIP_OK(hipDeviceGetStreamPriorityRange(&lo, &hi));
      HIP_OK(hipStreamCreateWithPriority(&d.lastream, hipStreamNonBlocking, lo));   // (measured: at normal priority 223 ms per step against 201 -- it then shares a hardware queue with the engine stream)
    }
    return 0;
  };
  // LOOK-AHEAD over the levels of one batch each (the chain of single launches towards the root; "direct_lookahead", round 6).  There the engine stream ran
  // inversion -> G -> Schur complement -> gather of the next level -> inversion ... one after the other: a latency-bound chain of block steps (a few workgroups busy at a
  // time) taking turns with a matrix-core-bound launch.  But the parent's pivot block F11 receives only the LEADING block of a child's S (rows and columns that are own dofs
  // of the parent: the first DsFrontDesc.lead of the boundary, sorted by elimination position -- a quarter of S two levels below cfg4's root, 16-45 % below that).  So:
  //   engine stream:  ... G_l, [S_l leading tiles], gather + entries of F11_{l+1}, inversion_{l+1}, (wait) G_{l+1}, ...
  //   side stream:        (after G_l) [S_l other tiles], gather + entries of F12 / F21 of level l + 1, ...
  // Every tile / panel entry is the same arithmetic whichever launch forms it: the factors keep their bits.  The persistent inversion launch next to the Schur tiles
  // is safe: the GEMM workgroups wait for nothing and end by themselves, the inversion's become resident as they drain.
  // (only in the context that holds the device's dataflow token: the drain rule and the gate below protect THIS context's persistent launches from its own side stream; the
  // low-priority workgroups of a context without the token would arrive next to the token holder's launches unguarded)
  const bool la_on = (d.lookahead & 1) && stop_sn < 0 && !P.blk_lmid.empty() && P.la_from < P.n_levels && ds_flow_token_held(d);
  if (la_on) TSL_TRY(side_streams());
  hipStream_t ls = d.lastream;
  enum { LA_W = 0, LA_GA = 1, LA_A = 2, LA_REST = 3, LA_DRAIN = 4 };   // engine stream: W stored, leading columns of G stored, leading Schur tiles stored; side stream: F12 / F21 of the next level written
  // The fronts of a level are independent: where a level was split into batches (by pivot-block size) the batches run on parallel
  // streams -- the latency-bound one (a few fronts in the LDS kernel, or the block steps of a handful of larger fronts) next to the
  // throughput-bound one (a thousand leaves).
  for (size_t bi = 0; bi < P.batches.size();) {
    size_t be = bi;
    while (be < P.batches.size() && P.batches[be].level == P.batches[bi].level) be++;
    bool stop_here = false;
    if (stop_sn >= 0) for (int q = P.batches[bi].first; q < P.batches[be - 1].first + P.batches[be - 1].count; q++) stop_here |= P.level_sn[q] == stop_sn;
    const int lvl = P.batches[bi].level;
    const bool la_level = la_on && lvl >= P.la_from;   // (one batch)
    const bool la_in = la_level && lvl > P.la_from;    // the Schur launches of the level below were split: F11 first, the other panels on the side stream
    if (la_in) {
      ds_launch_level_start(c, s, D, lvl, 1, &P.batches[bi]);
      HIP_OK(hipStreamWaitEvent(ls, d.ev_la[LA_A], 0));   // (tiles of the leading blocks hold entries of F12 / F21 too where lead is no multiple of the tile)
      ds_launch_level_start(c, ls, D, lvl, 2, &P.batches[bi]);
      HIP_OK(hipEventRecord(d.ev_la[LA_REST], ls));
    } else
    ds_launch_level_start(c, s, D, lvl);   // (the children's Schur complements of every lower level are stored)
    if (la_level) {
      const DsBatch& b = P.batches[bi];
      bool flow_free = true;
      // A dataflow launch of more workgroups than CUs needs SECOND slots on CUs that already hold one of its (waiting) workgroups.  Measured (round 6): next to queued
      // side-stream work -- thousands of small workgroups of the eager sweeps -- the last few workgroups of cfg4's root launch (289) were then not started for seconds,
      // once in ~100 time steps (DS_FLOW_SPINS ran out: 0.5 s and the context's dataflow path lost); launches of at most one workgroup per CU never were.  Such a
      // launch therefore starts with the side stream drained.
      // (Also measured: EVERY dataflow launch held back until the eager sweeps queued before it have ended -- 195.1 instead of 191.8 ms per step; not kept: in ~650 time
      // steps of cfg4 with the rule above alone no launch of at most one workgroup per CU stalled, against one stall per ~100 steps without it.)
      if (la_in && ds_flow_wgs(P, b) > P.n_cu) {
        HIP_OK(hipEventRecord(d.ev_la[LA_DRAIN], ls));
        HIP_OK(hipStreamWaitEvent(s, d.ev_la[LA_DRAIN], 0));
      }
      invert_batch(b, s, flow_free);
      if (la_in && eager_r != nullptr && (d.lookahead & 2) && lvl >= P.n_levels - 3) {
        // (the side stream has this level's F12 / F21 behind it and, through LA_A, every inversion below; the root's LA_REST, which the engine stream waits for, follows the sweeps)
        // in up to three parts: levels [0, c0) next to the chain of the third level from the top, [c0, c1) next to the chain below the root, [c1, root) next to the root's
        // chain.  Every part sits behind a gate (k_ds_flow_gate): the level's own chain is resident before the part's workgroups arrive; the root's launch (more workgroups
        // than CUs) also starts with the side stream drained (above), so that nothing of an earlier part is in flight when it is dispatched.
        // (cfg4, same-box pairs of 20 + 5 steps: c0 / c1 = 2 / none 193.0, 0 / 3 191.6, 1 / 3 191.5, 1 / 2 193.1, 1 / 5 193.1, 2 / 4 191.8 ms per step; 420 time steps of stress with 1 / 3: no stall)
        const int top = P.n_levels - 1;
        const int c0 = std::min((d.lookahead >> 2) & 7, top - 2);
        const int c1r = (d.lookahead >> 5) & 15, c1 = c1r == 0 ? top - 1 : std::min(std::max(c1r, c0), top - 1);
        const int stage = lvl - (top - 2);   // 0, 1, 2
        const int lo = stage == 0 ? 0 : stage == 1 ? c0 : c1, hi = stage == 0 ? c0 : stage == 1 ? c1 : (c1r == 0 ? c1 : top);
        if (hi > lo && lo == d.eager_n && (stage > 0 || P.la_from < lvl)) {
          hipLaunchKernelGGL(k_ds_flow_gate, dim3(1), dim3(64), 0, ls, (const int*)(d.bad.p + DS_FLOW_ARRIVE), flow_wgs_sent);   // (the chain of this level is resident before the flood)
          for (int l = lo; l < hi; l++) ds_sweep_up_level(ls, D, d, l, eager_r, eager_z);
          d.eager_n = hi; d.eager_r = eager_r; d.eager_z = eager_z;
          if (stage == 2) { HIP_OK(hipEventRecord(d.ev_la[LA_DRAIN], ls)); swept_at_root = true; }   // (joined below: the application goes on from level eager_n on the engine stream)
        }
      }
      if (la_in) HIP_OK(hipStreamWaitEvent(s, d.ev_la[LA_REST], 0));
      if (b.max_bp > 0 && be < P.batches.size()) {
        // engine stream: the columns of G under the leading Schur tiles, those tiles; side stream: the other columns of G, the other tiles
        int max_lead = 0;
        for (int q = 0; q < b.count; q++) max_lead = std::max(max_lead, P.fr[P.level_sn[b.first + q]].lead);
        HIP_OK(hipEventRecord(d.ev_la[LA_W], s));
        ds_launch_gemm(s, D, b, 0, 0, d.g32_below, 1, max_lead);
        HIP_OK(hipEventRecord(d.ev_la[LA_GA], s));
        ds_launch_gemm(s, D, b, 1, 0, 0, 1, max_lead);
        HIP_OK(hipEventRecord(d.ev_la[LA_A], s));
        HIP_OK(hipStreamWaitEvent(ls, d.ev_la[LA_W], 0));
        ds_launch_gemm(ls, D, b, 0, 0, d.g32_below, 2, max_lead);
        HIP_OK(hipStreamWaitEvent(ls, d.ev_la[LA_GA], 0));
        ds_launch_gemm(ls, D, b, 1, 0, 0, 2, max_lead);
      } else if (b.max_bp > 0) {
        ds_launch_gemm(s, D, b, 0, d.xcd_map, d.g32_below);
        ds_launch_gemm(s, D, b, 1, d.xcd_map);
      }
      bi = be;
      continue;
    }
    if (stop_here) {   // diagnostic: the top rows of the assembled front stop_sn (children added, not yet factorised) -> file
      const DsFrontDesc& f = P.fr[stop_sn];
      std::vector<double> h((size_t)f.pp * f.ld);
      HIP_OK(hipStreamSynchronize(s));
      HIP_OK(hipMemcpy(h.data(), d.arena.p + f.off, h.size() * sizeof(double), hipMemcpyDeviceToHost));
      if (FILE* fp = fopen(dump_path, "wb")) {
        const int hdr[8] = {f.p, f.pp, f.b, f.bp, f.ld, f.nv_own, f.nv_bnd, 0};
        fwrite(hdr, sizeof(int), 8, fp);
        fwrite(&P.vtx[f.vtx_off], sizeof(int), (size_t)f.nv_own + f.nv_bnd, fp);
        fwrite(h.data(), sizeof(double), h.size(), fp);
        fclose(fp);
      }
      d.numeric_valid = false; d.have_factor = false;
      return 0;
    }
    const int nside = (int)std::min<size_t>(be - bi - 1, DS_NSIDE);
    if (nside > 0) {
      TSL_TRY(side_streams());
      HIP_OK(hipEventRecord(d.ev_ffork, s));
      for (int k = 0; k < nside; k++) HIP_OK(hipStreamWaitEvent(d.fstream[k], d.ev_ffork, 0));
    }
    bool flow_free = true;
    for (size_t q = bi; q < be; q++) {
      const int k = (int)(q - bi);   // batch 0 of the level (the largest pivot blocks: the longest chain of block steps) stays on the engine stream
      run_batch(P.batches[q], (nside > 0 && k > 0) ? d.fstream[(k - 1) % nside] : s, flow_free);
    }
    for (int k = 0; k < nside; k++) { HIP_OK(hipEventRecord(d.ev_fjoin[k], d.fstream[k])); HIP_OK(hipStreamWaitEvent(s, d.ev_fjoin[k], 0)); }
    bi = be;
  }
  // |H|_inf, the yardstick of the backward errors, is formed when a refinement first asks for it (direct_anorm: a 23-us kernel and a synchronisation)
  // and kept for the 64 factorisations that follow -- the Newton iterations of a time step: the norm (m / dt^2 + the elastic stiffness) moves by a
  // few per cent between them, the quantity it is compared with spans five decades
  if (++d.anorm_age >= 64) d.anorm_valid = false;
  if (stop_sn >= 0 || c->verbose >= 2) HIP_OK(hipStreamSynchronize(s));
  if (d.dbg == 30) {   // diagnostic: when did the pivots of the LAST dataflow launch's first front get published, when did two far workgroups finish each step
    if (d.tlog.n < 1024) { if (d.tlog.alloc(1024)) return -1; HIP_OK(hipMemset(d.tlog.p, 0, 1024 * sizeof(unsigned long long))); HIP_OK(hipMemset(d.tlog.p + 194, 0xff, sizeof(unsigned long long))); }
    else {
      unsigned long long h[1024];
      HIP_OK(hipStreamSynchronize(s));
      HIP_OK(hipMemcpy(h, d.tlog.p, sizeof(h), hipMemcpyDeviceToHost));
      int khz = 100000;
      (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, d.device);
      auto us = [&](unsigned long long t) { return t ? (double)(long long)(t - h[192]) * 1e3 / khz : -1.0; };
      fprintf(stderr, "[tsl] dataflow chain of the last launch (us since its start): pivot published | step finished by the last workgroup | by the first of the last super-row; first workgroup started at %.2f, last ended at %.2f\n", us(h[194]), us(h[193]));
      for (int k = 0; k < 64 && (h[k] || h[64 + k]); k++) {
        fprintf(stderr, "[tsl]   %2d  %8.2f (+%5.2f)   %8.2f   %8.2f   | owner of the pivot, relative to the publication of the previous one: flags seen, operands in LDS, R' done, tile updated, inverted:", k, us(h[k]), k ? us(h[k]) - us(h[k - 1]) : 0.0, us(h[64 + k]), us(h[128 + k]));
        for (int i = 0; i < 5; i++) fprintf(stderr, " %6.2f", h[256 + 8 * k + i] && k ? us(h[256 + 8 * k + i]) - us(h[k - 1]) : 0.0);
        fprintf(stderr, "\n");
      }
      HIP_OK(hipMemset(d.tlog.p, 0, 1024 * sizeof(unsigned long long)));
      HIP_OK(hipMemset(d.tlog.p + 194, 0xff, sizeof(unsigned long long)));
    }
  }
  if (swept_at_root) HIP_OK(hipStreamWaitEvent(s, d.ev_la[LA_DRAIN], 0));
  HIP_OK(hipGetLastError());
  d.flow_wgs_last = flow_wgs_sent;
  d.numeric_valid = true;
  d.have_factor = true;
  d.n_factor++;
  return 0;
}

// |H|_inf of the operator of the last assemble (static part): the yardstick of the solve's normwise backward error.  Only
// refinements that stall above cg_tol need it (a few adjoint solves of a rollout), so it is computed on demand -- 23 us of kernel
// plus a clear and a copy at the end of EVERY factorisation before.
static int direct_anorm(tsl_ctx* c) {
  DirectSolver& d = c->ds;
  if (d.anorm_valid) return 0;
  hipStream_t s = c->stream;
  if (d.anorm_dev.n == 0 && d.anorm_dev.alloc(1)) return -1;
  if (d.h_anorm == nullptr) HIP_OK(hipHostMalloc((void**)&d.h_anorm, sizeof(double)));
  HIP_OK(hipMemsetAsync(d.anorm_dev.p, 0, sizeof(double), s));
  hipLaunchKernelGGL(k_ds_rownorm, dim3(ds_nblk(c->NV, 256)), dim3(256), 0, s, c->NV, c->slice_off.p, c->slice_len.p, c->vals.p, d.anorm_dev.p);
  HIP_OK(hipMemcpyAsync(d.h_anorm, d.anorm_dev.p, sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  d.anorm = *d.h_anorm;
  d.anorm_valid = true; d.anorm_age = 0;
  return 0;
}

// z = (LU)^-1 r (permuted solver vectors; r is not modified, z may not alias r).  Measured and dropped (round 4): the upward half of the
// first application on a side stream next to the factorisation of the levels above ("eager" right-hand side) -- 268-271 ms per step with
// the lower 3 / 5 levels overlapped, 288 ms with every level, against 269 without: the sweeps slow the block-step chains down by what they take.
static int direct_apply(tsl_ctx* c, const double* r, double* z) {
  DirectSolver& d = c->ds;
  hipStream_t s = c->stream;
  const DirectPlan& P = d.plan;
  const DsDev D = ds_dev(c);
  const int l0 = (d.eager_n > 0 && r == d.eager_r && z == d.eager_z) ? d.eager_n : 0;   // (levels swept next to the factorisation, direct_factor)
  d.eager_n = 0;
  for (int l = l0; l < P.n_levels; l++) ds_sweep_up_level(s, D, d, l, r, z);
  for (int l = P.n_levels - 2; l >= 0; l--) {   // the top level has no boundary
    const int o0 = P.wl_own_ptr[l], b0 = P.wl_bnd_ptr[l];
    ds_launch_gemv(s, D, d, o0, b0 - o0, 2, (const double*)z, z, P.wl_own_wide.empty() ? -1 : P.wl_own_wide[l]);
  }
  d.n_apply++;
  return 0;
}

// Timing of one kernel class of the factorisation / solve for bench.py's roofline object: the launches of that class of ONE
// factorisation (or one application), exactly as direct_factor / direct_apply issue them on the current plan, replayed `reps` times
// back to back between one hipEvent pair on the engine stream.  cls: 0 the Gauss-Jordan inversions W = F11^-1 on the block-step path
// (k_ds_pivot0 + k_ds_gj_step + k_ds_gj_finish), 1 k_ds_gemm mode 1 (Schur complements: product + gather of the children + store),
// 2 G = W F12, 3 the inversions of the batches in the LDS kernel (k_ds_inv_small), 4 k_ds_gemv (all sweeps of one application),
// 5 the inversions of the batches in the dataflow kernel (k_ds_gj_flow), 6 k_ds_extend_panels (the panels' share of the extend-add),
// 7 (probe, with "ds_bench_batch" b) the dataflow inversion of batch b next to G + Schur GEMM of batch b - 1 on a side stream (scripts/probe_overlap.py).
// The replays overwrite the factors (marked invalid afterwards).
// out: {us per launch, algorithmic flops per launch, algorithmic bytes per launch, launches per factorisation / application}
static int direct_bench(tsl_ctx* c, int cls, int reps, double* out) {
  DirectSolver& d = c->ds;
  hipStream_t s = c->stream;
  if (!d.plan_valid || d.arena.n == 0) return tsl_fail("tsl_bench_direct: no factorisation yet");
  if (d.prezero_pending) { HIP_OK(hipStreamWaitEvent(s, d.ev_zero, 0)); d.prezero_pending = false; }
  const DirectPlan& P = d.plan;
  // Inside a time step the panel arena is cleared right after every solve: panels, Schur complements and G get a finite non-zero
  // pattern (bytes 0x3F = 4.8e-4) unless real factors are in place.
  if (!d.have_factor) {
    HIP_OK(hipMemsetAsync(d.arena.p, 0x3F, (size_t)P.arena * sizeof(double), s));
    HIP_OK(hipMemsetAsync(d.sarena.p, 0x3F, (size_t)P.sarena * sizeof(double), s));
    HIP_OK(hipMemsetAsync(d.garena.p, 0x3F, (size_t)P.garena * sizeof(double), s));
  }
  const DsDev D = ds_dev(c);
  if (cls == 0 || cls == 3 || cls == 5) {
    // the inversions are replayed in place, each on the output of the one before: on I + 4.8e-4 (ones) -- diagonally dominant up to 2000 pivots, as is its inverse -- no tile trips
    // the inversion's guards.  (Round 6: the replays used to run on the REAL factors, and the inverse of an inverse of cfg4's contact-stiff pivot blocks sent nearly every
    // tile through the guarded form as well: 137 us per dataflow launch in the replays against 100 on these data and 97-113 in situ.)
    HIP_OK(hipMemsetAsync(d.arena.p, 0x3F, (size_t)P.arena * sizeof(double), s));
    hipLaunchKernelGGL(k_ds_bench_diag, dim3((unsigned)P.fr.size()), dim3(256), 0, s, D);
    d.have_factor = false;
  }
  double flops = 0, bytes = 0;
  long launches = 0;
  // entries of the children's Schur complements that land in the boundary part (-> S) / in the panels of front f
  auto child_entries = [&](const DsFrontDesc& f, double& to_s, double& to_panels) {
    to_s = 0; to_panels = 0;
    for (int q = f.ch_off; q < f.ch_off + f.nchild; q++) {
      const DsChildRec& cr = P.ch_rec[q];
      const double nb = cr.nb, b = cr.b;
      to_s += nb * nb; to_panels += b * b - nb * nb;
    }
  };
  auto issue = [&](bool count) {
    if (cls == 4) {   // one application as direct_apply issues it (the right-hand side copy included)
      (void)direct_apply(c, c->v_b.p, c->v_t4.p);
      d.n_apply--;
      if (count) {
        for (int l = 0; l < P.n_levels; l++) launches += 1 + (P.wl_own_ptr[l + 1] > P.wl_bnd_ptr[l]) + (l < P.n_levels - 1);
        for (const DsFrontDesc& f : P.fr) { bytes += 8.0 * ((double)f.p * f.p + 2.0 * (double)f.p * f.b); flops += 2.0 * ((double)f.p * f.p + 2.0 * (double)f.p * f.b); }
      }
      return;
    }
    if (cls == 7) {   // probe: the dataflow inversion of batch "ds_bench_batch" on the engine stream NEXT TO G and the Schur GEMM of the batch before it on a side stream
      const int bq = d.bench_batch;
      if (bq < 1 || bq >= (int)P.batches.size()) return;
      const DsBatch& bf = P.batches[bq];
      const DsBatch& bg = P.batches[bq - 1];
      DsFlowArgs fa;
      if (!ds_flow_prepare(d, P, bf, true, fa, s) || bg.max_bp == 0) return;
      (void)hipEventRecord(d.ev_ffork, s);
      (void)hipStreamWaitEvent(d.fstream[0], d.ev_ffork, 0);
      ds_flow_launch(s, D, bf.first, fa, d);
      ds_launch_gemm(d.fstream[0], D, bg, 0, d.xcd_map, d.g32_below);
      ds_launch_gemm(d.fstream[0], D, bg, 1, d.xcd_map);
      (void)hipEventRecord(d.ev_fjoin[0], d.fstream[0]);
      (void)hipStreamWaitEvent(s, d.ev_fjoin[0], 0);
      if (count) launches++;
      return;
    }
    int bi = -1;
    for (const DsBatch& b : P.batches) {
      bi++;
      if (d.bench_batch >= 0 && bi != d.bench_batch) continue;   // "ds_bench_batch": one batch only
      const int lv0 = b.first, nf = b.count;
      const int tp = b.max_pp / DS_T, tb = b.max_bp / DS_T;
      if (cls == 6) {
        if (b.level == 0) continue;
        ds_launch_extend(s, D, b.first, b.count, b.max_pp + b.max_bp);
        if (count) {
          launches++;
          for (int i = 0; i < nf; i++) {
            const DsFrontDesc& f = P.fr[P.level_sn[lv0 + i]];
            double to_s, to_p;
            child_entries(f, to_s, to_p);
            bytes += 8.0 * (to_p + (double)f.pp * f.ld + (double)f.bp * f.pp); flops += to_p;   // the children's entries read, every panel entry written once
          }
        }
      } else if (cls == 0 || cls == 3 || cls == 5) {   // W = F11^-1: cls 0 the batches on the block-step path (pivot0 + block steps + finish), cls 3 the batches in the LDS kernel, cls 5 those in the dataflow kernel
        bool flow_free = true;   // (one dataflow launch per level: the first batch of the level that can take it)
        for (int q = 0; q < bi; q++) if (P.batches[q].level == b.level && !ds_flow_pieces(d, P, P.batches[q], true, s).empty()) flow_free = false;
        const std::vector<DsBatch> pieces = ds_flow_pieces(d, P, b, flow_free, s);
        const int mine = !pieces.empty() ? 5 : (ds_use_small(d, b) ? 3 : 0);   // the class direct_factor runs this batch in
        if (mine != cls) continue;
        if (cls == 5) { for (const DsBatch& pc : pieces) { DsFlowArgs fa; if (ds_flow_prepare(d, P, pc, true, fa, s)) { ds_flow_launch(s, D, pc.first, fa, d); if (count) launches++; } } }
        else if (cls == 3) { hipLaunchKernelGGL(k_ds_inv_small, dim3(nf), dim3(256), ds_small_lds(b.max_pp), s, D, lv0, b.max_pp + 1); if (count) launches++; }
        else {
          hipLaunchKernelGGL(k_ds_pivot0, dim3(nf), dim3(256), 0, s, D, lv0);
          for (int k = 0; k < tp; k++) { const int na = P.act_n[b.act_off + k]; hipLaunchKernelGGL(k_ds_gj_step, dim3(na + na * tp * tp), dim3(256), 0, s, D, lv0, k, tp, na); }
          hipLaunchKernelGGL(k_ds_gj_finish, dim3(tp, nf), dim3(256), 0, s, D, lv0);
          if (count) launches += tp + 2;
        }
        if (count) for (int i = 0; i < nf; i++) {
          const DsFrontDesc& f = P.fr[P.level_sn[lv0 + i]];
          flops += 2.0 * (double)f.pp * f.pp * f.pp;
          bytes += 16.0 * (double)f.pp * f.pp * (cls != 0 ? 1.0 : f.pp / (double)DS_T);   // the block read and written once per launch that touches it
        }
      } else if (tb > 0) {
        if (cls == 1 || cls == 2) ds_launch_gemm(s, D, b, cls == 2 ? 0 : 1, d.xcd_map, cls == 2 ? d.g32_below : 0);
        else continue;
        if (count) {
          launches++;
          for (int i = 0; i < nf; i++) {
            const DsFrontDesc& f = P.fr[P.level_sn[lv0 + i]];
            if (cls == 1) {   // F21 and G read, S stored, the children's entries that land in the boundary part read
              double to_s, to_p;
              child_entries(f, to_s, to_p);
              flops += 2.0 * (double)f.bp * f.bp * f.pp; bytes += 8.0 * (2.0 * (double)f.bp * f.pp + (double)f.b * f.b + to_s);
            } else { flops += 2.0 * (double)f.pp * f.pp * f.bp; bytes += 8.0 * ((double)f.pp * f.pp + 2.0 * (double)f.pp * f.bp); }
          }
        }
      }
    }
  };
  if (d.ev0 == nullptr) { HIP_OK(hipEventCreate(&d.ev0)); HIP_OK(hipEventCreate(&d.ev1)); }
  if (cls == 7) {
    if (d.fstream[0] == nullptr)
      for (int k = 0; k < DS_NSIDE; k++) { HIP_OK(hipStreamCreateWithFlags(&d.fstream[k], hipStreamNonBlocking)); HIP_OK(hipEventCreateWithFlags(&d.ev_fjoin[k], hipEventDisableTiming)); }
    if (d.ev_ffork == nullptr) HIP_OK(hipEventCreateWithFlags(&d.ev_ffork, hipEventDisableTiming));
  }
  issue(true);  // warm-up and accounting
  HIP_OK(hipEventRecord(d.ev0, s));
  for (int r = 0; r < reps; r++) issue(false);
  HIP_OK(hipEventRecord(d.ev1, s));
  HIP_OK(hipEventSynchronize(d.ev1));
  HIP_OK(hipGetLastError());
  float ms = 0;
  HIP_OK(hipEventElapsedTime(&ms, d.ev0, d.ev1));
  d.numeric_valid = false; d.have_factor = false;
  out[0] = launches > 0 ? ms * 1e3 / ((double)launches * reps) : 0.0;
  out[1] = launches > 0 ? flops / launches : 0.0;
  out[2] = launches > 0 ? bytes / launches : 0.0;
  out[3] = (double)launches;
  return 0;
}
