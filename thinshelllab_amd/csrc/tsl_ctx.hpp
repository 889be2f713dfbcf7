// Context of libtsl_hip.so: device-resident topology, the SELL-64 block matrix, solver scratch.
// Host-side only (no kernels here).
#pragma once
#include <memory>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/tsl_hip.h"
#include "k_body.hpp"
#include "direct_plan.hpp"

#define TSL_SLICE 64  // rows per SELL slice == wavefront width on gfx950

extern thread_local std::string g_tsl_err;
int tsl_fail(const char* fmt, ...);

#define HIP_OK(call)                                                                     \
  do {                                                                                   \
    hipError_t e__ = (call);                                                             \
    if (e__ != hipSuccess) return tsl_fail("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  bool owned = true;   // false: a VIEW into memory of a scene group (direct_group.hpp): never freed or re-allocated here
  int alloc(size_t count) {
    if (!owned) return tsl_fail("internal: allocation of %zu B requested for a buffer that is a view into a scene group's memory", count * sizeof(T));
    release();
    n = count;
    if (count == 0) return 0;
    hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
    if (e != hipSuccess) { p = nullptr; return tsl_fail("hipMalloc(%zu B) failed: %s", count * sizeof(T), hipGetErrorString(e)); }
    return 0;
  }
  int upload(const std::vector<T>& h) {
    if (alloc(h.size())) return -1;
    if (h.empty()) return 0;
    hipError_t e = hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    if (e != hipSuccess) return tsl_fail("hipMemcpy H2D failed: %s", hipGetErrorString(e));
    return 0;
  }
  int zero(hipStream_t s = 0) {
    if (!n) return 0;
    hipError_t e = hipMemsetAsync(p, 0, n * sizeof(T), s);
    if (e != hipSuccess) return tsl_fail("hipMemset failed: %s", hipGetErrorString(e));
    return 0;
  }
  void release() { if (p && owned) (void)hipFree(p); p = nullptr; n = 0; owned = true; }
  void view(T* q, size_t count) { release(); p = q; n = count; owned = false; }
  void swap(DevBuf& o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(owned, o.owned); }
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
};

// per-cloth constants read by the kernels (device copy)
struct ClothDev {
  int face_start, NF, v_offset, NV;
  double dx, mass, Kl, Ka, Kb, k_angle;
};
struct ElasticDev {
  int kind, cell_start, n_cells, v_offset, n_verts;
  double mu, lam, alpha;
};

// device scalars shared by the solver / Newton kernels: one 256-B record, viewed as CgScal (k_solver.hpp)
struct SolverScalars {
  double raw[32];
};

// one stencil level (>= 1) of the cloth multigrid hierarchy (k_mg.hpp)
struct MgLevel {
  int N = 0, M = 0, n = 0;
  DevBuf<double> A, Dinv, x, x2, r, t;
  DevBuf<float> A32, S32;  // single-precision copies for the two cycle kernels of the level (mg_st_f32)
  DevBuf<double> S;      // P^T A Dinv towards the next level (49 slots per coarse node, k_st_build_ra); empty on the last level
  DevBuf<double> omega;  // [0] damping factor, [1] lambda_max estimate (device resident)
  DevBuf<float> Cinv;    // dense inverse of the last level of the hierarchy (blocked Gauss-Jordan per assembly), symmetrised, fp32
  DevBuf<int> cbad;
  DevBuf<double> gjD, gjR, gjC, gjP;  // blocked Gauss-Jordan workspace (dense levels with more than 192 unknowns)
  int gj_ld = 0;
};
struct MgCloth {
  int v_offset = 0, N0 = 0, M0 = 0;
  std::vector<MgLevel*> lv;  // lv[0] = level 1
  int dense_lv = -1;         // index into lv of the level that is solved exactly (dense inverse), -1: none
  ~MgCloth() { for (auto* l : lv) delete l; }
};

// sparse direct preconditioner (direct_sym.hpp / direct_plan.hpp / k_direct.hpp): multifrontal LU of the assembled operator
#define DS_NSIDE 2
// a plan parked for a constraint set seen earlier: the reverse sweep revisits the sets of the forward rollout step by step
// (analytic_grad_single.py:217-257 re-detects the contacts of every tape step), so its plans are found here instead of being
// rebuilt (3-4 ms of host work each on cfg4)
struct DsPlanSlot {
  DirectPlan plan;
  std::vector<int> h_cons, h_cset;   // constraint vertices in the engine's order (contact map) / as a sorted set (what the plan depends on)
  uint64_t key = 0;
  long stamp = 0;
  bool used = false;
  DevBuf<int> level_sn, pmap, vtx, blk_ld, wl_front, wl_row, blk_q, cgr_ptr, cgr_ent, cgr_ld;
  DevBuf<long long> blk_dst, cgr_dst;
  DevBuf<DsFrontDesc> fr, frl;
  DevBuf<DsChildRec> ch_rec;
};
struct DirectSolver {
  char* pin = nullptr; size_t pin_cap = 0, pin_off = 0;   // pinned staging arena of the plan uploads (pageable copies are staged by the runtime at ~6 GB/s, synchronously)
  int enable = -1;          // -1 auto (cloth grids of >= 1024 cells: the iterative hierarchy is probed first, the factorisation takes over when it fails), 0 off, 1 always
  bool hard = false;        // auto mode: the last probe of the iterative hierarchy failed
  int hard_steps = 0, probe_cap = 60, probe_every = 16;
  hipStream_t zstream = 0;  // the arena of the NEXT factorisation is cleared here, next to the line search and the assembly (direct_prezero)
  hipEvent_t ev_zfork = nullptr, ev_zero = nullptr;
  bool prezero_pending = false;
  size_t prezero_n = 0;
  int prezero = 1;          // "direct_prezero"
  hipStream_t fstream[2] = {nullptr, nullptr};   // batches of one level run next to each other (direct_factor)
  hipEvent_t ev_ffork = nullptr, ev_fjoin[2] = {nullptr, nullptr};
  hipEvent_t ev_la[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  hipStream_t lastream = nullptr;
  // look-ahead of the upper levels: G stored / leading Schur tiles stored (engine stream), other panels of the next level written (side stream)
  int eager_n = 0; const double* eager_r = nullptr; double* eager_z = nullptr;   // upward sweeps done next to the factorisation: levels [0, eager_n) of the application (eager_r -> eager_z)
  int lookahead = 103;      // "direct_lookahead" (bit 1: the upward sweep of the first application next to the chains of the last three levels; (value >> 2) & 7 = c0, (value >> 5) & 15 = c1:
                            // levels [0, c0) next to the third chain from the top, [c0, c1) next to the one below the root, [c1, root) next to the root's (c1 = 0: none there); default c0 = 1, c1 = 3): the levels of one batch each form the leading block of their Schur complements first and invert the parents' pivot blocks next to the rest (direct_factor)
  // "direct_flow": block steps of a batch alone on its level as one persistent dataflow launch (k_ds_gj_flow)
  int device = 0;          // HIP device of the context (tsl_ctx_create)
  int flow_wgs_last = 0;   // workgroups of the dataflow launches of the last factorisation (the abort report compares it with the number that started)
  int flow = 3, flow_cap = 0, flow_epoch = 0;   // flow_cap: workgroups of k_ds_gj_flow the device holds at once
  int flow_token = 0, flow_token_fd = -1;       // the device's dataflow token (direct_host.hpp): 0 not asked yet, 1 held, -1 refused
  long flow_token_asked = 0;                    // n_factor at the last request
  int small_rounds = 2;     // "direct_small_rounds": rounds of the chip a batch may take in the LDS kernel
  int xcd_map = 64;         // "direct_xcd": batches of at least this many fronts launch their GEMM tiles with the XCD-aware map; 0: never
  long n_flow = 0, n_flow_abort = 0;   // dataflow launches / launches that lost a flag (the solve then refactorises on the block-step path)
  DevBuf<double> flow_x;    // exchange slots (pivot inverses, row / column panel tiles)
  DevBuf<int> flow_f;       // their flags (epoch of the launch that published the slot)
  DevBuf<unsigned long long> tlog;   // "ds_dbg" 30: device-clock stamps of a dataflow launch (diagnostic)
  int gemv_wide_below = 300;   // "direct_gemv_wide_below": a sweep launch of fewer 16-row chunks than this runs four narrow workgroups per chunk (k_ds_gemv_wide; 0 = never).
                               // cfg4, one application: 404 us without, 385 / 380 / 387 / 388 / 409 / 523 us at 150 / 300 / 600 / 1200 / 2400 / always (scripts/archive_r02_r03/exp_gemv_wide.py)
  int g32_below = 1100;     // "direct_g32_below": G = W F12 of a batch with fewer 64 x 64 tiles than this runs in the 32 x 32-tile kernel (k_ds_gemm_g32; 0 = never).
                            // cfg4: 52 -> 34, 67 -> 52, 37 -> 32 us on the three top levels with boundaries; the batches of 1150+ tiles lose (31 -> 33 us)
  bool cons_checked = false; // the constraint list has not changed since direct_plan last looked (reset by tsl_contact_detect)
  double* h_anorm = nullptr; // pinned: |H|_inf of the last factorisation (valid after the next stream synchronisation)
  int bench_batch = -1;     // tsl_bench_direct: restrict the replay to one batch (-1: all)
  // static pivoting: pivots below piv_tol x their own scale are perturbed to that bound ("direct_piv_tol").  1e-11 since round 3: a pivot
  // that lost eleven digits to cancellation against its entry diagonal (a contact stiffness of 1e14 on a nearly collapsed pad triangle over
  // a true pivot of 1e4) still carries five -- enough for a factorisation that is refined --, while 1e-8 replaced such pivots by 1e6 and
  // sent the refinement into its fallback: driver's command 3 of 18 runs with flagged solves at 1e-8, 0 of 16 at 1e-11; 30 flagged solves in a
  // T = 50 run at 1e-13 (garbage pivots pass).  (The T = 50 rollout reaches collapsed pad triangles -- entries of 1e17..1e22 -- in a third of its
  // runs at any threshold: 5 of 15 at 1e-11, 3 of 11 at 1e-12.)
  double piv_tol = 1e-11;
  int fallback_cap = 1000;  // iteration cap of the hierarchy when the factorisation broke down ("direct_fallback_cap")
  int leaf = 64;            // vertices per leaf of the nested dissection (cfg4 sweep: 32 -> 452, 48 -> 420, 56 / 64 -> 407, 80 -> 538 ms per step)
  bool static_ready = false, numeric_valid = false;
  bool have_factor = false;  // factors of the current plan exist
  int dbg = 0;
  // "direct_berr": the first pass of a refined solve is accepted when its normwise backward error |b - Hx| / (|H|_inf |x| + |b|) is at most this
  // (0: forward-residual rule only) and its forward residual at most "direct_berr_rel_cap" x cg_tol (direct_refine)
  double berr_tol = 1e-12, berr_rel_cap = 50.0;
  long berr_seen = 0, berr_accepted = 0;
  double berr_max = 0, berr_rel_max = 0;   // largest backward error / forward residual accepted under the rule
  int n_setup_fail = 0;        // set-up failures of the direct path in automatic mode (three disable it)
  std::vector<std::unique_ptr<DsPlanSlot>> cache;   // plans of earlier constraint sets ("direct_plan_cache" slots, least recently used evicted)
  int cache_cap = 64, cache_mb = 1024;   // slots / MB of parked plans ("direct_plan_cache", "direct_plan_cache_mb")
  long cache_clock = 0, n_plan_hits = 0, n_plan_evicted = 0;
  DirectPlan plan;
  std::vector<DsGrid> grids;
  std::vector<DsBlock> blocks;
  std::vector<int> row_ptr;   // CSR numbering of the static block pattern (rows of h_rows)
  std::vector<int> h_c2s;     // SELL address of every CSR block
  std::vector<int> h_cons;    // constraint vertices the current plan's contact map was built for (engine order)
  std::vector<int> h_cset;    // the same constraints as a sorted set: what tree, fronts and static maps depend on
  bool plan_valid = false;
  DevBuf<int> level_sn, pmap, vtx, blk_ld, bad, wl_front, wl_row, blk_q, cgr_ptr, cgr_ent, cgr_ld;
  DevBuf<long long> blk_dst, cgr_dst;
  DevBuf<DsFrontDesc> fr, frl;   // front descriptors by supernode id / in level order
  DevBuf<DsChildRec> ch_rec;
  DevBuf<double> arena, sarena, garena, scr, w;   // panel arena (cleared per factorisation), Schur arena (never cleared), G arena
  long n_plans = 0, n_factor = 0, n_apply = 0, n_perturbed = 0;
  long plan_gen = 0;          // counts every change of the device arrays of the active plan (new plan, cached plan swapped in, contact map redone): a scene group re-merges on it
  bool keep_host_maps = false;   // member of a scene group: parked plans keep their host block lists / child tables (ds_swap_slot)
  bool merged = false;        // the solver of a scene group's pseudo-context: its plan is the merge of the members' plans (direct_group.hpp), direct_plan does nothing
  DirectSolver* token_lender = nullptr;   // member of a scene group: the group's solver, whose dataflow token the member may use (the host runs them in lock step)
  double t_plan = 0;          // host seconds spent in plan builds
  double anorm = 0;           // infinity norm of the factorised operator's static part (backward-error yardstick)
  bool anorm_valid = false;   // anorm belongs to an operator at most 64 factorisations old
  int anorm_age = 0;
  int anorm_spd = -1;         // projection mode of the assembly the norm belongs to (the adjoint's un-projected operator gets its own)
  DevBuf<double> anorm_dev;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

struct tsl_group;
struct tsl_ctx {
  tsl_group* group = nullptr;   // set while the context is a member of a scene group (its matrix, right-hand side / solution vectors and fronts are views into the group's memory)
  hipStream_t stream = 0;       // internal non-blocking work stream (graph capture needs a real stream)
  hipStream_t user_stream = 0;  // caller's stream (tsl_set_stream); ordered with `stream` through events at every entry point
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  int depth = 0;
  // captured PCG iteration chunk
  hipGraphExec_t pcg_graph = nullptr;
  long pcg_graph_key = -1;
  hipGraphExec_t mr_graph = nullptr;  // six MINRES iterations
  long mr_graph_key = -1;
  int use_graph = 1;
  int NV = 0, NF = 0;  // tot_NV, tot_NF
  double dt = 5e-3, k_contact = 1000, eps_contact = 1e-3, eps_v = 0.01, damping = 1.0, mu_cloth_elastic = 1.0, mu_cloth_cloth = 1.0;
  int max_n_constraints = 10000;
  int newton_cap = 1000, plastic = 0, contact_enable = 1;
  double cg_tol = 1e-10;
  int cg_maxit = 200000, cg_check = 32;
  double grid_h = 0.003;
  double grid_extent = 0.2;  // half-width of the broad-phase box (geometry.py:8-19 hard-codes 0.2 m)

  // ---- cloth
  std::vector<ClothDev> h_cloth;
  DevBuf<ClothDev> d_cloth;
  int n_cface = 0, n_hinge = 0;
  DevBuf<int> cf_f2v, cf_cf, cf_cp, cf_cloth;  // per cloth face (global ids): verts, counter_face, counter_point, cloth id
  DevBuf<double> cf_V, cf_li;                   // rest area, rest lengths
  DevBuf<int> cf_blk;                           // n_cface x 9 block offsets
  DevBuf<int> cf_order;                         // faces in the order the scattering kernels take them (one stencil class after the other)
  DevBuf<int> hg_info;                          // n_hinge x 8: f1, l, f2, p4, p21, v[..] unused
  DevBuf<int> hg_v;                             // n_hinge x 4 vertex ids (a,b,c,d)
  DevBuf<int> hg_blk;                           // n_hinge x 16 block offsets
  DevBuf<double> norm_dir;                      // n_cface x 3
  DevBuf<double> quirk;                         // n_cloth x 3 faces x 3 slots x 10 (c_i, mat_N)
  std::vector<int> h_cf_f2v, h_cf_cf, h_cf_cp;

  // ---- tets
  std::vector<ElasticDev> h_el;
  DevBuf<ElasticDev> d_el;
  int n_tet = 0;
  DevBuf<int> tet_v, tet_el, tet_blk;  // 4 global verts, elastic id, 16 block offsets
  DevBuf<double> tet_B, tet_W;

  // ---- vertices
  DevBuf<double> mass, grav, fext;  // NV, NV*3, NV*3
  DevBuf<int> frozen;               // 3*NV (original order)
  DevBuf<int> diag_blk;             // NV
  std::vector<int> h_frozen;
  std::vector<double> h_mass;

  // ---- matrix (SELL-64 of 3x3 blocks, rows permuted by length)
  int n_slices = 0;
  long n_slots = 0;  // block slots incl. padding
  std::vector<int> h_rowpos, h_perm, h_slice_off, h_slice_len, h_colidx;
  std::vector<std::vector<int>> h_rows;  // original-order adjacency (sorted)
  DevBuf<int> rowpos, perm, slice_off, slice_len, colidx, diag_perm;
  DevBuf<double> vals, vals_full;  // masked (solver) and unmasked (adjoint) copies
  DevBuf<float> vals32;            // single-precision copy of the preconditioner's matrix for the multigrid smoother products
  bool vals32_valid = false;
  int mg_f32 = 1;
  DevBuf<double> Dinv;             // NV x 9 (permuted)
  DevBuf<unsigned char> fzmask;    // NV (permuted) 3-bit frozen mask
  DevBuf<double> mdt2;             // NV (permuted) m/dt^2
  long nnzb = 0;

  // ---- solver vectors (permuted AoS, 3*NV)
  DevBuf<double> v_x, v_r, v_z, v_p, v_Ap, v_b, v_t0, v_t1, v_t2, v_t3, v_t4, v_mg;
  // dense inverse of the small FEM-body diagonal blocks (k_body.hpp)
  BodyDenseArgs bd_args{};
  int bd_rows_n = 0, bd_n3max = 0, bd_wg = 0, bd_scr_n = 0, bd_enable = -1;
  size_t bd_w_total = 0;
  bool bd_valid = false;
  DevBuf<int> bd_rows, bd_body_of, bd_local_of, bd_bad;
  DevBuf<double> bd_W, bd_scr;
  DevBuf<double> bd_rb;  // compact ping-pong copy of the PCG residual on the dense-body rows (body part of k_pcg_update)
  int pcg_body_fold = 1;
  int mg_st_f32 = 1;
  int mg_chunk = 0;      // multigrid-PCG iterations per hipGraph replay / host convergence read (0: 8 on long solves, else 4)
  double mr_eta = 0.3;   // MINRES stops a recurrence cycle at |eta| <= mr_eta * tol (scaled); the true residual decides afterwards
  int mg_fr_rows = 32;   // coarse nodes per workgroup of k_st_first_restrict (16 / 32 / 64)
  DevBuf<float> bd_Binv;
  DevBuf<double> gm_V, gm_h;  // GMRES basis ((m+1) vectors) and projection coefficients
  DevBuf<double> gm_Z;        // preconditioned basis of the flexible variant (direct preconditioner)
  DevBuf<unsigned> cg_ent;
  DevBuf<int> cg_base, cg_ptr;           // gather assembly of the cloth Hessian: block addresses (ascending), list offsets, packed (element, vertex pair)
  DevBuf<double> cg_hrec, cg_frec;       // per-hinge (13) and per-face (81) records, entry-major
  // Every sum of the step and of the adjoint has a fixed order -- element gradients, energies and the Hessian blocks of cloth AND tets go through staging
  // records and gathers, contact lists are ordered, contact sums walk them in order: two runs give the same bits.  (The f64-atomic scatter kernels of
  // rounds 1-3 and their "deterministic" = 0 switch left in round 6.)
  DevBuf<int> trans;                     // slot of a matrix block -> address of its transposed block
  DevBuf<int> vg_ptr, vg_idx;            // vertex -> staging slots of its incident faces / hinges / tets (k_vertex_gather)
  DevBuf<double> vg_stage, cg_trec;      // staged element gradients (3-vectors); per-tet 12 x 12 records (144)
  int vg_ns = 0, vg_hinge0 = 0, vg_tet0 = 0;
  DevBuf<double> vg_stage2;   // second staging array of the tet slots (tsl_param_grad: the two materials accumulate into different vectors)
  DevBuf<double> dot_part; DevBuf<int> dot_ticket;   // scratch of the dot products (per-workgroup partials joined in a fixed order)
  DevBuf<double> e_part;                 // per-workgroup partial energies
  int n_cgblk = 0, n_cgblk_cloth = 0;   // gather lists: blocks of the cloth first, blocks of the FEM bodies behind them
  DevBuf<double> tet_V;       // eigenvector bases of the clamped element blocks of the last assembly (81 x n_tet, entry-major): warm start of the next one
  long tet_V_count = 0;
  int tet_warm = 1;
  DevBuf<double> ir_part;     // direct_refine: per-block partial sums + the three results
  DevBuf<int> ir_ticket;
  double* h_ir = nullptr;     // pinned host copy of {r.r, x.x, b.b, max |x_i|}
  double last_xmax = 0;       // max |x_i| of the solution direct_refine returned (the Newton loop's |p|max)
  bool last_xmax_valid = false;
  int gmres_m = 300, use_gmres = 1, use_minres = 1, verbose = 0;
  // analytic_grad_system.Grad: pos_grad clamp (1 there, 1000 in analytic_grad_single) and whether angleref_grad is clamped too
  double adj_clamp = 1000.0;
  int adj_clamp_angleref = 1;
  DevBuf<double> dmu_accum;  // d_mu of the box / ball bodies (accumulates like the reference's field)
  // preconditioner built from a different (SPD-projected) assembly than the operator: adjoint solves (un-projected H)
  DevBuf<double> vals_pc, c_H_pc;
  bool pc_separate = false, pc_frozen = false, in_step = false;
  bool cdiag_valid = false;   // c_diag holds the diagonal blocks of the contact terms in place (an assembly for the direct path does not form them)
  bool dinv_valid = false;    // Dinv holds the block-Jacobi inverse of the operator in place (an assembly for the direct path does not form it)
  const double *st_pos = nullptr, *st_prev = nullptr, *st_vel = nullptr, *st_ref = nullptr;  // state of the last assemble
  int fwd_spd_pc = 1;
  int adj_spd_pc = 1;
  DevBuf<SolverScalars> scal;
  DevBuf<double> part_pAp, part_rz, part_rr;  // per-block partial sums of the two-kernel PCG iteration
  SolverScalars* h_scal = nullptr;  // pinned
  SolverScalars* h_scal2 = nullptr; // pinned, two records: read-back slots of the PCG chunks in flight
  hipEvent_t rb_event[2] = {nullptr, nullptr};
  hipStream_t side = 0;                         // second stream: contact blocks of an assembly run next to the cloth / tet kernels
  hipEvent_t ev_fork = nullptr, ev_fork0 = nullptr, ev_g2 = nullptr, ev_join = nullptr, ev_join2 = nullptr, ev_hh = nullptr, ev_gf = nullptr;
  hipStream_t side2 = 0;                        // third stream: the tet kernels next to the contact kernels (side) and the cloth kernels

  // ---- Newton scratch (original order)
  DevBuf<double> F, pdir, x1;

  // ---- contact
  int n_body = 0, n_pair = 0;
  std::vector<tsl_body> h_bodies;
  std::vector<tsl_contact_pair> h_pairs;
  std::vector<int> self_contact;  // per body: query its own vertices against its own triangles (geometry_self.project_pair_self)
  DevBuf<int> faces;  // NF x 3
  DevBuf<int> border; // NV (BaseScene.border_flag)
  DevBuf<double> vn;
  DevBuf<int> proj_flag, proj_dir, proj_idx;  // n_body x NV (x3)
  DevBuf<double> proj_w;
  DevBuf<int> nc_dev;
  int nc = 0;
  DevBuf<int> c_idx;                                  // max_nc x 4
  DevBuf<double> c_w, c_n, c_dx0, c_k, c_mu, c_T;     // 3,3,3,1,1,6
  DevBuf<int> cr_ptr, cr_cnt, cr_fill, cr_ent;        // row -> (constraint, slot) CSR of the step's constraints (ContactRows)
  DevBuf<int4> cr_rows;
  DevBuf<int> cr_tmp;
  DevBuf<int> c_kind;                                 // friction parameter of the constraint's pair (0 fixed, 1 / 2 live)
  DevBuf<double> c_H;                                 // max_nc x 144 (12x12, masked) for the matrix-free product
  DevBuf<double> c_Hfull;                             // unmasked copy (adjoint)
  DevBuf<double> c_diag;                              // NV x 9 (permuted): masked contact contribution to the diagonal blocks
  DevBuf<double> c_G;                                 // max_nc x 12 scratch
  DevBuf<int> grid_key, grid_val, grid_key2, grid_val2, grid_range;  // per target body: cell id / face id (sorted), active range (6 ints)
  DevBuf<int> vnf_ptr, vnf_lst;   // vertex -> incident surface triangles (vertex normals summed in a fixed order)
  DevBuf<int> cq_flag, cq_scan;   // activity flag / constraint slot of every query vertex of every contact pair (fixed constraint order)
  DevBuf<int> grid_scan;   // scan scratch of the grids (one region per target body)
  std::vector<size_t> gb_f0, gb_t0, gb_s0;   // first entry of body b's region in the key / bucket / scan-scratch buffers
  DevBuf<int> grid_cnt, grid_ptr, grid_cur, scan_tmp;   // hash buckets of the broad phase (histogram, offsets, cursors) and the scratch of the scan
  int grid_buckets_max = 0;
  int max_body_faces = 0;

  // ---- multigrid preconditioner
  std::vector<MgCloth*> mg;
  int mg_enable = -1;  // -1 auto (on when a cloth hierarchy exists), 0 off, 1 on
  double mg_omega = 0.0;   // > 0: fixed damping; 0: 1.5 / lambda_max(D^-1 A) per level from a power iteration
  int mg_pi_iters = 12;
  DevBuf<double> mg_omega0, mg_pi_part, mg_pi_norm;  // level 0
  int mg_nu = 1, mg_coarse_sweeps = 8, mg_fuse = 1, mg_fuse_restrict = 1, mg_max_levels = 16, mg_coarse_exact = 1, mg_coarse_lag = 0, mg_dense_nodes = 64, mg_dense_auto = 1;
  double last_step_iters_per_solve = 0.0;
  bool warm_valid = false;
  bool mg_cinv_valid = false;
  bool mg_ops_valid = false, mg_suspended = false, mg_omega_valid = false;

  // ---- profiling of the dominant kernel
  int prof_enable = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double prof_ms = 0;
  long prof_launches = 0, prof_samples = 0;
  DevBuf<unsigned long long> prof_dev;  // per sampled launch: min start / max end of the device wall clock
  long prof_dev_used = 0, prof_dev_cap = 512;
  size_t prof_waves = 0;
  double prof_event_ms = 0, prof_dev_ticks = 0;
  long prof_dev_n = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_used = 0;
  long prof_chunks = 0;
  double tm_loop = 0;  // verbose: host time spent in the PCG iteration loops of the current step
  bool ev_sample_next = false;

  DirectSolver ds;
  bool ds_probe = false;      // set while the iterative hierarchy runs as the capped probe of the auto mode
  bool ds_suspended = false;  // set while the iterative hierarchy runs as the fallback of a failed direct solve
  // stats
  tsl_step_stats step_stats{};
  ~tsl_ctx() { for (auto* m : mg) delete m; }
};

// Every entry point runs on the context's own stream; Enter/leave order it after / before the caller's stream.
struct Scope {
  tsl_ctx* c;
  explicit Scope(tsl_ctx* c_) : c(c_) {
    if (c->depth++ == 0) { (void)hipEventRecord(c->ev_in, c->user_stream); (void)hipStreamWaitEvent(c->stream, c->ev_in, 0); }
  }
  ~Scope() {
    if (--c->depth == 0) { (void)hipEventRecord(c->ev_out, c->stream); (void)hipStreamWaitEvent(c->user_stream, c->ev_out, 0); }
  }
};
