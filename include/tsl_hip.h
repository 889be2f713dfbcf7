/*
 * tsl_hip.h -- C ABI of libtsl_hip.so, the MI355X (gfx950) thin-shell engine.
 *
 * Drop-in boundary.  The reference (Genesis-Embodied-AI/ThinShellLab) has no FFI layer: its engine
 * is a set of Python objects (code/engine/BaseScene.py, model_fold_offset.py, ...) whose methods the
 * task scenes / Grad / trajopt scripts call.  This header declares what a Python `engine` package binds
 * with ctypes so that those callers keep working; every entry point names the reference method it
 * replaces (file:line under /root/reference/code).  See INTEGRATION.md for the binding stub.
 *
 * Conventions: extern "C"; plain pointers and sizes; all reals fp64 and all indices int32 like the
 * reference (ti.init(default_fp=ti.f64, default_ip=ti.i32)); return 0 on success, <0 on error
 * (message via tsl_last_error()).  Pointers named *_dev are device (HBM) pointers owned by the caller
 * (torch tensors); pointers named *_host are host memory.  No ownership is transferred.  The library
 * owns only the opaque tsl_ctx (topology, system matrix, scratch).  Work is enqueued on the stream set
 * with tsl_set_stream(); calls that return a host scalar synchronise that stream.
 * One context per GPU / per scene; contexts are independent (one process per GPU, no collectives).
 */
#ifndef TSL_HIP_H
#define TSL_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tsl_ctx tsl_ctx;

/* One cloth (engine/model_fold_offset.py:10-107).  Index tables are the ones Cloth.init_mesh builds
 * (model_fold_offset.py:928-1018), LOCAL vertex / face numbering; v_offset = Cloth.offset. */
typedef struct {
  int32_t N, M, NV, NF, v_offset;
  double dx, mass, Kl, Ka, Kb, k_angle;
  const int32_t* f2v_host;           /* NF x 3 */
  const int32_t* counter_face_host;  /* NF x 3 */
  const int32_t* counter_point_host; /* NF x 3 */
  const double* rest_area_host;      /* NF      Cloth.V   (model_fold_offset.py:835) */
  const double* rest_len_host;       /* NF x 3  Cloth.l_i (model_fold_offset.py:836-838) */
} tsl_cloth_desc;

/* One tetrahedral body.  kind 0: engine/model_elastic_tactile.py (stable Neo-Hookean, alpha);
 * kind 1: engine/model_elastic_offset.py (Neo-Hookean with log J).  B = F_B (Ds^-1), W = F_W. */
typedef struct {
  int32_t kind, n_verts, n_cells, v_offset;
  double mu, lam, alpha;
  const int32_t* tets_host; /* n_cells x 4, LOCAL vertex ids */
  const double* B_host;     /* n_cells x 9 row-major */
  const double* W_host;     /* n_cells */
} tsl_elastic_desc;

/* One BaseScene.contact_pair_analysis call of the scene's contact_analysis()
 * (BaseScene.py:818-835, Scene_folding.py:99-108): query vertices [v_start, v_end) against body b_idx. */
typedef struct {
  int32_t b_idx, v_start, v_end;
  int32_t mu_is_param; /* 0: use mu; 1 / 2: the live mu_cloth_elastic / mu_cloth_cloth parameter (times mu when mu > 0) */
  double mu;
} tsl_contact_pair;

/* Body ranges of BaseScene.body_list (BaseScene.py:91-99). */
typedef struct { int32_t v_start, v_end, f_start, f_end; } tsl_body;

typedef struct {
  int32_t tot_NV, tot_NF;
  double dt, k_contact, eps_contact, eps_v, damping; /* BaseScene.py:44-48, scene init_scene_parameters */
  int32_t max_n_constraints;
  int32_t n_cloth;   const tsl_cloth_desc* cloths;
  int32_t n_elastic; const tsl_elastic_desc* elastics;
  int32_t n_body;    const tsl_body* bodies;
  int32_t n_pair;    const tsl_contact_pair* pairs;
  const double* mass_host;      /* tot_NV      BaseScene.mass (BaseScene.py:332-346) */
  const double* gravity_host;   /* tot_NV x 3  per-vertex gravity acceleration of its body (BaseScene.py:361-379) */
  const int32_t* faces_host;    /* tot_NF x 3  BaseScene.faces, GLOBAL vertex ids (BaseScene.py:348-359) */
  const int32_t* frozen_host;   /* 3*tot_NV    BaseScene.frozen (BaseScene.py:80, set_frozen) */
  double grid_h;                /* geometry.py:8 (0.003) */
} tsl_scene_desc;

typedef struct {
  int32_t newton_iters, ls_evals, cg_iters, solves, restarts, fallback, nc;
  double last_delta, last_alpha, energy;
  /* convergence accounting of the step's linear solves (the reference's spsolve is exact every time, sparse_solver.py:85-105):
   * fallback = solves that needed a second solver and converged there (flag 1); unconverged = solves that ended with flag 3;
   * attained = solves accepted by the attainable-accuracy rule (iterative solvers: true residual stagnating within 100x of cg_tol;
   * direct path: stagnating with a normwise backward error <= 1e-12, what a backward-stable direct solver delivers);
   * factorizations = numeric factorisations of the direct preconditioner; max_rel_residual over the step's solves */
  int32_t unconverged, attained, factorizations, plans;
  double max_rel_residual, max_backward_error;
} tsl_step_stats;

typedef struct {
  int32_t iters, restarts, flag; /* flag 0 converged with the primary solver, 1 converged with a fallback solver, 3 NOT converged */
  double rel_residual;           /* true residual |b - Hx| / |b| of the returned solution */
  int32_t method;                /* solver that produced x: 0 PCG, 1 MINRES, 2 GMRES, 3 BiCGStab, 4 sparse LU + iterative refinement (GMRES where that stalls) */
  int32_t attained;              /* 1: accepted by the attainable-accuracy rule instead of rel_residual <= cg_tol */
  double backward_error;         /* method 4: |b - Hx| / (|H|_inf |x| + |b|) of the returned solution (0 if not evaluated) */
} tsl_solve_stats;

const char* tsl_version(void);
const char* tsl_last_error(void);

/* BaseScene.__init__ + init_objects + init_property + set_frozen: upload topology, build the BSR(3x3)
 * pattern (replaces SparseMatrix's dense n x n storage, sparse_solver.py:13-17). */
int tsl_ctx_create(const tsl_scene_desc* desc, tsl_ctx** out);
void tsl_ctx_destroy(tsl_ctx* ctx);
int tsl_set_stream(tsl_ctx* ctx, void* hip_stream);

/* 0-d field writes of the reference and the engine's own switches (39 keys + three patterns; an unknown key is an error).
 *  Scene (trajopt_folding.py:50,66; Scene_folding.py:30-31; geometry.py:8-19; geometry_self.py:166-230):
 *   "cloth<i>.Kb|Kl|Ka|k_angle", "elastic<i>.mu|lam|alpha", "mu_cloth_elastic", "mu_cloth_cloth", "k_contact", "eps_contact", "eps_v", "damping",
 *   "newton_cap", "plastic", "contact" (0: no detection in tsl_step), "grid_h", "grid_extent" (broad-phase cell and box), "self_contact<body>"
 *   (0 / 1: the body's vertices are also projected onto its own triangles), "adj_clamp", "adj_clamp_angleref" (the clamping of analytic_grad_single:
 *   1000, on; of analytic_grad_system: 1, off), "adj_spd_pc" (adjoint solves of the iterative hierarchy preconditioned from the projected assembly).
 *  Linear solve (no reference counterpart: the reference calls cupyx spsolve):
 *   "cg_tol" (1e-10), "cg_maxit", "direct" (-1 auto: cloth grids of >= 1024 cells / 0 iterative hierarchy only / 1 always: multifrontal LU on the GPU),
 *   "direct_leaf" (vertices per nested-dissection leaf, 64), "direct_piv_tol" (static pivoting: pivots below this fraction of their entry diagonal are
 *   perturbed to it, 1e-11), "direct_berr" (1e-12: a first pass of the factorised solve whose normwise backward error |b - Hx| / (|H|_inf |x| + |b|) is
 *   at most this is accepted; 0: every solve is refined to cg_tol), and whose forward residual is at most 50 x cg_tol (fixed),
 *   "direct_flow" (3; bit 0: the block steps of one batch per tree level as ONE persistent dataflow launch, k_ds_gj_flow; bit 1: also the batches the
 *   LDS kernel would take; 0: one launch per 32 pivots -- the same bits either way), "direct_flow_token" (0: hand the device's dataflow token back),
 *   "direct_small_rounds" (2: rounds of the chip a batch may take in the LDS kernel k_ds_inv_small), "direct_g32_below" (1100: G = W F12 of a batch
 *   with fewer 64 x 64 tiles than this uses 32 x 32 tiles), "direct_gemv_wide_below" (300: a sweep launch of fewer 16-row chunks than this runs four
 *   workgroups per chunk), "direct_lookahead" (103; bit 0: on the tree levels of one batch each the leading block of every Schur complement is formed first and
 *   the parents' pivot blocks are gathered and inverted next to the rest of the Schur launch on a low-priority second stream; bit 1: the upward sweep of the solve's
 *   first application runs on that stream next to the chains of the last three levels -- levels [0, c0) next to the third from the top, [c0, c1) next to the one below the
 *   root, [c1, root) next to the root's, c0 = (value >> 2) & 7, c1 = (value >> 5) & 15 (0: none next to the root's); 0: one after the other -- the same bits).
 *   (Fixed since round 6: batches of >= 64 fronts launch their GEMM tiles with the XCD-aware map, the leaf panels of the next
 *   factorisation are cleared on a side stream after each solve of a time step, 64 plans of earlier constraint sets are kept.)
 *   "mg" (-1 auto / 0 / 1), "mg_coarse_exact", "mg_dense_nodes" (largest multigrid level solved exactly; -1 = chosen per time step), "body_inv"
 *   (dense inverses of the small FEM-body blocks), "gmres_m" (GMRES restart length), "tet_warm" (1: the eigen-clamp of the element blocks starts from
 *   the eigenvectors of the element's previous assembly).
 *   (Element gradients / blocks and contact rows go to staging slots and records and are summed by gather kernels in a fixed order, constraint lists are
 *   compacted by scan, energies and dot products joined from per-workgroup partials: no f64 atomics on the step and adjoint path, two runs give the same
 *   bits.  There is no switch: the scattered-atomics assembly of rounds 1-3 and its "deterministic" key are gone.)
 *  Diagnostics: "verbose" (1 phase times per step, 2 plans, 3 batches, 4 Newton iterations, 5 refinement passes), "ds_dbg" (21 / 22: force the dataflow-abort
 *   branch, tests -- 22 with the rule that a first loss takes only the look-ahead away; 30: device-clock trace of a dataflow chain), "ds_bench_batch" (tsl_bench_direct on one batch).
 * The experiment switches of rounds 1-4 (factor lagging, alternative GEMM / Schur tilings, one-lane contact and 16-lane element assembly, PCG warm
 * start, ...) are gone with the code they selected; the measurements that retired them are in profiles/README.md and DESIGN.md section 9. */
int tsl_set_param(tsl_ctx* ctx, const char* key, double value);
int tsl_set_frozen(tsl_ctx* ctx, const int32_t* frozen_host);          /* BaseScene.set_frozen */
int tsl_set_ext_force(tsl_ctx* ctx, const double* ext_force_host);      /* BaseScene.ext_force / manipulate_force */
int tsl_set_gravity(tsl_ctx* ctx, const double* gravity_host);          /* per-vertex, tot_NV x 3 */

/* BaseScene.compute_energy (BaseScene.py:427-451) at the given state, constraints as last detected. */
int tsl_energy(tsl_ctx* ctx, const double* pos_dev, const double* prev_pos_dev, const double* vel_dev,
               const double* ref_angle_dev, double* energy_host);

/* newton_step_init + compute_residual_and_Hessian(spd) (BaseScene.py:976-1040) or, with grad_dev == NULL,
 * compute_Hessian(spd) (BaseScene.py:1042-1052).  grad_dev (3*tot_NV) receives BaseScene.F. */
int tsl_assemble(tsl_ctx* ctx, const double* pos_dev, const double* prev_pos_dev, const double* vel_dev,
                 const double* ref_angle_dev, int spd, double* grad_dev);

/* SparseMatrix.solve (sparse_solver.py:85-105): x = H^-1 rhs for the matrix of the last tsl_assemble. */
int tsl_solve(tsl_ctx* ctx, const double* rhs_dev, double* x_dev, tsl_solve_stats* stats_host);

/* BaseScene.time_step (BaseScene.py:1327-1370; Scene_folding.py:279-322): contact detection, Newton loop with
 * halving line search, velocity update, optional plastic ref-angle update.  State arrays are updated in place. */
int tsl_step(tsl_ctx* ctx, double* pos_dev, double* prev_pos_dev, double* vel_dev, double* ref_angle_dev,
             tsl_step_stats* stats_host);

/* calc_vn + projection_query + contact_analysis (BaseScene.py:837-850, geometry.py:223-229, BaseScene.py:818-835).
 * proj_flag / proj_dir persist inside the context between calls (geometry.py:210-219). */
int tsl_contact_detect(tsl_ctx* ctx, const double* pos_dev, const double* prev_pos_dev, int32_t* nc_host);
int tsl_contact_reset(tsl_ctx* ctx); /* BaseScene.reset: proj_flag.fill(0) (BaseScene.py:268) */

/* Cloth.update_ref_angle (model_fold_offset.py:176-185) for every cloth. */
int tsl_update_ref_angle(tsl_ctx* ctx, const double* pos_dev, double* ref_angle_dev);

/* Grad.transfer_grad (analytic_grad_single.py:217-257) for one step s.  Buffers are the Grad fields:
 * pos_buffer/pos_grad (T x tot_NV x 3), ref_angle_buffer/angleref_grad (T x cloth_faces x 3).
 * On return tmp_z_frozen_dev (3*tot_NV) holds BaseScene.tmp_z_frozen for gripper.gather_grad. */
int tsl_adjoint_step(tsl_ctx* ctx, int step, int tot_timestep, const double* pos_buffer_dev, double* pos_grad_dev,
                     const double* ref_angle_buffer_dev, double* angleref_grad_dev, double* tmp_z_frozen_dev,
                     double adjoint_damping, tsl_solve_stats* stats_host);

/* Elastic.get_force of every FEM body (model_elastic_tactile.py:144-164, model_elastic_offset.py:188-208): internal force +
 * gravity + external force per vertex; consumed by BaseScene.check_early_stop / gather_force (BaseScene.py:1541-1584).
 * force_dev: tot_NV x 3, rows of the cloths are zero. */
int tsl_elastic_force(tsl_ctx* ctx, const double* pos_dev, double* force_dev);

/* System identification (engine/analytic_grad_system.py:112-160): the reverse step is tsl_adjoint_step with
 * tsl_set_param "adj_clamp" = 1 and "adj_clamp_angleref" = 0 (that class clamps pos_grad to +-1 and nothing else, :104-109);
 * this call then returns the sums over the free dofs of p . d(force)/d(parameter) with p the solution of that step
 * (get_parameters_grad :69-80 with BaseScene.get_paramters_grad BaseScene.py:1513-1525, Cloth.compute_deri
 * model_fold_offset.py:1082-1127, Elastic.compute_deri model_elastic_tactile.py:329-347 / model_elastic_offset.py:415-431).
 * pos = tape state x_s, ref_angle = tape rest angles of step s-1 (copy_pos_and_refangle, BaseScene.py:284-292).
 * out_host = {grad_kb, grad_mu, grad_lam} contributions of this step (grad_lam is 0: the reference never pushes d_lam up). */
int tsl_param_grad(tsl_ctx* ctx, const double* pos_dev, const double* ref_angle_dev, double* out_host);

/* Scene_sliding.contact_energy_backprop_friction (Scene_sliding.py:139-176): contribution of the last tsl_adjoint_step to
 * d(loss)/d(mu_cloth_cloth), summed over the constraints of the pairs that use that parameter. pos = tape state x_s. */
int tsl_friction_grad(tsl_ctx* ctx, const double* pos_dev, double* out_host);

/* Introspection used by the parity tests (tests/ only): assembled matrix as BSR on the host. */
int tsl_matrix_nnzb(tsl_ctx* ctx, int32_t* nb_host, int32_t* nnzb_host);
int tsl_matrix_export(tsl_ctx* ctx, int32_t* row_ptr_host, int32_t* col_host, double* vals_host);
int tsl_constraints_export(tsl_ctx* ctx, int32_t* idx_host, double* w_host, double* k_host, double* dx0_host,
                           double* T_host, double* n_host, double* mu_host, int32_t max_n);
/* per-constraint dense 12x12 blocks (vertex order idx0..idx3) of the last assemble; masked = frozen rule applied */
int tsl_contact_blocks_export(tsl_ctx* ctx, double* blocks_host, int32_t max_n, int32_t masked);
int tsl_proj_export(tsl_ctx* ctx, int32_t* proj_flag_host, int32_t* proj_dir_host, int32_t* proj_idx_host, double* proj_w_host);
int tsl_proj_import(tsl_ctx* ctx, const int32_t* proj_flag_host, const int32_t* proj_dir_host);
/* BaseScene.border_flag (tot_NV; BaseScene.py:104, restored by Scene_balancing.load_all :213-222, read by project_pair geometry.py:194-201) */
int tsl_set_border(tsl_ctx* ctx, const int32_t* border_flag_host);

/* Batched SPD projections (linalg.py:5-12 and :15-148) on device arrays of D x D blocks, D in {2,3,9}. */
int tsl_spd_project(tsl_ctx* ctx, double* blocks_dev, int32_t n_blocks, int32_t D);

/* Timing of the dominant kernel for bench.py's roofline object: HIP-event time (ms) accumulated over the
 * PCG iteration kernels since the last reset, launch count and the bytes one iteration moves algorithmically.
 * Sampled launches (1 in 16) are timed twice: by the device wall clock inside the kernel (min start / max end over waves =
 * the duration a kernel trace reports) and by a hipEvent pair around the launch on the engine stream. */
int tsl_profile_reset(tsl_ctx* ctx, int enable);
int tsl_profile_read(tsl_ctx* ctx, double* spmv_ms_host, int64_t* spmv_launches_host, int64_t* spmv_bytes_per_launch_host);
int tsl_profile_read_events(tsl_ctx* ctx, double* spmv_ms_hip_events_host); /* same launches bracketed by hipEvents (includes launch gaps) */
/* `reps` back-to-back launches of one operator kernel on the currently assembled matrix (and the current step's contact rows),
 * bracketed by ONE hipEvent pair on the engine stream: average microseconds per launch.  variant 20 = k_pcg_spmv exactly as a
 * PCG iteration launches it (what bench.py's roofline object is priced on), 2 = the plain SELL-64 product k_spmv_mw,
 * 30 = k_stream_read over the matrix values only (streaming-read yardstick for the same bytes). */
int tsl_bench_spmv(tsl_ctx* ctx, int variant, int reps, double* us_per_launch_host);

/* Sparse direct path (multifrontal LU of the operator, the counterpart of the reference's spsolve, sparse_solver.py:85-105).
 * tsl_bench_direct: the launches of one kernel class of ONE factorisation (cls 0 the Gauss-Jordan inversions W = F11^-1 on the block-step
 * path: k_ds_pivot0 + k_ds_gj_step + k_ds_gj_finish; 1 k_ds_gemm in Schur mode: product, gather of the children's Schur complements,
 * store; 2 k_ds_gemm in G = W F12 mode; 3 the inversions in the LDS kernel k_ds_inv_small; 5 the inversions in the persistent dataflow
 * kernel k_ds_gj_flow; 6 k_ds_extend_panels, the panels' share of the extend-add) or of one application (4 k_ds_gemv) on the current
 * plan, replayed `reps` times between one hipEvent pair;
 * out4 = {us per launch, algorithmic flops per launch, algorithmic bytes per launch, launches per factorisation}.  The factors are
 * invalid afterwards.  tsl_direct_info: {plans, factorisations, applications, perturbed pivots of the last factorisation, host
 * seconds in plan builds, supernodes, levels, batches, flops per factorisation, bytes of fronts (panels + Schur complements)}.
 * tsl_direct_counters: the first n of {dataflow launches, dataflow launches that lost a flag (redone on the block-step path), plan-cache
 * hits, bytes of the panel arena (cleared per factorisation), of the Schur arena, of the G arena, Schur-complement entries stored per
 * factorisation, plans parked in the cache, first passes of refined solves whose normwise backward error |b - Hx| / (|H|_inf |x| + |b|)
 * was evaluated, first passes accepted on it ("direct_berr", default 1e-12), largest backward error / forward residual so accepted}. */
int tsl_bench_direct(tsl_ctx* ctx, int cls, int reps, double* out4_host);
int tsl_direct_info(tsl_ctx* ctx, double* out10_host);
int tsl_direct_counters(tsl_ctx* ctx, double* out_host, int32_t n);

/* ---- scene groups: several scenes of ONE device whose sparse direct solves share their launches --------------------------------------------
 * The reference's trajectory-optimisation drivers roll out independent copies of one scene (training/trajopt_*.py, one rollout per candidate
 * trajectory); a single 100k-triangle scene leaves most of the chip idle during the latency-bound parts of its factorisation.  A group takes the
 * members' matrices, solution / right-hand side vectors and fronts into memory of its own, merges their plans, and tsl_group_step advances all
 * members by one time step in lock step: per member exactly tsl_step's kernels and decisions (BaseScene.time_step, BaseScene.py:1327-1370),
 * the factorisation and the first application of the factors as ONE set of launches for all.  A member's result is bit-identical to the one
 * tsl_step gives for it alone.  Members stay ordinary contexts: every other entry point (tsl_adjoint_step, tsl_solve, ...) works on them as before.
 *   tsl_group_create: ctxs[n] contexts of one device that use the sparse direct solve and belong to no group; destroying a member destroys the group.
 *   tsl_group_step:   pos / prev / vel / ref_angle[n] device pointers (as tsl_step), stats[n] host records (null: none).
 *   tsl_group_adjoint_step: Grad.transfer_grad (analytic_grad_single.py:217-257) of every member for the same reverse step -- the arguments of
 *                     tsl_adjoint_step as arrays of n, the n adjoint systems through one merged factorisation; same bits per member; stats[n] or null.
 *   tsl_group_info:   {plan merges, arena re-layouts, host seconds in merges, bytes of the group's arenas, merged factorisations, merged applications,
 *                     solves in which a member went on from the merged first pass on its own path (refinement, GMRES), dataflow launches of the merged
 *                     factorisations, merged factorisations redone on the block-step path after such a launch lost a flag}; nine values. */
typedef struct tsl_group tsl_group;
int tsl_group_create(tsl_ctx* const* ctxs, int32_t n, tsl_group** out);
void tsl_group_destroy(tsl_group* g);
int tsl_group_step(tsl_group* g, double* const* pos, double* const* prev_pos, double* const* vel, double* const* ref_angle, tsl_step_stats* stats_host);
int tsl_group_adjoint_step(tsl_group* g, int step, int T, const double* const* pos_buffer, double* const* pos_grad, const double* const* ref_buffer,
                           double* const* angleref_grad, double* const* tmp_z_frozen, const double* adj_damping_host, tsl_solve_stats* stats_host);
int tsl_group_info(tsl_group* g, double* out9_host);

#ifdef __cplusplus
}
#endif
#endif /* TSL_HIP_H */
