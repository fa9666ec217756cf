/*
 * hmg.h -- C ABI of libhmg_hip.so: matrix-free geometric multigrid on the implicit fine grid,
 * MI355X (gfx950) native.  This is the drop-in boundary for the hot path (level L3) of
 * haampie/Homogenization.jl.  The reference has no FFI seam; its seam is Julia multiple dispatch on
 * `AbstractMatrix` level vectors (src/multigrid.jl:7-13).  Each entry point below names the reference
 * method it replaces (file:line in the reference checkout).  A Julia host binds these with `ccall`
 * (INTEGRATION.md); in this repository the executable host mirror is Python/ctypes
 * (homogenization.jl_amd/api.py).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; hmg_last_error() gives the message
 *     (thread-local).  No exception or longjmp crosses the boundary.
 *   - handles are opaque, created and destroyed by the library.  A vector keeps its grid alive and a grid its context
 *     (reference counts): the hmg_*_destroy calls may come in any order -- e.g. from the finalizers of a garbage-
 *     collected host -- and device memory goes back when the last dependant has been destroyed.
 *   - host arrays use the reference's API layout: level vectors are Nf x Ne column-major FP64 in
 *     the reference's hierarchical node order (src/multilevel_reference.jl:41-61); meshes use
 *     1-based node ids with every cell's tuple ascending (src/implicit_fine_grid.jl:14).
 *   - device work is enqueued on the context's HIP stream; calls that return a scalar to the host
 *     synchronise that stream, all others are asynchronous.
 *   - all COMPUTING calls on one context must come from one host thread at a time.  The hmg_*_destroy calls are the
 *     exception: they may come from any thread at any time (finalizers) -- reference counts, the registry of contexts
 *     and the pool of level-vector memory are guarded by a lock; the memory of a destroyed vector is reused only by
 *     work enqueued later on the context's stream.
 */
#ifndef HMG_H
#define HMG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hmg_ctx hmg_ctx;
typedef struct hmg_grid hmg_grid;
typedef struct hmg_vec hmg_vec;

const char *hmg_last_error(void);
int hmg_version(void);

/* ---- context ------------------------------------------------------------------------------ */
/* stream: a hipStream_t owned by the caller (e.g. torch's current stream) or NULL to create one. */
int hmg_ctx_create(int device, void *stream, hmg_ctx **out);
/* Always runs on the given stream handle, including the null (legacy default) stream -- the handle
 * torch.cuda.current_stream().cuda_stream returns for torch's default stream. */
int hmg_ctx_create_on_stream(int device, void *stream, hmg_ctx **out);
int hmg_ctx_destroy(hmg_ctx *ctx);
int hmg_ctx_sync(hmg_ctx *ctx);
/* Hands the blocks of destroyed level vectors that the context keeps for reuse (option "vec_pool") back to the device;
 * done automatically when an allocation of a level vector fails and when the context goes. */
int hmg_ctx_release_memory(hmg_ctx *ctx);
/* option names: "apply_threads" (workgroup size of the apply kernel, 0 = auto), "apply_wg512" (1 = default: level-6
 * cells -- 6545 nodes, 52 KB of LDS -- are applied by 512-thread workgroups, three resident per CU; 0 = by 1024-thread
 * workgroups, two per CU; same arithmetic per node), "fuse_cg" (1 = fused CG pass,
 * default; read when a grid is created), "fold_x" / "lazy_dead" / "swap_rp" / "fold_prolong" (1 = default: hmg_vcycle
 * folds the pre-smoother's last x-update into the local residual / lets the pre-smoother's last step write nothing
 * and folds both pending x-updates / lets CG step 0 take r itself as p by exchanging the two handles' device
 * pointers / folds the prolongation into the post-smoother's first residual -- exact savings, results unchanged;
 * 0 = the plain sequence), "lean_post" (1 = default: hmg_vcycle also drops the post-smoothers' dead tails -- the last
 * p-update and, below the top level, everything of the last CG step but x += alpha p; x of every level and r of the
 * top level are unchanged bit for bit, p / Ap (and r below the top level) are scratch on return, as they are for the
 * reference's own callers: the next smoothing_steps! overwrites them before reading, src/multigrid.jl:46-50;
 * 0 = they hold what the reference leaves), "lazy_post" (1 = default: below the top level the post-smoother's dead last step
 * writes nothing -- its direction is formed in LDS for the apply and again on the fly by the one pass that does both pending
 * x-updates; x unchanged bit for bit), "lazy_top" (the same on the top level, where r is live as well: 1 = the last
 * step's apply writes Ap alone and its r-update carries both x-updates; 2 = default: with three steps or more the step before
 * writes its direction into a spare vector of the top level's size -- reserved when the first vector of the finest level is
 * created or wrapped, or by hmg_grid_reserve_spare; never inside a V-cycle -- and leaves its x-update to
 * that pass too; x and r unchanged bit for bit), "apply_wave" (1 = default: cells of 969 nodes -- 3D level 5 -- are applied by one WAVE
 * per cell with the class weights taken from a cache that hmg_grid_set_operator fills, hmg_apply_wave.hip; 0 = the 256-thread
 * workgroup kernel; taken only where the mesh has at most 1024 distinct coefficient rows and |alpha| = 1; "wave_grid": its
 * persistent waves per CU, default 16 = what the LDS holds, "wave_grid_total": the same as an absolute number -- tests), "prolong_in_image" (1 = default: on level 6 the folded prolongation stages the
 * coarse column at the even nodes of the LDS lattice image itself instead of in LDS of its own behind it, which would cost
 * the third resident workgroup; "prolong_gather", the option's round-2 name, is still accepted), "fold_restrict" (1 = default:
 * inside hmg_vcycle the local residual of levels 6 and 5 restricts itself in its kernel's epilogue and is not stored; the
 * coarse right-hand side is the stand-alone restriction's to the last bit), "zero_entry" (1 = default: inside hmg_vcycle a
 * coarse level's zero initial guess is never written -- its first residual is the constrained copy of b and the local
 * residual that carries both pending x-updates does not read x), "fold_faces" (1 = default: the face part of A p's
 * interface sum rides in the CG r-update), "cell_order" (1 = default: the full-grid apply launches of cells of >= 969 nodes
 * walk the cells XCD by XCD -- workgroups are dispatched round-robin over the eight XCDs, XCD x takes the x-th contiguous eighth
 * of the cells; a performance hint, results unchanged), "persistent_waves" (one-wave apply workgroups per CU that walk the
 * cells of the small levels; default 32, 0 = one workgroup per cell), "overlap_min_doubles" / "comm_rehearsal" (multi-GPU, below),
 * "vec_pool" (1 = default: hmg_vec_destroy keeps the block for the next
 * hmg_vec_create of the same size -- re-allocating freed device memory costs ~35 ms per GB here; 0 = free at once and
 * release what is held; hmg_ctx_destroy releases it too), "coarse_maxit", "coarse_check", "coarse_poly" (4 = default: the
 * level-1 PCG is preconditioned by that many Chebyshev iterates of the Jacobi-scaled operator -- k - 1 sparse products without a
 * reduction per outer iteration --; 1 = plain Jacobi), "weight_cache" / "apply_small" / "apply_pack" (1 = default: class weights of levels 2-6
 * from the class-weight cache; levels 2-4 by the pipelined one-wave kernel, level 2 four cells to a wave, hmg_apply_small.hip),
 * "apply_slab2" (1 = default: cells larger than the LDS -- 3D level 7 -- are applied by ONE persistent 1024-thread workgroup per CU
 * whose waves have roles, hmg_apply_slab.hip: loader waves stream the next k-plane window from HBM while evaluator waves work on the
 * current one; 0 = the rolling-window kernel of rounds 1-4, same arithmetic per node; "slab2_grid": its workgroups, 0 = one per CU; "restrict_slab2": 1 = default, the stand-alone restriction of such a level goes
 * through it too; "slab2_force": experiment, level 6 through it -- needs HMG_SLAB_LDS_KB <= 30 when the grid is created),
 * "time_apply"; "coarse_rtol" and "coarse_poly_ratio" (20: the interval [lmax / ratio, lmax]) via hmg_ctx_set_option_f64.  Environment: HMG_SLAB_LDS_KB (LDS window of the slab
 * kernel for cells larger than the LDS, default 70). */
int hmg_ctx_set_option(hmg_ctx *ctx, const char *name, int64_t value);
int hmg_ctx_set_option_f64(hmg_ctx *ctx, const char *name, double value);
/* HIP-event timing of the operator-apply launches of levels >= the value given to option "time_apply"
 * (0 switches it off; setting it resets the counters).  Synchronises the stream. */
int hmg_ctx_apply_timing(hmg_ctx *ctx, int64_t *launches, double *total_ms, double *total_bytes);
/* ... the same, only the launches of one level ("time_apply" = 1 times every level). */
int hmg_ctx_apply_timing_level(hmg_ctx *ctx, int level, int64_t *launches, double *total_ms, double *total_bytes);
/* diagnostic counters: "wave_launches" (launches of the one-wave-per-cell level-5 apply, hmg_apply_wave.hip), "small_launches"
   (levels 2-4, hmg_apply_small.hip), "slab2_launches" (level 7, hmg_apply_slab.hip), "comm_calls", "comm_nranks" (ranks of the RCCL communicator made by hmg_comm_init, 0 without
   one), "device_allocs" (device / pinned allocations the library has made in this process: constant across hmg_vcycle once the
   grid, its operator, its level-1 system and the level vectors exist), "spare_bytes" (spare direction vectors held by this
   context's grids, see hmg_grid_reserve_spare), "lazy_top_form" (the form the last finest-level post-smoother inside hmg_vcycle
   took: 2 = three-update form with the spare vector, 1 = two-update form, 0 = plain); -1 for an unknown name.  No counterpart in
   the reference. */
int64_t hmg_ctx_counter(hmg_ctx *ctx, const char *name);

/* ---- grid: ImplicitFineGrid(base, levels)  (src/implicit_fine_grid.jl:13-18) ------------------ */
/* Also derives ZeroDirichletConstraint(list_boundary_nodes_edges_faces(base)...)
 * (src/interface.jl:207-284, src/implicit_fine_grid.jl:80-84). */
int hmg_grid_create(hmg_ctx *ctx, int dim, int nlevels, int64_t nnodes, const double *coords /* dim*nnodes */,
                    int64_t ncells, const int64_t *cells /* (dim+1)*ncells, 1-based */, hmg_grid **out);
int hmg_grid_destroy(hmg_grid *grid);
/* L2PlusDivAGrad(diff, mass, constraint, lambda, sigmas)  (src/build_local_operators.jl:26-32) */
int hmg_grid_set_operator(hmg_grid *grid, const double *sigma /* dim*ncells */, double lambda);
int hmg_grid_set_lambda(hmg_grid *grid, double lambda);
/* Domain shrink to a prefix of cells / nodes + new Dirichlet boundary
 * (src/examples/homogenized_coefficients.jl:309-336).  Level vectors keep their storage.  On a partitioned grid
 * the prefix lengths are GLOBAL; every rank keeps its cells below the prefix (a prefix of its own columns) and the
 * cut entities, masks and ownership are re-derived. */
int hmg_grid_shrink(hmg_grid *grid, int64_t ncells_prefix, int64_t nnodes_prefix);
/* The reference's LevelState holds five vectors per level (src/multigrid.jl:18-25).  With option "lazy_top" = 2 (the default) the
 * finest level's post-smoother inside hmg_vcycle uses a SIXTH vector of that level's size (+20 % on the finest level's footprint)
 * to save 8 B/DOF of traffic per V-cycle.  It belongs to the grid and is setup, not hot-path, memory: reserved automatically when
 * the first vector of the finest level is created or wrapped (if that allocation fails, V-cycles silently-but-reportedly take the
 * two-update form: hmg_ctx_counter "lazy_top_form" / "spare_bytes"), or explicitly here: enable = 1 reserves it now and FAILS if
 * the memory is not there; enable = 0 releases it and keeps it released (the five-vector footprint of the reference). */
int hmg_grid_reserve_spare(hmg_grid *grid, int enable);
int64_t hmg_grid_ncells(const hmg_grid *grid);
int64_t hmg_grid_nnodes(const hmg_grid *grid);
int hmg_grid_nlevels(const hmg_grid *grid);
int64_t hmg_grid_nf(const hmg_grid *grid, int level);   /* nnodes(refined_mesh(implicit, level)) */
int64_t hmg_grid_ld(const hmg_grid *grid, int level);   /* device column stride in doubles */
/* Table export for tests / host mirrors. which: "hier2slot" (int32[nf]), "slot_ijk" (int32[3*nf]),
 * "slot_cls" (int32[nf]), "par_a","par_b" (int32[nf]), "ctab" (f64), "dmask","dupmask" (int32[ncells]),
 * "interior_nodes" (int32, 0-based), "coef" (f64[8*ncells]).  Returns the element count via *count. */
int hmg_grid_table_i32(const hmg_grid *grid, int level, const char *which, int32_t *out, int64_t cap, int64_t *count);
int hmg_grid_table_f64(const hmg_grid *grid, int level, const char *which, double *out, int64_t cap, int64_t *count);

/* ---- level vectors: the five matrices of LevelState (src/multigrid.jl:7-25) ------------------- */
int hmg_vec_create(hmg_grid *grid, int level, hmg_vec **out);                 /* zeros(Nf, Ne) */
int hmg_vec_wrap(hmg_grid *grid, int level, void *device_ptr, hmg_vec **out); /* caller-owned ld*Ne doubles */
int hmg_vec_destroy(hmg_vec *v);
void *hmg_vec_device_ptr(hmg_vec *v);
int hmg_vec_upload(hmg_vec *v, const double *host);      /* Nf x Ne, hierarchical order */
int hmg_vec_download(hmg_vec *v, double *host);
int hmg_vec_fill(hmg_vec *v, double value);                                   /* fill!           */
int hmg_vec_fill_random(hmg_vec *v, uint64_t seed, int64_t cell_offset);      /* rand! (seeded)  */
int hmg_vec_copy(hmg_vec *dst, hmg_vec *src);                                 /* copyto!         */
int hmg_vec_axpy(double alpha, hmg_vec *x, hmg_vec *y);                       /* axpy!  multigrid.jl:65-66 */
int hmg_vec_xpby(hmg_vec *r, double beta, hmg_vec *p);                        /* p .= r .+ beta.*p  :68   */
int hmg_vec_dot(hmg_vec *x, hmg_vec *y, double *out);    /* dot over raw storage, copies counted  :54 */
int hmg_vec_norm_unique(hmg_vec *r, double *out);        /* norm after zero_out_all_but_one!, r kept */

/* ---- hot-path primitives -------------------------------------------------------------------- */
/* mul!(alpha, base, A::L2PlusDivAGrad, x, y): y += alpha*A*x   (src/apply_local_operators.jl:85-133) */
int hmg_apply(hmg_grid *grid, int level, double alpha, hmg_vec *x, hmg_vec *y);
/* general form behind mul!/local_residual!/the smoother's Ap = A*p: out = (src ? src : 0) + alpha*A*x, then
 * (constrain != 0) apply_constraint!(out).  src may be NULL or alias out. */
int hmg_apply_ex(hmg_grid *grid, int level, double alpha, hmg_vec *x, hmg_vec *src, hmg_vec *out, int constrain);
/* local_residual!: r = b - A*x, then constraint            (src/apply_local_operators.jl:18-27) */
int hmg_residual(hmg_grid *grid, int level, hmg_vec *x, hmg_vec *b, hmg_vec *r);
/* apply_constraint!                                          (src/implicit_fine_grid.jl:94-139) */
int hmg_constraint(hmg_grid *grid, int level, hmg_vec *x);
/* broadcast_interfaces!                                      (src/implicit_fine_grid.jl:209-328) */
int hmg_interface_sum(hmg_grid *grid, int level, hmg_vec *x);
/* zero_out_all_but_one!                                      (src/implicit_fine_grid.jl:334-386) */
int hmg_zero_duplicates(hmg_grid *grid, int level, hmg_vec *x);
/* restrict_to!(b_coarse, P, r_fine) / interpolate_and_sum_to!(x_fine, P, x_coarse), P = interops[level_fine-1]
 *                                                            (src/interpolation.jl:52-74) */
int hmg_restrict(hmg_grid *grid, int level_fine, hmg_vec *r_fine, hmg_vec *b_coarse);
int hmg_prolong_add(hmg_grid *grid, int level_fine, hmg_vec *x_coarse, hmg_vec *x_fine);
/* copy_to_base! / distribute!                                (src/implicit_fine_grid.jl:148-202) */
int hmg_gather_base(hmg_grid *grid, hmg_vec *v1, double *host_u /* nnodes */);
int hmg_scatter_base(hmg_grid *grid, const double *host_u, hmg_vec *v1);

/* ---- driver right-hand sides (run once per outer step) --------------------------------- */
/* rhs_a xi grad v!(b, dphis, implicit, sigmas, xi)   (src/examples/homogenized_coefficients.jl:449-474) */
int hmg_rhs_axi_grad(hmg_grid *grid, const double *xi /* dim */, hmg_vec *b);
/* local_rhs!(b, implicit): b[:, e] = |det J_e| * int phi over the refined reference cell (unit load;
 * src/implicit_fine_grid.jl:391-409, used by checkerboard_hypercube_multigrid, ...homogenized_coefficients.jl:543) */
int hmg_local_rhs(hmg_grid *grid, hmg_vec *b);
/* integrate_first_term (mode 0), integrate_terms (mode 1), integrate_area (mode 2) over the first ncells_subset cells
 * (src/examples/homogenized_coefficients.jl:592-689).  `second` is, for mode 0, the vector rhs_a xi grad v! produced for
 * the same xi (hmg_rhs_axi_grad: the reference recomputes dot(dphi_i, P) per node, :621 -- it is that vector entry);
 * for mode 1 the previous iterate v_{k-1}.  One 16 B/DOF pass of the operator-apply kernel in reductions-only form, on
 * every level the apply supports.  Partitioned grid: local cells, this rank's share (the host adds the shares). */
int hmg_integrate(hmg_grid *grid, int mode, hmg_vec *v, hmg_vec *second, int64_t ncells_subset, const double *xi,
                  double *out);
/* next_rhs!(b, x, implicit, ops): b = lambda*|J|*M*x  (src/examples/homogenized_coefficients.jl:695-713) */
int hmg_next_rhs(hmg_grid *grid, hmg_vec *x, hmg_vec *b);

/* ---- fused fast path ------------------------------------------------------------------------ */
/* smoothing_steps!(steps, implicit, ops, curr, k)            (src/multigrid.jl:46-71) */
int hmg_smooth(hmg_grid *grid, int level, int steps, hmg_vec *x, hmg_vec *b, hmg_vec *r, hmg_vec *p, hmg_vec *Ap);
/* Optional, at setup: choose by measurement which memory block plays which role.  `states` as for hmg_vcycle (5 handles
 * per level, level-major); the five hmg_vec_create'd vectors x, b, r, p, Ap of `level` >= 2 (LevelState,
 * src/multigrid.jl:7-25) are tuned, those of level - 1 serve as the coarse side; operator set.  The passes that stream
 * five or six finest-level vectors at once run up to 8 % slower or faster depending on where the blocks lie in HBM
 * relative to each other -- a property of the physical pages, not of anything an address shows --
 * so the library times this level's share of a V-cycle (hmg_vcycle_down + hmg_vcycle_up with `steps` smoothing steps)
 * for `trials` assignments of the five blocks plus `extra` freshly allocated ones to the five roles, keeps the fastest
 * (the handles' device pointers are exchanged; pointers obtained from hmg_vec_device_ptr before are stale), frees the
 * blocks left over and zero-fills the vectors of both levels: CALL IT BEFORE THE VECTORS HOLD DATA.  ms_out (may be
 * NULL): [0] time of the assignment the handles came with, [1] of the one they leave with.  Costs (trials + 3) x that
 * share of a V-cycle; on a partitioned grid every rank must call it with the same arguments. */
int hmg_level_tune_placement(hmg_grid *grid, int level, int steps, hmg_vec **states, int extra, int trials, double *ms_out);
/* Coarse operator for the current sigma/lambda/boundary: replaces
 * cholesky(assemble_checkerboard(base, cond, lambda)[interior, interior])
 * (src/examples/homogenized_coefficients.jl:259-261) by a device-resident CG, preconditioned by Chebyshev iterates of the Jacobi-scaled operator (option "coarse_poly"; 1 = plain Jacobi-PCG). */
int hmg_coarse_setup(hmg_grid *grid);
/* level-1 branch of vcycle!                                  (src/multigrid.jl:74-93) */
int hmg_coarse_solve(hmg_grid *grid, hmg_vec *b1, hmg_vec *x1);
/* Iterations of the last level-1 solve; blocks until its probe has landed.  -1 (hmg_last_error set): that solve did not
 * reach coarse_rtol.  How a solve is policed: the first solve on a new matrix (operator, lambda or domain changed) looks at
 * its residual every "coarse_check" iterations and fails with an error after "coarse_maxit"; later solves enqueue
 * 1.5 x the largest count seen + 16 iterations without a host round trip and leave a probe behind, which is judged by the
 * next call that synchronises the stream anyway (hmg_vec_norm_unique, hmg_vec_dot, hmg_integrate, hmg_ctx_sync, this
 * function) or by the next solve: an unconverged solve is an ERROR of that call (the V-cycle that used it was inexact),
 * the budget is dropped, and repeating the V-cycle solves the slow, checked way.  The reference's CHOLMOD solve
 * (src/multigrid.jl:84) cannot fail this way; the error keeps that contract visible. */
int hmg_coarse_last_iterations(const hmg_grid *grid);
int64_t hmg_coarse_misses(const hmg_grid *grid);   /* budgeted solves that ran out of iterations so far */
/* vcycle!(implicit, base, ops, levels, k, steps); coarser levels use steps_coarse (the reference
 * does not forward `steps`, src/multigrid.jl:109, so pass 2 for parity).
 * states: 5*nlevels handles ordered level-major as x,b,r,p,Ap of level 1, then level 2, ... */
int hmg_vcycle(hmg_grid *grid, int top_level, int steps, int steps_coarse, hmg_vec **states);
/* The two halves of one level of vcycle! (same `states` layout; only levels `level` and `level - 1` are touched):
 *   down  smoothing_steps!, local_residual!, restrict_to!(next.b, P, curr.r), fill!(next.x, 0)   (src/multigrid.jl:100-106)
 *   up    interpolate_and_sum_to!(curr.x, P, next.x), smoothing_steps!                          (src/multigrid.jl:112-115)
 * hmg_vcycle(k) == down(k); hmg_vcycle(k-1); up(k).  After `down`, x, r (the cell-local residual) and the coarse b
 * hold what the reference leaves; p and Ap are scratch (the library drops the pre-smoother's dead tail).  After `up`,
 * x and r hold what the reference leaves; p and Ap too with option "lean_post" = 0 (see above; `up` acts as the top
 * level of a V-cycle).  With
 * option "swap_rp" each half exchanges the device pointers of the r and p handles once: call them in pairs when
 * r / p wrap caller-owned memory. */
int hmg_vcycle_down(hmg_grid *grid, int level, int steps, hmg_vec **states);
int hmg_vcycle_up(hmg_grid *grid, int level, int steps, hmg_vec **states);

/* ---- multi-GPU hooks (one process per GPU; the host layer owns the communicator) ------------------
 * The grid of a rank holds the cells that rank owns.  Entities shared with other ranks are listed by
 * hmg_grid_set_cut(): per cut entity a global cut id and the local first copy.  After the local
 * interface sum the library packs one value per cut DOF into the exchange buffer, calls `exchange`
 * (an in-place sum over ranks, e.g. RCCL allreduce issued by the host on the same stream) and writes
 * the result back to every local copy.  `scalar_sum` does the same for `count` device doubles
 * (the CG dot products). */
typedef int (*hmg_exchange_fn)(void *user, void *device_buf, int64_t count);
int hmg_grid_set_cut(hmg_grid *grid, int64_t ncut_global_faces, int64_t ncut_global_edges, int64_t ncut_global_nodes,
                     int64_t nlocal_faces, const int64_t *face_gid, const int32_t *face_cell_lid,
                     int64_t nlocal_edges, const int64_t *edge_gid, const int32_t *edge_cell_lid,
                     int64_t nlocal_nodes, const int64_t *node_gid, const int32_t *node_cell_lid);
int hmg_grid_set_exchange(hmg_grid *grid, hmg_exchange_fn exchange, hmg_exchange_fn scalar_sum, void *user,
                          void *device_exchange_buf, int64_t exchange_buf_doubles);
/* Asynchronous form of the exchange: begin() starts the in-place sum over ranks of device_buf[0..count) and
 * returns, end() makes the context's stream wait for it.  With it the smoother overlaps the exchange with the
 * apply / interface sums of the cells that do not touch a partition cut (hmg_grid_set_overlap, default on). */
int hmg_grid_set_exchange_async(hmg_grid *grid, hmg_exchange_fn begin, int (*end)(void *user));
int hmg_grid_set_overlap(hmg_grid *grid, int enabled);
/* (context option "overlap_min_doubles", default 524288: levels whose GLOBAL cut -- one value per cut DOF, the same number
 *  on every rank, so that all ranks decide alike and issue their collectives in one order -- is smaller than 4 MiB run in
 *  the plain form even with the overlap on: splitting the launches of a launch-bound small level costs more than its
 *  messages hide) */
/* Exchange among the sharers only (SURVEY 8e's cheaper alternative; the default of hmg_grid_use_comm).  The library groups
 * the cut entities into SEGMENTS by the set of ranks that share them (octants: a quarter of a cut plane = 2 ranks, half an
 * axis line = 4, the centre node = 8), lays this rank's segments out one after the other in the exchange buffer (a segment
 * has the same length and order on each of its members) and, per exchange, hands the transport a list of messages
 * msgs[4 i ..] = {peer rank, buffer offset, count, staging offset} (doubles): send device_buf[offset .. +count) to the
 * peer, receive the peer's partial segment into device_stage[staging offset .. +count).  Two ranks list the segments they
 * share in the same order, so the k-th message a -> b meets the k-th b <- a.  The library then adds the members' partials
 * in ascending rank order (identical bits on every member).  p2p runs on the context's stream; p2p_begin only starts the
 * messages and the `end` of hmg_grid_set_exchange_async joins them.  enabled = 0: back to the all-reduce over the global
 * cut buffer.  hmg_grid_set_exchange still supplies user, the exchange buffer, scalar_sum and `exchange` (level-1 gather). */
typedef int (*hmg_p2p_fn)(void *user, void *device_buf, void *device_stage, int64_t nmsgs, const int64_t *msgs);
int hmg_grid_set_exchange_p2p(hmg_grid *grid, int enabled, hmg_p2p_fn p2p, hmg_p2p_fn p2p_begin, void *device_stage_buf,
                              int64_t stage_buf_doubles);
int64_t hmg_grid_cut_stage_doubles(const hmg_grid *grid);          /* staging capacity needed (max over levels) */
/* the message list of one level: 4 int64 per message (count = numbers written) */
int hmg_grid_exchange_messages(const hmg_grid *grid, int level, int64_t *out, int64_t cap, int64_t *count);
/* level = 0: capacity needed for every level and for the coarse gather */
int64_t hmg_grid_cut_buffer_doubles(const hmg_grid *grid, int level);
void *hmg_ctx_scalar_bank(hmg_ctx *ctx);
/* device_doubles16 = NULL restores the library's own bank (call it before the caller-owned memory goes away) */
int hmg_ctx_set_scalar_bank(hmg_ctx *ctx, void *device_doubles16);
/* the context's HIP stream (a hipStream_t): a host layer that issues its own collectives must issue them here */
void *hmg_ctx_stream(hmg_ctx *ctx);

/* ---- in-library communicator: RCCL over xGMI, one rank per GPU ------------------------------------------------------
 * The reference's exchange point is broadcast_interfaces! (src/implicit_fine_grid.jl:209-328; called at
 * src/multigrid.jl:51,61,75) plus the three dot products of a CG step (src/multigrid.jl:54,64,67).  With a communicator
 * the library does them itself: ncclAllReduce of the packed cut buffer (on a second HIP stream, event-ordered, when the
 * overlap is on) and of the CG scalars (two neighbouring slots of the scalar bank at CG step 0: one call), all enqueued
 * on streams -- no host code runs between two kernels of a V-cycle.  Usage: rank 0 calls hmg_comm_unique_id and hands
 * the 128 bytes to every rank (any channel: MPI, a file, torch.distributed), every rank calls hmg_comm_init on its
 * context, creates its grid with hmg_grid_create_partition and calls hmg_grid_use_comm.  librccl is opened at run time
 * (dlopen), so single-GPU use does not need it. */
#define HMG_COMM_ID_BYTES 128
int hmg_comm_unique_id(void *out128);
int hmg_comm_init(hmg_ctx *ctx, int nranks, int rank, const void *unique_id128);
int hmg_comm_destroy(hmg_ctx *ctx);
int hmg_comm_stats(hmg_ctx *ctx, int64_t *calls, int64_t *doubles);   /* collectives issued so far, doubles moved */
int hmg_grid_use_comm(hmg_grid *grid);
/* sum over ranks of `count` (<= 4) host doubles, in place, blocking: the driver's per-cycle integrals */
int hmg_comm_sum_host(hmg_ctx *ctx, double *vals, int count);
/* Grid of the cells owner[c] == rank of a global base mesh.  The library derives the local mesh, the cut
 * entities (global ids identical on all ranks), global multiplicities and Dirichlet / first-copy masks, and
 * keeps the global mesh for a replicated level-1 solve.  hmg_grid_set_operator takes the GLOBAL sigma.
 * hmg_grid_set_cut is called internally; the host only supplies hmg_grid_set_exchange. */
int hmg_grid_create_partition(hmg_ctx *ctx, int dim, int nlevels, int64_t nnodes, const double *coords, int64_t ncells,
                              const int64_t *cells, const int32_t *owner, int rank, int nranks, hmg_grid **out);
/* Rehearsal of a larger partition on fewer GPUs (measurement / test aid, not a production entry point).
 * cut_owner[c] decides which entities count as cut (their copies' cells differ in cut_owner) while owner[] still decides
 * which cells are local: owner = 0 everywhere, cut_owner = the octant of a cell and a 1-rank communicator walk the
 * cut-first cell lists, pack / unpack kernels, events and the all-reduce of an 8-rank partition with every copy held
 * locally -- results equal the unpartitioned grid's bit for bit.  cut_owner = NULL: as hmg_grid_create_partition.
 * Context option "comm_rehearsal" = 1 lets a grid that holds rank r's share of an N-rank partition use a communicator
 * of another size (timing only: the neighbours' contributions are missing from the sums). */
int hmg_grid_create_partition_rehearsal(hmg_ctx *ctx, int dim, int nlevels, int64_t nnodes, const double *coords,
                                        int64_t ncells, const int64_t *cells, const int32_t *owner, const int32_t *cut_owner,
                                        int rank, int nranks, hmg_grid **out);

/* ---- host-side problem synthesis (threaded; HMG_SETUP_THREADS, default: all cores up to 16) -------------------------
 * hypercube(ElT, n; origin) + order_nodes_and_elements_by_magnitude (src/tet/generate_grid.jl:6-45,
 * src/tri/generate_grid.jl:6-35, src/examples/homogenized_coefficients.jl:21-28) in one call: a box of shape[] unit
 * cubes (6 tetrahedra / 2 triangles each), node ids with the last coordinate fastest.  transposed_lookup = 1 reproduces
 * the reference's corner lookup (first-index-fastest id table, cubes only: the geometry comes out transposed, as in
 * the reference); 0 keeps node id and coordinates aligned (boxes for multi-GPU weak scaling).  ordered = 1 sorts nodes
 * and cells by infinity norm (stable) so that every centred sub-box is a prefix.  cells: 1-based ascending tuples. */
int hmg_checkerboard_mesh_size(int dim, const int64_t *shape, int64_t *nnodes, int64_t *ncells);
int hmg_checkerboard_mesh(int dim, const int64_t *shape, const double *origin, int transposed_lookup, int ordered,
                          double *coords /* dim*nnodes */, int64_t *cells /* (dim+1)*ncells */);
/* conductivity_per_element(mesh, sigma, offset) (src/examples/homogenized_coefficients.jl:494-503): sigma[c] =
 * sigma_grid[trunc(centre_c + offset) - 1]; sigma_grid: grid_shape[0] x .. x grid_shape[dim-1] x dim, C order */
int hmg_conductivity_per_element(int dim, int64_t nnodes, const double *coords, int64_t ncells, const int64_t *cells,
                                 const int64_t *grid_shape, const double *sigma_grid, const double *offset, double *sigma);

/* owner[c] = row-major index of the width^dim block (blocks[] of them per axis, counted from origin) that holds the
 * centre of cell c -- the ownership partition of the multi-GPU runs (halves / quadrants / octants about the origin) */
int hmg_block_owner(int dim, int64_t nnodes, const double *coords, int64_t ncells, const int64_t *cells,
                    const int64_t *blocks, double width, const double *origin, int32_t *owner);

#ifdef __cplusplus
}
#endif
#endif /* HMG_H */
