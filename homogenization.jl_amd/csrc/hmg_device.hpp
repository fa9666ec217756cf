// Device-side view of the tables and the kernel launch API (implemented in hmg_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace hmg {

// pos32 / sweep32 / sweep_slot are followed by this many padding entries (0 / 0 / 0xffff): k_apply prefetches
// its addressing words two iterations (<= 2 x 1024 threads) ahead of a trip count rounded up to the block size
constexpr int TABLE_PAD = 3 * 1024;

struct LevelDev {
    int dim, level, m;
    int nf, ld;
    int ncorner, nedge, nface;
    int nei, nfi, nint;
    int off_edge, off_face, off_int;
    int ncls, ndir, nterm;
    int lds_g0, lds_g1;
    int nf_coarse;                 // nf of level-1 (0 on level 1)
    const uint64_t *meta;          // [nf]
    const uint16_t *lpos;          // [nf] LDS lattice index of every storage slot
    const uint16_t *sweep_slot;    // [nsweep + TABLE_PAD] interior sweep in LDS lattice order: storage slot, or 0xffff for
                                   //          the two surface end positions of a row (idle lanes)
    int nsweep;
    const uint32_t *pos32;         // [nf + TABLE_PAD] L | j<<16 | k<<22 | cls<<28 (levels whose cell fits the LDS)
    const uint32_t *pos32w;        // [nf]     i | j<<7 | k<<14 | cls<<21  (3D, any level)
    const uint32_t *sweep32;       // [nsweep + TABLE_PAD] same packing, cls = 0
    // register-blocked interior (3D, see interior_block): one entry per interior (i,j) of every blk_R-th k-plane
    const uint32_t *blk_word;      // [nblk + TABLE_PAD] L | j<<16 | k0<<22 | nv<<28
    const uint16_t *blk_slot;      // [nblk + TABLE_PAD] storage slot of node (i,j,k0)
    int nblk, blk_R;               // blk_R = 0: no blocked tables for this level
    const double *ctab;            // [ncls*ndir*nterm]
    const int32_t *hier2slot;      // [nf]
    const int32_t *par_a, *par_b;  // [nf]     (level > 1)
    const uint32_t *par32;         // [nf]     par_a | par_b << 16 (coarse storage slots; level > 1)
    // folded prolongation with the coarse column staged at the EVEN nodes of the lattice image (3D levels with blocked tables):
    const uint64_t *par64;         // [nf]     LDS lattice position of parent a | of parent b << 16 | of the slot itself << 32
    const uint16_t *clpos;         // [nf_coarse] LDS lattice position (fine lattice, node (2i,2j,2k)) of every coarse slot
    // restriction in the epilogue of the local residual (k_apply<.., RS>): per coarse slot the addressing word of its fine
    // node (decode32w: i | j<<7 | k<<14 | cls<<21) and the weights 1 / 0.5 / 0 of the 15 taps per entity class
    const uint32_t *rs_word;       // [nf_coarse]
    const double *rs_w;            // [ncls * ndir]
    // ... or, on levels whose stand-alone restriction is k_restrict (summation in the reference's order: ascending fine
    // hierarchical id), the same lists with lattice positions in place of storage slots: entry e of ridx -> rs_lp[e]
    const uint16_t *rs_lp;         // [size of ridx] or null (levels restricted by the slab kernel: stencil order above)
    const int32_t *rptr, *ridx;    // [nf_coarse+1], [..] (level > 1)
    const double *dphi;            // [3*nf]
    // one-wave-per-cell apply of level 5 (k_apply_wave in hmg_apply_wave.hip): per-lane tables, loaded once per wave
    const uint32_t *wave_tab;      // [WAVE_TAB_ROWS * 64] addressing words of the lane's surface runs and interior blocks
    const uint32_t *wave_lpos;     // [8 * 64]  LDS byte offsets of the lane's 16 slots, two per word
    const uint32_t *wave_par;      // [16 * 64] lattice positions of the two parents of the lane's 16 slots (a | b << 16)
    const uint32_t *wave_cl;       // [3 * 64]  LDS byte offset of the lane's coarse slots (even lattice nodes of the image)
    const uint32_t *wave_rs;       // [192 * 8] per coarse slot: 15 lattice positions (u16) of its restriction sum + their number
    // class-weight cache (hmg_grid_set_operator): W[cell class][sign of alpha][entity class][16] = sum_t ctab * scale_t
    const double *wcache;          // null: not available on this level
};

constexpr int WAVE_TAB_ROWS = 15;
constexpr int WAVE_ROW = 16;                  // doubles per class row of the weight cache (15 taps + 1 pad: 128-B rows)
constexpr int WAVE_WSTRIDE = 15 * WAVE_ROW;   // doubles per (cell class, sign)

// Rolling-window tables of k_apply_slab (cells larger than the LDS), one set per such level.
struct SlabTables {
    const int *head;             // per slab 8 ints: k0, ld_off, ld_cnt, cp_off, cp_cnt, cp_surf (surface entries first), 0, 0
    const uint32_t *ld_word;     // slots new in the window: LDS lattice position L | slot << 16   (+ TABLE_PAD)
    const uint32_t *cp_word;     // evaluated slots: i | j<<7 | k<<14 | cls<<21 (decode32w)        (+ TABLE_PAD)
    const uint16_t *cp_slot;     // evaluated slots: storage slot                                  (+ TABLE_PAD)
    int nslab, lds_nodes;
    int max_surf;                // most surface entries (cp_surf) of any slab
    int max_int;                 // most interior entries (cp_cnt - cp_surf) of any slab
};

struct MeshDev {
    int dim;
    int64_t ncells, nnodes;
    const int32_t *cells;        // (dim+1)*ncells
    const int32_t *face_pairs;   // 3*nfacepairs
    int64_t nfacepairs;
    const int32_t *face_partner; // 4*ncells: partner cell << 2 | partner face of every face (-1: boundary / cut / 2D)
    const int32_t *edge_ptr, *edge_ent;
    int64_t nsharededges;
    const int32_t *node_ptr, *node_ent;
    int64_t nsharednodes;
    const int32_t *node_first;   // nnodes
    const uint16_t *dmask, *dupmask;
    const double *coef;          // 8 per cell
    int64_t ncut_edge_groups, ncut_node_groups;   // leading groups of the edge / node CSR that are cut
    int64_t ncut_face_pairs;                      // leading face pairs that count as cut (rehearsal partitions only, else 0)
    const int32_t *cells_cut, *cells_inner;       // partitioned grids: cell lists for the overlapped exchange
    const int32_t *cell_perm;                     // option cell_order: workgroup b of a full-grid register-blocked apply works on cell_perm[b]
    int64_t ncells_cut, ncells_inner;
    const uint8_t *mult;         // 16 per cell: number of copies of each entity (bit order of the masks)
    // class-weight cache: cells with bitwise equal coefficient rows share a class (checkerboards: <= 48); null = no cache
    const int32_t *cell_class;
    int nclasses;
    double wc_lambda;            // the lambda the cached weights were formed with
    double *blockpart;           // 2 per cell: scratch for the fused apply's block sums
    // slab decomposition of the finest level when one cell exceeds the LDS (set per launch by the host)
    SlabTables slab;             // set per level before an apply of a level whose cell exceeds the LDS
};

struct ApplyArgs {
    double alpha, lambda;
    const double *x;       // input column; with x2: xin = x + beta * x2, beta = scal[s_num] / scal[s_den]
    const double *x2;
    double *xout;          // optional: xin written back (p-update / p = r)
    double *xacc;          // optional: xacc += (scal[a_num] / scal[a_den]) * x2  (the previous step's x-update)
    int a_num, a_den;
    const double *x3;      // optional (k_apply FUSED, with x2, without xacc): xin = (x + ax*x2) + c*(x3 + beta*x2),
    int c_num, c_den;      //   ax = scal[a_num]/scal[a_den], c = scal[c_num]/scal[c_den]: two CG x-updates at once
    const double *xcoarse; // optional (fused kernel, not the slab one): xin = x + P xcoarse first (prolongation of the
    int64_t ldc;           //   coarse-grid correction; column stride ldc), written back through xout
    const double *src;     // optional: out = src + alpha * A * xin
    double *out;
    double *rcoarse;       // optional (k_apply RS instantiation): restrict_to!(rcoarse, P, out) in the epilogue -- column stride
    int64_t ldrc;          //   ldrc; `out` itself may then be null (the cell-local residual is not stored at all)
    const double *scal;
    int s_num, s_den;
    double *blockpart;
    const uint8_t *mult;
    int flags;             // bit 0: Dirichlet constraint on out; bit 1: mass term only; bit 2 (slab kernel): weights =
                           // last term of the class table, no per-cell scaling (restriction); bit 3 (fused): src
                           // multiplies instead of being added, sum mult (x + src) out; bit 4: mass term not scaled
                           // by |J|; bit 5 (fused): unit multiplicities; bit 6 (fused, xcoarse): the coarse column is staged
                           // in the lattice image; bit 7 (fused, x3 mode): x is zero and is not read
    int64_t ncells_prefix; // > 0: only the first ncells_prefix cells
    int64_t out_ld;        // column stride of out if it differs from the level's (slab restriction), else 0
    const int32_t *cell_list;   // optional: workgroup b works on cell cell_list[b] (ncell_list of them)
    int64_t ncell_list;
    int64_t nwork;         // set by the launcher: work items (cells) of this launch -- one-wave workgroups loop over them
    const int32_t *cell_class;  // set by the launcher (WC instantiations): the cell's class in the class-weight cache
};

struct CoarseDev {
    int64_t n;
    const int32_t *rowptr, *colidx;
    const double *val, *diag;
    const int32_t *interior;     // n
};

// scalar bank slots (device doubles)
// (S_DONE / S_ITER / S_CRR: device-side state of the coarse PCG -- converged flag, iterations done, last r.r;
//  slots 12..15: host sums of hmg_comm_sum_host; S_RS3 / S_PAP3: the third pair of a smoother that defers three x-updates --
//  the bank is 16 doubles by contract (hmg_ctx_set_scalar_bank), so they share S_C1, which the coarse PCG leaves alone, and S_TMP, a
//  temporary of calls that cannot run inside a smoother)
enum { S_RS = 0, S_PAP = 1, S_RS2 = 2, S_TMP = 3, S_C0 = 4, S_C1 = 5, S_C2 = 6, S_C3 = 7, S_PAP2 = 8, S_DONE = 9, S_ITER = 10,
       S_CRR = 11, S_HOST = 12, S_COUNT = 16,
       S_RS3 = S_C1, S_PAP3 = S_TMP };

struct Launch {
    hipStream_t stream;
    double *partials;     // >= 4096 doubles
    double *rpart;        // block partials of the streaming reductions: one per 512 vector entries
    int64_t rpart_cap;
    double *scal;         // S_COUNT doubles
    int num_cu;
    int apply_threads;    // 0 = auto
    int apply_mass_only;  // 1: only the mass term (next_rhs!), set around a single launch
    int apply_unblocked;  // 1: node-per-thread interior sweep instead of the register-blocked one (dev / A-B knob)
    int apply_wg512;      // 1 (default): cells that would take the 1024-thread register-blocked instantiation take the 512-thread one:
                          // three workgroups (three columns in flight) per CU instead of two
    int cell_order;       // 1 (default): full-grid register-blocked apply launches walk the cells XCD by XCD (MeshDev::cell_perm)
    int weight_cache;     // 1 (default): level 6 takes its class weight rows from the class-weight cache where it exists
    int apply_pack;       // 1 (default): cells of at most 16 nodes (3D level 2) four to a wave (k_apply_pack)
    int apply_small;      // 1 (default): levels 2-4 (3D, <= 192 nodes per cell) take the pipelined one-wave kernel where the cache exists
    int64_t *n_small_launches;
    int apply_wave;       // 1 (default): level 5 takes the one-wave-per-cell kernel where the class-weight cache exists
    int64_t wave_grid;    // its grid: waves resident at once (16 per CU)
    int64_t *n_wave_launches;   // counts its launches (hmg_ctx_counter "wave_launches"; tests check that the path is taken)
    int apply_slab2;      // 1 (default): cells larger than the LDS (level 7) take the role-split persistent kernel (hmg_apply_slab.hip)
    int64_t *n_slab2_launches;
    int64_t slab2_grid;   // its grid (0: one workgroup per CU)
    int restrict_slab2;   // 1 (default): the stand-alone restriction of levels with slab tables goes through it too (eight loader waves)
    int slab2_force;      // 1 (experiment): every 3D level with slab tables of two slabs or more takes it (level 6 with HMG_SLAB_LDS_KB <= 30)
    int64_t persistent_waves;   // grid of the one-wave apply instantiations (default 32 per CU: what is resident at once); they
                          // loop over the cells.  Larger than the number of cells = one workgroup per cell
};

// out = (src ? src : 0) + alpha * A x, then (use_mask) zero Dirichlet DOFs.  src may alias out.
void launch_apply(const Launch &L, const LevelDev &lv, const MeshDev &mesh, double alpha, double lambda,
                  const double *x, const double *src, double *out, int use_mask);
size_t apply_lds_bytes(const LevelDev &lv);
// can the fused apply of this level restrict its output in its epilogue (ApplyArgs::rcoarse)?
bool apply_restricts(const Launch &L, const LevelDev &lv);
// Fused CG pass, see k_apply<.., FUSED>: the kernel (possibly over a cell list, several launches) leaves
// per-cell partial sums in mesh.blockpart; the reduce step turns them into scal[slot_pap] = sum mult*xin*out and
// scal[slot_rr] = sum xin*xin (slot_rr < 0: not wanted).  a.scal / a.mult / a.blockpart are filled in here.
void launch_apply_fused_kernel(const Launch &L, const LevelDev &lv, const MeshDev &mesh, ApplyArgs a);
void launch_apply_fused_reduce(const Launch &L, const MeshDev &mesh, int slot_pap, int slot_rr);
void launch_apply_args(const Launch &L, const LevelDev &lv, const MeshDev &mesh, ApplyArgs a);
// one-wave-per-cell path (hmg_apply_wave.hip): can this launch take it / launch it (a.scal, a.mult, a.blockpart filled in)
bool apply_wave_ok(const Launch &L, const LevelDev &lv, const MeshDev &mesh, const ApplyArgs &a, bool fused);
void launch_apply_wave(const Launch &L, const LevelDev &lv, const MeshDev &mesh, const ApplyArgs &a, bool fused);
// cells larger than the LDS (hmg_apply_slab.hip): one persistent workgroup per CU, loader and evaluator waves
bool apply_slab2_ok(const Launch &L, const LevelDev &lv, const MeshDev &mesh, const ApplyArgs &a);
void launch_apply_slab2(const Launch &L, const LevelDev &lv, const MeshDev &mesh, const ApplyArgs &a, bool fused);
// small 3D levels (hmg_apply_small.hip): one persistent, software-pipelined wave per cell
bool apply_small_ok(const Launch &L, const LevelDev &lv, const MeshDev &mesh, const ApplyArgs &a, bool fused);
void launch_apply_small(const Launch &L, const LevelDev &lv, const MeshDev &mesh, const ApplyArgs &a, bool fused);
// W[class][sign][entity class][16] for every distinct coefficient row (coef_rep: 8 doubles per class)
void launch_weight_cache(const Launch &L, const LevelDev &lv, const double *coef_rep, int nclasses, double lambda, double *wcache);

// which: 0 everything; 1 only the cut edge / node groups (and cut face pairs); 2 everything else (faces, non-cut groups)
// faces = false leaves the non-cut shared faces alone (their sum then rides in launch_cg_rupdate_faces)
void launch_interface_sum(const Launch &L, const LevelDev &lv, const MeshDev &mesh, double *x, int which = 0,
                          bool faces = true);
void launch_mask(const Launch &L, const LevelDev &lv, const MeshDev &mesh, double *x, int which /*0 dmask,1 dupmask*/);
void launch_restrict_slab(const Launch &L, const LevelDev &fine_rtab, const MeshDev &mesh, const SlabTables &st, int ldc,
                          const double *rf, double *bc);
void launch_restrict(const Launch &L, const LevelDev &fine, const LevelDev &coarse, int64_t ncells,
                     const double *rf, double *bc);
void launch_prolong_add(const Launch &L, const LevelDev &fine, const LevelDev &coarse, int64_t ncells,
                        const double *xc, double *xf);

void launch_fill(const Launch &L, double *x, int64_t n, double v);
void launch_copy(const Launch &L, double *dst, const double *src, int64_t n);
void launch_axpy(const Launch &L, double a, const double *x, double *y, int64_t n);
void launch_xpby(const Launch &L, const double *r, double b, double *p, int64_t n);
// scal[slot] = sum x*y   (deterministic two-stage reduction)
void launch_dot(const Launch &L, const double *x, const double *y, int64_t n, int slot);
// scal[slot] = sum over first copies only of x*x
void launch_norm2_unique(const Launch &L, const LevelDev &lv, const MeshDev &mesh, const double *x, int slot);
// p = r; scal[slot] = r.r
void launch_copy_dot(const Launch &L, double *p, const double *r, int64_t n, int slot);
// alpha = scal[s_num]/scal[s_den]; x += alpha p; r -= alpha q; scal[s_out] = r.r
void launch_cg_update(const Launch &L, double *x, double *r, const double *p, const double *q, int64_t n,
                      int s_num, int s_den, int s_out);
// alpha = scal[s_num]/scal[s_den]; r -= alpha q; scal[s_out] = r.r
void launch_cg_rupdate(const Launch &L, const double *r, double *rout, const double *q, int64_t n, int s_num, int s_den,
                       int s_out);
// the same with the face part of q's interface sum taken on the fly (q unsummed on the shared faces)
void launch_cg_rupdate_faces(const Launch &L, const LevelDev &lv, const MeshDev &mesh, const double *r, double *rout,
                             const double *q, int64_t n, int s_num, int s_den, int s_out);
void launch_cg_rupdate_faces_x(const Launch &L, const LevelDev &lv, const MeshDev &mesh, const double *r, double *rout,
                               const double *q, int64_t n, int s_num, int s_den, int s_out, double *x, const double *p, int a_num,
                               int a_den, int b_num, int b_den, const double *p0, int z_num, int z_den);
// x += (scal[a_num]/scal[a_den]) p; with_p: p = r + (scal[s_num]/scal[s_den]) p
void launch_cg_xp_update(const Launch &L, double *x, double *p, const double *r, int64_t n, int a_num, int a_den,
                         int s_num, int s_den, int with_p);
// x = (x + (scal[a_num]/scal[a_den]) p) + (scal[c_num]/scal[c_den]) (r + (scal[b_num]/scal[b_den]) p)
void launch_cg_x2_update(const Launch &L, double *x, const double *p, const double *r, int64_t n, int a_num, int a_den, int b_num,
                         int b_den, int c_num, int c_den);
// beta = scal[s_num]/scal[s_den]; p = r + beta p
void launch_cg_pupdate(const Launch &L, double *p, const double *r, int64_t n, int s_num, int s_den);

void launch_gather_base(const Launch &L, const MeshDev &mesh, int ld1, const double *v1, double *u);
void launch_scatter_base(const Launch &L, const MeshDev &mesh, int ld1, const double *u, double *v1);

void launch_gather_owned(const Launch &L, const MeshDev &mesh, const int32_t *nodes_g, const int32_t *owned, int ld1,
                         const double *v1, double *ug);
void launch_scatter_cells(const Launch &L, const int32_t *cell_nodes, int64_t ncells, int npc, int ld1, const double *u,
                          double *v1);

void launch_permute(const Launch &L, const LevelDev &lv, int64_t ncells, const double *src, double *dst,
                    int to_storage);
void launch_fill_random(const Launch &L, const LevelDev &lv, int64_t ncells, double *x, uint64_t seed,
                        int64_t cell_offset);

// preconditioned-CG pieces (Jacobi / Chebyshev) for the level-1 system
void launch_coarse_gather_rhs(const Launch &L, const CoarseDev &A, const double *u, double *b);
void launch_coarse_scatter_sol(const Launch &L, const CoarseDev &A, int64_t nnodes, const double *x, double *u);
void launch_coarse_init(const Launch &L, const CoarseDev &A, const double *b, double *x, double *r, double *z,
                        double *p, double zscale = 1.0, double *dcheb = nullptr);   // x=0,r=b,z=r/d,p=z, scal[S_C0]=r.z, scal[S_C2]=b.b, S_DONE = S_ITER = 0
// polynomial (Chebyshev) preconditioner of the level-1 PCG, see k_coarse_cheb: with dcheb the init / update kernels leave
// z = d = zscale D^-1 r (the first iterate), launch_coarse_cheb does one more step, the last one leaves the partials of r.z
void launch_coarse_cheb(const Launch &L, const CoarseDev &A, const double *r, const double *zin, double *zout, double *d, double c1,
                        double c2, int last);
void launch_coarse_rz_from_cheb(const Launch &L, const CoarseDev &A);
// one iteration = launch_coarse_direction + launch_coarse_update; r.z lives in scal[slot_old] -> scal[slot_new] (S_C0 / S_C3,
// exchanged by the caller every iteration), the other dot products stay in block partials that the consumer kernel sums itself
void launch_coarse_update(const Launch &L, const CoarseDev &A, double *x, double *r, double *z, const double *p,
                          const double *q, int slot_old, double zscale = 1.0, double *dcheb = nullptr);   // alpha = rz/p.q; partials of r.z (new), r.r
void launch_coarse_residual_norm(const Launch &L, const CoarseDev &A);
// two-launch iteration: direction (beta, convergence bookkeeping, p = z + beta p, q = A z + beta q, partials of p.q) + update;
// mode 0 regular, 1 first of a solve, 2 bookkeeping only; count_it: see k_coarse_direction
void launch_coarse_direction(const Launch &L, const CoarseDev &A, double *p, double *q, const double *z, int slot_old, int slot_new,
                             double rtol2, int mode, int count_it, int rz_from_cheb = 0);   // scal[S_TMP] = r.r of the last update

// b[slot, cell] = dot(dphi[slot], pvec[cell])   (rhs_a xi grad v)
void launch_rhs_dphi(const Launch &L, const LevelDev &lv, int64_t ncells, const double *pvec, double *b);

// scal[slot] = integral over the first nsub cells; second = b (mode 0, first term) / the previous iterate (mode 1)
void launch_integrate(const Launch &L, const LevelDev &lv, const MeshDev &mesh, int mode, int64_t nsub, const double *v,
                      const double *second, int slot);

// multi-GPU cut exchange, all three kinds (faces, edges, nodes) at once: unpack = 0 packs buf[pos[e] + k] <- x (first
// local copy), 1 writes x <- buf[pos[e] + k]; pos[e] = first buffer position of the run of local cut copy e (layout:
// CutLevel in hmg_capi.cpp)
struct CutPackArgs {
    int64_t n[3];                 // local cut copies per kind
    const int64_t *pos[3];
    const int32_t *cell_lid[3];   // cell * 8 + local entity
    const uint8_t *first[3];      // 1: the first local copy of its entity (the one that is packed)
};
void launch_cut_pack(const Launch &L, const LevelDev &lv, const CutPackArgs &c, double *buf, double *x, int unpack);
// sharers-only exchange: every buffer position of this rank's segments <- sum over the segment's members in ascending
// rank order (own partial from buf, the others' from stage); plan: CutLevel::plan
void launch_seg_sum(const Launch &L, const int64_t *plan, int64_t ndoubles, double *buf, const double *stage);

}  // namespace hmg
