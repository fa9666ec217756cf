// C ABI (include/hmg.h): object lifetimes, table upload, primitive dispatch, smoother and V-cycle
// orchestration on one HIP stream with device-resident CG scalars.
#include "../../include/hmg.h"
#include "hmg_device.hpp"
#include "hmg_host.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

using namespace hmg;


#define HIPCHK(expr)                                                                            \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " (" #expr ")"); \
    } while (0)

namespace {

void release_pooled_memory();   // every live context hands its pooled level-vector blocks back (defined behind hmg_ctx)

// Handles may be destroyed from another thread than the one that computes (finalizers of a garbage-collected host: Julia
// runs them where it likes, Python's weakref.finalize on the collecting thread): reference counts, the registry of live
// contexts and the pooled level-vector blocks are guarded by this lock.  (Recursive: a failed allocation inside a guarded
// region hands the pools back.)  Everything else on one context is for one host thread at a time (include/hmg.h).
std::recursive_mutex &lifetime_mutex()
{
    static std::recursive_mutex m;
    return m;
}
using LifetimeLock = std::lock_guard<std::recursive_mutex>;

// Device / pinned allocations the library has made so far (hmg_ctx_counter "device_allocs"): after setup -- grid, operator, level-1
// system, level vectors -- a V-cycle makes none (tests/test_gpu_parity.py::test_no_allocation_inside_a_vcycle).
std::atomic<int64_t> &device_allocs()
{
    static std::atomic<int64_t> n{0};
    return n;
}

// Host-only grids (hmg_grid_create with a NULL context: table queries, and the CPU sanitizer job of tests/test_sanitizers.py)
// run every table builder as a device grid does; inside a DryUploads scope the uploads keep a running checksum of what WOULD have
// gone to the device instead of touching the HIP runtime (hmg_grid_table_i32 "upload_hash": the same mesh must give the same
// tables whatever the allocator hands out -- an uninitialised read shows up as a checksum that moves with ASan's malloc fill).
struct DryUploads {
    static DryUploads *&current()
    {
        static thread_local DryUploads *c = nullptr;
        return c;
    }
    bool dry;
    uint64_t *hash;
    DryUploads *prev;
    DryUploads(bool dry_, uint64_t *hash_) : dry(dry_), hash(hash_), prev(current()) { current() = this; }
    ~DryUploads() { current() = prev; }
    DryUploads(const DryUploads &) = delete;
    DryUploads &operator=(const DryUploads &) = delete;
    static bool active() { return current() && current()->dry; }
    static void note(const void *data, size_t bytes)
    {
        uint64_t h = *current()->hash ^ (bytes * 0x9e3779b97f4a7c15ull);
        const unsigned char *b = (const unsigned char *)data;
        for (size_t i = 0; i < bytes; ++i) h = (h ^ b[i]) * 1099511628211ull;
        *current()->hash = h;
    }
};

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    void alloc(size_t count)
    {
        release();
        n = count;
        if (!count || DryUploads::active()) return;
        if (hipMalloc((void **)&p, count * sizeof(T)) != hipSuccess) {
            (void)hipGetLastError();
            release_pooled_memory();             // blocks the contexts keep for reuse may be what is in the way
            HIPCHK(hipMalloc((void **)&p, count * sizeof(T)));
        }
        device_allocs() += 1;
    }
    void upload(const std::vector<T> &h, hipStream_t s)
    {
        alloc(h.size());
        if (DryUploads::active()) {
            DryUploads::note(h.data(), h.size() * sizeof(T));
            return;
        }
        if (!h.empty()) {
            HIPCHK(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
            HIPCHK(hipStreamSynchronize(s));
        }
    }
};

struct LevelBufs {
    DevBuf<uint64_t> meta;
    DevBuf<uint16_t> lpos, sweep_slot;
    DevBuf<int> slab_head, slab_rs_head;
    DevBuf<uint32_t> slab_ld_word, slab_cp_word, slab_rs_word;
    DevBuf<uint16_t> slab_cp_slot, slab_rs_slot;
    DevBuf<double> rtab;   // restriction weights in class-table layout (slab levels)
    int nslab = 0, slab_lds_nodes = 0, slab_max_surf = 0, slab_rs_max_surf = 0, slab_max_int = 0, slab_rs_max_int = 0;
    DevBuf<uint32_t> pos32, pos32w, sweep32, par32, blk_word;
    DevBuf<uint64_t> par64;
    DevBuf<uint16_t> clpos;
    DevBuf<uint32_t> rs_word;
    DevBuf<double> rs_w;
    DevBuf<uint16_t> rs_lp;
    DevBuf<uint16_t> blk_slot;
    int nblk = 0, blk_R = 0;
    // one-wave-per-cell apply of level 5 (k_apply_wave): per-lane tables + the class-weight cache of the current operator
    DevBuf<uint32_t> wave_tab, wave_lpos, wave_par, wave_cl, wave_rs;
    DevBuf<double> wcache;
    DevBuf<double> ctab;
    DevBuf<int32_t> hier2slot, par_a, par_b, rptr, ridx;
    DevBuf<double> dphi;
};

struct CutKind {
    int64_t nglobal = 0;
    int64_t nentries = 0;
    std::vector<int64_t> gid;                    // host: global cut id of every local copy
    std::vector<int32_t> seg;                    // host: segment / index inside it (sharers-only exchange), may be empty
    std::vector<int64_t> sidx;
    DevBuf<int32_t> cell_lid;
    DevBuf<uint8_t> first;
};

// Exchange-buffer layout of one level (built at the first exchange on that level).
//   global layout (all-reduce over every rank): [faces | edges | nodes], a run per GLOBAL cut id -- identical on all ranks;
//   segment layout (exchange among the sharers only): this rank's segments one after the other, inside a segment faces,
//   edges, nodes -- a segment has the same length and order on each of its members.
struct CutLevel {
    bool ready = false;
    DevBuf<int64_t> pos[3];                      // per local cut copy: first buffer position of its run
    int64_t ndoubles = 0;                        // buffer positions used on this level
    // segment layout only:
    std::vector<int64_t> ops;                    // messages, 4 numbers each: peer rank, buffer offset, count, stage offset
    int64_t nstage = 0;                          // staging doubles (the peers' partial segments land there)
    DevBuf<int64_t> plan;                        // k_seg_sum: nseg, then per segment off, size, nmembers, mtab offset; then mtab
};

}  // namespace

struct ApplyTimer {
    bool on = false;
    int min_level = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    std::vector<int> ev_level;          // per used event pair: the level of the launch and its algorithmic bytes
    std::vector<double> ev_bytes;
    size_t used = 0;
    double bytes = 0.0;
    int64_t launches = 0;
};

// Lifetimes: a vector keeps its grid alive, a grid its context (reference counts, single host thread): hmg_*_destroy
// hands the caller's reference back, the object goes when the last dependant has gone -- the order in which a host
// (finalizers of a garbage-collected language in particular) destroys handles does not matter.
struct hmg_ctx {
    int refs = 1;
    ApplyTimer timer;
    bool fuse_cg_default = true;
    bool fold_x = true;   // V-cycle: pre-smoother's last x-update rides with the local residual
    bool swap_rp = true;  // V-cycle: step 0 of a smoother takes r itself as p (pointer exchange), see smooth()
    bool fold_prolong = true;   // V-cycle: prolongation folded into the post-smoother's first residual
    bool lazy_dead = true;      // V-cycle: the pre-smoother's dead last step writes nothing (see smooth())
    bool fold_faces = true;     // fused CG: the face part of Ap's interface sum rides in the r-update (all steps but a live last one)
    bool lean_post = true;      // V-cycle: the post-smoother's dead tail is dropped too (see vcycle_up())
    bool lazy_post = true;      // ... and below the finest level its dead last step writes nothing: both x-updates in one pass
    int lazy_top = 2;           // ... on the finest level its last step leaves both x-updates to the r-update, 2: and the step before its own (see smooth())
    bool zero_entry = true;        // V-cycle: a coarse level's zero initial guess is never materialised (see vcycle_down())
    bool fold_restrict = true;     // V-cycle: the restriction rides in the epilogue of the local residual, which is then not stored
    bool prolong_in_image = true;  // folded prolongation, level 6: the coarse column is staged at the even nodes of the lattice image
                                   // instead of in LDS of its own behind it (three workgroups per CU stay resident)
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    DevBuf<double> partials, scal, rpart;
    Launch L{};
    int coarse_maxit = 5000;
    int coarse_check = 25;
    bool coarse_probe = true;   // budgeted level-1 solves leave a probe behind (off: stream-capture experiments)
    double coarse_rtol = 1e-13;
    int coarse_poly = 4;            // level-1 PCG: Chebyshev iterates per preconditioner application (1 = plain Jacobi)
    double coarse_poly_ratio = 20.0;   // ... on the interval [lmax / ratio, lmax] of D^-1 A, lmax = its Gershgorin bound
    // in-library communicator (one rank per GPU, RCCL over xGMI): hmg_comm_init
    ncclComm_t comm = nullptr;
    int comm_nranks = 1, comm_rank = 0;
    hipStream_t comm_stream = nullptr;       // the overlapped cut exchange runs here
    hipEvent_t ev_packed = nullptr, ev_summed = nullptr;
    int64_t comm_calls = 0, comm_doubles = 0;
    int64_t small_launches = 0;              // launches of the pipelined small-level apply
    int64_t wave_launches = 0;               // launches of the one-wave-per-cell apply (hmg_ctx_counter)
    int64_t slab2_launches = 0;              // launches of the role-split slab apply (hmg_apply_slab.hip)
    int64_t spare_bytes = 0;                 // spare direction vectors held by this context's grids (reserve_top_spare)
    int last_top_form = 0;                   // form the last finest-level post-smoother inside hmg_vcycle took: 0 plain, 1 two-update, 2 three-update
    // Level-vector memory handed back by hmg_vec_destroy, kept for the next hmg_vec_create of the same size: on this
    // platform hipMalloc of memory the process has freed before costs ~35 ms per GB (tools/dev/alloc_probe.hip: 6 x 10 GB
    // 0.001 s fresh, 2.05 s after a hipFree), i.e. 1.9 s of the 71 GB a second driver call allocates.
    bool vec_pool_on = true;
    std::vector<std::pair<size_t, void *>> vec_pool;
    // rehearsal on fewer GPUs than the partition is meant for: a grid that holds rank r's share of an N-rank partition
    // may use a communicator of another size (the neighbours' contributions are then simply missing from the sums --
    // the work per rank, the message sizes and the stream choreography are the real ones, the numbers are not)
    bool comm_rehearsal = false;
    int64_t overlap_min_doubles = 524288;   // levels whose GLOBAL cut is below 4 MiB (about 1 MiB per rank at octants) are
                                            // exchanged in the plain form (see apply_then_sum)
    // grids of this context whose last budgeted level-1 solve still has its probe in flight: judged at the next call that
    // synchronises the stream anyway (norms, dot products, integrals, hmg_ctx_sync, downloads)
    std::vector<struct hmg_grid *> probe_grids;
};

namespace {

void vec_pool_trim(hmg_ctx *c)
{
    LifetimeLock lock(lifetime_mutex());
    if (c->vec_pool.empty()) return;
    (void)hipStreamSynchronize(c->stream);
    for (auto &b : c->vec_pool) (void)hipFree(b.second);
    c->vec_pool.clear();
}

std::vector<hmg_ctx *> &live_contexts()
{
    static std::vector<hmg_ctx *> v;
    return v;
}

void release_pooled_memory()
{
    LifetimeLock lock(lifetime_mutex());
    for (hmg_ctx *c : live_contexts()) vec_pool_trim(c);
}

// zero-filled device memory for one level vector (stream-ordered: kernels of the previous owner were enqueued on the
// same stream, or joined to it by events, before the block came back)
double *vec_alloc(hmg_ctx *c, size_t bytes)
{
    LifetimeLock lock(lifetime_mutex());
    void *p = nullptr;
    for (size_t i = 0; i < c->vec_pool.size(); ++i)
        if (c->vec_pool[i].first == bytes) {
            p = c->vec_pool[i].second;
            c->vec_pool[i] = c->vec_pool.back();
            c->vec_pool.pop_back();
            break;
        }
    const bool pooled = p != nullptr;
    if (!p && hipMalloc(&p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        release_pooled_memory();                 // pooled blocks of other sizes (any context's) may be what is in the way
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess)
            throw std::runtime_error(std::string("hipMalloc of a level vector (") + std::to_string(bytes >> 20) +
                                     " MiB) failed: " + hipGetErrorString(e));
    }
    if (!pooled) device_allocs() += 1;
    hipError_t e = hipMemsetAsync(p, 0, bytes, c->stream);
    if (e != hipSuccess) {
        (void)hipFree(p);
        throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e));
    }
    return (double *)p;
}

void vec_release(hmg_ctx *c, void *p, size_t bytes)
{
    LifetimeLock lock(lifetime_mutex());
    if (c->vec_pool_on && bytes > 0) {
        c->vec_pool.emplace_back(bytes, p);
        return;
    }
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(p);
}

}  // namespace

namespace {
// state of the last coarse solve, copied to pinned host memory behind the solve and read when somebody asks
struct CoarseProbe {
    double *h = nullptr;            // pinned: S_DONE, S_ITER, S_CRR, b.b
    hipEvent_t ev = nullptr;
    bool pending = false;
    int budget = 0;                 // iterations launched by the solve the probe belongs to
    int generation = 0;             // the level-1 matrix (hmg_grid::coarse_generation) that solve used
};

}  // namespace

struct hmg_grid {
    int refs = 1;
    hmg_ctx *ctx = nullptr;
    int dim = 0, nlevels = 0;
    std::vector<LevelTables> lt;
    std::vector<std::unique_ptr<LevelBufs>> lb;
    std::vector<LevelDev> ld;
    MeshTables mesh_full, mesh;
    bool shrunk = false;
    MeshDev md{};
    DevBuf<int32_t> d_cells, d_face_pairs, d_face_partner, d_edge_ptr, d_edge_ent, d_node_ptr, d_node_ent, d_node_first;
    DevBuf<uint16_t> d_dmask, d_dupmask;
    DevBuf<uint8_t> d_mult;
    DevBuf<double> d_blockpart;
    bool fuse_cg = true;
    DevBuf<double> d_coef;
    std::vector<double> sigma, coef;
    // class-weight cache (k_apply_wave): cells with bitwise equal coefficient rows share a class
    DevBuf<int32_t> d_cell_class;
    DevBuf<double> d_coef_rep;
    int nclasses = 0;
    double wc_lambda = 0.0;
    bool wc_ready = false;
    double lambda = 0.0;
    bool has_op = false;
    // coarse system
    CoarseMatrix cm;
    CoarseDev cd{};
    bool coarse_ready = false;
    DevBuf<int32_t> c_rowptr, c_colidx, c_interior;
    DevBuf<double> c_val, c_diag, c_b, c_x, c_r, c_z, c_p, c_q, c_u, c_z2, c_d;   // (c_z2, c_d: Chebyshev preconditioner)
    DevBuf<double> top_spare;           // second direction vector of the finest level's post-smoother (smooth(), lazy_top = 2)
    bool top_spare_refused = false;
    double c_lmax = 2.0;                         // Gershgorin bound of D^-1 A of the level-1 matrix
    int coarse_last_it = 0;
    int coarse_budget = 0;                       // iterations a solve enqueues blindly (0: not known yet)
    int coarse_generation = 0;                   // counts the level-1 matrices assembled for this grid
    int64_t coarse_misses = 0;                   // budgeted solves that ran out of iterations (each one was reported or, with a
                                                 // new matrix in between, only counted)
    std::unique_ptr<CoarseProbe> probe{new CoarseProbe};
    // multi-GPU
    std::unique_ptr<Partition> part;
    std::vector<double> sigma_global;
    // inputs of the partition analysis, kept for a domain shrink (re-analysis of the prefix mesh)
    std::vector<double> part_coords;
    std::vector<int64_t> part_cells;
    std::vector<int32_t> part_owner, part_cut_owner;     // (part_cut_owner: rehearsal partitions only, else empty)
    bool part_halo = true;                               // partition analysis on this rank's cells + one-cell halo (see create_partition)
    int64_t part_nnodes = 0, part_ncells = 0;
    DevBuf<int32_t> d_nodes_g, d_owned, d_cells_gnode;
    CutKind cut[3];   // faces, edges, nodes
    // Number of cut entities per kind OVER ALL RANKS, agreed once per partition analysis (agree_on_cut): what the overlap
    // decision of apply_then_sum looks at.  (CutKind::nglobal is rank-local after a halo-only analysis.)  -1: no agreement
    // possible (no scalar_sum callback) -- the plain form everywhere.
    int64_t cut_agreed[3] = {0, 0, 0};
    bool cut_agreed_ready = false;
    std::vector<std::unique_ptr<CutLevel>> cutlv;   // [nlevels]
    bool sharers = false;                        // exchange among the sharers of each cut entity (segments) instead of one
                                                 // all-reduce over the global cut buffer; needs a p2p transport (below)
    hmg_exchange_fn exchange = nullptr, scalar_sum = nullptr;
    hmg_exchange_fn ex_begin = nullptr;          // asynchronous form: begin issues the sum, end waits for it
    int (*ex_end)(void *) = nullptr;
    hmg_p2p_fn p2p = nullptr, p2p_begin = nullptr;   // segment layout: the messages of one exchange (sync / begin; ex_end ends it)
    double *stage = nullptr;
    int64_t stage_cap = 0;
    DevBuf<double> own_stage;
    bool overlap = true;
    DevBuf<int32_t> d_cells_cut, d_cells_inner, d_cell_perm;
    void *ex_user = nullptr;
    double *ex_buf = nullptr;
    int64_t ex_cap = 0;
    DevBuf<double> own_exbuf;                    // hmg_grid_use_comm: library-owned exchange buffer

    uint64_t upload_hash = 1469598103934665603ull;   // host-only grids: checksum of every table a device grid would upload (DryUploads)

    const MeshTables &cur() const { return shrunk ? mesh : mesh_full; }
};

struct hmg_vec {
    hmg_grid *g = nullptr;
    int level = 0;
    double *d = nullptr;
    bool own = false;
    int64_t alloc_cells = 0;
    size_t bytes = 0;        // own: size of the allocation behind d
};

namespace {

int fail(const std::exception &e)
{
    last_error() = e.what();
    return 1;
}

#define HMG_TRY try {
#define HMG_END                      \
    }                                \
    catch (const std::exception &e)  \
    {                                \
        return fail(e);              \
    }                                \
    catch (...)                      \
    {                                \
        last_error() = "unknown error"; \
        return 1;                    \
    }                                \
    return 0;

void need(bool c, const char *msg)
{
    if (!c) throw std::runtime_error(msg);
}

const LevelDev &lev(const hmg_grid *g, int level)
{
    need(g != nullptr, "null grid");
    need(g->ctx != nullptr, "this grid was created without a device context (host tables only): no compute path exists on the CPU");
    need(level >= 1 && level <= g->nlevels, "level out of range");
    return g->ld[level - 1];
}

void check_vec(const hmg_grid *g, int level, const hmg_vec *v, const char *name)
{
    if (!v) throw std::runtime_error(std::string("null vector: ") + name);
    if (v->g != g) throw std::runtime_error(std::string("vector belongs to another grid: ") + name);
    if (v->level != level) throw std::runtime_error(std::string("vector has the wrong level: ") + name);
    if (v->alloc_cells < g->md.ncells) throw std::runtime_error(std::string("vector too small: ") + name);
}

int64_t vec_len(const hmg_vec *v) { return (int64_t)v->g->ld[v->level - 1].ld * v->g->md.ncells; }

// scratch of the streaming reductions (one partial per 256-thread block = per 512 entries, see hmg_kernels.hip)
void ensure_reduce_scratch(hmg_ctx *c, int64_t nentries)
{
    const int64_t need_blocks = nentries / 512 + 2;
    if (need_blocks <= 2048 || need_blocks <= c->L.rpart_cap) return;
    HIPCHK(hipStreamSynchronize(c->stream));
    c->rpart.alloc((size_t)need_blocks);
    c->L.rpart = c->rpart.p;
    c->L.rpart_cap = need_blocks;
}

void upload_mesh(hmg_grid *g)
{
    const MeshTables &M = g->cur();
    MeshDev &d0 = g->md;
    d0.dim = M.dim;
    d0.ncells = M.ncells;
    d0.nnodes = M.nnodes;
    DryUploads dry_scope(!g->ctx, &g->upload_hash);   // host-only grid: the tables are built and checksummed, nothing is uploaded
    hipStream_t s = g->ctx ? g->ctx->stream : nullptr;
    g->d_cells.upload(M.cells, s);
    g->d_face_pairs.upload(M.face_pairs, s);
    {
        std::vector<int32_t> fp((size_t)M.ncells * 4, -1);
        for (size_t q = 3 * (size_t)M.ncut_face_pairs; q + 2 < M.face_pairs.size(); q += 3) {   // (cut pairs never ride in the r-update)
            const int32_t ca = M.face_pairs[q], cb = M.face_pairs[q + 1], la = M.face_pairs[q + 2] & 15, lb = M.face_pairs[q + 2] >> 4;
            fp[(size_t)ca * 4 + la] = (cb << 2) | lb;
            fp[(size_t)cb * 4 + lb] = (ca << 2) | la;
        }
        g->d_face_partner.upload(fp, s);
    }
    g->d_edge_ptr.upload(M.edge_ptr, s);
    g->d_edge_ent.upload(M.edge_ent, s);
    g->d_node_ptr.upload(M.node_ptr, s);
    g->d_node_ent.upload(M.node_ent, s);
    g->d_node_first.upload(M.node_first, s);
    {   // the class-weight-cache kernels read the masks of two neighbouring cells as ONE 32-bit word (HMG_KP(uint32_t, dmask)[cell >> 1]
        // in hmg_kernels.hip / hmg_apply_wave.hip / hmg_apply_small.hip): an even number of entries, whatever the cell count
        std::vector<uint16_t> dm(M.dmask);
        if (dm.size() & 1) dm.push_back((uint16_t)0);
        g->d_dmask.upload(dm, s);
    }
    g->d_dupmask.upload(M.dupmask, s);
    g->d_mult.upload(M.mult, s);
#ifdef HMG_PHASE_TIMING
    constexpr size_t BP = 10;   // 2 reduction partials + 8 time stamps per workgroup
#else
    constexpr size_t BP = 2;
#endif
    if (g->d_blockpart.n < (size_t)M.ncells * BP) g->d_blockpart.alloc((size_t)M.ncells * BP);
    MeshDev &d = g->md;
    d.dim = M.dim;
    d.ncells = M.ncells;
    d.nnodes = M.nnodes;
    d.cells = g->d_cells.p;
    d.face_pairs = g->d_face_pairs.p;
    d.face_partner = g->d_face_partner.p;
    d.nfacepairs = (int64_t)M.face_pairs.size() / 3;
    d.edge_ptr = g->d_edge_ptr.p;
    d.edge_ent = g->d_edge_ent.p;
    d.nsharededges = (int64_t)M.edge_ptr.size() - 1;
    d.node_ptr = g->d_node_ptr.p;
    d.node_ent = g->d_node_ent.p;
    d.nsharednodes = (int64_t)M.node_ptr.size() - 1;
    d.node_first = g->d_node_first.p;
    d.dmask = g->d_dmask.p;
    d.dupmask = g->d_dupmask.p;
    d.mult = g->d_mult.p;
    d.blockpart = g->d_blockpart.p;
    {   // (the cell lists of the overlapped exchange in the XCD-aware order too: workgroup b -> XCD b % 8 walks the
        //  (b % 8)-th contiguous eighth of the list, see cell_perm below)
        const char *env = std::getenv("HMG_XCD_LISTS");          // (dev knob: 0 = the lists as the partition analysis made them)
        const bool on = !(env && env[0] == '0');
        auto xcd_order = [on](const std::vector<int32_t> &v) {
            if (!on) return v;
            const int64_t n = (int64_t)v.size(), len = (n + 7) / 8;
            std::vector<int32_t> o((size_t)n);
            int64_t k = 0;
            for (int64_t b = 0; k < n; ++b) {
                const int64_t pos = b / 8, c = (b % 8) * len + pos;
                if (pos < len && c < n) o[(size_t)k++] = v[(size_t)c];
            }
            return o;
        };
        g->d_cells_cut.upload(xcd_order(M.cells_cut), s);
        g->d_cells_inner.upload(xcd_order(M.cells_inner), s);
    }
    d.cells_cut = g->d_cells_cut.p;
    d.cells_inner = g->d_cells_inner.p;
    d.ncells_cut = (int64_t)M.cells_cut.size();
    d.ncells_inner = (int64_t)M.cells_inner.size();
    {   // XCD-aware cell order of the full-grid apply launches (option cell_order): workgroups are dispatched round-robin over
        // the 8 XCDs (workgroup b -> XCD b % 8), so with cell = b every XCD's L2 sees every eighth column of every vector.  Here
        // XCD x walks the x-th contiguous eighth of the cells instead: -0.7 ... -1.3 ms per V-cycle (2, 4, 16 regions: -0.1 ... -0.4;
        // 64 regions or runs of 8 cells per XCD: slower; profiles/r03_experiments.txt).  A performance hint only: any mapping is correct.
        const int64_t n = M.ncells, len = (n + 7) / 8;
        std::vector<int32_t> perm((size_t)n);
        int64_t k = 0;
        for (int64_t b = 0; k < n; ++b) {
            const int64_t pos = b / 8, c = (b % 8) * len + pos;
            if (pos < len && c < n) perm[(size_t)k++] = (int32_t)c;
        }
        g->d_cell_perm.upload(perm, s);
        d.cell_perm = g->d_cell_perm.p;
    }
    d.ncut_edge_groups = M.ncut_edge_groups;
    d.ncut_node_groups = M.ncut_node_groups;
    d.ncut_face_pairs = M.ncut_face_pairs;
    d.coef = g->d_coef.p;
}

}  // namespace
static void upload_levels(hmg_grid *g)
{
    DryUploads dry_scope(!g->ctx, &g->upload_hash);
    hipStream_t s = g->ctx ? g->ctx->stream : nullptr;
    g->ld.resize(g->nlevels);
    for (int l = 0; l < g->nlevels; ++l) {
        const LevelTables &T = g->lt[l];
        g->lb.emplace_back(new LevelBufs);
        LevelBufs &B = *g->lb.back();
        std::vector<uint32_t> B_blk_word_host;     // host copies for the tables of k_apply_wave (below)
        std::vector<uint16_t> B_blk_slot_host, wave_cl_host;
        {
            B.meta.upload(T.meta, s);
            {
                std::vector<uint16_t> lp(T.meta.size());
                for (size_t q = 0; q < lp.size(); ++q) lp[q] = (uint16_t)(T.meta[q] & 0xffffu);
                B.lpos.upload(lp, s);
            }
            {
                // compact / wide addressing words (decode32 / decode32w in hmg_kernels.hip)
                std::vector<int32_t> slot_of_L(T.nf, -1);
                for (int q = 0; q < T.nf; ++q) slot_of_L[(size_t)(T.meta[q] & 0xffffu)] = q;
                const bool compact_ok = T.dim == 3 ? T.m <= 63 : T.m <= 255;
                auto pack32 = [&](uint64_t mt, int cls) -> uint32_t {
                    if (!compact_ok) return 0u;
                    const int sl = slot_of_L[(size_t)(mt & 0xffffu)];
                    const uint32_t j = T.slot_ijk[3 * sl + 1], k = T.slot_ijk[3 * sl + 2];
                    return (uint32_t)(mt & 0xffffu) | (j << 16) | (T.dim == 3 ? (k << 22) : 0u) | ((uint32_t)cls << 28);
                };
                std::vector<uint32_t> p32(T.meta.size()), s32(T.sweep_meta.size()), p32w(T.meta.size());
                for (size_t q = 0; q < p32.size(); ++q) {
                    p32[q] = pack32(T.meta[q], T.slot_cls[q]);
                    const uint32_t i = T.slot_ijk[3 * q], j = T.slot_ijk[3 * q + 1], k = T.slot_ijk[3 * q + 2];
                    if (T.dim == 3 && (i > 127 || j > 127 || k > 127))
                        throw std::runtime_error("lattice coordinate exceeds 127");
                    p32w[q] = (i & 127u) | ((j & 127u) << 7) | ((k & 127u) << 14) | ((uint32_t)T.slot_cls[q] << 21);
                }
                for (size_t q = 0; q < s32.size(); ++q) s32[q] = pack32(T.sweep_meta[q], 0);
                // padding read (never used) by k_apply's two-ahead table prefetch: see TABLE_PAD there
                p32.resize(p32.size() + TABLE_PAD, 0u);
                s32.resize(s32.size() + TABLE_PAD, 0u);
                B.pos32.upload(p32, s);
                B.pos32w.upload(p32w, s);
                B.sweep32.upload(s32, s);
            }
            // register-blocked interior of k_apply (interior_block in hmg_kernels.hip): every R-th interior k-plane,
            // all its interior (i,j) in lattice order; R = 6 makes the 4495 interior nodes of level 6 945 entries,
            // one pass of a 1024-thread workgroup
            // (level 5: R = 4, 152 entries for its 256-thread workgroup)
            if (T.dim == 3 && T.nint > 0 && T.m <= 63 && (T.nf > 2048 || (T.nf > 256 && T.nf <= 1024)) &&
                sizeof(double) * (size_t)(T.nf + 512) <= 160 * 1024) {
                const int R = T.nf > 2048 ? 6 : 4, m = T.m;
                std::vector<int32_t> slot_of_L(T.nf, -1);
                for (int q = 0; q < T.nf; ++q) slot_of_L[(size_t)(T.meta[q] & 0xffffu)] = q;
                auto tri = [](int n) { return (n + 1) * (n + 2) / 2; };
                std::vector<int> PO(m + 2, 0);
                for (int k = 0; k <= m; ++k) PO[k + 1] = PO[k] + tri(m - k);
                auto lin = [&](int i, int j, int k) { return PO[k] + j * (m - k + 1) - j * (j - 1) / 2 + i; };
                std::vector<uint32_t> bw;
                std::vector<uint16_t> bs;
                size_t covered = 0;
                for (int k0 = 1; k0 <= m - 3; k0 += R)
                    for (int j = 1; j + k0 <= m - 2; ++j)
                        for (int i = 1; i + j + k0 <= m - 1; ++i) {
                            const int nv = std::min(R, m - i - j - k0);
                            const int L = lin(i, j, k0), sl = slot_of_L[L];
                            if (sl < T.off_int || T.slot_cls[sl] != 0) throw std::runtime_error("blocked interior: not an interior node");
                            // the kernel derives the slots of the R nodes from the first: check that rule here
                            int ds = tri(m - k0 - 3) - (j - 1), cur = sl;
                            for (int r = 1; r < nv; ++r) {
                                cur += ds;
                                ds -= (m - k0) - 1 - r;
                                if (cur != slot_of_L[lin(i, j, k0 + r)]) throw std::runtime_error("blocked interior: slot rule broken");
                            }
                            bw.push_back((uint32_t)L | ((uint32_t)j << 16) | ((uint32_t)k0 << 22) | ((uint32_t)nv << 28));
                            bs.push_back((uint16_t)sl);
                            covered += nv;
                        }
                if ((int)covered != T.nint) throw std::runtime_error("blocked interior: tables do not cover the interior");
                // the same instantiation evaluates the faces one class per wave and skips the taps that leave the cell
                // (face_tap_mask in hmg_kernels.hip: f0 k = 0, f1 j = 0, f2 i = 0, f3 i+j+k = m): check them against
                // the class table, and the run counts the kernel is compiled for (4 waves x 2 runs of 64 per face,
                // 3 runs of 64 for corners + edges)
                static const uint32_t absent[4] = {1u << 8 | 1u << 10 | 1u << 12 | 1u << 14, 1u << 4 | 1u << 6 | 1u << 7 | 1u << 13,
                                                   1u << 2 | 1u << 3 | 1u << 9 | 1u << 14, 1u << 1 | 1u << 5 | 1u << 11 | 1u << 13};
                const int nw = T.nf > 2048 ? 16 : 4;     // waves of the workgroup that runs this level
                bool ok = T.nface == 4 && T.nfi <= 128 * std::max(nw / 4, 1) && T.nei <= 64 && (int)bw.size() <= (nw - 1) * 64;
                for (int f = 0; ok && f < 4; ++f)
                    for (int d = 0; d < T.ndir; ++d) {
                        bool zero = true;
                        for (int t = 0; t < T.nterm; ++t) zero = zero && T.ctab[((size_t)(1 + f) * T.ndir + d) * T.nterm + t] == 0.0;
                        if (((absent[f] >> d) & 1u) && !zero) ok = false;   // a tap the kernel skips carries weight
                    }
                // edges: an edge node keeps the taps both of its faces keep (edge_tap_mask)
                static const int ef[6][2] = {{0, 1}, {0, 2}, {1, 2}, {0, 3}, {1, 3}, {2, 3}};
                for (int e = 0; ok && e < T.nedge && T.nedge == 6; ++e)
                    for (int d = 0; d < T.ndir; ++d) {
                        bool zero = true;
                        for (int t = 0; t < T.nterm; ++t) zero = zero && T.ctab[((size_t)(1 + T.nface + e) * T.ndir + d) * T.nterm + t] == 0.0;
                        if ((((absent[ef[e][0]] | absent[ef[e][1]]) >> d) & 1u) && !zero) ok = false;
                    }
                if (!ok) throw std::runtime_error("blocked apply: the class table does not match the kernel's face / edge tap masks");
                B.nblk = (int)bw.size();
                B.blk_R = R;
                B_blk_word_host = bw;
                B_blk_slot_host = bs;
                bw.resize(bw.size() + TABLE_PAD, 0u);
                bs.resize(bs.size() + TABLE_PAD, (uint16_t)0);
                B.blk_word.upload(bw, s);
                B.blk_slot.upload(bs, s);
            }
            // slab tables: needed by the apply of levels whose cell exceeds the LDS (level 7), and used by the
            // restriction of every large 3D level (level 6: the whole cell is one slab)
            if (T.dim == 3 && T.nf > 2048) {
                // greedy slabs of k-planes: the rolling window [k0-1, k1] (+ zero guard) of k_apply_slab must fit
                // half of the CU's LDS, so that two workgroups are resident (HMG_SLAB_LDS_KB overrides, dev knob)
                int kb = 70;
                if (const char *e = std::getenv("HMG_SLAB_LDS_KB")) kb = std::max(16, std::min(158, std::atoi(e)));
                const int cap = (kb * 1024) / 8 - 232;
                const int slab_g0 = 0;
                auto po = [&](int k) {
                    long long n1 = T.m + 1, n2 = T.m + 1 - std::min(std::max(k, 0), T.m + 1);
                    return (int)((n1 * (n1 + 1) * (n1 + 2) - n2 * (n2 + 1) * (n2 + 2)) / 6);
                };
                std::vector<int> sk{0};
                int maxn = 0;
                while (sk.back() <= T.m) {
                    int k0 = sk.back(), k1 = k0 + 1;
                    while (k1 <= T.m && slab_g0 + T.lds_g1 + po(k1 + 2) - po(k0 - 1) <= cap) ++k1;
                    int nn = slab_g0 + T.lds_g1 + po(k1 + 1) - po(k0 - 1);
                    if (nn > cap) throw std::runtime_error("apply slabs: a single plane does not fit the LDS");
                    maxn = std::max(maxn, nn);
                    sk.push_back(k1);
                }
                B.nslab = (int)sk.size() - 1;
                B.slab_lds_nodes = maxn;
                // Inside every entity segment the slots are ordered by plane k, so the slots of planes [ka, kb) are
                // one contiguous run.  15 segments: 4 corners, 6 edges, 4 faces, interior.
                std::vector<std::pair<int, int>> seg;
                for (int c = 0; c < T.ncorner; ++c) seg.push_back({c, c + 1});
                for (int e = 0; e < T.nedge; ++e) seg.push_back({T.off_edge + e * T.nei, T.off_edge + (e + 1) * T.nei});
                for (int f = 0; f < T.nface; ++f) seg.push_back({T.off_face + f * T.nfi, T.off_face + (f + 1) * T.nfi});
                seg.push_back({T.off_int, T.nf});                    // interior last: the kernel's fast path
                auto run = [&](std::pair<int, int> sg, int ka, int kb) {   // slots of the segment with ka <= k < kb
                    int b = sg.second, e = sg.first;
                    for (int t = sg.first; t < sg.second; ++t) {
                        int k = T.slot_ijk[3 * t + 2];
                        if (k >= ka && k < kb) {
                            b = std::min(b, t);
                            e = std::max(e, t + 1);
                        }
                    }
                    if (b >= e) return std::pair<int, int>{0, 0};
                    for (int t = b; t < e; ++t) {
                        int k = T.slot_ijk[3 * t + 2];
                        if (k < ka || k >= kb) throw std::runtime_error("apply slabs: plane range is not contiguous");
                    }
                    return std::pair<int, int>{b, e};
                };
                // flat lists per slab (k_apply_slab): slots new in the rolling window (planes k0-1 and k0 come
                // from the previous slab's LDS image) and slots evaluated (surface entities first)
                std::vector<int> head((size_t)B.nslab * 8, 0);
                std::vector<uint32_t> ldw, cpw;
                std::vector<uint16_t> cps;
                for (int sl = 0; sl < B.nslab; ++sl) {
                    head[sl * 8 + 0] = sk[sl];
                    head[sl * 8 + 1] = (int)ldw.size();
                    head[sl * 8 + 3] = (int)cpw.size();
                    for (size_t si = 0; si < seg.size(); ++si) {
                        const auto &sg = seg[si];
                        if (si + 1 == seg.size()) head[sl * 8 + 5] = (int)cpw.size() - head[sl * 8 + 3];   // surface entries
                        auto ld = run(sg, sl == 0 ? 0 : sk[sl] + 1, sk[sl + 1] + 1);
                        for (int t = ld.first; t < ld.second; ++t)
                            ldw.push_back((uint32_t)(T.meta[t] & 0xffffu) | ((uint32_t)t << 16));
                        auto cp = run(sg, sk[sl], sk[sl + 1]);
                        const bool interior = si + 1 == seg.size();
                        for (int t = cp.first; t < cp.second; ++t) {
                            // surface entries: i | j << 7 | k << 14 | cls << 21 (decode32w); cell interior (round 4): the lattice
                            // position itself, L | j << 16 | k << 23 (decode_lattice: no tetrahedral-number arithmetic per node)
                            if (interior)
                                cpw.push_back((uint32_t)(T.meta[t] & 0xffffu) | (((uint32_t)T.slot_ijk[3 * t + 1] & 127u) << 16) |
                                              (((uint32_t)T.slot_ijk[3 * t + 2] & 127u) << 23));
                            else
                            cpw.push_back(((uint32_t)T.slot_ijk[3 * t] & 127u) | (((uint32_t)T.slot_ijk[3 * t + 1] & 127u) << 7) |
                                          (((uint32_t)T.slot_ijk[3 * t + 2] & 127u) << 14) | ((uint32_t)T.slot_cls[t] << 21));
                            cps.push_back((uint16_t)t);
                        }
                    }
                    head[sl * 8 + 2] = (int)ldw.size() - head[sl * 8 + 1];
                    head[sl * 8 + 4] = (int)cpw.size() - head[sl * 8 + 3];
                }
                if ((int)ldw.size() != T.nf || (int)cpw.size() != T.nf)
                    throw std::runtime_error("apply slabs: the slab lists do not cover the cell exactly once");
                // k_apply_slab2 (hmg_apply_slab.hip) takes the interior slots of a slab as one run of consecutive slots
                for (int sl = 0; sl < B.nslab; ++sl) {
                    const int b0 = head[sl * 8 + 3] + head[sl * 8 + 5], e0 = head[sl * 8 + 3] + head[sl * 8 + 4];
                    for (int q = b0 + 1; q < e0; ++q)
                        if (cps[(size_t)q] != cps[(size_t)q - 1] + 1) throw std::runtime_error("apply slabs: interior slots of a slab are not consecutive");
                    B.slab_max_surf = std::max(B.slab_max_surf, head[sl * 8 + 5]);
                    B.slab_max_int = std::max(B.slab_max_int, head[sl * 8 + 4] - head[sl * 8 + 5]);
                }
                ldw.resize(ldw.size() + TABLE_PAD, 0u);
                cpw.resize(cpw.size() + TABLE_PAD, 0u);
                cps.resize(cps.size() + TABLE_PAD, (uint16_t)0);
                B.slab_head.upload(head, s);
                B.slab_ld_word.upload(ldw, s);
                B.slab_cp_word.upload(cpw, s);
                B.slab_cp_slot.upload(cps, s);
                if (l > 0) {
                    // restriction through the same window (launch_restrict_slab): evaluated nodes = the even
                    // lattice nodes (= nodes of the coarser level), output slot = their COARSE storage slot;
                    // weights 1 / 0.5 on the taps that exist (nonzero mass entry of the class table)
                    const LevelTables &C = g->lt[l - 1];
                    const int mc = C.m;
                    std::vector<int32_t> cslot((size_t)(mc + 1) * (mc + 1) * (mc + 1), -1);
                    auto cidx = [&](int i, int j, int k) { return ((size_t)k * (mc + 1) + j) * (mc + 1) + i; };
                    for (int q = 0; q < C.nf; ++q) cslot[cidx(C.slot_ijk[3 * q], C.slot_ijk[3 * q + 1], C.slot_ijk[3 * q + 2])] = q;
                    std::vector<int> rhead(head);
                    std::vector<uint32_t> rsw;
                    std::vector<uint16_t> rss;
                    for (int sl = 0; sl < B.nslab; ++sl) {
                        rhead[sl * 8 + 3] = (int)rsw.size();
                        for (size_t si = 0; si < seg.size(); ++si) {
                            if (si + 1 == seg.size()) rhead[sl * 8 + 5] = (int)rsw.size() - rhead[sl * 8 + 3];
                            auto cp = run(seg[si], sk[sl], sk[sl + 1]);
                            for (int t = cp.first; t < cp.second; ++t) {
                                const int i = T.slot_ijk[3 * t], j = T.slot_ijk[3 * t + 1], k = T.slot_ijk[3 * t + 2];
                                if ((i | j | k) & 1) continue;
                                const int cs = cslot[cidx(i / 2, j / 2, k / 2)];
                                if (cs < 0) throw std::runtime_error("slab restriction: even node without a coarse slot");
                                if (si + 1 == seg.size())     // (cell interior: lattice-position form, as above)
                                    rsw.push_back((uint32_t)(T.meta[t] & 0xffffu) | (((uint32_t)j & 127u) << 16) | (((uint32_t)k & 127u) << 23));
                                else
                                rsw.push_back(((uint32_t)i & 127u) | (((uint32_t)j & 127u) << 7) | (((uint32_t)k & 127u) << 14) |
                                              ((uint32_t)T.slot_cls[t] << 21));
                                rss.push_back((uint16_t)cs);
                            }
                        }
                        rhead[sl * 8 + 4] = (int)rsw.size() - rhead[sl * 8 + 3];
                        B.slab_rs_max_surf = std::max(B.slab_rs_max_surf, rhead[sl * 8 + 5]);
                        B.slab_rs_max_int = std::max(B.slab_rs_max_int, rhead[sl * 8 + 4] - rhead[sl * 8 + 5]);
                    }
                    if ((int)rsw.size() != C.nf) throw std::runtime_error("slab restriction: lists do not cover the coarse cell");
                    rsw.resize(rsw.size() + TABLE_PAD, 0u);
                    rss.resize(rss.size() + TABLE_PAD, (uint16_t)0);
                    std::vector<double> rt(T.ctab.size(), 0.0);
                    for (int c = 0; c < T.ncls; ++c)
                        for (int d = 0; d < T.ndir; ++d) {
                            const size_t e = ((size_t)c * T.ndir + d) * T.nterm + T.nterm - 1;
                            rt[e] = T.ctab[e] != 0.0 ? (d == 0 ? 1.0 : 0.5) : 0.0;
                        }
                    B.slab_rs_head.upload(rhead, s);
                    B.slab_rs_word.upload(rsw, s);
                    B.slab_rs_slot.upload(rss, s);
                    B.rtab.upload(rt, s);
                }
            }
            {
                std::vector<uint16_t> ss(T.sweep_slot);
                ss.resize(ss.size() + TABLE_PAD, (uint16_t)0xffff);
                B.sweep_slot.upload(ss, s);
            }
            B.ctab.upload(T.ctab, s);
            B.hier2slot.upload(T.hier2slot, s);
            B.par_a.upload(T.par_a, s);
            {
                std::vector<uint32_t> pp(T.par_a.size());
                for (size_t q = 0; q < pp.size(); ++q) {
                    if ((uint32_t)T.par_a[q] > 0xffffu || (uint32_t)T.par_b[q] > 0xffffu)
                        throw std::runtime_error("coarse slot exceeds 16 bits");
                    pp[q] = (uint32_t)T.par_a[q] | ((uint32_t)T.par_b[q] << 16);
                }
                B.par32.upload(pp, s);
            }
            B.par_b.upload(T.par_b, s);
            if (l > 0 && T.dim == 3 && B.nblk > 0 && T.nf <= 0xffff) {
                // folded prolongation, coarse column staged in the image itself (k_apply<.., CG>): coarse slot c = lattice
                // node (ci,cj,ck) of the coarser level sits at the fine lattice node (2ci,2cj,2ck)
                const LevelTables &C = g->lt[l - 1];
                const int m = T.m;
                std::vector<int32_t> lat((size_t)(m + 1) * (m + 1) * (m + 1), -1);
                auto at = [&](int i, int j, int k) -> int32_t & { return lat[((size_t)k * (m + 1) + j) * (m + 1) + i]; };
                for (int q = 0; q < T.nf; ++q)
                    at(T.slot_ijk[3 * q], T.slot_ijk[3 * q + 1], T.slot_ijk[3 * q + 2]) = (int32_t)(T.meta[q] & 0xffffu);
                std::vector<uint16_t> cl((size_t)C.nf);
                for (int c = 0; c < C.nf; ++c) {
                    const int32_t Lp = at(2 * C.slot_ijk[3 * c], 2 * C.slot_ijk[3 * c + 1], 2 * C.slot_ijk[3 * c + 2]);
                    if (Lp < 0) throw std::runtime_error("prolongation tables: coarse node without a fine lattice node");
                    cl[c] = (uint16_t)Lp;
                }
                std::vector<uint64_t> p64((size_t)T.nf);
                for (int q = 0; q < T.nf; ++q) {
                    const uint64_t a = cl[(size_t)T.par_a[q]], b = cl[(size_t)T.par_b[q]], self = T.meta[q] & 0xffffu;
                    // (an identity row is its own parent: the coarse value sits where the slot's own value will go)
                    if (T.par_a[q] == T.par_b[q] && a != self) throw std::runtime_error("prolongation tables: identity row off its node");
                    p64[q] = a | (b << 16) | (self << 32);
                }
                B.clpos.upload(cl, s);
                B.par64.upload(p64, s);
                wave_cl_host = cl;
                // restriction in the epilogue of the local residual (k_apply<.., RS>): addressing word of every coarse slot's
                // fine node and the weights 1 / 0.5 on the taps that exist (nonzero mass entry of the class table) -- the
                // weights of the stand-alone restriction (rtab above / launch_restrict_slab)
                std::vector<int32_t> fslot((size_t)(m + 1) * (m + 1) * (m + 1), -1);
                for (int q = 0; q < T.nf; ++q)
                    fslot[((size_t)T.slot_ijk[3 * q + 2] * (m + 1) + T.slot_ijk[3 * q + 1]) * (m + 1) + T.slot_ijk[3 * q]] = q;
                std::vector<uint32_t> rw((size_t)C.nf);
                for (int c = 0; c < C.nf; ++c) {
                    const int i = 2 * C.slot_ijk[3 * c], j = 2 * C.slot_ijk[3 * c + 1], k = 2 * C.slot_ijk[3 * c + 2];
                    const int fs = fslot[((size_t)k * (m + 1) + j) * (m + 1) + i];
                    if (fs < 0 || i > 127 || j > 127 || k > 127) throw std::runtime_error("restriction tables: bad coarse node");
                    rw[c] = ((uint32_t)i & 127u) | (((uint32_t)j & 127u) << 7) | (((uint32_t)k & 127u) << 14) |
                            ((uint32_t)T.slot_cls[fs] << 21);
                }
                std::vector<double> wts((size_t)T.ncls * T.ndir, 0.0);
                for (int c = 0; c < T.ncls; ++c)
                    for (int d = 0; d < T.ndir; ++d)
                        wts[(size_t)c * T.ndir + d] =
                            T.ctab[((size_t)c * T.ndir + d) * T.nterm + T.nterm - 1] != 0.0 ? (d == 0 ? 1.0 : 0.5) : 0.0;
                B.rs_word.upload(rw, s);
                B.rs_w.upload(wts, s);
                if (T.nf <= 2048) {
                    // levels whose stand-alone restriction is k_restrict: the epilogue sums in ITS order (the reference's:
                    // identity row first, then the midpoints in ascending fine hierarchical id) -- same bits on both paths
                    std::vector<uint16_t> lp(T.ridx.size());
                    for (size_t e = 0; e < lp.size(); ++e) lp[e] = (uint16_t)(T.meta[(size_t)T.ridx[e]] & 0xffffu);
                    B.rs_lp.upload(lp, s);
                }
            }
            if (T.dim == 3 && T.m == 16 && T.nf == 969 && B.nblk == 152 && B.blk_R == 4 && T.lds_g0 == 0 && T.nfi == 105 && T.nei == 15 &&
                T.ncorner == 4 && T.nedge == 6 && T.nface == 4 && T.off_edge == 4 && T.off_face == 94) {
                // Tables of k_apply_wave (hmg_apply_wave.hip): what lane l of the one wave that owns a cell needs, row by row.
                //   wave_tab rows 0..7   faces: run r = 2 f + h covers nodes h * 64 + l of face f
                //            rows 8..9   corners and edges: slot r * 64 + l (< 94)
                //            rows 10..12 interior blocks u = r * 64 + l (< 152): the block word of the blocked tables
                //            row 13      storage slot of block l | of block 64 + l << 16
                //            row 14      storage slot of block 128 + l | class of slot l << 16 | class of slot 64 + l << 24
                //   surface words: L | len << 10 | A << 15 | B << 23 | valid << 31 (rows have len = m+1-j-k nodes, A / B = offsets
                //   to the same (i,j) in the plane above / below, as decode32 derives them)
                const int m = T.m, WVZ = 168, WDUMMY = 167;
                auto surf_word = [&](int t, bool valid) -> uint32_t {
                    const uint32_t L = (uint32_t)(T.meta[(size_t)t] & 0xffffu);
                    const int j = T.slot_ijk[3 * t + 1], k = T.slot_ijk[3 * t + 2];
                    const int len = m + 1 - j - k, n = m - k, Tk = (n + 1) * (n + 2) / 2;
                    const int A = Tk - j, Bo = Tk + n + 2 - j;
                    if (L > 1023u || len < 0 || len > 31 || A < 0 || A > 255 || Bo < 0 || Bo > 255)
                        throw std::runtime_error("wave tables: addressing word out of range");
                    return L | ((uint32_t)len << 10) | ((uint32_t)A << 15) | ((uint32_t)Bo << 23) | (valid ? 1u << 31 : 0u);
                };
                std::vector<uint32_t> wt((size_t)WAVE_TAB_ROWS * 64, 0u), wl(8 * 64, 0u);
                std::vector<uint32_t> bwv(B_blk_word_host.begin(), B_blk_word_host.begin() + B.nblk);
                for (int l = 0; l < 64; ++l) {
                    for (int r = 0; r < 8; ++r) {
                        const int f = r >> 1, ti = (r & 1) * 64 + l;
                        const bool valid = ti < T.nfi;
                        wt[(size_t)r * 64 + l] = surf_word(T.off_face + f * T.nfi + (valid ? ti : 0), valid);
                    }
                    uint32_t cls[2];
                    for (int r = 0; r < 2; ++r) {
                        const int t = r * 64 + l;
                        const bool valid = t < T.off_face;
                        wt[(size_t)(8 + r) * 64 + l] = surf_word(valid ? t : 0, valid);
                        cls[r] = T.slot_cls[(size_t)(valid ? t : T.off_edge)];
                        if (cls[r] < 5 || cls[r] > 14) throw std::runtime_error("wave tables: edge / corner class out of range");
                    }
                    uint32_t bsl[3];
                    for (int r = 0; r < 3; ++r) {
                        const int u = r * 64 + l;
                        const bool valid = u < B.nblk;
                        // (no block: block 0's addresses with no valid node -- nothing is stored)
                        wt[(size_t)(10 + r) * 64 + l] = valid ? bwv[(size_t)u] : (bwv[0] & 0x0fffffffu);
                        bsl[r] = valid ? B_blk_slot_host[(size_t)u] : B_blk_slot_host[0];
                    }
                    wt[(size_t)13 * 64 + l] = bsl[0] | (bsl[1] << 16);
                    wt[(size_t)14 * 64 + l] = bsl[2] | (cls[0] << 16) | (cls[1] << 24);
                    for (int i = 0; i < 8; ++i) {
                        uint32_t off[2];
                        for (int h = 0; h < 2; ++h) {
                            const int t = l + 64 * (2 * i + h);
                            off[h] = 8u * (uint32_t)(t < T.nf ? WVZ + (int)(T.meta[(size_t)t] & 0xffffu) : WDUMMY);
                        }
                        wl[(size_t)i * 64 + l] = off[0] | (off[1] << 16);
                    }
                }
                B.wave_tab.upload(wt, s);
                B.wave_lpos.upload(wl, s);
                if (l > 0 && !wave_cl_host.empty() && g->lt[l - 1].nf == 165) {
                    const LevelTables &C = g->lt[l - 1];
                    std::vector<uint32_t> wp(16 * 64, 0u), wc(3 * 64, 8u * (uint32_t)WDUMMY), wr((size_t)192 * 8, 0u);
                    for (int t = 0; t < T.nf; ++t)
                        wp[(size_t)(t / 64) * 64 + t % 64] = (uint32_t)wave_cl_host[(size_t)T.par_a[t]] | ((uint32_t)wave_cl_host[(size_t)T.par_b[t]] << 16);
                    for (int c = 0; c < C.nf; ++c) wc[(size_t)(c / 64) * 64 + c % 64] = 8u * (uint32_t)(WVZ + wave_cl_host[(size_t)c]);
                    for (int c = 0; c < C.nf; ++c) {
                        const int b = T.rptr[c], n = T.rptr[c + 1] - b;
                        if (n < 1 || n > 15) throw std::runtime_error("wave tables: restriction row longer than 15");
                        uint16_t e[16] = {0};
                        for (int q = 0; q < n; ++q) e[q] = (uint16_t)(T.meta[(size_t)T.ridx[b + q]] & 0xffffu);
                        e[15] = (uint16_t)n;
                        for (int q = 0; q < 8; ++q) wr[(size_t)c * 8 + q] = (uint32_t)e[2 * q] | ((uint32_t)e[2 * q + 1] << 16);
                    }
                    B.wave_par.upload(wp, s);
                    B.wave_cl.upload(wc, s);
                    B.wave_rs.upload(wr, s);
                }
            }
            B.rptr.upload(T.rptr, s);
            B.ridx.upload(T.ridx, s);
            B.dphi.upload(T.dphi, s);
        }
        LevelDev &D = g->ld[l];
        D.dim = T.dim;
        D.level = T.level;
        D.m = T.m;
        D.nf = T.nf;
        D.ld = T.ld;
        D.ncorner = T.ncorner;
        D.nedge = T.nedge;
        D.nface = T.nface;
        D.nei = T.nei;
        D.nfi = T.nfi;
        D.nint = T.nint;
        D.off_edge = T.off_edge;
        D.off_face = T.off_face;
        D.off_int = T.off_int;
        D.ncls = T.ncls;
        D.ndir = T.ndir;
        D.nterm = T.nterm;
        D.lds_g0 = T.lds_g0;
        D.lds_g1 = T.lds_g1;
        D.nf_coarse = l > 0 ? g->lt[l - 1].nf : 0;
        D.meta = B.meta.p;
        D.lpos = B.lpos.p;
        D.sweep_slot = B.sweep_slot.p;
        D.pos32 = B.pos32.p;
        D.pos32w = B.pos32w.p;
        D.sweep32 = B.sweep32.p;
        D.nsweep = (int)T.sweep_meta.size();
        D.blk_word = B.blk_word.p;
        D.blk_slot = B.blk_slot.p;
        D.nblk = B.nblk;
        D.blk_R = B.blk_R;
        D.ctab = B.ctab.p;
        D.hier2slot = B.hier2slot.p;
        D.par_a = B.par_a.p;
        D.par32 = B.par32.p;
        D.par_b = B.par_b.p;
        D.par64 = B.par64.p;
        D.clpos = B.clpos.p;
        D.rs_word = B.rs_word.p;
        D.rs_w = B.rs_w.p;
        D.rs_lp = B.rs_lp.p;
        D.rptr = B.rptr.p;
        D.ridx = B.ridx.p;
        D.dphi = B.dphi.p;
        D.wave_tab = B.wave_tab.p;
        D.wave_lpos = B.wave_lpos.p;
        D.wave_par = B.wave_par.p;
        D.wave_cl = B.wave_cl.p;
        D.wave_rs = B.wave_rs.p;
        D.wcache = nullptr;                      // (set with the operator: build_weight_cache)
    }
}
namespace {


// Class-weight cache of the one-wave apply (hmg_apply_wave.hip).  The per-cell weights of the lattice stencil are linear in
// the cell's coefficient row (|J| P_kl, |J|), and on the meshes this library is built for most rows repeat: a checkerboard
// has at most 8 sigma triples x 6 tetrahedron orientations = 48 distinct ones.  Cells are classed by the BITS of their row;
// per class, sign of alpha and level the 15 x 15 weights are formed once on the device (launch_weight_cache), by the same
// products in the same order as the kernels form them per cell.  More than WC_MAX_CLASSES distinct rows (perturbed or
// unstructured meshes): no cache, the level keeps the 256-thread kernel.
constexpr int WC_MAX_CLASSES = 1024;

void build_cell_classes(hmg_grid *g)
{
    g->nclasses = 0;
    g->wc_ready = false;
    g->md.cell_class = nullptr;
    g->md.nclasses = 0;
    for (auto &d : g->ld) d.wcache = nullptr;
    bool any = false;
    for (const auto &d : g->ld) any = any || d.level >= 2;
    if (!any || g->dim != 3) return;
    const int64_t n = g->cur().ncells;
    struct Key {
        uint64_t b[8];
        bool operator==(const Key &o) const { return std::memcmp(b, o.b, sizeof(b)) == 0; }
    };
    struct Hash {
        size_t operator()(const Key &k) const
        {
            uint64_t h = 1469598103934665603ull;
            for (int q = 0; q < 8; ++q) h = (h ^ k.b[q]) * 1099511628211ull;
            return (size_t)h;
        }
    };
    std::unordered_map<Key, int32_t, Hash> ids;
    std::vector<int32_t> cls((size_t)n);
    std::vector<double> rep;
    for (int64_t c = 0; c < n; ++c) {
        Key k;
        std::memcpy(k.b, g->coef.data() + (size_t)c * 8, sizeof(k.b));
        auto it = ids.find(k);
        if (it == ids.end()) {
            if ((int)ids.size() >= WC_MAX_CLASSES) return;           // too many distinct rows: no cache
            it = ids.emplace(k, (int32_t)ids.size()).first;
            rep.insert(rep.end(), g->coef.begin() + (size_t)c * 8, g->coef.begin() + (size_t)c * 8 + 8);
        }
        cls[(size_t)c] = it->second;
    }
    g->nclasses = (int)ids.size();
    hipStream_t s = g->ctx ? g->ctx->stream : nullptr;
    g->d_cell_class.upload(cls, s);
    g->d_coef_rep.upload(rep, s);
    g->md.cell_class = g->d_cell_class.p;
    g->md.nclasses = g->nclasses;
}

// (re)forms the cached weights when the operator or lambda has changed since they were formed; called in front of every
// apply (a host comparison when nothing has changed)
void ensure_weight_cache(hmg_grid *g)
{
    if (!g->md.cell_class || (g->wc_ready && g->wc_lambda == g->lambda)) return;
    for (int l = 0; l < g->nlevels; ++l) {
        LevelDev &D = g->ld[l];
        if (D.level < 2 || D.ncls != 15) continue;           // (every 3D level an operator is applied on: k_apply<.., WC>, k_apply_wave)
        LevelBufs &B = *g->lb[l];
        if (B.wcache.n != (size_t)g->nclasses * 2 * WAVE_WSTRIDE) {
            HIPCHK(hipStreamSynchronize(g->ctx->stream));   // (kernels that read the old cache)
            B.wcache.alloc((size_t)g->nclasses * 2 * WAVE_WSTRIDE);
        }
        launch_weight_cache(g->ctx->L, D, g->d_coef_rep.p, g->nclasses, g->lambda, B.wcache.p);
        D.wcache = B.wcache.p;
    }
    g->wc_lambda = g->lambda;
    g->md.wc_lambda = g->lambda;
    g->wc_ready = true;
}

void upload_operator(hmg_grid *g)
{
    const MeshTables &M = g->cur();
    if (g->part) {   // local sigma = rows of the global field
        const int dim = g->dim;
        g->sigma.resize((size_t)M.ncells * dim);
        for (int64_t q = 0; q < M.ncells; ++q)
            for (int a = 0; a < dim; ++a)
                g->sigma[(size_t)q * dim + a] = g->sigma_global[(size_t)g->part->cells_g[q] * dim + a];
    }
    build_cell_coefficients(M, g->sigma.data(), g->coef);
    g->coarse_ready = false;
    DryUploads dry_scope(!g->ctx, &g->upload_hash);
    g->d_coef.upload(g->coef, g->ctx ? g->ctx->stream : nullptr);
    g->md.coef = g->d_coef.p;
    build_cell_classes(g);
    if (g->ctx) ensure_weight_cache(g);          // formed here, not in front of the first apply: a V-cycle allocates nothing
}

void exchange_cut(hmg_grid *g, const LevelDev &lv, double *x);   // defined below
bool has_exchange(const hmg_grid *g) { return g->exchange || g->ex_begin || g->p2p || g->p2p_begin; }

void set_slab(hmg_grid *g, const LevelDev &lv)
{
    const LevelBufs &B = *g->lb[lv.level - 1];
    g->md.slab.head = B.slab_head.p;
    g->md.slab.ld_word = B.slab_ld_word.p;
    g->md.slab.cp_word = B.slab_cp_word.p;
    g->md.slab.cp_slot = B.slab_cp_slot.p;
    g->md.slab.nslab = B.nslab;
    g->md.slab.lds_nodes = B.slab_lds_nodes;
    g->md.slab.max_surf = B.slab_max_surf;
    g->md.slab.max_int = B.slab_max_int;
}

// every operator apply goes through here: optional HIP-event bracketing for bench.py's roofline
void apply(hmg_grid *g, const LevelDev &lv, double alpha, const double *x, const double *src, double *out, int mask)
{
    hmg_ctx *c = g->ctx;
    ApplyTimer &t = c->timer;
    const bool timed = t.on && lv.level >= t.min_level;
    if (timed) {
        if (t.used == t.pool.size()) {
            hipEvent_t a, b;
            HIPCHK(hipEventCreate(&a));
            HIPCHK(hipEventCreate(&b));
            t.pool.emplace_back(a, b);
        }
        HIPCHK(hipEventRecord(t.pool[t.used].first, c->stream));
    }
    set_slab(g, lv);
    ensure_weight_cache(g);
    launch_apply(c->L, lv, g->md, alpha, g->lambda, x, src, out, mask);
    if (timed) {
        HIPCHK(hipEventRecord(t.pool[t.used].second, c->stream));
        t.used += 1;
        t.launches += 1;
        const double by = 8.0 * (double)lv.nf * (double)g->md.ncells * (src ? 3.0 : 2.0);
        t.bytes += by;
        t.ev_level.push_back(lv.level);
        t.ev_bytes.push_back(by);
    }
}

void restrict_level(hmg_grid *g, int level_fine, const double *rf, double *bc)
{
    // ref: src/interpolation.jl:52-62
    const LevelDev &fine = lev(g, level_fine), &coarse = lev(g, level_fine - 1);
    const LevelBufs &B = *g->lb[level_fine - 1];
    if (B.slab_rs_head.p) {
        LevelDev fr = fine;
        fr.ctab = B.rtab.p;
        SlabTables st{};
        st.head = B.slab_rs_head.p;
        st.ld_word = B.slab_ld_word.p;
        st.cp_word = B.slab_rs_word.p;
        st.cp_slot = B.slab_rs_slot.p;
        st.nslab = B.nslab;
        st.lds_nodes = B.slab_lds_nodes;
        st.max_surf = B.slab_rs_max_surf;
        st.max_int = B.slab_rs_max_int;
        launch_restrict_slab(g->ctx->L, fr, g->md, st, coarse.ld, rf, bc);
        return;
    }
    launch_restrict(g->ctx->L, fine, coarse, g->md.ncells, rf, bc);
}

void interface_sum(hmg_grid *g, const LevelDev &lv, double *x, bool faces = true)
{
    launch_interface_sum(g->ctx->L, lv, g->md, x, 0, faces);
    if (has_exchange(g)) exchange_cut(g, lv, x);
}

void scalar_sum(hmg_grid *g, int slot, int count)
{
    if (g->scalar_sum) {
        if (g->scalar_sum(g->ex_user, g->ctx->L.scal + slot, count) != 0)
            throw std::runtime_error("scalar_sum callback failed");
    }
}

int64_t cut_doubles(hmg_grid *g, const LevelDev &lv);
void cut_pack(hmg_grid *g, const LevelDev &lv, double *x, int unpack);
void exchange_prepare(hmg_grid *g, const LevelDev &lv);
void exchange_run(hmg_grid *g, const LevelDev &lv, bool async);
void exchange_finish(hmg_grid *g, const LevelDev &lv, bool async);


// The size of the cut as EVERY rank sees it: per kind the number of cut entities of the whole partition.  A global analysis
// knows it (Partition::nglobal); a halo-only analysis knows the entities this rank shares -- each is counted by the lowest
// rank among its sharers (the segment's first member) and the counts are summed over the ranks through the scalar_sum
// callback, once per partition analysis (first apply on the partitioned grid, and again after a domain shrink).  Collective:
// every rank reaches it at the same point of the same call sequence.
void agree_on_cut(hmg_grid *g)
{
    if (g->cut_agreed_ready) return;
    const Partition &P = *g->part;
    hmg_ctx *c = g->ctx;
    if (P.global_ids) {
        for (int k = 0; k < 3; ++k) g->cut_agreed[k] = g->cut[k].nglobal;
    } else if (!g->scalar_sum) {
        for (int k = 0; k < 3; ++k) g->cut_agreed[k] = -1;        // nothing to agree with: the plain form on every rank
    } else {
        double lead[3] = {0.0, 0.0, 0.0};
        for (const Partition::Segment &S : P.segs)
            if (!S.members.empty() && S.members.front() == P.rank)
                for (int k = 0; k < 3; ++k) lead[k] += (double)S.count[k];
        double *d = c->L.scal + S_HOST;           // (the last slots of the scalar bank are not used by the kernels)
        HIPCHK(hipMemcpyAsync(d, lead, sizeof(lead), hipMemcpyHostToDevice, c->stream));
        scalar_sum(g, S_HOST, 3);
        HIPCHK(hipMemcpyAsync(lead, d, sizeof(lead), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        for (int k = 0; k < 3; ++k) g->cut_agreed[k] = (int64_t)std::llround(lead[k]);
    }
    g->cut_agreed_ready = true;
}

struct TimedRegion {   // HIP-event bracket of the finest-level operator applies (bench.py roofline)
    hmg_ctx *c;
    bool timed;
    TimedRegion(hmg_grid *g, const LevelDev &lv, double bytes) : c(g->ctx)
    {
        ApplyTimer &t = c->timer;
        timed = t.on && lv.level >= t.min_level;
        if (!timed) return;
        if (t.used == t.pool.size()) {
            hipEvent_t a, b;
            HIPCHK(hipEventCreate(&a));
            HIPCHK(hipEventCreate(&b));
            t.pool.emplace_back(a, b);
        }
        HIPCHK(hipEventRecord(t.pool[t.used].first, c->stream));
        t.bytes += bytes;
        t.ev_level.push_back(lv.level);
        t.ev_bytes.push_back(bytes);
    }
    void stop()
    {
        if (!timed) return;
        ApplyTimer &t = c->timer;
        HIPCHK(hipEventRecord(t.pool[t.used].second, c->stream));
        t.used += 1;
        t.launches += 1;
        timed = false;
    }
};

// out = (src ? src : 0) + alpha*A*xin with the Dirichlet constraint, followed by the interface sum of
// `out` (local + across ranks).  fused: the CG extras of k_apply<.., FUSED> (xin = x + beta*x2 written to
// xout, scal[slot_pap] = sum mult*xin*out, scal[slot_rr] = sum xin*xin), summed over ranks.
// On a partitioned grid the cells that own a copy of a cut entity go first; their cut DOFs are packed and
// the sum over ranks is started, then the remaining cells and interface entities are processed while it
// is in flight.
void apply_then_sum(hmg_grid *g, const LevelDev &lv, ApplyArgs a, bool fused, int slot_pap, int slot_rr,
                    bool sum_out = true, bool faces = true)
{
    hmg_ctx *c = g->ctx;
    const Launch &L = c->L;
    set_slab(g, lv);
    ensure_weight_cache(g);
    // algorithmic HBM streams of this launch: x in, out, + src, + x2 (p_old), + xout (p), + xacc (x read and write)
    // (flags bit 7: x is a zero that is not read)
    const double streams = ((a.flags & 128) ? 0.0 : 1.0) + (a.out ? 1.0 : 0.0) + (a.src ? 1.0 : 0.0) + (a.x2 ? 1.0 : 0.0) + (a.xout ? 1.0 : 0.0) +
                           (a.xacc ? 2.0 : 0.0) + (a.x3 ? 1.0 : 0.0) + (a.xcoarse ? (double)lv.nf_coarse / (double)lv.nf : 0.0) +
                           (a.rcoarse ? (double)lv.nf_coarse / (double)lv.nf : 0.0);
    TimedRegion tr(g, lv, 8.0 * (double)lv.nf * (double)g->md.ncells * streams);
    // (overlapping costs launches -- the apply in two parts, the interface sums in two parts: worth it where the exchange
    //  moves real data; on the small levels, launch-bound as they are, it only adds to the chain.  Rehearsal on one rank,
    //  profiles/r03_partitioned_overhead.txt: every level overlapped +3.6 ms per V-cycle, none +0.0.)
    // The decision must be the SAME ON EVERY RANK: the overlapped form issues the exchange before the scalar sums, the plain
    // form after them, and calls on one communicator have to come in one order everywhere.  So it looks at the global size of
    // the cut on this level (identical on all ranks), never at what this rank happens to own; a rank without cut cells walks
    // the overlapped form with empty lists.
    // (round 3 read CutKind::nglobal here, which counts only the cut entities THIS rank has a copy of once the partition is
    //  analysed on the rank's halo: ranks on either side of the threshold chose different forms -- ADVICE r3.  The counts are
    //  now summed over the ranks once per partition analysis, every entity counted by the lowest rank that shares it.)
    if (has_exchange(g) && g->part) agree_on_cut(g);
    const bool has_cut = has_exchange(g) && g->part && g->cut_agreed[0] >= 0 &&
                         g->cut_agreed[0] + g->cut_agreed[1] + g->cut_agreed[2] > 0;
    const int64_t global_cut = has_cut ? g->cut_agreed[0] * lv.nfi + g->cut_agreed[1] * lv.nei + g->cut_agreed[2] : 0;
    const bool overlap = has_cut && (g->sharers ? g->p2p_begin != nullptr : g->ex_begin != nullptr) && g->ex_end && g->overlap &&
                         global_cut >= std::max<int64_t>(1, c->overlap_min_doubles);
    auto launch = [&](const int32_t *list, int64_t n) {
        ApplyArgs b = a;
        b.cell_list = list;
        b.ncell_list = n;
        if (fused)
            launch_apply_fused_kernel(L, lv, g->md, b);
        else
            launch_apply_args(L, lv, g->md, b);
    };
    auto sums = [&]() {
        if (!fused || slot_pap < 0) return;
        launch_apply_fused_reduce(L, g->md, slot_pap, slot_rr);
        // (r.r and p.Ap of CG step 0 sit in neighbouring slots of the scalar bank: one sum over ranks for both)
        if (slot_rr >= 0 && (slot_rr == slot_pap + 1 || slot_rr + 1 == slot_pap))
            scalar_sum(g, std::min(slot_rr, slot_pap), 2);
        else {
            if (slot_rr >= 0) scalar_sum(g, slot_rr, 1);
            scalar_sum(g, slot_pap, 1);
        }
    };
    if (!a.out && a.rcoarse) {       // local residual restricted in the kernel's epilogue: nothing is stored, summed or reduced
        need(fused && !sum_out, "epilogue restriction belongs to the cell-local residual");
        launch(nullptr, 0);
        tr.stop();
        return;
    }
    if (!a.out) {                    // reductions only (dead-tail step of a pre-smoother): nothing to sum or exchange
        need(fused, "apply without an output vector");
        launch(nullptr, 0);
        tr.stop();
        sums();
        return;
    }
    if (!sum_out) {                  // cell-local result wanted (local residual before the restriction): no reductions
        launch(nullptr, 0);
        tr.stop();
        return;
    }
    if (!overlap) {
        launch(nullptr, 0);
        tr.stop();
        sums();
        interface_sum(g, lv, a.out, faces);
        return;
    }
    if (g->md.ncells_cut > 0) launch(g->md.cells_cut, g->md.ncells_cut);
    launch_interface_sum(L, lv, g->md, a.out, 1);        // local copies of the cut entities
    exchange_prepare(g, lv);
    cut_pack(g, lv, a.out, 0);
    exchange_run(g, lv, true);
    if (g->md.ncells_inner > 0)                           // (an empty list must not read as "all cells")
        launch(g->md.cells_inner, g->md.ncells_inner);    // overlaps the sum over ranks
    tr.stop();
    launch_interface_sum(L, lv, g->md, a.out, 2, faces);
    sums();
    exchange_finish(g, lv, true);
    cut_pack(g, lv, a.out, 1);
}

// What a pre-smoother leaves to its caller when defer_x is set (see smooth()).
struct DeferredX {
    int rs = -1;         // >= 0: x += (scal[rs] / scal[pap]) * p_last is still to be done
    int pap = S_PAP;
    // lazy form (two_updates): p_last itself was not formed either.  With p1 = the p handle, r2 = the r handle:
    //   x += (scal[a_num]/scal[a_den]) p1;  p2 = r2 + (scal[b_num]/scal[b_den]) p1;  x += (scal[rs]/scal[pap]) p2
    bool two_updates = false;
    int a_num = -1, a_den = -1, b_num = -1, b_den = -1;
};

// live_tail = false drops the work of the last CG step whose results nobody can read: inside a V-cycle the
// pre-smoother's r, p and Ap are overwritten (local residual, post-smoother's `p = r`, its first `Ap`) before
// control returns to the caller, so of step `steps-1` only alpha = rs / p.Ap and x += alpha p are live -- the
// interface sum of Ap, r -= alpha Ap, r.r and the last p-update (src/multigrid.jl:60-68) are skipped, and the fused
// kernel does not even store Ap (p.Ap comes from the cell-local products and the multiplicities).
// defer_x (with live_tail = false, fused path): the last x += alpha p is left to the caller, which folds it into
// its next operator apply (vcycle: the local residual); returns the scalar slot of rs (alpha = scal[slot] /
// scal[S_PAP]), or -1 if x is already up to date.
// swap_rp (fused path): step 0 does not copy r into p.  p_0 = r_0 stays where it is, r_1 = r_0 - alpha Ap is written
// into the other buffer and the device pointers of the two handles are exchanged (8 B/DOF less at step 0).  Only
// for callers that smooth a level an even number of times before anybody looks at the handles' memory (vcycle: pre-
// and post-smoother), so that wrapped external buffers end up holding what their names say.
// xcoarse (fused path, cells that fit the LDS): x += P xcoarse (the coarse-grid correction, src/multigrid.jl:113) is
// applied in the load phase of the first residual instead of by a separate prolongation pass.
// With defer_x and lazy a dead last step i > 0 writes nothing at all: it forms p_i only in
// LDS for the operator apply and the p.Ap reduction, and leaves both pending x-updates to the caller (DeferredX).
// scratch_p (with live_tail = true): x and r are what the reference leaves, p and Ap are not -- the last p-update
// (src/multigrid.jl:68) is skipped and Ap stays unsummed on the faces (its face sums ride in the r-update).  For the
// post-smoother of the finest level inside hmg_vcycle: the caller reads x and r (the driver's residual norm,
// src/examples/homogenized_coefficients.jl:286), the next smoothing_steps! starts with p <- r and Ap <- 0.
// The spare direction vector of smooth()'s three-update form (lazy_top = 2) is SETUP: reserved when the first vector of the finest
// level is created or wrapped (or by hmg_grid_reserve_spare), never inside a smoother.  Without it (reservation refused for lack of
// memory, option lazy_top < 2 at that time, hmg_grid_reserve_spare(grid, 0)) smooth() takes the two-update form -- which form the
// last finest-level post-smoother took is reported by hmg_ctx_counter "lazy_top_form", the bytes held by "spare_bytes".
bool reserve_top_spare(hmg_grid *g, bool must)
{
    hmg_ctx *c = g->ctx;
    const LevelDev &lv = g->ld[(size_t)g->nlevels - 1];
    const size_t n = (size_t)lv.ld * (size_t)g->mesh_full.ncells;
    if (g->top_spare.n >= n && n > 0) return true;
    if (g->top_spare_refused && !must) return false;
    HIPCHK(hipSetDevice(c->device));
    double *q = nullptr;
    if (hipMalloc((void **)&q, n * sizeof(double)) != hipSuccess) {
        (void)hipGetLastError();
        release_pooled_memory();
        if (hipMalloc((void **)&q, n * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();
            g->top_spare_refused = true;
            if (must)
                throw std::runtime_error("the spare direction vector of the finest level (" + std::to_string((n * sizeof(double)) >> 20) +
                                         " MiB, option lazy_top = 2) does not fit the device memory; V-cycles take the two-update form");
            return false;
        }
    }
    device_allocs() += 1;
    c->spare_bytes -= (int64_t)(g->top_spare.n * sizeof(double));
    g->top_spare.release();
    g->top_spare.p = q;
    g->top_spare.n = n;
    g->top_spare_refused = false;
    c->spare_bytes += (int64_t)(n * sizeof(double));
    return true;
}

void release_top_spare(hmg_grid *g)
{
    if (g->ctx) {
        if (g->top_spare.p) (void)hipStreamSynchronize(g->ctx->stream);
        g->ctx->spare_bytes -= (int64_t)(g->top_spare.n * sizeof(double));
    }
    g->top_spare.release();
}

// the three-update form applies to: 3D grids whose finest level has face interiors (smooth(): top_form)
bool wants_top_spare(const hmg_grid *g, int level)
{
    return g->ctx && g->ctx->lazy_top > 1 && level == g->nlevels && g->dim == 3 && g->ld[(size_t)level - 1].nfi > 0;
}

DeferredX smooth(hmg_grid *g, int level, int steps, hmg_vec *x, hmg_vec *b, hmg_vec *r, hmg_vec *p, hmg_vec *Ap,
                 bool live_tail = true, bool defer_x = false, bool swap_rp = false, const hmg_vec *xcoarse = nullptr,
                 bool lazy = false, bool scratch_p = false, bool x_zero = false)
{
    // x_zero: x is the zero initial guess of a coarse level (src/multigrid.jl:106) and its memory does not hold the zeros:
    // the first residual is r = b under the constraint (b - A 0 = b to the last bit), and the caller must not let anything
    // read x before the deferred x-updates write it (vcycle_down checks that the step pattern guarantees this)
    DeferredX none;
    // ref: src/multigrid.jl:46-71
    const LevelDev &lv = lev(g, level);
    const Launch &L = g->ctx->L;
    const int64_t n = vec_len(x);
    {
        ApplyArgs a{};                                                    // r = b - A x, constraint, interface sum
        a.alpha = -1.0;
        a.lambda = g->lambda;
        a.x = x->d;
        a.src = b->d;
        a.out = r->d;
        a.flags = 1;
        if (x_zero) {
            need(!xcoarse, "zero initial guess with a coarse-grid correction");
            launch_copy(L, r->d, b->d, n);
            launch_mask(L, lv, g->md, r->d, 0);
            interface_sum(g, lv, r->d);
        } else if (xcoarse) {
            a.xcoarse = xcoarse->d;
            a.ldc = lev(g, level - 1).ld;
            a.xout = x->d;
            // (cells that fill a third of the LDS: no room for the coarse column next to three resident images)
            if (g->ctx->prolong_in_image && apply_lds_bytes(lv) > 48 * 1024) a.flags |= 64;
            apply_then_sum(g, lv, a, true, -1, -1);
        } else {
            apply_then_sum(g, lv, a, false, -1, -1);
        }
    }
    int cur = S_RS, other = S_RS2;
    if (g->fuse_cg) {
        // p-update and both reductions ride along with the operator apply (see k_apply<.., FUSED>)
        // Per step:  fused apply  [x += alpha_prev p_old;  p = r + beta p_old;  Ap = A p;  p.Ap (, r.r)]
        //            interface sum of Ap
        //            r -= alpha Ap;  r.r'
        // The x-update of step i rides with the fused apply of step i+1 (which reads p anyway); the last one
        // is done together with the reference's final p-update.
        bool top3 = false;
        for (int i = 0; i < steps; ++i) {
            const bool dead = !live_tail && i == steps - 1;
            ApplyArgs a{};
            a.alpha = 1.0;
            a.lambda = g->lambda;
            a.x = r->d;
            const bool lazy_dead = dead && defer_x && lazy && i > 0;
            // the same dead step when nobody defers the x-updates (a post-smoother below the finest level inside hmg_vcycle, of
            // which only x is read): p_i is formed in LDS only, neither p nor x is written by the apply (16 B/DOF instead of 40),
            // and one pass does both x-updates, p_i formed on the fly (32 B/DOF instead of 24): 48 instead of 64 B/DOF, x the
            // same to the last bit (round 4, option lazy_post)
            const bool lazy_x2 = dead && !defer_x && g->ctx->lazy_post && i > 0;
            // the face part of Ap's interface sum rides in the r-update below, except on the last step of a smoother
            // whose state is handed back (Ap must then hold what the reference leaves)
            const bool ride = g->ctx->fold_faces && lv.dim == 3 && lv.nfi > 0 && !dead &&
                              !(live_tail && !scratch_p && i == steps - 1);
            // the last step of a smoother of which x and r are read and p is scratch (the finest level's post-smoother inside
            // hmg_vcycle): the apply forms p_i in LDS only and writes Ap alone (24 instead of 48 B/DOF), the r-update carries both
            // pending x-updates with p_i formed on the fly (48 B/DOF instead of 26 + the 24 of the final x-update): 72 instead of 98
            // B/DOF for that step, x and r the same to the last bit (round 4, option lazy_top)
            const bool top_form = live_tail && scratch_p && steps >= 2 && g->ctx->lazy_top > 0 && g->ctx->fold_faces && lv.dim == 3 && lv.nfi > 0;
            const bool lazy_top = top_form && i == steps - 1;
            // ... and the step before it (three steps or more) writes its direction into a spare vector next to the previous one
            // instead of over it, so that ITS x-update can wait as well (the apply no longer reads and writes x: 32 instead of
            // 48 B/DOF; the last pass reads one stream more: 56 instead of 48): 8 B/DOF less again (option lazy_top = 2, the
            // default; the spare vector is allocated at first use -- if that fails the form above is taken)
            if (top_form && steps >= 3 && g->ctx->lazy_top > 1 && i == steps - 2 && g->top_spare.n >= (size_t)n) top3 = true;
            if (top_form && i == steps - 1) g->ctx->last_top_form = top3 ? 2 : 1;   // (hmg_vcycle sets it to 0 in front of the post-smoother)
            const bool top3_here = top3 && i == steps - 2;
            a.x2 = i == 0 ? nullptr : (top3 && lazy_top) ? g->top_spare.p : p->d;       // p = r  /  p = r + beta p, beta = rs'/rs
            a.xout = (i == 0 && swap_rp) || lazy_dead || lazy_x2 || lazy_top ? nullptr : top3_here ? g->top_spare.p : p->d;   // (swap_rp: r_0 itself becomes p_0)
            a.xacc = i == 0 || lazy_dead || lazy_x2 || lazy_top || top3_here ? nullptr : x->d;     // x += alpha_{i-1} p_{i-1}
            a.a_num = other;                                              // rs_{i-1} (after the swap below)
            a.a_den = S_PAP;                                              // p_{i-1}.Ap_{i-1}: still the old value here
            a.out = dead ? nullptr : Ap->d;
            a.s_num = cur;
            a.s_den = other;
            a.flags = 1;
            // (the kernels above read the previous p.Ap from S_PAP; the reduction that overwrites it is enqueued
            //  behind them on the same stream)
            if (lazy_x2) {
                apply_then_sum(g, lv, a, true, S_PAP2, -1);        // (S_PAP keeps the previous step's p.Ap, as below)
                launch_cg_x2_update(L, x->d, p->d, r->d, n, other, S_PAP, cur, other, cur, S_PAP2);
                return none;
            }
            if (lazy_dead) {
                // p.Ap of this step goes to its own slot: S_PAP still holds the previous step's, which the caller
                // needs for the first of the two pending x-updates
                apply_then_sum(g, lv, a, true, S_PAP2, -1);
                DeferredX d;
                d.rs = cur;
                d.pap = S_PAP2;
                d.two_updates = true;
                d.a_num = other;   // rs_{i-1}
                d.a_den = S_PAP;   // p_{i-1}.Ap_{i-1}
                d.b_num = cur;     // beta_i = rs_i / rs_{i-1}
                d.b_den = other;
                return d;
            }
            if (top3_here) {
                // slots: rs_{i-1} in `other`, p.Ap_{i-1} in S_PAP (both kept for the deferred x-update), rs_i in `cur`;
                // this step's p.Ap goes to S_PAP2 and its r.r to S_RS3
                apply_then_sum(g, lv, a, true, S_PAP2, -1, true, false);
                launch_cg_rupdate_faces(L, lv, g->md, r->d, r->d, Ap->d, n, cur, S_PAP2, S_RS3);
                scalar_sum(g, S_RS3, 1);
                continue;
            }
            if (lazy_top && top3) {
                // rs_{i-2} in `other`, rs_{i-1} in `cur`, rs_i in S_RS3;  p.Ap_{i-2} in S_PAP, p.Ap_{i-1} in S_PAP2;  p_{i-2} in p, p_{i-1} in the spare
                a.s_num = S_RS3;
                a.s_den = cur;
                apply_then_sum(g, lv, a, true, S_PAP3, -1, true, false);
                launch_cg_rupdate_faces_x(L, lv, g->md, r->d, r->d, Ap->d, n, S_RS3, S_PAP3, other, x->d, g->top_spare.p, cur, S_PAP2,
                                          S_RS3, cur, p->d, other, S_PAP);
                scalar_sum(g, other, 1);
                return none;
            }
            if (lazy_top) {
                // (p.Ap of this step to its own slot: S_PAP keeps the previous step's for the first of the two x-updates)
                apply_then_sum(g, lv, a, true, S_PAP2, -1, true, false);
                launch_cg_rupdate_faces_x(L, lv, g->md, r->d, r->d, Ap->d, n, cur, S_PAP2, other, x->d, p->d, other, S_PAP, cur, other,
                                          nullptr, 0, 0);
                scalar_sum(g, other, 1);
                return none;
            }
            apply_then_sum(g, lv, a, true, S_PAP, i == 0 ? cur : -1, true, !ride);
            const double *r_in = r->d;
            if (i == 0 && swap_rp) {
                std::swap(r->d, p->d);                                    // p now names r_0, r the spare buffer
                std::swap(r->own, p->own);
                std::swap(r->alloc_cells, p->alloc_cells);
                std::swap(r->bytes, p->bytes);
                r_in = p->d;
            }
            if (dead) {
                if (defer_x) {
                    DeferredX d;
                    d.rs = cur;
                    return d;
                }
                launch_cg_xp_update(L, x->d, p->d, r->d, n, cur, S_PAP, cur, other, 0);   // x += (rs / p.Ap) p
                return none;
            }
            if (ride)
                launch_cg_rupdate_faces(L, lv, g->md, r_in, r->d, Ap->d, n, cur, S_PAP, other);
            else
                launch_cg_rupdate(L, r_in, r->d, Ap->d, n, cur, S_PAP, other);   // alpha = rs / p.Ap
            scalar_sum(g, other, 1);
            std::swap(cur, other);
        }
        if (steps > 0) {
            // x += alpha_last p (alpha_last = rs_{s-1} / p.Ap: `other` holds rs_{s-1} after the swap) and the
            // reference's last p-update p = r + (rs_s / rs_{s-1}) p
            launch_cg_xp_update(L, x->d, p->d, r->d, n, other, S_PAP, cur, other, scratch_p ? 0 : 1);
        } else {
            launch_copy_dot(L, p->d, r->d, n, cur);
            scalar_sum(g, cur, 1);
        }
        return none;
    }
    launch_copy_dot(L, p->d, r->d, n, cur);                              // p = r; rs = r.r
    scalar_sum(g, cur, 1);
    for (int i = 0; i < steps; ++i) {
        apply(g, lv, 1.0, p->d, nullptr, Ap->d, 1);                            // Ap = A p, constraint
        interface_sum(g, lv, Ap->d);
        launch_dot(L, p->d, Ap->d, n, S_PAP);
        scalar_sum(g, S_PAP, 1);
        if (!live_tail && i == steps - 1) {
            launch_cg_xp_update(L, x->d, p->d, r->d, n, cur, S_PAP, cur, other, 0);       // x += (rs / p.Ap) p
            return none;
        }
        launch_cg_update(L, x->d, r->d, p->d, Ap->d, n, cur, S_PAP, other);   // alpha = rs/pAp
        scalar_sum(g, other, 1);
        if (scratch_p && i == steps - 1) return none;
        launch_cg_pupdate(L, p->d, r->d, n, other, cur);                       // beta = rs'/rs
        std::swap(cur, other);
    }
    return none;
}

void coarse_probe_drop(hmg_grid *g);

void coarse_setup(hmg_grid *g)
{
    need(g->has_op, "hmg_grid_set_operator must be called first");
    const MeshTables &M = g->part ? g->part->global : g->cur();
    assemble_coarse_matrix(M, g->part ? g->sigma_global.data() : g->sigma.data(), g->lambda, g->cm);
    DryUploads dry_scope(!g->ctx, &g->upload_hash);
    hipStream_t s = g->ctx ? g->ctx->stream : nullptr;
    g->c_rowptr.upload(g->cm.rowptr, s);
    g->c_colidx.upload(g->cm.colidx, s);
    g->c_val.upload(g->cm.val, s);
    g->c_diag.upload(g->cm.diag, s);
    g->c_interior.upload(g->cm.interior, s);
    size_t n = (size_t)std::max<int64_t>(g->cm.n, 1);
    g->c_b.alloc(n);
    g->c_x.alloc(n);
    g->c_r.alloc(n);
    g->c_z.alloc(n);
    g->c_p.alloc(n);
    g->c_q.alloc(n);
    g->c_z2.alloc(n);
    g->c_d.alloc(n);
    {
        // lmax(D^-1 A) <= max_i sum_j |a_ij| / a_ii: an upper bound that HOLDS (the Chebyshev polynomial of the preconditioner must
        // stay positive on the whole spectrum)
        double lmax = 0.0;
        for (int64_t i = 0; i < g->cm.n; ++i) {
            double sabs = 0.0;
            for (int32_t k = g->cm.rowptr[(size_t)i]; k < g->cm.rowptr[(size_t)i + 1]; ++k) sabs += std::fabs(g->cm.val[(size_t)k]);
            if (g->cm.diag[(size_t)i] > 0.0) lmax = std::max(lmax, sabs / g->cm.diag[(size_t)i]);
        }
        g->c_lmax = lmax > 0.0 ? lmax : 2.0;
    }
    g->c_u.alloc((size_t)M.nnodes);
    g->cd.n = g->cm.n;
    g->cd.rowptr = g->c_rowptr.p;
    g->cd.colidx = g->c_colidx.p;
    g->cd.val = g->c_val.p;
    g->cd.diag = g->c_diag.p;
    g->cd.interior = g->c_interior.p;
    if (!g->ctx) return;                           // (host-only grid: the matrix is assembled and checksummed, there is nothing to solve on)
    if (!g->probe->h) {
        HIPCHK(hipHostMalloc((void **)&g->probe->h, 4 * sizeof(double), hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&g->probe->ev, hipEventDisableTiming));
        device_allocs() += 1;
    }
    g->coarse_ready = true;
    // New matrix: the first solve counts its iterations again.  A probe the previous matrix's last solve left behind is
    // waited for and dropped here -- judged by coarse_pcg() it would put the old matrix's count back into the budget
    // (max), and the first solve on the new, possibly harder, system would be enqueued blindly with it.
    coarse_probe_drop(g);
    g->coarse_generation += 1;
    g->coarse_budget = 0;
}

void coarse_probe_wait(hmg_grid *g);
static void probe_unlist(hmg_grid *g);

// Iterations a later solve enqueues blindly, from the count the last judged solve needed.  Plain Jacobi-PCG: 1.5 x + 16 (the
// count moves by 10-20 % from one right-hand side to the next; a no-op iteration costs two launches of ~3 us).  With the
// polynomial preconditioner an iteration is k + 1 launches and takes the residual down by a larger, steadier factor (config 3:
// 27 iterations where plain PCG needs 99): 1.25 x + 4 -- at 1.5 x + 16 the no-op tail was a third of the solve.
int coarse_budget_for(const hmg_ctx *c, int last_it)
{
    return c->coarse_poly > 1 ? last_it + last_it / 4 + 4 : last_it + last_it / 2 + 16;
}

void coarse_pcg(hmg_grid *g)
{
    // CG (preconditioner: coarse_poly Chebyshev iterates of the Jacobi-scaled operator; 1 = plain Jacobi) on (lambda M + K_sigma)[interior, interior] x = b to a relative residual of
    // coarse_rtol; stands in for the reference's CHOLMOD solve (src/multigrid.jl:84).
    // Convergence is decided on the device: k_coarse_pupdate sets a flag once r.r <= rtol^2 b.b and every kernel of
    // the later iterations returns at once, so a fixed number of iterations can be enqueued without a host round trip.
    // The first solve after a (re)assembly finds that number the slow way (a look every coarse_check iterations);
    // later solves enqueue 1.5 x the largest count seen + 16 (the count moves by 10-20 % from one right-hand side to
    // the next; a no-op iteration costs ~3 us of launches), leave a probe (flag, count, r.r) behind in pinned
    // memory and return; the probe is checked at the next solve (or when the iteration count is asked for).
    hmg_ctx *c = g->ctx;
    const Launch &L = c->L;
    const CoarseDev &A = g->cd;
    if (A.n == 0) {
        g->coarse_last_it = 0;
        return;
    }
    coarse_probe_wait(g);                          // the previous solve's verdict (throws if it did not converge)
    CoarseProbe &pr = *g->probe;
    need(pr.h != nullptr, "level-1 solve without a level-1 system (coarse_setup)");
    const double rtol2 = c->coarse_rtol * c->coarse_rtol;
    // Polynomial preconditioner (round 4): z = p_{k-1}(D^-1 A) D^-1 r by k - 1 Chebyshev steps behind the init / update kernel (each
    // one sparse product, no reduction) -- an outer iteration is k + 1 launches for k products instead of two launches and two
    // grid-wide sums per product; about a third fewer launches to the same residual at config 3, half the time at 64^3 cubes.
    const int kpoly = std::max(1, c->coarse_poly);
    const double lmax = 1.02 * g->c_lmax, lmin = lmax / std::max(2.0, c->coarse_poly_ratio);
    const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma1 = theta / delta;
    double *dch = kpoly > 1 ? g->c_d.p : nullptr;
    const double zscale = 1.0 / theta;
    const double *zfinal = g->c_z.p;
    auto cheb_steps = [&]() {            // z_1 (in c_z, d in c_d) -> z_k; returns where it is
        double rho = 1.0 / sigma1;
        const double *zin = g->c_z.p;
        double *zout = g->c_z2.p;
        for (int j = 1; j < kpoly; ++j) {
            const double rho_n = 1.0 / (2.0 * sigma1 - rho);
            launch_coarse_cheb(L, A, g->c_r.p, zin, zout, g->c_d.p, rho_n * rho, 2.0 * rho_n / delta, j == kpoly - 1 ? 1 : 0);
            rho = rho_n;
            const double *t = zin;
            zin = zout;
            zout = const_cast<double *>(t);
        }
        zfinal = zin;
    };
    launch_coarse_init(L, A, g->c_b.p, g->c_x.p, g->c_r.p, g->c_z.p, g->c_p.p, zscale, dch);
    if (kpoly > 1) {
        cheb_steps();
        launch_coarse_rz_from_cheb(L, A);       // r.z of the first iteration
    }
    int slot_old = S_C0, slot_new = S_C3;          // r.z of the current / next iteration
    // One iteration = two launches (k_coarse_direction, k_coarse_update; three until round 3).  The direction launch of
    // iteration j does the bookkeeping of update j-1 (beta, convergence flag, count); a batch ends with a bookkeeping-only
    // launch so that the flag and the count the host (or the probe) reads are those of its last update.
    bool first = true, counted = true;
    auto iterate = [&](int count) {
        for (int q = 0; q < count; ++q) {
            if (first)
                launch_coarse_direction(L, A, g->c_p.p, g->c_q.p, zfinal, slot_old, slot_new, rtol2, 1, 0, kpoly > 1);
            else {
                launch_coarse_direction(L, A, g->c_p.p, g->c_q.p, zfinal, slot_old, slot_new, rtol2, 0, counted ? 0 : 1, kpoly > 1);
                std::swap(slot_old, slot_new);     // (the launch has published the new r.z in the other slot)
            }
            first = false;
            launch_coarse_update(L, A, g->c_x.p, g->c_r.p, g->c_z.p, g->c_p.p, g->c_q.p, slot_old, zscale, dch);
            if (kpoly > 1) cheb_steps();
            counted = false;
        }
        // bookkeeping of the batch's last update (leaves the slots alone: the next regular launch publishes the same value again)
        launch_coarse_direction(L, A, g->c_p.p, g->c_q.p, zfinal, slot_old, slot_new, rtol2, 2, 1, kpoly > 1);
        counted = true;
    };
    auto probe = [&]() {
        HIPCHK(hipMemcpyAsync(pr.h, L.scal + S_DONE, 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(pr.h + 3, L.scal + S_C2, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipEventRecord(pr.ev, c->stream));
        pr.pending = true;
        pr.generation = g->coarse_generation;
        if (std::find(c->probe_grids.begin(), c->probe_grids.end(), g) == c->probe_grids.end()) c->probe_grids.push_back(g);
    };
    if (g->coarse_budget > 0) {
        iterate(g->coarse_budget);
        pr.budget = g->coarse_budget;
        if (c->coarse_probe) probe();
        return;
    }
    int it = 0;
    while (it < c->coarse_maxit) {
        const int chunk = std::min(c->coarse_check, c->coarse_maxit - it);
        iterate(chunk);
        it += chunk;
        pr.budget = it;
        probe();
        HIPCHK(hipEventSynchronize(pr.ev));
        pr.pending = false;                        // judged right here
        probe_unlist(g);
        const double done = pr.h[0], rr = pr.h[2], bb = pr.h[3];
        if (!std::isfinite(rr) || !std::isfinite(bb)) throw std::runtime_error("coarse PCG diverged (non-finite residual)");
        if (done != 0.0 || !(bb > 0.0)) break;
        if (it >= c->coarse_maxit)
            throw std::runtime_error("coarse PCG: no convergence to coarse_rtol within coarse_maxit iterations");
    }
    g->coarse_last_it = (int)pr.h[1];
    g->coarse_budget = std::min(c->coarse_maxit, coarse_budget_for(c, g->coarse_last_it));
}

static void probe_unlist(hmg_grid *g)
{
    if (!g->ctx) return;
    auto &v = g->ctx->probe_grids;
    v.erase(std::remove(v.begin(), v.end(), g), v.end());
}

// Blocks until the probe of the last budgeted solve has landed and judges it.  A solve that ran out of its budget is an
// error (its unconverged x has already been prolonged): the budget is reset, so the caller may simply repeat the V-cycle
// -- the next solve counts its iterations the slow way.
void coarse_probe_wait(hmg_grid *g)
{
    if (!g->probe || !g->probe->pending) return;
    CoarseProbe &pr = *g->probe;
    HIPCHK(hipEventSynchronize(pr.ev));
    pr.pending = false;
    probe_unlist(g);
    if (pr.generation != g->coarse_generation) return;      // (a solve on a matrix that is gone: coarse_probe_drop counts those)
    const double done = pr.h[0], rr = pr.h[2], bb = pr.h[3];
    g->coarse_last_it = (int)pr.h[1];
    if (!std::isfinite(rr) || !std::isfinite(bb)) throw std::runtime_error("coarse PCG diverged (non-finite residual)");
    if (done == 0.0 && bb > 0.0) {
        g->coarse_budget = 0;                      // next solve: find the count the slow way again
        g->coarse_misses += 1;
        throw std::runtime_error("coarse PCG: the last level-1 solve did not reach coarse_rtol within the " +
                                 std::to_string(pr.budget) + " iterations enqueued for it (relative residual " +
                                 std::to_string(std::sqrt(rr / bb)) + "); the V-cycle that used it is inexact -- repeat it, "
                                 "the next solve counts its iterations again");
    }
    g->coarse_budget = std::max(g->coarse_budget, std::min(g->ctx->coarse_maxit, coarse_budget_for(g->ctx, g->coarse_last_it)));
}

// A new level-1 matrix is about to replace the one the pending probe belongs to: wait for it, count a miss, drop it.
void coarse_probe_drop(hmg_grid *g)
{
    if (!g->probe || !g->probe->pending) return;
    CoarseProbe &pr = *g->probe;
    (void)hipEventSynchronize(pr.ev);
    pr.pending = false;
    probe_unlist(g);
    if (pr.h[0] == 0.0 && pr.h[3] > 0.0) g->coarse_misses += 1;
}

// Called wherever the API has just synchronised the context's stream: probes that have landed by then are judged at once,
// so an unconverged budgeted solve is reported by the call that follows its V-cycle (the driver's integrals / residual
// norm), not by the next V-cycle -- and the last V-cycle of a run is judged as well.
void judge_probes(hmg_ctx *c)
{
    while (!c->probe_grids.empty()) {
        hmg_grid *g = c->probe_grids.back();
        c->probe_grids.pop_back();                 // first: the verdict may throw, and a listed grid need not be pending
        coarse_probe_wait(g);
    }
}

void coarse_solve(hmg_grid *g, hmg_vec *b1, hmg_vec *x1)
{
    // ref: src/multigrid.jl:74-93
    if (!g->coarse_ready) coarse_setup(g);
    const LevelDev &lv = lev(g, 1);
    const Launch &L = g->ctx->L;
    interface_sum(g, lv, b1->d);
    if (g->part) {
        // Replicated coarse solve: every rank contributes the nodes it owns to a global nodal vector
        // (one sum over ranks), solves the whole level-1 system, and scatters to its own cells.
        need(g->exchange != nullptr, "partitioned grid: hmg_grid_set_exchange must be called before a coarse solve");
        const int64_t ng = g->part->global.nnodes;
        need(ng <= g->ex_cap, "exchange buffer too small for the coarse gather");
        launch_fill(L, g->ex_buf, ng, 0.0);
        launch_gather_owned(L, g->md, g->d_nodes_g.p, g->d_owned.p, lv.ld, b1->d, g->ex_buf);
        if (g->exchange(g->ex_user, g->ex_buf, ng) != 0) throw std::runtime_error("exchange callback failed");
        launch_coarse_gather_rhs(L, g->cd, g->ex_buf, g->c_b.p);
        coarse_pcg(g);
        launch_coarse_scatter_sol(L, g->cd, ng, g->c_x.p, g->c_u.p);
        launch_scatter_cells(L, g->d_cells_gnode.p, g->md.ncells, g->dim + 1, lv.ld, g->c_u.p, x1->d);
        return;
    }
    launch_gather_base(L, g->md, lv.ld, b1->d, g->c_u.p);
    launch_coarse_gather_rhs(L, g->cd, g->c_u.p, g->c_b.p);
    coarse_pcg(g);
    launch_coarse_scatter_sol(L, g->cd, g->md.nnodes, g->c_x.p, g->c_u.p);
    launch_scatter_base(L, g->md, lv.ld, g->c_u.p, x1->d);
}

// Down leg of one level of the V-cycle (src/multigrid.jl:100-106): pre-smoother, local residual, restriction,
// zero initial guess on the coarser level.  Inside the library the pre-smoother's dead tail is dropped and its pending
// x-update(s) ride in the load phase of the local residual (see smooth()); x, the local residual in r and the
// coarse right-hand side are what the reference leaves, p and Ap are scratch.
// Can level k be entered with a zero initial guess that is never written to memory?  Level 1: the scatter of the coarse
// solution overwrites every entry of x.  Levels above: with the default two CG steps and the lazy dead tail the pre-smoother
// never touches x, and the local residual that carries both pending x-updates writes it (flags bit 7: x is not read).
bool zero_entry_ok(const hmg_grid *g, int k, int steps)
{
    if (!g->ctx->zero_entry) return false;
    if (k == 1) return true;
    if (!(g->fuse_cg && g->ctx->fold_x && g->ctx->lazy_dead && steps == 2)) return false;
    // (register-blocked levels take that residual through the instantiation that also restricts in its epilogue -- the only one
    //  of theirs the zero-input form is compiled into)
    const LevelDev &lv = g->ld[k - 1];
    if (lv.blk_R > 0) return g->ctx->fold_restrict && apply_restricts(g->ctx->L, lv);
    return true;
}

void vcycle_down(hmg_grid *g, int k, int steps, hmg_vec **st, bool inside = false, bool x_zero = false, int steps_next = -1)
{
    // x_zero: this level's x is a zero nobody has written (see zero_entry_ok); steps_next: the CG steps the next coarser
    // level will take (inside hmg_vcycle; decides whether ITS zero initial guess has to be written)
    // inside (hmg_vcycle): nobody can read this level's r before the post-smoother's first residual overwrites it
    // (src/multigrid.jl:104-113), so where the apply kernel can restrict in its epilogue the cell-local residual is
    // never stored -- the coarse right-hand side is the same to the last bit (option fold_restrict)
    hmg_vec **cur = st + 5 * (k - 1);
    hmg_vec **nxt = st + 5 * (k - 2);
    const Launch &L = g->ctx->L;
    // (an odd number of pointer exchanges would leave r and p swapped: both smoother calls take the same `steps`)
    const bool swap_rp = g->ctx->swap_rp && g->fuse_cg && steps > 0;
    const DeferredX dx = smooth(g, k, steps, cur[0], cur[1], cur[2], cur[3], cur[4], /*live_tail=*/false,
                                /*defer_x=*/g->ctx->fold_x, swap_rp, nullptr, /*lazy=*/g->ctx->lazy_dead, false, x_zero);
    need(!x_zero || (dx.rs >= 0 && dx.two_updates), "zero initial guess: the pre-smoother did not defer both x-updates");
    const bool skip_fill = inside && steps_next >= 0 && zero_entry_ok(g, k - 1, steps_next);
    if (dx.rs >= 0) {
        // local residual with the pre-smoother's pending x-update(s) folded into its load phase (x written back),
        // r = b - A x: 40 B/DOF instead of 24 + 24; in the lazy form the dead step wrote neither x nor p (16 B/DOF
        // instead of 40) and this pass reads r as well (48 B/DOF)
        ApplyArgs a{};
        a.alpha = -1.0;
        a.lambda = g->lambda;
        a.x = cur[0]->d;
        a.x2 = cur[3]->d;
        a.xout = cur[0]->d;
        if (dx.two_updates) {
            a.x3 = cur[2]->d;
            a.a_num = dx.a_num;
            a.a_den = dx.a_den;
            a.s_num = dx.b_num;
            a.s_den = dx.b_den;
            a.c_num = dx.rs;
            a.c_den = dx.pap;
        } else {
            a.s_num = dx.rs;
            a.s_den = dx.pap;
        }
        a.src = cur[1]->d;
        a.out = cur[2]->d;
        a.flags = 1 | (x_zero ? 128 : 0);
        if (g->ctx->fold_restrict && apply_restricts(L, lev(g, k))) {
            if (inside) a.out = nullptr;          // (hmg_vcycle_down hands r back: there it is stored as well)
            a.rcoarse = nxt[1]->d;
            a.ldrc = lev(g, k - 1).ld;
            apply_then_sum(g, lev(g, k), a, true, -1, -1, /*sum_out=*/false);
            if (!skip_fill) launch_fill(L, nxt[0]->d, vec_len(nxt[0]), 0.0);
            return;
        }
        apply_then_sum(g, lev(g, k), a, true, -1, -1, /*sum_out=*/false);
    } else {
        apply(g, lev(g, k), -1.0, cur[0]->d, cur[1]->d, cur[2]->d, 1);                        // local residual
    }
    restrict_level(g, k, cur[2]->d, nxt[1]->d);
    if (!skip_fill) launch_fill(L, nxt[0]->d, vec_len(nxt[0]), 0.0);
}

// Up leg (src/multigrid.jl:112-115): coarse-grid correction x_k += P x_{k-1}, post-smoother.
// Inside hmg_vcycle (option lean_post) the post-smoother's dead tail is dropped as the pre-smoother's is: on the finest
// level the caller can read x and r, so only the last p-update and the face sums of the last Ap go (scratch_p); on the
// levels below nothing but x is read before the next visit overwrites r, p and Ap (local residual, p <- r, Ap <- 0:
// src/multigrid.jl:46-50,104), so of the last CG step only alpha and x += alpha p are done.  x (and r on the finest
// level) are bit-identical with the full sequence.
void vcycle_up(hmg_grid *g, int k, int steps, hmg_vec **st, int lean = 0)
{
    hmg_vec **cur = st + 5 * (k - 1);
    hmg_vec **nxt = st + 5 * (k - 2);
    const Launch &L = g->ctx->L;
    const bool swap_rp = g->ctx->swap_rp && g->fuse_cg && steps > 0;
    // coarse-grid correction: folded into the post-smoother's first residual where the fused kernel can hold the
    // coarse column in LDS next to the lattice image (two workgroups per CU must still fit), else a separate pass
    const bool fold_p = g->ctx->fold_prolong && g->fuse_cg && apply_lds_bytes(lev(g, k)) <= 160 * 1024 &&
                        apply_lds_bytes(lev(g, k)) + sizeof(double) * (size_t)lev(g, k - 1).nf <= 80 * 1024;
    if (!fold_p) launch_prolong_add(L, lev(g, k), lev(g, k - 1), g->md.ncells, nxt[0]->d, cur[0]->d);
    const bool x_only = lean == 2 && steps > 0;
    smooth(g, k, steps, cur[0], cur[1], cur[2], cur[3], cur[4], /*live_tail=*/!x_only, false, swap_rp, fold_p ? nxt[0] : nullptr,
           false, /*scratch_p=*/lean == 1);
}

void vcycle(hmg_grid *g, int k, int steps, int steps_coarse, hmg_vec **st, bool top = true)
{
    // ref: src/multigrid.jl:73-119
    if (k == 1) {
        hmg_vec **cur = st;
        coarse_solve(g, cur[1], cur[0]);
        return;
    }
    // (below the top level x is the zero initial guess the level above left -- written only if this level needs it in memory)
    vcycle_down(g, k, steps, st, /*inside=*/true, /*x_zero=*/!top && zero_entry_ok(g, k, steps), steps_coarse);
    vcycle(g, k - 1, steps_coarse, steps_coarse, st, false);
    if (top) g->ctx->last_top_form = 0;
    vcycle_up(g, k, steps, st, g->ctx->lean_post ? (top ? 1 : 2) : 0);
}

// ---- multi-GPU cut exchange -------------------------------------------------------------------
void exchange_cut(hmg_grid *g, const LevelDev &lv, double *x);

}  // namespace

namespace {

// Buffer layout of one level (see CutLevel), built at the first exchange on that level and after every re-partition.
CutLevel &cut_level(hmg_grid *g, const LevelDev &lv)
{
    if (g->cutlv.size() != (size_t)g->nlevels) {
        g->cutlv.clear();
        for (int l = 0; l < g->nlevels; ++l) g->cutlv.emplace_back(new CutLevel);
    }
    CutLevel &C = *g->cutlv[lv.level - 1];
    if (C.ready) return C;
    const int64_t per[3] = {lv.nfi, lv.nei, 1};
    DryUploads dry_scope(!g->ctx, &g->upload_hash);
    hipStream_t st = g->ctx ? g->ctx->stream : nullptr;
    std::vector<int64_t> pos;
    if (!g->sharers) {
        need(!g->part || g->part->global_ids,
             "this partition was analysed on the rank's halo only: its cut ids mean nothing to other ranks -- exchange among "
             "the sharers (hmg_grid_use_comm / hmg_grid_set_exchange_p2p) or create the grid with HMG_PARTITION_ANALYSIS=global");
        int64_t off = 0;
        for (int k = 0; k < 3; ++k) {
            pos.resize(g->cut[k].gid.size());
            for (size_t e = 0; e < pos.size(); ++e) pos[e] = off + g->cut[k].gid[e] * per[k];
            C.pos[k].upload(pos, st);
            off += g->cut[k].nglobal * per[k];
        }
        C.ndoubles = off;
    } else {
        need(g->part != nullptr, "the sharers-only exchange needs the library's own partition analysis");
        const Partition &P = *g->part;
        const size_t nseg = P.segs.size();
        std::vector<int64_t> soff(nseg + 1, 0);
        for (size_t q = 0; q < nseg; ++q)
            soff[q + 1] = soff[q] + P.segs[q].count[0] * per[0] + P.segs[q].count[1] * per[1] + P.segs[q].count[2] * per[2];
        for (int k = 0; k < 3; ++k) {
            pos.resize(g->cut[k].seg.size());
            for (size_t e = 0; e < pos.size(); ++e) {
                const Partition::Segment &S = P.segs[(size_t)g->cut[k].seg[e]];
                const int64_t kbase = k == 0 ? 0 : k == 1 ? S.count[0] * per[0] : S.count[0] * per[0] + S.count[1] * per[1];
                pos[e] = soff[(size_t)g->cut[k].seg[e]] + kbase + g->cut[k].sidx[e] * per[k];
            }
            C.pos[k].upload(pos, st);
        }
        C.ndoubles = soff[nseg];
        // messages (one per segment and peer) and the summation plan: every member adds the members' partial segments in
        // ascending rank order, its own from the buffer, the others' from the staging area -- the same bits on every member
        C.ops.clear();
        std::vector<int64_t> plan{(int64_t)nseg}, mtab;
        plan.resize(1 + 4 * nseg);
        int64_t stage = 0;
        for (size_t q = 0; q < nseg; ++q) {
            const Partition::Segment &S = P.segs[q];
            const int64_t size = soff[q + 1] - soff[q];
            plan[1 + 4 * q] = soff[q];
            plan[2 + 4 * q] = size;
            plan[3 + 4 * q] = (int64_t)S.members.size();
            plan[4 + 4 * q] = (int64_t)mtab.size();
            for (int32_t m : S.members) {
                if (m == P.rank) {
                    mtab.push_back(-1);
                    continue;
                }
                mtab.push_back(stage);
                if (size > 0) {
                    C.ops.push_back(m);
                    C.ops.push_back(soff[q]);
                    C.ops.push_back(size);
                    C.ops.push_back(stage);
                }
                stage += size;
            }
        }
        C.nstage = stage;
        for (size_t q = 0; q < nseg; ++q) plan[4 + 4 * q] += (int64_t)(1 + 4 * nseg);    // absolute offsets of the member tables
        plan.insert(plan.end(), mtab.begin(), mtab.end());
        C.plan.upload(plan, st);
    }
    C.ready = true;
    return C;
}

int64_t cut_doubles(hmg_grid *g, const LevelDev &lv) { return cut_level(g, lv).ndoubles; }

// unpack = 0: buffer <- first local copy of every cut entity;  unpack = 1: every local copy <- buffer
void cut_pack(hmg_grid *g, const LevelDev &lv, double *x, int unpack)
{
    const Launch &L = g->ctx->L;
    CutLevel &C = cut_level(g, lv);
    CutPackArgs a{};
    for (int k = 0; k < 3; ++k) {
        a.n[k] = g->cut[k].nentries;
        a.pos[k] = C.pos[k].p;
        a.cell_lid[k] = g->cut[k].cell_lid.p;
        a.first[k] = g->cut[k].first.p;
    }
    launch_cut_pack(L, lv, a, g->ex_buf, x, unpack);
}

// The sum over ranks of the packed buffer, started (begin) and joined (finish) -- or both at once on the context's stream.
// Global layout: one in-place all-reduce (positions no local copy writes must be zero: filled first).  Segment layout:
// the partial segments travel to the other members, then every member adds them up in rank order.
void exchange_prepare(hmg_grid *g, const LevelDev &lv)
{
    CutLevel &C = cut_level(g, lv);
    need(C.ndoubles <= g->ex_cap, "exchange buffer too small for this level");
    if (g->sharers)
        need(C.nstage <= g->stage_cap, "staging buffer too small for this level");
    else
        launch_fill(g->ctx->L, g->ex_buf, C.ndoubles, 0.0);
}

void exchange_run(hmg_grid *g, const LevelDev &lv, bool async)
{
    CutLevel &C = cut_level(g, lv);
    int rc;
    if (g->sharers) {
        hmg_p2p_fn f = async ? g->p2p_begin : g->p2p;
        need(f != nullptr, "sharers-only exchange: no p2p transport set (hmg_grid_use_comm / hmg_grid_set_exchange_p2p)");
        rc = f(g->ex_user, g->ex_buf, g->stage, (int64_t)C.ops.size() / 4, C.ops.data());
    } else if (async)
        rc = g->ex_begin(g->ex_user, g->ex_buf, C.ndoubles);
    else if (g->exchange)
        rc = g->exchange(g->ex_user, g->ex_buf, C.ndoubles);
    else
        rc = g->ex_begin(g->ex_user, g->ex_buf, C.ndoubles) || g->ex_end(g->ex_user);
    if (rc != 0) throw std::runtime_error("exchange callback failed");
}

void exchange_finish(hmg_grid *g, const LevelDev &lv, bool async)
{
    if (async && g->ex_end(g->ex_user) != 0) throw std::runtime_error("exchange (end) callback failed");
    if (g->sharers) {
        CutLevel &C = cut_level(g, lv);
        launch_seg_sum(g->ctx->L, C.plan.p, C.ndoubles, g->ex_buf, g->stage);
    }
}

void exchange_cut(hmg_grid *g, const LevelDev &lv, double *x)
{
    if (cut_doubles(g, lv) == 0) return;
    exchange_prepare(g, lv);
    cut_pack(g, lv, x, 0);
    exchange_run(g, lv, false);
    exchange_finish(g, lv, false);
    cut_pack(g, lv, x, 1);
}

}  // namespace

// ---- in-library communicator: RCCL, resolved at run time ------------------------------------------
// librccl is opened with dlopen when a communicator is first asked for (a host process that already holds a copy --
// torch bundles one -- shares it), so the library loads and runs single-GPU without RCCL present.
namespace {

struct RcclApi {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi &rccl()
{
    static RcclApi api;
    if (api.h) return api;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names)
        if ((api.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!api.h)
        for (const char *n : names)
            if ((api.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!api.h) throw std::runtime_error(std::string("RCCL not found (dlopen librccl.so.1): ") + dlerror());
    auto sym = [&](const char *n) {
        void *p = dlsym(api.h, n);
        if (!p) throw std::runtime_error(std::string("RCCL symbol missing: ") + n);
        return p;
    };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    return api;
}

void nccl_check(ncclResult_t r, const char *what)
{
    if (r != ncclSuccess) throw std::runtime_error(std::string("RCCL error in ") + what + ": " + rccl().GetErrorString(r));
}

// in-place sum over ranks of n doubles, enqueued on `s`
void comm_allreduce(hmg_ctx *c, double *buf, int64_t n, hipStream_t s)
{
    need(c->comm != nullptr, "this context's communicator has been destroyed (hmg_comm_destroy)");
    nccl_check(rccl().AllReduce(buf, buf, (size_t)n, ncclDouble, ncclSum, c->comm, s), "ncclAllReduce");
    c->comm_calls += 1;
    c->comm_doubles += n;
}

// The messages of one sharers-only exchange as ONE grouped RCCL call: per message a send of this rank's partial segment and
// a receive of the peer's into the staging area.  Two members of a segment list their common segments in the same order
// (Partition::segs), and RCCL matches the k-th send a -> b with the k-th receive of b from a.
void comm_p2p(hmg_ctx *c, double *buf, double *stage, int64_t nmsg, const int64_t *m, hipStream_t s)
{
    if (nmsg == 0) return;
    need(c->comm != nullptr, "this context's communicator has been destroyed (hmg_comm_destroy)");
    int64_t sent = 0;
    nccl_check(rccl().GroupStart(), "ncclGroupStart");
    for (int64_t i = 0; i < nmsg; ++i) {
        // (rehearsal of one rank's share of a larger partition: every peer is this rank itself -- a local copy stands in
        //  for the link)
        const int peer = c->comm_rehearsal ? c->comm_rank : (int)m[4 * i];
        const size_t n = (size_t)m[4 * i + 2];
        nccl_check(rccl().Send(buf + m[4 * i + 1], n, ncclDouble, peer, c->comm, s), "ncclSend");
        nccl_check(rccl().Recv(stage + m[4 * i + 3], n, ncclDouble, peer, c->comm, s), "ncclRecv");
        sent += (int64_t)n;
    }
    nccl_check(rccl().GroupEnd(), "ncclGroupEnd");
    c->comm_calls += 1;
    c->comm_doubles += sent;
}

int comm_p2p_sync(void *user, void *buf, void *stage, int64_t nmsg, const int64_t *msgs)
{
    hmg_grid *g = (hmg_grid *)user;
    try {
        comm_p2p(g->ctx, (double *)buf, (double *)stage, nmsg, msgs, g->ctx->stream);
    } catch (const std::exception &e) {
        last_error() = e.what();
        return 1;
    }
    return 0;
}

int comm_p2p_begin(void *user, void *buf, void *stage, int64_t nmsg, const int64_t *msgs)
{
    hmg_grid *g = (hmg_grid *)user;
    hmg_ctx *c = g->ctx;
    try {
        need(c->comm != nullptr, "this context's communicator has been destroyed (hmg_comm_destroy)");
        HIPCHK(hipEventRecord(c->ev_packed, c->stream));             // the pack kernel
        HIPCHK(hipStreamWaitEvent(c->comm_stream, c->ev_packed, 0));
        comm_p2p(c, (double *)buf, (double *)stage, nmsg, msgs, c->comm_stream);
        HIPCHK(hipEventRecord(c->ev_summed, c->comm_stream));
    } catch (const std::exception &e) {
        last_error() = e.what();
        return 1;
    }
    return 0;
}

// the built-in forms of the exchange callbacks (user = the grid): everything is enqueued on HIP streams, no host
// synchronisation and no foreign code between two kernels of a V-cycle
int comm_exchange(void *user, void *buf, int64_t n)
{
    hmg_grid *g = (hmg_grid *)user;
    try {
        comm_allreduce(g->ctx, (double *)buf, n, g->ctx->stream);
    } catch (const std::exception &e) {
        last_error() = e.what();
        return 1;
    }
    return 0;
}

int comm_exchange_begin(void *user, void *buf, int64_t n)
{
    hmg_grid *g = (hmg_grid *)user;
    hmg_ctx *c = g->ctx;
    try {
        need(c->comm != nullptr, "this context's communicator has been destroyed (hmg_comm_destroy)");
        HIPCHK(hipEventRecord(c->ev_packed, c->stream));             // the pack kernels
        HIPCHK(hipStreamWaitEvent(c->comm_stream, c->ev_packed, 0));
        comm_allreduce(c, (double *)buf, n, c->comm_stream);
        HIPCHK(hipEventRecord(c->ev_summed, c->comm_stream));
    } catch (const std::exception &e) {
        last_error() = e.what();
        return 1;
    }
    return 0;
}

int comm_exchange_end(void *user)
{
    hmg_grid *g = (hmg_grid *)user;
    hipError_t e = hipStreamWaitEvent(g->ctx->stream, g->ctx->ev_summed, 0);
    if (e != hipSuccess) {
        last_error() = std::string("HIP error: ") + hipGetErrorString(e);
        return 1;
    }
    return 0;
}

}  // namespace

// =============================================================================================
static void ctx_unref(hmg_ctx *ctx)
{
    LifetimeLock lock(lifetime_mutex());
    if (!ctx || --ctx->refs > 0) return;
    auto &lc = live_contexts();
    lc.erase(std::remove(lc.begin(), lc.end(), ctx), lc.end());
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &ev : ctx->timer.pool) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    vec_pool_trim(ctx);
    if (ctx->comm) (void)rccl().CommDestroy(ctx->comm);
    if (ctx->ev_packed) (void)hipEventDestroy(ctx->ev_packed);
    if (ctx->ev_summed) (void)hipEventDestroy(ctx->ev_summed);
    if (ctx->comm_stream) (void)hipStreamDestroy(ctx->comm_stream);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

static void grid_unref(hmg_grid *grid)
{
    LifetimeLock lock(lifetime_mutex());
    if (!grid || --grid->refs > 0) return;
    hmg_ctx *c = grid->ctx;
    if (c) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
    }
    probe_unlist(grid);
    release_top_spare(grid);
    if (grid->probe && grid->probe->h) (void)hipHostFree(grid->probe->h);
    if (grid->probe && grid->probe->ev) (void)hipEventDestroy(grid->probe->ev);
    delete grid;
    ctx_unref(c);
}

static double read_scalar(hmg_ctx *c, int slot);
namespace { void judge_probes(hmg_ctx *c); }

extern "C" {

const char *hmg_last_error(void) { return last_error().c_str(); }
int hmg_version(void) { return 1; }

static int ctx_create(int device, void *stream, bool use_given, hmg_ctx **out);

int hmg_ctx_create(int device, void *stream, hmg_ctx **out) { return ctx_create(device, stream, stream != nullptr, out); }

int hmg_ctx_create_on_stream(int device, void *stream, hmg_ctx **out) { return ctx_create(device, stream, true, out); }

static int ctx_create(int device, void *stream, bool use_given, hmg_ctx **out)
{
    HMG_TRY
    need(out != nullptr, "null out pointer");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        throw std::runtime_error("no HIP device available: libhmg_hip needs an MI355X (gfx950); there is no CPU fallback");
    need(device >= 0 && device < ndev, "device index out of range");
    HIPCHK(hipSetDevice(device));
    std::unique_ptr<hmg_ctx> c(new hmg_ctx);
    c->device = device;
    if (use_given) {
        c->stream = (hipStream_t)stream;   // may be the null (legacy default) stream
    } else {
        HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    c->partials.alloc(4096);
    c->scal.alloc(S_COUNT);
    HIPCHK(hipMemsetAsync(c->scal.p, 0, S_COUNT * sizeof(double), c->stream));
    c->L.stream = c->stream;
    c->L.partials = c->partials.p;
    c->L.rpart = nullptr;
    c->L.rpart_cap = 0;
    c->L.scal = c->scal.p;
    c->L.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->L.apply_threads = 0;
    c->L.apply_mass_only = 0;
    c->L.apply_unblocked = 0;
    c->L.persistent_waves = 32 * (int64_t)c->L.num_cu;
    c->L.cell_order = 1;
    c->L.weight_cache = 1;  // level 6: class weight rows from the class-weight cache (k_apply<.., WC>)
    c->L.apply_wave = 1;    // level 5: one wave per cell where the class-weight cache exists (hmg_apply_wave.hip)
    c->L.wave_grid = 16 * (int64_t)c->L.num_cu;
    c->L.n_wave_launches = &c->wave_launches;
    c->L.apply_slab2 = 1;   // level 7: one persistent workgroup per CU, loader and evaluator waves (hmg_apply_slab.hip)
    c->L.n_slab2_launches = &c->slab2_launches;
    c->L.slab2_grid = 0;
    c->L.slab2_force = 0;
    c->L.restrict_slab2 = 1;
    c->L.apply_pack = 1;    // level 2: four cells per wave
    c->L.apply_small = 1;   // levels 2-4: pipelined one-wave kernel (hmg_apply_small.hip)
    c->L.n_small_launches = &c->small_launches;
    c->L.apply_wg512 = 1;   // level 6: three 512-thread workgroups per CU (measured: V-cycle 149.5 -> 141 ms; 3 x 640 threads do not fit the wave slots: 174 ms)
    {
        LifetimeLock lock(lifetime_mutex());
        live_contexts().push_back(c.get());
    }
    *out = c.release();
    HMG_END
}

int hmg_ctx_destroy(hmg_ctx *ctx)
{
    HMG_TRY
    ctx_unref(ctx);
    HMG_END
}

int hmg_ctx_sync(hmg_ctx *ctx)
{
    HMG_TRY
    need(ctx != nullptr, "null ctx");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    judge_probes(ctx);
    HMG_END
}

int64_t hmg_ctx_counter(hmg_ctx *ctx, const char *name)
{
    if (!ctx || !name) return -1;
    const std::string n(name);
    if (n == "wave_launches") return ctx->wave_launches;
    if (n == "slab2_launches") return ctx->slab2_launches;
    if (n == "small_launches") return ctx->small_launches;
    if (n == "comm_calls") return ctx->comm_calls;
    if (n == "device_allocs") return device_allocs().load();
    if (n == "spare_bytes") return ctx->spare_bytes;
    if (n == "lazy_top_form") return ctx->last_top_form;
    if (n == "comm_nranks") return ctx->comm ? ctx->comm_nranks : 0;     // as the RCCL communicator was created; 0: none
    return -1;
}

int hmg_ctx_release_memory(hmg_ctx *ctx)
{
    HMG_TRY
    need(ctx != nullptr, "null ctx");
    vec_pool_trim(ctx);
    HMG_END
}

int hmg_ctx_set_option(hmg_ctx *ctx, const char *name, int64_t value)
{
    HMG_TRY
    need(ctx && name, "null argument");
    std::string n(name);
    if (n == "apply_threads")
        ctx->L.apply_threads = (int)value;

    else if (n == "apply_unblocked")
        ctx->L.apply_unblocked = value != 0;
    else if (n == "apply_wg512")
        ctx->L.apply_wg512 = value != 0;
    else if (n == "apply_pack")            // 1 = default; 0: level 2 one cell per wave like levels 3-4 (A/B knob)
        ctx->L.apply_pack = value != 0;
    else if (n == "apply_small")           // 1 = default; 0: levels 2-4 keep k_apply<3,64,*> (A/B knob)
        ctx->L.apply_small = value != 0;
    else if (n == "weight_cache")          // 1 = default; 0: level 6 combines its class weights per cell (A/B knob)
        ctx->L.weight_cache = value != 0;
    else if (n == "apply_slab2")           // 1 = default; 0: cells larger than the LDS keep k_apply_slab (A/B knob)
        ctx->L.apply_slab2 = value != 0;
    else if (n == "restrict_slab2")        // 1 = default; 0: the stand-alone restriction keeps k_apply_slab (A/B knob)
        ctx->L.restrict_slab2 = value != 0;
    else if (n == "slab2_force")           // experiment: level 6 through the window kernel (needs HMG_SLAB_LDS_KB <= 30 at grid creation)
        ctx->L.slab2_force = value != 0;
    else if (n == "slab2_grid")            // its persistent workgroups (0 = default: one per CU; tests: fewer, many cells each)
        ctx->L.slab2_grid = std::max<int64_t>(0, value);
    else if (n == "apply_wave")            // 1 = default; 0: level 5 keeps the 256-thread kernel (A/B knob)
        ctx->L.apply_wave = value != 0;
    else if (n == "wave_grid")             // persistent waves per CU of the one-wave apply (default 16: what the LDS holds)
        ctx->L.wave_grid = std::max<int64_t>(1, value) * (int64_t)ctx->L.num_cu;
    else if (n == "wave_grid_total")       // ... as an absolute number of waves (tests: fewer waves than cells)
        ctx->L.wave_grid = std::max<int64_t>(1, value);
    else if (n == "cell_order")            // 1 = default: XCD-aware cell order of the register-blocked full-grid apply launches
        ctx->L.cell_order = value != 0;
    else if (n == "persistent_waves")        // per CU; 0 = one workgroup per cell (dev / A-B knob)
        ctx->L.persistent_waves = value > 0 ? value * (int64_t)ctx->L.num_cu : (int64_t)1 << 40;
    else if (n == "coarse_poly")           // Chebyshev iterates per preconditioner application of the level-1 PCG (1 = Jacobi)
        ctx->coarse_poly = std::max<int>(1, std::min<int>(16, (int)value));
    else if (n == "coarse_maxit")
        ctx->coarse_maxit = (int)value;
    else if (n == "coarse_check")
        ctx->coarse_check = std::max<int>(1, (int)value);
    else if (n == "coarse_probe")
        ctx->coarse_probe = value != 0;
    else if (n == "fuse_cg")
        ctx->fuse_cg_default = value != 0;
    else if (n == "fold_x")
        ctx->fold_x = value != 0;
    else if (n == "swap_rp")
        ctx->swap_rp = value != 0;
    else if (n == "fold_prolong")
        ctx->fold_prolong = value != 0;
    else if (n == "zero_entry")
        ctx->zero_entry = value != 0;
    else if (n == "fold_restrict")
        ctx->fold_restrict = value != 0;
    else if (n == "lazy_dead")
        ctx->lazy_dead = value != 0;
    else if (n == "fold_faces")
        ctx->fold_faces = value != 0;
    else if (n == "lean_post")
        ctx->lean_post = value != 0;
    else if (n == "lazy_post")
        ctx->lazy_post = value != 0;
    else if (n == "lazy_top")
        ctx->lazy_top = (int)value;
    else if (n == "prolong_in_image" || n == "prolong_gather")   // (prolong_gather: the option's name in round 2)
        ctx->prolong_in_image = value != 0;
    else if (n == "overlap_min_doubles")
        ctx->overlap_min_doubles = value;
    else if (n == "comm_rehearsal")
        ctx->comm_rehearsal = value != 0;
    else if (n == "vec_pool") {
        ctx->vec_pool_on = value != 0;
        if (!ctx->vec_pool_on) vec_pool_trim(ctx);
    }
    else if (n == "time_apply") {   // value = minimum level to time, 0 = off; resets the counters
        ctx->timer.on = value > 0;
        ctx->timer.min_level = (int)value;
        ctx->timer.used = 0;
        ctx->timer.bytes = 0.0;
        ctx->timer.launches = 0;
        ctx->timer.ev_level.clear();
        ctx->timer.ev_bytes.clear();
    }
    else
        throw std::runtime_error("unknown option: " + n);
    HMG_END
}

int hmg_ctx_set_option_f64(hmg_ctx *ctx, const char *name, double value)
{
    HMG_TRY
    need(ctx && name, "null argument");
    std::string n(name);
    if (n == "coarse_rtol")
        ctx->coarse_rtol = value;
    else if (n == "coarse_poly_ratio")     // lmax / lmin of the interval the level-1 PCG's Chebyshev preconditioner is built for
        ctx->coarse_poly_ratio = value;
    else
        throw std::runtime_error("unknown option: " + n);
    HMG_END
}

void *hmg_ctx_scalar_bank(hmg_ctx *ctx) { return ctx ? (void *)ctx->L.scal : nullptr; }

/* Replace the library's scalar bank (16 device doubles) by caller-owned device memory, e.g. a torch
 * tensor that torch.distributed can all-reduce in place. */
int hmg_ctx_set_scalar_bank(hmg_ctx *ctx, void *device_doubles16)
{
    HMG_TRY
    need(ctx != nullptr, "null argument");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    double *to = device_doubles16 ? (double *)device_doubles16 : ctx->scal.p;   // NULL: back to the library's own bank
    if (to != ctx->L.scal) HIPCHK(hipMemcpy(to, ctx->L.scal, S_COUNT * sizeof(double), hipMemcpyDeviceToDevice));
    ctx->L.scal = to;
    HMG_END
}

int hmg_ctx_apply_timing(hmg_ctx *ctx, int64_t *launches, double *total_ms, double *total_bytes)
{
    HMG_TRY
    need(ctx && launches && total_ms && total_bytes, "null argument");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    double ms = 0.0;
    for (size_t i = 0; i < ctx->timer.used; ++i) {
        float t = 0.f;
        HIPCHK(hipEventElapsedTime(&t, ctx->timer.pool[i].first, ctx->timer.pool[i].second));
        ms += t;
    }
    *launches = ctx->timer.launches;
    *total_ms = ms;
    *total_bytes = ctx->timer.bytes;
    HMG_END
}

int hmg_ctx_apply_timing_level(hmg_ctx *ctx, int level, int64_t *launches, double *total_ms, double *total_bytes)
{
    HMG_TRY
    need(ctx && launches && total_ms && total_bytes, "null argument");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const ApplyTimer &tm = ctx->timer;
    double ms = 0.0, by = 0.0;
    int64_t n = 0;
    for (size_t i = 0; i < tm.used && i < tm.ev_level.size(); ++i) {
        if (tm.ev_level[i] != level) continue;
        float t = 0.f;
        HIPCHK(hipEventElapsedTime(&t, tm.pool[i].first, tm.pool[i].second));
        ms += t;
        by += tm.ev_bytes[i];
        n += 1;
    }
    *launches = n;
    *total_ms = ms;
    *total_bytes = by;
    HMG_END
}

int hmg_grid_create(hmg_ctx *ctx, int dim, int nlevels, int64_t nnodes, const double *coords, int64_t ncells,
                    const int64_t *cells, hmg_grid **out)
{
    HMG_TRY
    need(coords && cells && out, "null argument");
    if (ctx) HIPCHK(hipSetDevice(ctx->device));
    std::unique_ptr<hmg_grid> g(new hmg_grid);
    g->ctx = ctx;
    g->fuse_cg = ctx ? ctx->fuse_cg_default : true;
    g->dim = dim;
    g->nlevels = nlevels;
    g->lt = build_level_tables(dim, nlevels);
    build_mesh_tables(dim, nnodes, coords, ncells, cells, g->mesh_full);
    upload_levels(g.get());
    upload_mesh(g.get());
    if (ctx) {
        LifetimeLock lock(lifetime_mutex());
        ctx->refs += 1;
    }
    *out = g.release();
    HMG_END
}

static void set_cut_kind(hmg_grid *g, int k, int64_t nglobal, int64_t n, const int64_t *gid, const int32_t *cell_lid,
                         const int32_t *seg, const int64_t *sidx);
static void upload_levels(hmg_grid *g);

// device side of the partition tables (after upload_mesh): cut lists, node ownership, global node ids of the cells
static void finish_partition(hmg_grid *g)
{
    const Partition &P = *g->part;
    const MeshTables &M = g->cur();
    for (int k = 0; k < 3; ++k)
        set_cut_kind(g, k, P.nglobal[k], (int64_t)P.gid[k].size(), P.gid[k].data(), P.cell_lid[k].data(), P.seg_of[k].data(),
                     P.seg_idx[k].data());
    DryUploads dry_scope(!g->ctx, &g->upload_hash);
    hipStream_t s = g->ctx ? g->ctx->stream : nullptr;
    g->d_nodes_g.upload(P.nodes_g, s);
    g->d_owned.upload(P.owned_node, s);
    std::vector<int32_t> cg(M.cells.size());
    for (size_t q = 0; q < cg.size(); ++q) cg[q] = P.nodes_g[M.cells[q]];
    g->d_cells_gnode.upload(cg, s);
}

static int create_partition(hmg_ctx *ctx, int dim, int nlevels, int64_t nnodes, const double *coords, int64_t ncells,
                            const int64_t *cells, const int32_t *owner, const int32_t *cut_owner, int rank, int nranks,
                            hmg_grid **out)
{
    HMG_TRY
    need(coords && cells && owner && out, "null argument");
    if (ctx) HIPCHK(hipSetDevice(ctx->device));
    std::unique_ptr<hmg_grid> g(new hmg_grid);
    g->ctx = ctx;
    g->fuse_cg = ctx ? ctx->fuse_cg_default : true;
    g->dim = dim;
    g->nlevels = nlevels;
    g->lt = build_level_tables(dim, nlevels);
    g->part.reset(new Partition);
    // The analysis looks at this rank's cells and their one-cell halo (the global pass keeps only what the replicated level-1
    // system needs); HMG_PARTITION_ANALYSIS=global, or HMG_EXCHANGE=allreduce -- which needs cut ids every rank agrees on --,
    // bring back the analysis of the whole mesh on every rank.
    {
        const char *pa = std::getenv("HMG_PARTITION_ANALYSIS"), *ex = std::getenv("HMG_EXCHANGE");
        g->part_halo = !((pa && std::string(pa) == "global") || (ex && std::string(ex) == "allreduce"));
    }
    build_partition(dim, nnodes, coords, ncells, cells, owner, rank, nranks, g->mesh_full, *g->part, cut_owner, g->part_halo);
    g->part_coords.assign(coords, coords + (size_t)dim * nnodes);
    g->part_cells.assign(cells, cells + (size_t)(dim + 1) * ncells);
    g->part_owner.assign(owner, owner + ncells);
    if (cut_owner) g->part_cut_owner.assign(cut_owner, cut_owner + ncells);
    g->part_nnodes = nnodes;
    g->part_ncells = ncells;
    upload_levels(g.get());
    upload_mesh(g.get());
    finish_partition(g.get());
    if (ctx) {
        LifetimeLock lock(lifetime_mutex());
        ctx->refs += 1;
    }
    *out = g.release();
    HMG_END
}

int hmg_grid_create_partition(hmg_ctx *ctx, int dim, int nlevels, int64_t nnodes, const double *coords, int64_t ncells,
                              const int64_t *cells, const int32_t *owner, int rank, int nranks, hmg_grid **out)
{
    return create_partition(ctx, dim, nlevels, nnodes, coords, ncells, cells, owner, nullptr, rank, nranks, out);
}

int hmg_grid_create_partition_rehearsal(hmg_ctx *ctx, int dim, int nlevels, int64_t nnodes, const double *coords,
                                        int64_t ncells, const int64_t *cells, const int32_t *owner, const int32_t *cut_owner,
                                        int rank, int nranks, hmg_grid **out)
{
    return create_partition(ctx, dim, nlevels, nnodes, coords, ncells, cells, owner, cut_owner, rank, nranks, out);
}

int hmg_grid_destroy(hmg_grid *grid)
{
    HMG_TRY
    grid_unref(grid);
    HMG_END
}

int hmg_grid_set_operator(hmg_grid *g, const double *sigma, double lambda)
{
    HMG_TRY
    need(g && sigma, "null argument");
    if (g->part)
        g->sigma_global.assign(sigma, sigma + (size_t)g->part->global.ncells * g->dim);
    else
        g->sigma.assign(sigma, sigma + (size_t)g->mesh_full.ncells * g->dim);
    g->lambda = lambda;
    g->has_op = true;
    upload_operator(g);
    HMG_END
}

int hmg_grid_set_lambda(hmg_grid *g, double lambda)
{
    HMG_TRY
    need(g != nullptr, "null grid");
    g->lambda = lambda;
    g->coarse_ready = false;
    if (g->ctx && g->has_op) ensure_weight_cache(g);
    HMG_END
}

int hmg_grid_reserve_spare(hmg_grid *g, int enable)
{
    HMG_TRY
    need(g != nullptr, "null grid");
    need(g->ctx != nullptr, "this grid was created without a device context (host tables only)");
    if (enable) {
        (void)reserve_top_spare(g, true);
    } else {
        release_top_spare(g);
        g->top_spare_refused = true;             // ... and the first vector of the finest level does not bring it back
    }
    HMG_END
}

int hmg_grid_shrink(hmg_grid *g, int64_t ncells_prefix, int64_t nnodes_prefix)
{
    HMG_TRY
    need(g != nullptr, "null grid");
    if (g->part) {
        // prefix of the GLOBAL mesh: this rank keeps its cells with a global id below the prefix length (local
        // cells are in ascending global order, so that is a prefix of every local level vector as well); cut
        // entities, Dirichlet masks, multiplicities and node ownership are re-derived from the smaller global mesh
        need(ncells_prefix >= 1 && ncells_prefix <= g->part_ncells && nnodes_prefix >= 1 && nnodes_prefix <= g->part_nnodes,
             "prefix out of range");
        const int rank = g->part->rank, nranks = g->part->nranks;
        std::unique_ptr<Partition> np(new Partition);
        MeshTables local;
        build_partition(g->dim, nnodes_prefix, g->part_coords.data(), ncells_prefix, g->part_cells.data(),
                        g->part_owner.data(), rank, nranks, local, *np,
                        g->part_cut_owner.empty() ? nullptr : g->part_cut_owner.data(), g->part_halo);
        need(local.ncells <= g->mesh_full.ncells, "shrunk partition is larger than the original one");
        g->mesh = std::move(local);
        g->part = std::move(np);
        g->shrunk = true;
        upload_mesh(g);
        finish_partition(g);
        if (g->has_op) upload_operator(g);
        return 0;
    }
    restrict_mesh_tables(g->mesh_full, ncells_prefix, nnodes_prefix, g->mesh);
    g->shrunk = true;
    upload_mesh(g);
    if (g->has_op) upload_operator(g);
    HMG_END
}

int64_t hmg_grid_ncells(const hmg_grid *g) { return g ? g->md.ncells : -1; }
int64_t hmg_grid_nnodes(const hmg_grid *g) { return g ? g->md.nnodes : -1; }
int hmg_grid_nlevels(const hmg_grid *g) { return g ? g->nlevels : -1; }
int64_t hmg_grid_nf(const hmg_grid *g, int level)
{
    return (g && level >= 1 && level <= g->nlevels) ? g->lt[level - 1].nf : -1;
}
int64_t hmg_grid_ld(const hmg_grid *g, int level)
{
    return (g && level >= 1 && level <= g->nlevels) ? g->lt[level - 1].ld : -1;
}

int hmg_grid_table_i32(const hmg_grid *g, int level, const char *which, int32_t *out, int64_t cap, int64_t *count)
{
    HMG_TRY
    need(g && which && count, "null argument");
    std::string w(which);
    std::vector<int32_t> tmp;
    const std::vector<int32_t> *src = nullptr;
    if (w == "dmask" || w == "dupmask") {
        const auto &m = w == "dmask" ? g->cur().dmask : g->cur().dupmask;
        tmp.assign(m.begin(), m.end());
        src = &tmp;
    } else if (w == "upload_hash") {   // host-only grids: checksum of every table a device grid would have uploaded so far (two halves)
        tmp = {(int32_t)(uint32_t)(g->upload_hash & 0xffffffffu), (int32_t)(uint32_t)(g->upload_hash >> 32)};
        src = &tmp;
    } else if (w == "face_pairs") {
        src = &g->cur().face_pairs;
    } else if (w == "edge_ptr") {
        src = &g->cur().edge_ptr;
    } else if (w == "edge_ent") {
        src = &g->cur().edge_ent;
    } else if (w == "node_ptr") {
        src = &g->cur().node_ptr;
    } else if (w == "node_ent") {
        src = &g->cur().node_ent;
    } else if (w == "node_first") {
        src = &g->cur().node_first;
    } else if (w == "coarse_rowptr") {
        src = &g->cm.rowptr;
    } else if (w == "coarse_colidx") {
        src = &g->cm.colidx;
    } else if (w == "part_cells") {
        need(g->part != nullptr, "not a partitioned grid");
        src = &g->part->cells_g;
    } else if (w == "part_nodes") {
        need(g->part != nullptr, "not a partitioned grid");
        src = &g->part->nodes_g;
    } else if (w == "part_owned") {
        need(g->part != nullptr, "not a partitioned grid");
        src = &g->part->owned_node;
    } else if (w == "cut_counts") {
        need(g->part != nullptr, "not a partitioned grid");
        tmp = {(int32_t)g->part->nglobal[0], (int32_t)g->part->nglobal[1], (int32_t)g->part->nglobal[2],
               (int32_t)g->part->gid[0].size(), (int32_t)g->part->gid[1].size(), (int32_t)g->part->gid[2].size(),
               (int32_t)g->cur().ncut_face_pairs, (int32_t)g->cur().ncut_edge_groups, (int32_t)g->cur().ncut_node_groups,
               (int32_t)g->cur().cells_cut.size(), (int32_t)g->cur().cells_inner.size()};
        src = &tmp;
    } else if (w == "seg_ptr" || w == "seg_members" || w == "seg_counts") {
        need(g->part != nullptr, "not a partitioned grid");
        if (w == "seg_ptr") tmp.push_back(0);
        for (const auto &S : g->part->segs) {
            if (w == "seg_ptr")
                tmp.push_back(tmp.back() + (int32_t)S.members.size());
            else if (w == "seg_members")
                tmp.insert(tmp.end(), S.members.begin(), S.members.end());
            else
                for (int k = 0; k < 3; ++k) tmp.push_back((int32_t)S.count[k]);
        }
        src = &tmp;
    } else if (w == "cut_seg_faces" || w == "cut_seg_edges" || w == "cut_seg_nodes" || w == "cut_sidx_faces" ||
               w == "cut_sidx_edges" || w == "cut_sidx_nodes") {
        need(g->part != nullptr, "not a partitioned grid");
        const int k = w.find("faces") != std::string::npos ? 0 : w.find("edges") != std::string::npos ? 1 : 2;
        if (w.find("sidx") != std::string::npos)
            tmp.assign(g->part->seg_idx[k].begin(), g->part->seg_idx[k].end());
        else
            tmp = g->part->seg_of[k];
        src = &tmp;
    } else if (w == "cut_gid_faces" || w == "cut_gid_edges" || w == "cut_gid_nodes" || w == "cut_ent_faces" ||
               w == "cut_ent_edges" || w == "cut_ent_nodes") {
        need(g->part != nullptr, "not a partitioned grid");
        const int k = w.find("faces") != std::string::npos ? 0 : w.find("edges") != std::string::npos ? 1 : 2;
        if (w.find("gid") != std::string::npos)
            tmp.assign(g->part->gid[k].begin(), g->part->gid[k].end());
        else
            tmp = g->part->cell_lid[k];
        src = &tmp;
    } else if (w == "mult") {
        tmp.assign(g->cur().mult.begin(), g->cur().mult.end());
        src = &tmp;
    } else if (w == "interior_nodes") {
        const MeshTables &M = g->cur();
        for (int64_t i = 0; i < M.nnodes; ++i)
            if (!M.node_on_boundary[i] && M.node_first[i] >= 0) tmp.push_back((int32_t)i);
        src = &tmp;
    } else {
        need(level >= 1 && level <= g->nlevels, "level out of range");
        const LevelTables &T = g->lt[level - 1];
        if (w == "hier2slot")
            src = &T.hier2slot;
        else if (w == "slot_ijk")
            src = &T.slot_ijk;
        else if (w == "ref_cells")
            src = &T.ref_cells;
        else if (w == "par_a")
            src = &T.par_a;
        else if (w == "par_b")
            src = &T.par_b;
        else if (w == "rptr")
            src = &T.rptr;
        else if (w == "ridx")
            src = &T.ridx;
        else if (w == "slot_cls") {
            tmp.assign(T.slot_cls.begin(), T.slot_cls.end());
            src = &tmp;
        } else if (w == "layout") {
            tmp = {T.nf, T.ld, T.ncorner, T.nedge, T.nface, T.nei, T.nfi, T.nint, T.off_edge, T.off_face, T.off_int,
                   T.ncls, T.ndir, T.nterm, T.lds_g0, T.lds_g1, T.m};
            src = &tmp;
        } else
            throw std::runtime_error("unknown i32 table: " + w);
    }
    *count = (int64_t)src->size();
    if (out) {
        need(cap >= *count, "output buffer too small");
        std::copy(src->begin(), src->end(), out);
    }
    HMG_END
}

int hmg_grid_table_f64(const hmg_grid *g, int level, const char *which, double *out, int64_t cap, int64_t *count)
{
    HMG_TRY
    need(g && which && count, "null argument");
    std::string w(which);
    const std::vector<double> *src = nullptr;
    if (w == "coef")
        src = &g->coef;
    else if (w == "ctab") {
        need(level >= 1 && level <= g->nlevels, "level out of range");
        src = &g->lt[level - 1].ctab;
    } else if (w == "coarse_val")
        src = &g->cm.val;
    else if (w == "load") {
        need(level >= 1 && level <= g->nlevels, "level out of range");
        src = &g->lt[level - 1].load;
    }
#ifdef HMG_PHASE_TIMING
    else if (w == "phase_stamps") {
        need(g->ctx != nullptr, "no device");
        *count = (int64_t)g->d_blockpart.n;
        if (out) {
            need(cap >= *count, "output buffer too small");
            HIPCHK(hipStreamSynchronize(g->ctx->stream));
            HIPCHK(hipMemcpy(out, g->d_blockpart.p, sizeof(double) * g->d_blockpart.n, hipMemcpyDeviceToHost));
        }
        return 0;
    }
#endif
    else
        throw std::runtime_error("unknown f64 table: " + w);
    *count = (int64_t)src->size();
    if (out) {
        need(cap >= *count, "output buffer too small");
        std::copy(src->begin(), src->end(), out);
    }
    HMG_END
}

// ---- vectors ----------------------------------------------------------------------------------
int hmg_vec_create(hmg_grid *g, int level, hmg_vec **out)
{
    HMG_TRY
    need(g && out, "null argument");
    const LevelDev &lv = lev(g, level);
    HIPCHK(hipSetDevice(g->ctx->device));
    std::unique_ptr<hmg_vec> v(new hmg_vec);
    v->g = g;
    v->level = level;
    v->own = true;
    v->alloc_cells = g->md.ncells;
    size_t bytes = sizeof(double) * (size_t)lv.ld * (size_t)g->md.ncells;
    ensure_reduce_scratch(g->ctx, (int64_t)lv.ld * g->md.ncells);
    v->d = vec_alloc(g->ctx, bytes);
    v->bytes = bytes;
    if (wants_top_spare(g, level)) (void)reserve_top_spare(g, false);
    {
        LifetimeLock lock(lifetime_mutex());
        g->refs += 1;
    }
    *out = v.release();
    HMG_END
}

int hmg_vec_wrap(hmg_grid *g, int level, void *device_ptr, hmg_vec **out)
{
    HMG_TRY
    need(g && out && device_ptr, "null argument");
    ensure_reduce_scratch(g->ctx, (int64_t)lev(g, level).ld * g->md.ncells);
    std::unique_ptr<hmg_vec> v(new hmg_vec);
    v->g = g;
    v->level = level;
    v->own = false;
    v->alloc_cells = g->md.ncells;
    v->d = (double *)device_ptr;
    if (wants_top_spare(g, level)) (void)reserve_top_spare(g, false);
    {
        LifetimeLock lock(lifetime_mutex());
        g->refs += 1;
    }
    *out = v.release();
    HMG_END
}

int hmg_vec_destroy(hmg_vec *v)
{
    HMG_TRY
    if (v) {
        if (v->own && v->d) vec_release(v->g->ctx, v->d, v->bytes);
        hmg_grid *g = v->g;
        delete v;
        grid_unref(g);
    }
    HMG_END
}

void *hmg_vec_device_ptr(hmg_vec *v) { return v ? (void *)v->d : nullptr; }

static const int64_t STAGE_DOUBLES = (int64_t)32 << 20;   // 256 MiB staging chunks

int hmg_vec_upload(hmg_vec *v, const double *host)
{
    HMG_TRY
    need(v && host, "null argument");
    hmg_grid *g = v->g;
    const LevelDev &lv = lev(g, v->level);
    const int64_t ncells = g->md.ncells;
    int64_t cells_per = std::max<int64_t>(1, STAGE_DOUBLES / lv.nf);
    cells_per = std::min(cells_per, ncells);
    DevBuf<double> stage;
    stage.alloc((size_t)cells_per * lv.nf);
    for (int64_t c0 = 0; c0 < ncells; c0 += cells_per) {
        int64_t nc = std::min(cells_per, ncells - c0);
        HIPCHK(hipMemcpyAsync(stage.p, host + c0 * lv.nf, sizeof(double) * nc * lv.nf, hipMemcpyHostToDevice,
                              g->ctx->stream));
        launch_permute(g->ctx->L, lv, nc, stage.p, v->d + c0 * lv.ld, 1);
        HIPCHK(hipStreamSynchronize(g->ctx->stream));
    }
    HMG_END
}

int hmg_vec_download(hmg_vec *v, double *host)
{
    HMG_TRY
    need(v && host, "null argument");
    hmg_grid *g = v->g;
    const LevelDev &lv = lev(g, v->level);
    const int64_t ncells = g->md.ncells;
    int64_t cells_per = std::max<int64_t>(1, STAGE_DOUBLES / lv.nf);
    cells_per = std::min(cells_per, ncells);
    DevBuf<double> stage;
    stage.alloc((size_t)cells_per * lv.nf);
    for (int64_t c0 = 0; c0 < ncells; c0 += cells_per) {
        int64_t nc = std::min(cells_per, ncells - c0);
        launch_permute(g->ctx->L, lv, nc, v->d + c0 * lv.ld, stage.p, 0);
        HIPCHK(hipMemcpyAsync(host + c0 * lv.nf, stage.p, sizeof(double) * nc * lv.nf, hipMemcpyDeviceToHost,
                              g->ctx->stream));
        HIPCHK(hipStreamSynchronize(g->ctx->stream));
    }
    HMG_END
}

int hmg_vec_fill(hmg_vec *v, double value)
{
    HMG_TRY
    need(v != nullptr, "null vector");
    launch_fill(v->g->ctx->L, v->d, vec_len(v), value);
    HMG_END
}

int hmg_vec_fill_random(hmg_vec *v, uint64_t seed, int64_t cell_offset)
{
    HMG_TRY
    need(v != nullptr, "null vector");
    launch_fill_random(v->g->ctx->L, lev(v->g, v->level), v->g->md.ncells, v->d, seed, cell_offset);
    HMG_END
}

int hmg_vec_copy(hmg_vec *dst, hmg_vec *src)
{
    HMG_TRY
    need(dst && src, "null vector");
    check_vec(dst->g, dst->level, src, "src");
    launch_copy(dst->g->ctx->L, dst->d, src->d, vec_len(dst));
    HMG_END
}

int hmg_vec_axpy(double alpha, hmg_vec *x, hmg_vec *y)
{
    HMG_TRY
    need(x && y, "null vector");
    check_vec(y->g, y->level, x, "x");
    launch_axpy(y->g->ctx->L, alpha, x->d, y->d, vec_len(y));
    HMG_END
}

int hmg_vec_xpby(hmg_vec *r, double beta, hmg_vec *p)
{
    HMG_TRY
    need(r && p, "null vector");
    check_vec(p->g, p->level, r, "r");
    launch_xpby(p->g->ctx->L, r->d, beta, p->d, vec_len(p));
    HMG_END
}

static double read_scalar(hmg_ctx *c, int slot)
{
    double h = 0.0;
    HIPCHK(hipMemcpyAsync(&h, c->L.scal + slot, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    judge_probes(c);
    return h;
}

int hmg_vec_dot(hmg_vec *x, hmg_vec *y, double *out)
{
    HMG_TRY
    need(x && y && out, "null argument");
    check_vec(x->g, x->level, y, "y");
    launch_dot(x->g->ctx->L, x->d, y->d, vec_len(x), S_TMP);
    scalar_sum(x->g, S_TMP, 1);
    *out = read_scalar(x->g->ctx, S_TMP);
    HMG_END
}

int hmg_vec_norm_unique(hmg_vec *r, double *out)
{
    HMG_TRY
    need(r && out, "null argument");
    launch_norm2_unique(r->g->ctx->L, lev(r->g, r->level), r->g->md, r->d, S_TMP);
    scalar_sum(r->g, S_TMP, 1);
    *out = std::sqrt(read_scalar(r->g->ctx, S_TMP));
    HMG_END
}

// ---- primitives ---------------------------------------------------------------------------------
int hmg_apply(hmg_grid *g, int level, double alpha, hmg_vec *x, hmg_vec *y)
{
    HMG_TRY
    need(g && g->has_op, "operator not set");
    check_vec(g, level, x, "x");
    check_vec(g, level, y, "y");
    need(x->d != y->d, "x and y must not alias");
    apply(g, lev(g, level), alpha, x->d, y->d, y->d, 0);
    HMG_END
}

int hmg_apply_ex(hmg_grid *g, int level, double alpha, hmg_vec *x, hmg_vec *src, hmg_vec *out, int constrain)
{
    HMG_TRY
    need(g && g->has_op, "operator not set");
    check_vec(g, level, x, "x");
    check_vec(g, level, out, "out");
    if (src) check_vec(g, level, src, "src");
    need(x->d != out->d, "x and out must not alias");
    apply(g, lev(g, level), alpha, x->d, src ? src->d : nullptr, out->d, constrain ? 1 : 0);
    HMG_END
}

int hmg_residual(hmg_grid *g, int level, hmg_vec *x, hmg_vec *b, hmg_vec *r)
{
    HMG_TRY
    need(g && g->has_op, "operator not set");
    check_vec(g, level, x, "x");
    check_vec(g, level, b, "b");
    check_vec(g, level, r, "r");
    need(x->d != r->d, "x and r must not alias");
    apply(g, lev(g, level), -1.0, x->d, b->d, r->d, 1);
    HMG_END
}

int hmg_constraint(hmg_grid *g, int level, hmg_vec *x)
{
    HMG_TRY
    check_vec(g, level, x, "x");
    launch_mask(g->ctx->L, lev(g, level), g->md, x->d, 0);
    HMG_END
}

int hmg_interface_sum(hmg_grid *g, int level, hmg_vec *x)
{
    HMG_TRY
    check_vec(g, level, x, "x");
    interface_sum(g, lev(g, level), x->d);
    HMG_END
}

int hmg_zero_duplicates(hmg_grid *g, int level, hmg_vec *x)
{
    HMG_TRY
    check_vec(g, level, x, "x");
    launch_mask(g->ctx->L, lev(g, level), g->md, x->d, 1);
    HMG_END
}

int hmg_restrict(hmg_grid *g, int level_fine, hmg_vec *r_fine, hmg_vec *b_coarse)
{
    HMG_TRY
    need(g && level_fine >= 2, "restriction needs level_fine >= 2");
    check_vec(g, level_fine, r_fine, "r_fine");
    check_vec(g, level_fine - 1, b_coarse, "b_coarse");
    restrict_level(g, level_fine, r_fine->d, b_coarse->d);
    HMG_END
}

int hmg_prolong_add(hmg_grid *g, int level_fine, hmg_vec *x_coarse, hmg_vec *x_fine)
{
    HMG_TRY
    need(g && level_fine >= 2, "prolongation needs level_fine >= 2");
    check_vec(g, level_fine, x_fine, "x_fine");
    check_vec(g, level_fine - 1, x_coarse, "x_coarse");
    launch_prolong_add(g->ctx->L, lev(g, level_fine), lev(g, level_fine - 1), g->md.ncells, x_coarse->d, x_fine->d);
    HMG_END
}

int hmg_gather_base(hmg_grid *g, hmg_vec *v1, double *host_u)
{
    HMG_TRY
    need(g && host_u, "null argument");
    check_vec(g, 1, v1, "v1");
    DevBuf<double> u;
    u.alloc((size_t)g->md.nnodes);
    launch_gather_base(g->ctx->L, g->md, lev(g, 1).ld, v1->d, u.p);
    HIPCHK(hipMemcpyAsync(host_u, u.p, sizeof(double) * g->md.nnodes, hipMemcpyDeviceToHost, g->ctx->stream));
    HIPCHK(hipStreamSynchronize(g->ctx->stream));
    HMG_END
}

int hmg_scatter_base(hmg_grid *g, const double *host_u, hmg_vec *v1)
{
    HMG_TRY
    need(g && host_u, "null argument");
    check_vec(g, 1, v1, "v1");
    DevBuf<double> u;
    u.alloc((size_t)g->md.nnodes);
    HIPCHK(hipMemcpyAsync(u.p, host_u, sizeof(double) * g->md.nnodes, hipMemcpyHostToDevice, g->ctx->stream));
    launch_scatter_base(g->ctx->L, g->md, lev(g, 1).ld, u.p, v1->d);
    HIPCHK(hipStreamSynchronize(g->ctx->stream));
    HMG_END
}

// ---- driver right-hand sides -------------------------------------------------------
int hmg_rhs_axi_grad(hmg_grid *g, const double *xi, hmg_vec *b)
{
    HMG_TRY
    need(g && g->has_op && xi && b, "null argument or operator not set");
    check_vec(g, b->level, b, "b");
    const MeshTables &M = g->cur();
    const int dim = g->dim;
    // P = -detJ * (Jinv' * (sigma .* xi))   (ref: ...homogenized_coefficients.jl:468)
    std::vector<double> pv((size_t)M.ncells * 3, 0.0);
    for (int64_t c = 0; c < M.ncells; ++c) {
        const double *Ji = &M.jinv[(size_t)c * dim * dim];
        const double *sg = &g->sigma[(size_t)c * dim];
        for (int a = 0; a < dim; ++a) {
            double s = 0.0;
            for (int k = 0; k < dim; ++k) s += Ji[k + dim * a] * (sg[k] * xi[k]);
            pv[(size_t)c * 3 + a] = -M.detj[c] * s;
        }
    }
    DevBuf<double> d;
    d.upload(pv, g->ctx->stream);
    launch_rhs_dphi(g->ctx->L, lev(g, b->level), M.ncells, d.p, b->d);
    HIPCHK(hipStreamSynchronize(g->ctx->stream));
    HMG_END
}

int hmg_local_rhs(hmg_grid *g, hmg_vec *b)
{
    HMG_TRY
    need(g && b, "null argument");
    check_vec(g, b->level, b, "b");
    const MeshTables &M = g->cur();
    LevelDev lv = lev(g, b->level);
    const LevelTables &T = g->lt[b->level - 1];
    // b[:, e] = (int phi over the refined reference simplex) * |det J_e|   (ref: src/implicit_fine_grid.jl:391-409);
    // evaluated by the d.p kernel of the other right-hand sides with d = (load, 0, 0), p = (|det J|, 0, 0)
    std::vector<double> tab((size_t)T.nf * 3, 0.0), pv((size_t)M.ncells * 3, 0.0);
    for (int t = 0; t < T.nf; ++t) tab[(size_t)t * 3] = T.load[t];
    for (int64_t c = 0; c < M.ncells; ++c) pv[(size_t)c * 3] = M.detj[c];
    DevBuf<double> dt, dp;
    dt.upload(tab, g->ctx->stream);
    dp.upload(pv, g->ctx->stream);
    lv.dphi = dt.p;
    launch_rhs_dphi(g->ctx->L, lv, M.ncells, dp.p, b->d);
    HIPCHK(hipStreamSynchronize(g->ctx->stream));
    HMG_END
}

int hmg_integrate(hmg_grid *g, int mode, hmg_vec *v, hmg_vec *vprev, int64_t ncells_subset, const double *xi, double *out)
{
    HMG_TRY
    need(g && g->has_op && v && out, "null argument or operator not set");
    // (partitioned grid: the subset counts LOCAL cells and the result is this rank's share; the host sums over ranks)
    check_vec(g, v->level, v, "v");
    need(ncells_subset >= 0 && ncells_subset <= g->md.ncells, "subset out of range");
    const MeshTables &M = g->cur();
    const int dim = g->dim;
    if (mode == 2) {   // integrate_area: sum(mass) * sum |J|   (ref: ...:673-689)
        const double m_total = dim == 3 ? 1.0 / 6.0 : 0.5;
        double area = 0.0;
        for (int64_t c = 0; c < ncells_subset; ++c) area += m_total * M.detj[c];
        *out = area;
        return 0;
    }
    need(mode == 0 || mode == 1, "mode must be 0 (first term), 1 (terms) or 2 (area)");
    if (ncells_subset == 0) {
        *out = 0.0;
        return 0;
    }
    need(vprev != nullptr, mode == 0 ? "mode 0 needs the right-hand side rhs_a.xi.grad(v) (hmg_rhs_axi_grad) as second vector"
                                     : "mode 1 needs the previous iterate as second vector");
    check_vec(g, v->level, vprev, "second vector");
    need(v->d != vprev->d, "the two vectors must not alias");
    (void)xi;   // (mode 0: the direction already sits in the right-hand side the caller passes)
    set_slab(g, lev(g, v->level));
    launch_integrate(g->ctx->L, lev(g, v->level), g->md, mode, ncells_subset, v->d, vprev->d, S_TMP);
    *out = read_scalar(g->ctx, S_TMP);
    HMG_END
}

int hmg_next_rhs(hmg_grid *g, hmg_vec *x, hmg_vec *b)
{
    HMG_TRY
    need(g && g->has_op && x && b, "null argument or operator not set");
    check_vec(g, x->level, b, "b");
    check_vec(g, x->level, x, "x");
    need(x->d != b->d, "x and b must not alias");
    // b = lambda*|J|*M*x  (ref: ...homogenized_coefficients.jl:695-713)
    g->ctx->L.apply_mass_only = 1;
    set_slab(g, lev(g, x->level));
    try {
        launch_apply(g->ctx->L, lev(g, x->level), g->md, 1.0, g->lambda, x->d, nullptr, b->d, 0);
    } catch (...) {
        g->ctx->L.apply_mass_only = 0;
        throw;
    }
    g->ctx->L.apply_mass_only = 0;
    HMG_END
}

// ---- fused fast path ----------------------------------------------------------------------------
int hmg_smooth(hmg_grid *g, int level, int steps, hmg_vec *x, hmg_vec *b, hmg_vec *r, hmg_vec *p, hmg_vec *Ap)
{
    HMG_TRY
    need(g && g->has_op, "operator not set");
    check_vec(g, level, x, "x");
    check_vec(g, level, b, "b");
    check_vec(g, level, r, "r");
    check_vec(g, level, p, "p");
    check_vec(g, level, Ap, "Ap");
    smooth(g, level, steps, x, b, r, p, Ap);
    HMG_END
}

// Where the five vectors of a level lie in HBM relative to each other moves the passes that stream five or six of them
// at once by up to 8 % (tools/dev/placement_pick.py: three smoothing steps at config 3 take 61.2 ... 66.5 ms depending on
// which of ten equal allocations plays which role; moving a vector INSIDE its allocation changes nothing, no rule in the
// virtual addresses, another process on the same box draws another table -- the physical pages decide).  So the choice is
// made by measurement, FFTW style: time this level's share of a V-cycle (the down and the up half, src/multigrid.jl:100-115,
// with the pointer exchanges and folds the real cycle uses) for `trials` assignments of the 5 + `extra` blocks to the roles
// x, b, r, p, Ap and keep the fastest.
int hmg_level_tune_placement(hmg_grid *g, int level, int steps, hmg_vec **states, int extra, int trials, double *ms_out)
{
    HMG_TRY
    need(g && g->has_op && states, "null argument or operator not set");
    need(level >= 2 && level <= g->nlevels, "level must be 2..nlevels (level 1 has no smoother)");
    need(steps >= 1, "steps must be positive");
    need(extra >= 0 && extra <= 8, "extra must be 0..8");
    need(trials >= 1 && trials <= 512, "trials must be 1..512");
    hmg_vec **state = states + 5 * (level - 1);
    hmg_vec **below = states + 5 * (level - 2);
    for (int q = 0; q < 5; ++q) {
        check_vec(g, level, state[q], "states[] (level)");
        check_vec(g, level - 1, below[q], "states[] (level - 1)");
        need(state[q]->own && state[q]->bytes == state[0]->bytes && state[q]->alloc_cells == state[0]->alloc_cells,
             "the five vectors of the level must be hmg_vec_create'd vectors");
        for (int p = 0; p < q; ++p) need(state[p] != state[q] && state[p]->d != state[q]->d, "the five vectors must be distinct");
    }
    hmg_ctx *c = g->ctx;
    HIPCHK(hipSetDevice(c->device));
    const size_t bytes = state[0]->bytes;
    const int64_t n = vec_len(state[0]);
    std::vector<double *> blk;
    for (int q = 0; q < 5; ++q) blk.push_back(state[q]->d);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::vector<int> first = {0, 1, 2, 3, 4}, best = first;
    auto assign = [&](const std::vector<int> &a) {
        for (int q = 0; q < 5; ++q) state[q]->d = blk[(size_t)a[q]];
    };
    auto cleanup = [&](const std::vector<int> &keep) {      // the blocks no handle refers to go back
        assign(keep);
        (void)hipStreamSynchronize(c->stream);
        for (size_t i = 0; i < blk.size(); ++i)
            if (std::find(keep.begin(), keep.end(), (int)i) == keep.end()) (void)hipFree(blk[i]);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
    double t_first = 0.0, t_best = 0.0;
    try {
        // Everything that can fail on ONE rank only (the spare blocks, the events) comes before the first collective, and on a
        // partitioned grid the ranks agree on the outcome: run() below holds exchanges, and a rank that threw here alone would
        // leave the others blocked in them.
        std::string local_failure;
        try {
            for (int e = 0; e < extra; ++e) blk.push_back(vec_alloc(c, bytes));
            HIPCHK(hipEventCreate(&e0));
            HIPCHK(hipEventCreate(&e1));
        } catch (const std::exception &ex) {
            local_failure = ex.what();
        }
        if (has_exchange(g) && g->part && g->scalar_sum) {
            double failed = local_failure.empty() ? 0.0 : 1.0;
            double *d = c->L.scal + S_HOST;
            HIPCHK(hipMemcpyAsync(d, &failed, sizeof(double), hipMemcpyHostToDevice, c->stream));
            scalar_sum(g, S_HOST, 1);
            HIPCHK(hipMemcpyAsync(&failed, d, sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            if (failed > 0.0 && local_failure.empty())
                local_failure = "placement tuning: another rank could not set it up (nothing was changed on this rank)";
        }
        if (!local_failure.empty()) throw std::runtime_error(local_failure);
        const int nb = (int)blk.size();
        auto run = [&](const std::vector<int> &a) {
            assign(a);
            launch_fill(c->L, state[0]->d, n, 0.5);       // (r, p, Ap are written before they are read)
            launch_fill(c->L, state[1]->d, n, 1.0);
            HIPCHK(hipEventRecord(e0, c->stream));
            // as hmg_vcycle runs them on its top level (each half exchanges the r and p pointers: restored by the pair; the
            // coarse x stays the zero it is)
            vcycle_down(g, level, steps, states, /*inside=*/true, /*x_zero=*/false, /*steps_next=*/2);
            vcycle_up(g, level, steps, states, c->lean_post ? 1 : 0);
            HIPCHK(hipEventRecord(e1, c->stream));
            HIPCHK(hipEventSynchronize(e1));
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, e0, e1));
            return (double)ms;
        };
        (void)run(first);                                  // first touch of every kernel and table
        uint64_t rng = 0x9E3779B97F4A7C15ull;              // (the same sequence on every rank: the halves may hold collectives)
        auto next = [&]() {
            rng ^= rng << 13;
            rng ^= rng >> 7;
            rng ^= rng << 17;
            return rng;
        };
        t_best = 1e300;
        for (int t = 0; t < trials; ++t) {
            std::vector<int> a(nb);
            for (int i = 0; i < nb; ++i) a[i] = i;
            if (t > 0)
                for (int i = 0; i < 5; ++i) std::swap(a[i], a[i + (int)(next() % (uint64_t)(nb - i))]);
            a.resize(5);
            const double ms = run(a);
            if (t == 0) t_first = ms;
            if (ms < t_best) {
                t_best = ms;
                best = a;
            }
        }
        {   // a second look at both (always: every rank must issue the same number of calls): the handles keep what they
            // had unless the gain is real
            const double again_first = run(first), again_best = run(best);
            t_first = std::min(t_first, again_first);
            t_best = std::max(t_best, again_best);
            if (!(t_best < 0.995 * t_first)) {
                best = first;
                t_best = t_first;
            }
        }
    } catch (...) {
        cleanup(first);
        throw;
    }
    cleanup(best);
    for (int q = 0; q < 5; ++q) {
        HIPCHK(hipMemsetAsync(state[q]->d, 0, bytes, c->stream));
        HIPCHK(hipMemsetAsync(below[q]->d, 0, sizeof(double) * (size_t)vec_len(below[q]), c->stream));
    }
    if (ms_out) {
        ms_out[0] = t_first;
        ms_out[1] = t_best;
    }
    HMG_END
}

int hmg_coarse_setup(hmg_grid *g)
{
    HMG_TRY
    need(g != nullptr, "null grid");
    coarse_setup(g);
    HMG_END
}

int hmg_coarse_solve(hmg_grid *g, hmg_vec *b1, hmg_vec *x1)
{
    HMG_TRY
    need(g && g->has_op, "operator not set");
    check_vec(g, 1, b1, "b1");
    check_vec(g, 1, x1, "x1");
    coarse_solve(g, b1, x1);
    HMG_END
}

int hmg_coarse_last_iterations(const hmg_grid *g)
{
    if (!g) return -1;
    try {
        coarse_probe_wait(const_cast<hmg_grid *>(g));
    } catch (const std::exception &e) {
        last_error() = e.what();
        return -1;
    }
    return g->coarse_last_it;
}

int64_t hmg_coarse_misses(const hmg_grid *g) { return g ? g->coarse_misses : -1; }

int hmg_vcycle(hmg_grid *g, int top_level, int steps, int steps_coarse, hmg_vec **states)
{
    HMG_TRY
    need(g && g->has_op && states, "null argument or operator not set");
    need(top_level >= 1 && top_level <= g->nlevels, "top_level out of range");
    for (int l = 1; l <= top_level; ++l)
        for (int q = 0; q < 5; ++q) check_vec(g, l, states[5 * (l - 1) + q], "states[]");
    vcycle(g, top_level, steps, steps_coarse, states);
    HMG_END
}

static void check_two_levels(hmg_grid *g, int level, hmg_vec **states)
{
    need(g && g->has_op && states, "null argument or operator not set");
    need(level >= 2 && level <= g->nlevels, "level out of range (2..nlevels)");
    for (int l = level - 1; l <= level; ++l)
        for (int q = 0; q < 5; ++q) check_vec(g, l, states[5 * (l - 1) + q], "states[]");
}

int hmg_vcycle_down(hmg_grid *g, int level, int steps, hmg_vec **states)
{
    HMG_TRY
    check_two_levels(g, level, states);
    vcycle_down(g, level, steps, states);
    HMG_END
}

int hmg_vcycle_up(hmg_grid *g, int level, int steps, hmg_vec **states)
{
    HMG_TRY
    check_two_levels(g, level, states);
    vcycle_up(g, level, steps, states, g->ctx->lean_post ? 1 : 0);   // as the finest level of hmg_vcycle
    HMG_END
}

// ---- multi-GPU hooks ------------------------------------------------------------------------------
static void set_cut_kind(hmg_grid *g, int k, int64_t nglobal, int64_t n, const int64_t *gid, const int32_t *cell_lid,
                         const int32_t *seg, const int64_t *sidx)
{
    CutKind &c = g->cut[k];
    c.nglobal = nglobal;
    c.nentries = n;
    c.gid.assign(gid, gid + n);
    c.seg.clear();
    c.sidx.clear();
    if (seg && sidx) {
        c.seg.assign(seg, seg + n);
        c.sidx.assign(sidx, sidx + n);
    }
    std::vector<int32_t> hc(cell_lid, cell_lid + n);
    std::vector<uint8_t> first(n, 0);
    std::unordered_map<int64_t, int> seen;
    for (int64_t i = 0; i < n; ++i) {
        need(c.gid[i] >= 0 && c.gid[i] < nglobal, "cut id out of range");
        need((hc[i] >> 3) >= 0 && (hc[i] >> 3) < g->md.ncells, "cut entry references a cell outside the grid");
        if (seen.emplace(c.gid[i], 1).second) first[i] = 1;
    }
    g->cutlv.clear();                            // buffer layouts are rebuilt at the next exchange
    g->cut_agreed_ready = false;                 // ... and the ranks agree on the size of the new cut at the next apply
    DryUploads dry_scope(!g->ctx, &g->upload_hash);
    c.cell_lid.upload(hc, g->ctx ? g->ctx->stream : nullptr);
    c.first.upload(first, g->ctx ? g->ctx->stream : nullptr);
}

int hmg_grid_set_cut(hmg_grid *g, int64_t ngf, int64_t nge, int64_t ngn, int64_t nlf, const int64_t *face_gid,
                     const int32_t *face_cell_lid, int64_t nle, const int64_t *edge_gid, const int32_t *edge_cell_lid,
                     int64_t nln, const int64_t *node_gid, const int32_t *node_cell_lid)
{
    HMG_TRY
    need(g != nullptr, "null grid");
    // (a host-made cut list knows global ids only: all-reduce over the global cut buffer)
    g->sharers = false;
    set_cut_kind(g, 0, ngf, nlf, face_gid, face_cell_lid, nullptr, nullptr);
    set_cut_kind(g, 1, nge, nle, edge_gid, edge_cell_lid, nullptr, nullptr);
    set_cut_kind(g, 2, ngn, nln, node_gid, node_cell_lid, nullptr, nullptr);
    HMG_END
}

int hmg_grid_set_exchange(hmg_grid *g, hmg_exchange_fn exchange, hmg_exchange_fn scalar_sum_fn, void *user,
                          void *device_exchange_buf, int64_t exchange_buf_doubles)
{
    HMG_TRY
    need(g != nullptr, "null grid");
    g->exchange = exchange;
    g->scalar_sum = scalar_sum_fn;
    g->cut_agreed_ready = false;
    g->ex_user = user;
    g->ex_buf = (double *)device_exchange_buf;
    g->ex_cap = exchange_buf_doubles;
    HMG_END
}

int hmg_grid_set_exchange_async(hmg_grid *g, hmg_exchange_fn begin, int (*end)(void *user))
{
    HMG_TRY
    need(g != nullptr, "null grid");
    g->ex_begin = begin;
    g->ex_end = end;
    HMG_END
}

int hmg_grid_set_overlap(hmg_grid *g, int enabled)
{
    HMG_TRY
    need(g != nullptr, "null grid");
    g->overlap = enabled != 0;
    HMG_END
}

int hmg_comm_unique_id(void *out128)
{
    HMG_TRY
    need(out128 != nullptr, "null argument");
    ncclUniqueId id;
    nccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    static_assert(sizeof(id) == HMG_COMM_ID_BYTES, "ncclUniqueId size");
    std::memcpy(out128, &id, sizeof(id));
    HMG_END
}

int hmg_comm_init(hmg_ctx *ctx, int nranks, int rank, const void *unique_id128)
{
    HMG_TRY
    need(ctx && unique_id128, "null argument");
    need(nranks >= 1 && rank >= 0 && rank < nranks, "rank out of range");
    need(ctx->comm == nullptr, "this context already has a communicator");
    HIPCHK(hipSetDevice(ctx->device));
    ncclUniqueId id;
    std::memcpy(&id, unique_id128, sizeof(id));
    nccl_check(rccl().CommInitRank(&ctx->comm, nranks, id, rank), "ncclCommInitRank");
    ctx->comm_nranks = nranks;
    ctx->comm_rank = rank;
    HIPCHK(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&ctx->ev_packed, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ctx->ev_summed, hipEventDisableTiming));
    HMG_END
}

int hmg_comm_destroy(hmg_ctx *ctx)
{
    HMG_TRY
    need(ctx != nullptr, "null ctx");
    if (ctx->comm) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->comm_stream));
        nccl_check(rccl().CommDestroy(ctx->comm), "ncclCommDestroy");
        ctx->comm = nullptr;
        ctx->comm_nranks = 1;
        ctx->comm_rank = 0;
        // the second stream and its events go with the communicator (a later hmg_comm_init makes new ones); grids that
        // still point at the built-in exchange callbacks fail cleanly in comm_allreduce / comm_p2p from now on
        (void)hipEventDestroy(ctx->ev_packed);
        (void)hipEventDestroy(ctx->ev_summed);
        (void)hipStreamDestroy(ctx->comm_stream);
        ctx->ev_packed = ctx->ev_summed = nullptr;
        ctx->comm_stream = nullptr;
    }
    HMG_END
}

int hmg_comm_stats(hmg_ctx *ctx, int64_t *calls, int64_t *doubles)
{
    HMG_TRY
    need(ctx && calls && doubles, "null argument");
    *calls = ctx->comm_calls;
    *doubles = ctx->comm_doubles;
    HMG_END
}

int hmg_comm_sum_host(hmg_ctx *ctx, double *vals, int count)
{
    HMG_TRY
    need(ctx && vals, "null argument");
    need(ctx->comm != nullptr, "hmg_comm_init must be called first");
    need(count >= 1 && count <= S_COUNT - S_HOST, "count must be 1..4");
    double *d = ctx->L.scal + S_HOST;     // (the last slots of the scalar bank are not used by the kernels)
    HIPCHK(hipMemcpyAsync(d, vals, sizeof(double) * count, hipMemcpyHostToDevice, ctx->stream));
    comm_allreduce(ctx, d, count, ctx->stream);
    HIPCHK(hipMemcpyAsync(vals, d, sizeof(double) * count, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HMG_END
}

int64_t hmg_grid_cut_stage_doubles(const hmg_grid *cg);

int hmg_grid_use_comm(hmg_grid *g)
{
    HMG_TRY
    need(g && g->ctx, "null grid or host-only grid");
    need(g->ctx->comm != nullptr, "hmg_comm_init must be called on the grid's context first");
    need(g->part != nullptr, "not a partitioned grid (hmg_grid_create_partition)");
    need(g->ctx->comm_rehearsal || (g->part->nranks == g->ctx->comm_nranks && g->part->rank == g->ctx->comm_rank),
         "the grid's partition and the context's communicator disagree on rank / size");
    // exchange among the sharers of each cut entity (grouped ncclSend / ncclRecv) unless HMG_EXCHANGE=allreduce asks for
    // round 2's single all-reduce over the global cut buffer
    const char *mode = std::getenv("HMG_EXCHANGE");
    g->sharers = !(mode && std::string(mode) == "allreduce");
    g->cutlv.clear();
    const int64_t cap = std::max<int64_t>(hmg_grid_cut_buffer_doubles(g, 0), 1);
    g->own_exbuf.alloc((size_t)cap);
    g->ex_buf = g->own_exbuf.p;
    g->ex_cap = cap;
    const int64_t scap = std::max<int64_t>(hmg_grid_cut_stage_doubles(g), 1);
    g->own_stage.alloc((size_t)scap);
    g->stage = g->own_stage.p;
    g->stage_cap = scap;
    g->ex_user = g;
    g->exchange = comm_exchange;             // (the level-1 gather stays an all-reduce of the global nodal vector)
    g->scalar_sum = comm_exchange;           // (the same in-place sum, on the scalar bank)
    g->cut_agreed_ready = false;
    g->ex_begin = comm_exchange_begin;
    g->ex_end = comm_exchange_end;
    g->p2p = comm_p2p_sync;
    g->p2p_begin = comm_p2p_begin;
    HMG_END
}

void *hmg_ctx_stream(hmg_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int64_t hmg_grid_cut_buffer_doubles(const hmg_grid *cg, int level)
{
    hmg_grid *g = const_cast<hmg_grid *>(cg);
    if (!g || level < 0 || level > g->nlevels) return -1;
    try {
        if (level == 0) {   // required capacity: max over levels and the coarse gather
            int64_t n = g->part ? g->part->global.nnodes : 0;
            for (int l = 0; l < g->nlevels; ++l) n = std::max(n, cut_doubles(g, g->ld[l]));
            return n;
        }
        return cut_doubles(g, g->ld[level - 1]);
    } catch (const std::exception &e) {
        last_error() = e.what();
        return -1;
    }
}

int64_t hmg_grid_cut_stage_doubles(const hmg_grid *cg)
{
    hmg_grid *g = const_cast<hmg_grid *>(cg);
    if (!g) return -1;
    try {
        int64_t n = 0;
        for (int l = 0; l < g->nlevels; ++l) n = std::max(n, cut_level(g, g->ld[l]).nstage);
        return n;
    } catch (const std::exception &e) {
        last_error() = e.what();
        return -1;
    }
}

int hmg_grid_set_exchange_p2p(hmg_grid *g, int enabled, hmg_p2p_fn p2p, hmg_p2p_fn p2p_begin, void *device_stage_buf,
                              int64_t stage_buf_doubles)
{
    HMG_TRY
    need(g != nullptr, "null grid");
    need(!enabled || g->part != nullptr, "the sharers-only exchange needs a grid made by hmg_grid_create_partition");
    g->sharers = enabled != 0;
    g->cutlv.clear();
    g->p2p = p2p;
    g->p2p_begin = p2p_begin;
    g->stage = (double *)device_stage_buf;
    g->stage_cap = stage_buf_doubles;
    HMG_END
}

int hmg_grid_exchange_messages(const hmg_grid *cg, int level, int64_t *out, int64_t cap, int64_t *count)
{
    HMG_TRY
    hmg_grid *g = const_cast<hmg_grid *>(cg);
    need(g && count, "null argument");
    need(level >= 1 && level <= g->nlevels, "level out of range");
    const CutLevel &C = cut_level(g, g->ld[level - 1]);
    *count = (int64_t)C.ops.size();
    if (out) {
        need(cap >= *count, "output buffer too small");
        std::copy(C.ops.begin(), C.ops.end(), out);
    }
    HMG_END
}

}  // extern "C"
