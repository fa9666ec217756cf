// Host-side data model of the MI355X implicit-grid multigrid library.
//
// Nothing in here is a translation of the reference's Julia containers: the reference keeps
// per-level CSC matrices and ascending node-id lists (src/build_local_operators.jl,
// src/multilevel_reference.jl); this library keeps, per level, (1) an *entity-major* storage
// order of a cell's DOFs so that every shared face / edge of two cells is one contiguous,
// identically ordered run in both columns, and (2) a 15-point (3D) / 7-point (2D) lattice stencil
// in "class" form (one coefficient row per entity of the reference simplex).
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include <thread>
#include <vector>

namespace hmg {

// Host threads for the setup work (table builders, mesh synthesis): HMG_SETUP_THREADS, else the hardware's, at most 16.
int setup_threads();

// f(begin, end) over [0, n) in contiguous pieces, one per thread.
template <class F>
void parallel_for(int64_t n, F f)
{
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(setup_threads(), n / 4096));
    if (T == 1) {
        f((int64_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([&, t] { f(n * t / T, n * (t + 1) / T); });
    for (auto &x : th) x.join();
}

// Sort with a strict weak order: the pieces are sorted by one thread each, then merged pairwise (in parallel per
// round).  Equal elements keep no particular order -- callers use total orders.
template <class T, class Cmp>
void parallel_sort(std::vector<T> &v, Cmp cmp)
{
    const int64_t n = (int64_t)v.size();
    int P = 1;
    while (P * 2 <= setup_threads() && n / (P * 2) >= 32768) P *= 2;
    if (P == 1) {
        std::sort(v.begin(), v.end(), cmp);
        return;
    }
    std::vector<int64_t> cut(P + 1);
    for (int p = 0; p <= P; ++p) cut[p] = n * p / P;
    {
        std::vector<std::thread> th;
        for (int p = 0; p < P; ++p) th.emplace_back([&, p] { std::sort(v.begin() + cut[p], v.begin() + cut[p + 1], cmp); });
        for (auto &x : th) x.join();
    }
    for (int w = 1; w < P; w *= 2) {
        std::vector<std::thread> th;
        for (int p = 0; p + w < P; p += 2 * w)
            th.emplace_back([&, p, w] {
                std::inplace_merge(v.begin() + cut[p], v.begin() + cut[p + w], v.begin() + cut[std::min(p + 2 * w, P)], cmp);
            });
        for (auto &x : th) x.join();
    }
}

// ---------------------------------------------------------------------------------------------
// Per-level tables of the refined reference simplex.
// ---------------------------------------------------------------------------------------------
struct LevelTables {
    int dim = 0;
    int level = 0;       // 1-based, as in the reference
    int m = 0;           // lattice intervals per edge: 2^(level-1)
    int nf = 0;          // DOFs per coarse cell (nodes of the refined reference simplex)
    int ld = 0;          // column stride of a level vector in doubles
    int ncorner = 0, nedge = 0, nface = 0;
    int nei = 0, nfi = 0, nint = 0;             // interior DOFs per edge / face / cell
    int off_edge = 0, off_face = 0, off_int = 0; // storage offsets of the entity segments

    // hierarchical (reference / API) node id <-> storage slot
    std::vector<int32_t> hier2slot, slot2hier;
    // lattice coordinates (i,j,k) of every storage slot (k = 0 in 2D)
    std::vector<int32_t> slot_ijk;
    // entity class of every slot: 0 interior, 1..nface faces, then edges, then corners
    std::vector<uint8_t> slot_cls;
    // packed per-slot stencil addressing word (see pack_meta)
    std::vector<uint64_t> meta;
    int lds_g0 = 0, lds_g1 = 0;   // LDS guard entries in front of / behind the lattice image
    // cell-interior sweep in LDS lattice order: every interior row with its two end (surface)
    // positions kept as idle lanes so that a half-wave reads contiguous LDS words.
    // sweep_meta: L | len<<16 | A<<32 | B<<48 ; sweep_slot: storage slot or 0xffff (idle)
    std::vector<uint64_t> sweep_meta;
    std::vector<uint16_t> sweep_slot;

    // stencil in class form: ctab[(cls*ndir + dir)*nterm + term]
    int ncls = 0, ndir = 0, nterm = 0;
    std::vector<double> ctab;

    // cells of the refined reference simplex (Kuhn sub-simplices), dim+1 hierarchical node ids each; the set of
    // cells equals refined_element(level) of the reference (src/multilevel_reference.jl:41-61), the order is the
    // lattice enumeration, not the red-refinement order (used for export only)
    std::vector<int32_t> ref_cells;

    // load[slot] = int phi_slot over the refined reference simplex (assemble_vector(fine, identity), src/assembly.jl:121-155)
    std::vector<double> load;

    // dphi[3*slot + a] = int d phi_slot / d x_a over the refined reference simplex
    // (ref: src/examples/homogenized_coefficients.jl:407-442, partial_derivatives_functionals)
    std::vector<double> dphi;

    // transfer operators between level-1 (coarse) and this level (fine); empty on level 1.
    // prolongation: fine slot <- 0.5*coarse[pa] + 0.5*coarse[pb] (pa == pb: identity row);
    // pa is the parent with the smaller hierarchical id (the reference's CSC column order).
    std::vector<int32_t> par_a, par_b;
    // restriction (gather form): coarse slot c sums fine slots ridx[rptr[c]..rptr[c+1]) in
    // ascending fine hierarchical id; the first entry is the identity row (weight 1), the rest 0.5.
    std::vector<int32_t> rptr, ridx;
};

inline uint64_t pack_meta(uint32_t L, uint32_t len, uint32_t cls, uint32_t A, uint32_t B)
{
    return (uint64_t)L | ((uint64_t)len << 16) | ((uint64_t)cls << 24) | ((uint64_t)A << 32) |
           ((uint64_t)B << 48);
}

// Builds tables for levels 1..nlevels of the reference simplex of dimension dim (2 or 3).
// Numbering contract (normative, from the reference): src/multilevel_reference.jl:9-13,41-61,
// src/tet/refine.jl:16-21, src/tri/refine.jl:21-25, src/sparse_graph.jl:20-48.
std::vector<LevelTables> build_level_tables(int dim, int nlevels);

// Stencil direction list: offsets (di,dj,dk); entry 0 is the node itself.
int stencil_dirs(int dim, const int (**dirs)[3]);

// ---------------------------------------------------------------------------------------------
// Base-mesh tables.
// ---------------------------------------------------------------------------------------------
struct MeshTables {
    int dim = 0;
    int64_t nnodes = 0, ncells = 0;
    std::vector<double> coords;    // dim * nnodes (node-major)
    std::vector<int32_t> cells;    // (dim+1) * ncells, 0-based, sorted per cell

    // shared faces (3D): exactly two copies each. Packed: cellA, cellB, (lfA | lfB<<4)
    std::vector<int32_t> face_pairs;   // 3 ints per shared face
    // shared edges / nodes in CSR form; entry = cell * 8 + local id
    std::vector<int32_t> edge_ptr, edge_ent;
    std::vector<int32_t> node_ptr, node_ent;
    // every base node: first listed copy (cell*8 + local node), for the level-1 gather
    std::vector<int32_t> node_first;
    // every base node: all its copies (cell*8 + local node, ascending cell), CSR -- row-wise assembly of the level-1 matrix
    std::vector<int32_t> node_all_ptr, node_all_ent;
    // per cell: Dirichlet entity bitmask and "not the first copy" bitmask; bit = cls-1
    std::vector<uint16_t> dmask, dupmask;
    // per cell, 16 bytes: number of copies of each of the cell's entities (same bit order as the masks)
    std::vector<uint8_t> mult;
    // boundary (Dirichlet) base nodes flag
    std::vector<uint8_t> node_on_boundary;

    // partitioned grids only: shared edge / node groups whose entity is cut by the partition are listed
    // first (ncut_*_groups of them); cells_cut = local cells that own a copy of any cut entity,
    // cells_inner = the others.  Used to overlap the exchange with the work that does not feed it.
    int64_t ncut_edge_groups = 0, ncut_node_groups = 0;
    // ... and shared faces whose BOTH copies are local although the face counts as cut (only with a rehearsal's
    // `cut_owner`, see build_partition: in a real partition a cut face has one local copy and is no face pair at all)
    int64_t ncut_face_pairs = 0;
    std::vector<int32_t> cells_cut, cells_inner;

    // cell geometry: detJ and Jinv = inv(J') (column-major dim x dim)
    std::vector<double> detj, jinv;
};

// cells_1based: (dim+1) x ncells, 1-based, each column ascending (reference contract,
// src/implicit_fine_grid.jl:14).
void build_mesh_tables(int dim, int64_t nnodes, const double *coords, int64_t ncells,
                       const int64_t *cells_1based, MeshTables &out);

// Recomputes Dirichlet masks / boundary flags for the mesh restricted to the first
// `ncells_prefix` cells (domain shrink, ref: src/examples/homogenized_coefficients.jl:309-316),
// and rebuilds the shared-entity lists for that prefix.
void restrict_mesh_tables(const MeshTables &full, int64_t ncells_prefix, int64_t nnodes_prefix,
                          MeshTables &out);

// Per-cell operator coefficients: coef[cell*8 + t] = |J| * P_t for the unique entries of
// P = J^-1 diag(sigma) J^-T (3D: 11,12,13,22,23,33; 2D: 11,12,22), then |J| at t = nterm-1.
// ref: src/apply_local_operators.jl:101-118.
void build_cell_coefficients(const MeshTables &mesh, const double *sigma, std::vector<double> &coef);

// Coarse (level-1) operator lambda*M + K_sigma on the base mesh, interior rows/cols only, CSR.
// ref: src/examples/homogenized_coefficients.jl:358-402 (assemble_checkerboard), src/grid.jl:176-202.
struct CoarseMatrix {
    int64_t n = 0;                      // interior unknowns
    std::vector<int32_t> interior;      // interior node ids (ascending)
    std::vector<int32_t> node2int;      // base node -> interior index or -1
    std::vector<int32_t> rowptr, colidx;
    std::vector<double> val, diag;
};
void assemble_coarse_matrix(const MeshTables &mesh, const double *sigma, double lambda,
                            CoarseMatrix &out);

// ---------------------------------------------------------------------------------------------
// Partition by coarse-cell ownership (one rank per GPU).
// ---------------------------------------------------------------------------------------------
struct Partition {
    int rank = 0, nranks = 1;
    MeshTables global;                       // the whole base mesh (kept for the replicated coarse solve)
    std::vector<int32_t> cells_g;            // global id of every local cell (ascending)
    std::vector<int32_t> nodes_g;            // global id of every local node (ascending)
    std::vector<int32_t> owned_node;         // per local node: 1 if its globally first copy is in a local cell
    // entities shared between ranks: global cut id + local copy, per kind (0 faces, 1 edges, 2 nodes)
    int64_t nglobal[3] = {0, 0, 0};
    std::vector<int64_t> gid[3];
    std::vector<int32_t> cell_lid[3];
    // false: the analysis looked at this rank's cells and their one-cell halo only (the default).  The ids in gid[] then
    // number this rank's cut entities and mean nothing to other ranks: the all-reduce over a global cut buffer is not
    // available, the exchange runs among the sharers (segments below).  `global` then carries only what the replicated
    // level-1 system needs (cells, geometry, node -> cells, first copies, boundary nodes).
    bool global_ids = true;
    // The same cut entities grouped by WHO shares them (exchange among the sharers only): a segment is the set
    // of cut entities with one set of member ranks -- at octants: a quarter of a cut plane (2 ranks), half an axis line
    // (4 ranks), the centre node (8 ranks).  Only the segments this rank is a member of are listed, in an order every
    // rank derives alike (first appearance in the global entity order: faces, edges, nodes), so that two members post
    // their messages for each other in the same sequence.  Inside a segment: faces, then edges, then nodes, each in
    // global entity order.
    struct Segment {
        std::vector<int32_t> members;        // ascending ranks, this rank among them (a rehearsal's segment may hold it alone)
        int64_t count[3] = {0, 0, 0};        // entities per kind
    };
    std::vector<Segment> segs;
    std::vector<int32_t> seg_of[3];          // per local cut copy (parallel to gid / cell_lid): its segment ...
    std::vector<int64_t> seg_idx[3];         // ... and the entity's index among the segment's entities of that kind
};

// Splits `global` by owner[cell] and fills `local` (tables of this rank's cells, with Dirichlet mask,
// duplicate mask and multiplicities taken from the GLOBAL mesh) and `part`.
// cut_owner (optional, rehearsals on fewer GPUs than the partition is meant for): an entity counts as cut when its
// copies' cells differ in cut_owner[] instead of owner[] -- with owner = 0 everywhere and cut_owner = the octant of a
// cell, one rank walks the cut-first cell lists, the pack / unpack kernels and the exchange of an 8-rank partition while
// holding every copy itself (the sum over ranks is then the identity and the results equal the unpartitioned ones).
// halo_only: analyse this rank's cells plus every cell that shares a node with one of them instead of the whole mesh --
// all copies of every entity of a local cell live there, which is all that masks, multiplicities, first copies and the
// sharer sets need; the global pass shrinks to what the replicated level-1 matrix needs (no entity lists are sorted for
// it).  Same tables for the local cells, same segments in the same order; no global cut ids (Partition::global_ids).
void build_partition(int dim, int64_t nnodes, const double *coords, int64_t ncells, const int64_t *cells_1based,
                     const int32_t *owner, int rank, int nranks, MeshTables &local, Partition &part,
                     const int32_t *cut_owner = nullptr, bool halo_only = false);

std::string &last_error();

}  // namespace hmg
