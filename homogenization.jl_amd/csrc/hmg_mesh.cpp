// Base-mesh tables: shared-entity lists for the interface sum, Dirichlet / duplicate masks,
// cell geometry and operator coefficients, coarse-level matrix.
//
// Behavioural contract (what the lists contain and in which order copies are summed) follows
// src/interface.jl:65-117,124-197,207-284 and src/implicit_fine_grid.jl:94-139,209-386 of the
// reference; the data structures are this library's own: faces as explicit pairs, per-cell entity
// bitmasks instead of (element, local id) value lists.
#include "hmg_host.hpp"

#include <algorithm>
#include <map>
#include <array>
#include <cmath>
#include <cstdlib>
#include <stdexcept>
#include <unordered_set>

namespace hmg {

int setup_threads()
{
    static const int T = [] {
        int t = 0;
        if (const char *e = std::getenv("HMG_SETUP_THREADS")) t = std::atoi(e);
        if (t <= 0) t = (int)std::thread::hardware_concurrency();
        return std::max(1, std::min(t, 16));
    }();
    return T;
}

namespace {

struct Ent {
    std::array<int32_t, 3> key;
    int32_t cell;
    int32_t lid;
};

inline bool ent_less(const Ent &a, const Ent &b)
{
    if (a.key != b.key) return a.key < b.key;
    return a.cell < b.cell;   // element-major listing + stable sort in the reference
}

const int TET_FACES[4][3] = {{0, 1, 2}, {0, 1, 3}, {0, 2, 3}, {1, 2, 3}};
const int TET_EDGES[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
const int TRI_EDGES[3][2] = {{0, 1}, {0, 2}, {1, 2}};

std::vector<Ent> list_entities(const MeshTables &M, int kind /*0 node,1 edge,2 face*/)
{
    const int N = M.dim + 1;
    const int per = kind == 0 ? N : kind == 1 ? (M.dim == 3 ? 6 : 3) : 4;
    std::vector<Ent> v((size_t)M.ncells * per);
    parallel_for(M.ncells, [&](int64_t c0, int64_t c1) {
        for (int64_t c = c0; c < c1; ++c) {
            const int32_t *el = &M.cells[c * N];
            for (int l = 0; l < per; ++l) {
                Ent e;
                e.cell = (int32_t)c;
                e.lid = l;
                if (kind == 0)
                    e.key = {el[l], -1, -1};
                else if (kind == 1) {
                    const int *t = M.dim == 3 ? TET_EDGES[l] : TRI_EDGES[l];
                    e.key = {el[t[0]], el[t[1]], -1};
                } else
                    e.key = {el[TET_FACES[l][0]], el[TET_FACES[l][1]], el[TET_FACES[l][2]]};
                v[(size_t)c * per + l] = e;
            }
        }
    });
    // ((key, cell) is a total order up to the local id, and one cell lists an entity once)
    parallel_sort(v, ent_less);
    return v;
}

template <class F>
void for_groups(const std::vector<Ent> &v, F f)
{
    size_t i = 0;
    while (i < v.size()) {
        size_t j = i + 1;
        while (j < v.size() && v[j].key == v[i].key) ++j;
        f(i, j);
        i = j;
    }
}

struct Geo {
    double J[3][3], Jinv[3][3], det;   // Jinv = inv(J')
};

Geo cell_geo(const MeshTables &M, int64_t c)
{
    const int dim = M.dim, N = dim + 1;
    const int32_t *el = &M.cells[c * N];
    Geo g{};
    const double *p0 = &M.coords[(size_t)el[0] * dim];
    for (int col = 0; col < dim; ++col) {
        const double *p = &M.coords[(size_t)el[col + 1] * dim];
        for (int a = 0; a < dim; ++a) g.J[a][col] = p[a] - p0[a];
    }
    double inv[3][3] = {{0}};
    if (dim == 2) {
        double det = g.J[0][0] * g.J[1][1] - g.J[0][1] * g.J[1][0];
        inv[0][0] = g.J[1][1] / det;
        inv[0][1] = -g.J[0][1] / det;
        inv[1][0] = -g.J[1][0] / det;
        inv[1][1] = g.J[0][0] / det;
        g.det = std::fabs(det);
    } else {
        const double(*J)[3] = g.J;
        double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
        double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
        double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
        double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
        inv[0][0] = c00 / det;
        inv[1][0] = c01 / det;
        inv[2][0] = c02 / det;
        inv[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
        inv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
        inv[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
        inv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
        inv[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
        inv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
        g.det = std::fabs(det);
    }
    for (int a = 0; a < dim; ++a)
        for (int b = 0; b < dim; ++b) g.Jinv[a][b] = inv[b][a];   // inv(J') = inv(J)'
    return g;
}

struct EntLists {
    std::vector<Ent> faces, edges, nodes;   // sorted by (key, cell)
};

void build_from_cells0(MeshTables &M, EntLists *keep = nullptr)
{
    const int dim = M.dim, N = dim + 1;
    const int nface = dim == 3 ? 4 : 0, nedge = dim == 3 ? 6 : 3;
    auto bit_face = [&](int f) { return (uint16_t)(1u << f); };
    auto bit_edge = [&](int e) { return (uint16_t)(1u << (nface + e)); };
    auto bit_node = [&](int n) { return (uint16_t)(1u << (nface + nedge + n)); };

    for (int64_t c = 0; c < M.ncells; ++c)
        for (int l = 0; l + 1 < N; ++l)
            if (!(M.cells[c * N + l] < M.cells[c * N + l + 1]))
                throw std::runtime_error("base mesh: every cell's node tuple must be strictly ascending");

    M.dmask.assign(M.ncells, 0);
    M.dupmask.assign(M.ncells, 0);
    M.mult.assign((size_t)M.ncells * 16, 1);
    auto set_mult = [&](int32_t cell, int bit, size_t n) {
        if (n > 255) throw std::runtime_error("base mesh: an entity is shared by more than 255 cells");
        M.mult[(size_t)cell * 16 + bit] = (uint8_t)n;
    };
    M.node_on_boundary.assign(M.nnodes, 0);
    M.face_pairs.clear();
    M.edge_ptr.assign(1, 0);
    M.edge_ent.clear();
    M.node_ptr.assign(1, 0);
    M.node_ent.clear();
    M.node_first.assign(M.nnodes, -1);

    std::unordered_set<uint64_t> bedges;
    auto ekey = [&](int32_t a, int32_t b) { return (uint64_t)a * (uint64_t)M.nnodes + (uint64_t)b; };

    if (dim == 3) {
        auto faces = list_entities(M, 2);
        for_groups(faces, [&](size_t i, size_t j) {
            if (j - i == 1) {   // boundary face (src/interface.jl:209-215)
                M.dmask[faces[i].cell] |= bit_face(faces[i].lid);
                const auto &k = faces[i].key;
                bedges.insert(ekey(k[0], k[1]));
                bedges.insert(ekey(k[0], k[2]));
                bedges.insert(ekey(k[1], k[2]));
            } else if (j - i == 2) {
                M.face_pairs.push_back(faces[i].cell);
                M.face_pairs.push_back(faces[i + 1].cell);
                M.face_pairs.push_back(faces[i].lid | (faces[i + 1].lid << 4));
                M.dupmask[faces[i + 1].cell] |= bit_face(faces[i + 1].lid);
                set_mult(faces[i].cell, faces[i].lid, 2);
                set_mult(faces[i + 1].cell, faces[i + 1].lid, 2);
            } else
                throw std::runtime_error("base mesh: a face is shared by more than two cells");
        });
        if (keep) keep->faces.swap(faces);
        // pairs in ascending order of their first cell: consecutive wavefronts of the face kernel then walk the level
        // vector front to back (the key order above follows the node numbering instead)
        {
            struct P3 {
                int32_t a, b, l;
            };
            std::vector<P3> pr(M.face_pairs.size() / 3);
            for (size_t q = 0; q < pr.size(); ++q) pr[q] = {M.face_pairs[3 * q], M.face_pairs[3 * q + 1], M.face_pairs[3 * q + 2]};
            parallel_sort(pr, [](const P3 &x, const P3 &y) { return x.a != y.a ? x.a < y.a : x.b != y.b ? x.b < y.b : x.l < y.l; });
            for (size_t q = 0; q < pr.size(); ++q) {
                M.face_pairs[3 * q] = pr[q].a;
                M.face_pairs[3 * q + 1] = pr[q].b;
                M.face_pairs[3 * q + 2] = pr[q].l;
            }
        }
    }
    {
        auto edges = list_entities(M, 1);
        if (dim == 2)
            for_groups(edges, [&](size_t i, size_t j) {
                if (j - i == 1) bedges.insert(ekey(edges[i].key[0], edges[i].key[1]));
                if (j - i > 2) throw std::runtime_error("base mesh: an edge is shared by more than two triangles");
            });
        for_groups(edges, [&](size_t i, size_t j) {
            bool bnd = bedges.count(ekey(edges[i].key[0], edges[i].key[1])) != 0;
            if (bnd) {
                M.node_on_boundary[edges[i].key[0]] = 1;
                M.node_on_boundary[edges[i].key[1]] = 1;
            }
            for (size_t q = i; q < j; ++q) {
                if (bnd) M.dmask[edges[q].cell] |= bit_edge(edges[q].lid);
                if (q > i) M.dupmask[edges[q].cell] |= bit_edge(edges[q].lid);
                set_mult(edges[q].cell, nface + edges[q].lid, j - i);
            }
            if (j - i >= 2) {   // singletons removed (src/interface.jl:99)
                for (size_t q = i; q < j; ++q) M.edge_ent.push_back(edges[q].cell * 8 + edges[q].lid);
                M.edge_ptr.push_back((int32_t)M.edge_ent.size());
            }
        });
        if (keep) keep->edges.swap(edges);
    }
    {
        auto nodes = list_entities(M, 0);
        M.node_all_ptr.assign(M.nnodes + 1, 0);
        M.node_all_ent.resize(nodes.size());
        for (size_t q = 0; q < nodes.size(); ++q) {
            M.node_all_ptr[nodes[q].key[0] + 1] += 1;
            M.node_all_ent[q] = nodes[q].cell * 8 + nodes[q].lid;
        }
        for (int64_t g = 0; g < M.nnodes; ++g) M.node_all_ptr[g + 1] += M.node_all_ptr[g];
        for_groups(nodes, [&](size_t i, size_t j) {
            int32_t g = nodes[i].key[0];
            M.node_first[g] = nodes[i].cell * 8 + nodes[i].lid;
            bool bnd = M.node_on_boundary[g] != 0;
            for (size_t q = i; q < j; ++q) {
                if (bnd) M.dmask[nodes[q].cell] |= bit_node(nodes[q].lid);
                if (q > i) M.dupmask[nodes[q].cell] |= bit_node(nodes[q].lid);
                set_mult(nodes[q].cell, nface + nedge + nodes[q].lid, j - i);
            }
            if (j - i >= 2) {
                for (size_t q = i; q < j; ++q) M.node_ent.push_back(nodes[q].cell * 8 + nodes[q].lid);
                M.node_ptr.push_back((int32_t)M.node_ent.size());
            }
        });
        if (keep) keep->nodes.swap(nodes);
    }

    M.detj.resize(M.ncells);
    M.jinv.resize((size_t)M.ncells * dim * dim);
    std::vector<char> bad(1, 0);
    parallel_for(M.ncells, [&](int64_t c0, int64_t c1) {
        for (int64_t c = c0; c < c1; ++c) {
            Geo g = cell_geo(M, c);
            if (!(g.det > 0.0)) bad[0] = 1;
            M.detj[c] = g.det;
            for (int b = 0; b < dim; ++b)
                for (int a = 0; a < dim; ++a) M.jinv[(size_t)c * dim * dim + a + dim * b] = g.Jinv[a][b];
        }
    });
    if (bad[0]) throw std::runtime_error("base mesh: degenerate cell");
}

}  // namespace

static void build_mesh_tables_keep(int dim, int64_t nnodes, const double *coords, int64_t ncells,
                                   const int64_t *cells_1based, MeshTables &M, EntLists *keep);

void build_mesh_tables(int dim, int64_t nnodes, const double *coords, int64_t ncells,
                       const int64_t *cells_1based, MeshTables &M)
{
    build_mesh_tables_keep(dim, nnodes, coords, ncells, cells_1based, M, nullptr);
}

static void build_mesh_tables_keep(int dim, int64_t nnodes, const double *coords, int64_t ncells,
                                   const int64_t *cells_1based, MeshTables &M, EntLists *keep)
{
    if (dim != 2 && dim != 3) throw std::runtime_error("dim must be 2 or 3");
    if (ncells <= 0 || nnodes <= 0) throw std::runtime_error("empty base mesh");
    if (ncells >= (int64_t(1) << 27)) throw std::runtime_error("too many cells for one device partition");
    const int N = dim + 1;
    M.dim = dim;
    M.nnodes = nnodes;
    M.ncells = ncells;
    M.coords.assign(coords, coords + (size_t)nnodes * dim);
    M.cells.resize((size_t)ncells * N);
    for (int64_t q = 0; q < ncells * N; ++q) {
        int64_t v = cells_1based[q] - 1;
        if (v < 0 || v >= nnodes) throw std::runtime_error("base mesh: node index out of range (cells are 1-based)");
        M.cells[q] = (int32_t)v;
    }
    build_from_cells0(M, keep);
}

void restrict_mesh_tables(const MeshTables &full, int64_t ncells_prefix, int64_t nnodes_prefix,
                          MeshTables &M)
{
    if (ncells_prefix <= 0 || ncells_prefix > full.ncells || nnodes_prefix <= 0 || nnodes_prefix > full.nnodes)
        throw std::runtime_error("shrink: prefix out of range");
    const int N = full.dim + 1;
    M.dim = full.dim;
    M.nnodes = nnodes_prefix;
    M.ncells = ncells_prefix;
    M.coords.assign(full.coords.begin(), full.coords.begin() + (size_t)nnodes_prefix * full.dim);
    M.cells.assign(full.cells.begin(), full.cells.begin() + (size_t)ncells_prefix * N);
    for (int32_t v : M.cells)
        if (v >= nnodes_prefix) throw std::runtime_error("shrink: a kept cell references a dropped node");
    build_from_cells0(M);
}

namespace {

// What the replicated level-1 system needs of the GLOBAL mesh, without sorting a single entity list: cells, geometry, the
// node -> cells map (cells are visited in ascending order: a counting sort), first copies, and the boundary nodes -- a face
// (3D) / edge (2D) lies on the boundary iff no other cell around its first node contains all its nodes.
void build_global_lean(int dim, int64_t nnodes, const double *coords, int64_t ncells, const int64_t *cells_1based, MeshTables &M)
{
    if (dim != 2 && dim != 3) throw std::runtime_error("dim must be 2 or 3");
    if (ncells <= 0 || nnodes <= 0) throw std::runtime_error("empty base mesh");
    const int N = dim + 1;
    M = MeshTables();
    M.dim = dim;
    M.nnodes = nnodes;
    M.ncells = ncells;
    M.coords.assign(coords, coords + (size_t)nnodes * dim);
    M.cells.resize((size_t)ncells * N);
    for (int64_t q = 0; q < ncells * N; ++q) {
        const int64_t v = cells_1based[q] - 1;
        if (v < 0 || v >= nnodes) throw std::runtime_error("base mesh: node index out of range (cells are 1-based)");
        M.cells[q] = (int32_t)v;
    }
    for (int64_t c = 0; c < ncells; ++c)
        for (int l = 0; l + 1 < N; ++l)
            if (!(M.cells[c * N + l] < M.cells[c * N + l + 1]))
                throw std::runtime_error("base mesh: every cell's node tuple must be strictly ascending");
    M.node_all_ptr.assign(nnodes + 1, 0);
    for (int64_t q = 0; q < ncells * N; ++q) M.node_all_ptr[M.cells[q] + 1] += 1;
    for (int64_t g = 0; g < nnodes; ++g) M.node_all_ptr[g + 1] += M.node_all_ptr[g];
    M.node_all_ent.resize((size_t)ncells * N);
    {
        std::vector<int32_t> fill(M.node_all_ptr.begin(), M.node_all_ptr.end() - 1);
        for (int64_t c = 0; c < ncells; ++c)
            for (int l = 0; l < N; ++l) M.node_all_ent[fill[M.cells[c * N + l]]++] = (int32_t)(c * 8 + l);
    }
    M.node_first.assign(nnodes, -1);
    for (int64_t g = 0; g < nnodes; ++g)
        if (M.node_all_ptr[g + 1] > M.node_all_ptr[g]) M.node_first[g] = M.node_all_ent[M.node_all_ptr[g]];
    // boundary nodes
    std::vector<uint8_t> flag(nnodes, 0);
    const int nsub = dim == 3 ? 4 : 3, sublen = dim;        // faces of a tetrahedron / edges of a triangle
    auto sub = [&](int f, int q) { return dim == 3 ? TET_FACES[f][q] : TRI_EDGES[f][q]; };
    parallel_for(ncells, [&](int64_t c0, int64_t c1) {
        for (int64_t c = c0; c < c1; ++c) {
            const int32_t *el = &M.cells[c * N];
            for (int f = 0; f < nsub; ++f) {
                int32_t nd[3];
                for (int q = 0; q < sublen; ++q) nd[q] = el[sub(f, q)];
                bool shared = false;
                for (int32_t e = M.node_all_ptr[nd[0]]; e < M.node_all_ptr[nd[0] + 1] && !shared; ++e) {
                    const int64_t o = M.node_all_ent[e] >> 3;
                    if (o == c) continue;
                    const int32_t *oe = &M.cells[o * N];
                    int hit = 0;
                    for (int q = 0; q < sublen; ++q)
                        for (int l = 0; l < N; ++l) hit += oe[l] == nd[q];
                    shared = hit == sublen;
                }
                if (!shared)
                    for (int q = 0; q < sublen; ++q) flag[nd[q]] = 1;     // (benign race: every writer stores 1)
            }
        }
    });
    M.node_on_boundary.assign(flag.begin(), flag.end());
    M.detj.resize(ncells);
    M.jinv.resize((size_t)ncells * dim * dim);
    std::vector<char> bad(1, 0);
    parallel_for(ncells, [&](int64_t c0, int64_t c1) {
        for (int64_t c = c0; c < c1; ++c) {
            Geo g = cell_geo(M, c);
            if (!(g.det > 0.0)) bad[0] = 1;
            M.detj[c] = g.det;
            for (int b = 0; b < dim; ++b)
                for (int a = 0; a < dim; ++a) M.jinv[(size_t)c * dim * dim + a + dim * b] = g.Jinv[a][b];
        }
    });
    if (bad[0]) throw std::runtime_error("base mesh: degenerate cell");
}

}  // namespace

void build_partition(int dim, int64_t nnodes, const double *coords, int64_t ncells, const int64_t *cells_1based,
                     const int32_t *owner, int rank, int nranks, MeshTables &local, Partition &part,
                     const int32_t *cut_owner, bool halo_only)
{
    const int32_t *cown = cut_owner ? cut_owner : owner;
    part.rank = rank;
    part.nranks = nranks;
    part.global_ids = !halo_only;
    MeshTables &G = part.global;
    EntLists glists;                         // the entity lists, sorted once, reused for the cut analysis below
    const int N = dim + 1;
    // halo mode: H = the local cells and every cell that shares a node with one (hcells: their global ids, ascending),
    // analysed with node ids compressed monotonically -- the order of entity keys, hence of segments, is the global one
    MeshTables H;
    std::vector<int32_t> hcells, hcell_of;   // H cell -> global cell; global cell -> H cell or -1
    if (halo_only) {
        build_global_lean(dim, nnodes, coords, ncells, cells_1based, G);
        std::vector<uint8_t> mark(nnodes, 0);
        for (int64_t c = 0; c < ncells; ++c)
            if (owner[c] == rank)
                for (int l = 0; l < N; ++l) mark[G.cells[c * N + l]] = 1;
        hcell_of.assign(ncells, -1);
        for (int64_t c = 0; c < ncells; ++c) {
            bool in = false;
            for (int l = 0; l < N; ++l) in = in || mark[G.cells[c * N + l]];
            if (in) {
                hcell_of[c] = (int32_t)hcells.size();
                hcells.push_back((int32_t)c);
            }
        }
        std::vector<int32_t> hnode(nnodes, -1);
        int32_t nh = 0;
        {
            std::vector<uint8_t> used(nnodes, 0);
            for (int32_t c : hcells)
                for (int l = 0; l < N; ++l) used[G.cells[(size_t)c * N + l]] = 1;
            for (int64_t g = 0; g < nnodes; ++g)
                if (used[g]) hnode[g] = nh++;
        }
        H.dim = dim;
        H.nnodes = nh;
        H.ncells = (int64_t)hcells.size();
        H.coords.resize((size_t)nh * dim);
        for (int64_t g = 0; g < nnodes; ++g)
            if (hnode[g] >= 0)
                for (int a = 0; a < dim; ++a) H.coords[(size_t)hnode[g] * dim + a] = coords[(size_t)g * dim + a];
        H.cells.resize(hcells.size() * N);
        for (size_t q = 0; q < hcells.size(); ++q)
            for (int l = 0; l < N; ++l) H.cells[q * N + l] = hnode[G.cells[(size_t)hcells[q] * N + l]];
        if (H.ncells >= (int64_t(1) << 27)) throw std::runtime_error("too many cells for one device partition");
        build_from_cells0(H, &glists);
    } else {
        build_mesh_tables_keep(dim, nnodes, coords, ncells, cells_1based, G, &glists);
    }
    // tables the analysis reads (A) and the map from their cell numbering to global cell ids
    const MeshTables &A = halo_only ? H : G;
    auto gcell = [&](int32_t acell) -> int32_t { return halo_only ? hcells[(size_t)acell] : acell; };
    const int nface = dim == 3 ? 4 : 0, nedge = dim == 3 ? 6 : 3;

    // local cells / nodes in ascending global order (keeps every cell tuple ascending)
    std::vector<int32_t> cell_l(ncells, -1), node_l(nnodes, -1);
    part.cells_g.clear();
    for (int64_t c = 0; c < ncells; ++c) {
        if (owner[c] < 0 || owner[c] >= nranks) throw std::runtime_error("partition: owner out of range");
        if (owner[c] == rank) {
            cell_l[c] = (int32_t)part.cells_g.size();
            part.cells_g.push_back((int32_t)c);
            for (int l = 0; l < N; ++l) node_l[G.cells[c * N + l]] = 0;
        }
    }
    if (part.cells_g.empty()) throw std::runtime_error("partition: this rank owns no cell");
    part.nodes_g.clear();
    for (int64_t g = 0; g < nnodes; ++g)
        if (node_l[g] == 0) {
            node_l[g] = (int32_t)part.nodes_g.size();
            part.nodes_g.push_back((int32_t)g);
        }
    std::vector<double> lc((size_t)part.nodes_g.size() * dim);
    for (size_t q = 0; q < part.nodes_g.size(); ++q)
        for (int a = 0; a < dim; ++a) lc[q * dim + a] = coords[(size_t)part.nodes_g[q] * dim + a];
    std::vector<int64_t> lcells((size_t)part.cells_g.size() * N);
    for (size_t q = 0; q < part.cells_g.size(); ++q)
        for (int l = 0; l < N; ++l) lcells[q * N + l] = (int64_t)node_l[G.cells[(size_t)part.cells_g[q] * N + l]] + 1;
    build_mesh_tables(dim, (int64_t)part.nodes_g.size(), lc.data(), (int64_t)part.cells_g.size(), lcells.data(), local);

    // masks and multiplicities are properties of the GLOBAL mesh (in halo mode: every copy of a local cell's entity is in H)
    for (size_t q = 0; q < part.cells_g.size(); ++q) {
        const int64_t c = halo_only ? hcell_of[part.cells_g[q]] : part.cells_g[q];
        local.dmask[q] = A.dmask[c];
        local.dupmask[q] = A.dupmask[c];
        for (int b = 0; b < 16; ++b) local.mult[q * 16 + b] = A.mult[(size_t)c * 16 + b];
    }
    part.owned_node.assign(part.nodes_g.size(), 0);
    for (size_t q = 0; q < part.nodes_g.size(); ++q) {
        const int32_t first = G.node_first[part.nodes_g[q]];
        part.owned_node[q] = first >= 0 && owner[first >> 3] == rank;
        local.node_on_boundary[q] = G.node_on_boundary[part.nodes_g[q]];
    }

    auto cut_first = [&](std::vector<int32_t> &ptr, std::vector<int32_t> &ent, const std::vector<int32_t> &cut_entries) {
        // stable partition of the CSR groups: groups containing a cut entry first
        std::unordered_set<int32_t> cs(cut_entries.begin(), cut_entries.end());
        std::vector<int32_t> nptr{0}, nent;
        int64_t ncut = 0;
        for (int pass = 0; pass < 2; ++pass)
            for (size_t gidx = 0; gidx + 1 < ptr.size(); ++gidx) {
                bool is_cut = cs.count(ent[ptr[gidx]]) != 0;
                if (is_cut != (pass == 0)) continue;
                for (int32_t q = ptr[gidx]; q < ptr[gidx + 1]; ++q) nent.push_back(ent[q]);
                nptr.push_back((int32_t)nent.size());
                if (pass == 0) ++ncut;
            }
        ptr.swap(nptr);
        ent.swap(nent);
        return ncut;
    };

    // entities whose copies live on more than one rank get a global cut id (same on every rank)
    part.segs.clear();
    std::map<std::vector<int32_t>, int32_t> seg_by_members;
    std::vector<int32_t> members;
    for (int kind = 0; kind < 3; ++kind) {
        part.nglobal[kind] = 0;
        part.gid[kind].clear();
        part.cell_lid[kind].clear();
        part.seg_of[kind].clear();
        part.seg_idx[kind].clear();
        if (kind == 0 && dim != 3) continue;
        const std::vector<Ent> &ents = kind == 0 ? glists.faces : kind == 1 ? glists.edges : glists.nodes;
        (void)nface;
        (void)nedge;
        for_groups(ents, [&](size_t i, size_t j) {
            bool cut = false, mine = !halo_only;
            for (size_t q = i + 1; q < j; ++q)
                if (cown[gcell(ents[q].cell)] != cown[gcell(ents[i].cell)]) cut = true;
            if (!cut) return;
            if (halo_only) {                     // (groups without a local copy may be incomplete in H: not ours to judge)
                for (size_t q = i; q < j; ++q) mine = mine || owner[gcell(ents[q].cell)] == rank;
                if (!mine) return;
            }
            const int64_t id = part.nglobal[kind]++;
            members.clear();
            for (size_t q = i; q < j; ++q) members.push_back(owner[gcell(ents[q].cell)]);
            std::sort(members.begin(), members.end());
            members.erase(std::unique(members.begin(), members.end()), members.end());
            if (!std::binary_search(members.begin(), members.end(), (int32_t)rank)) return;
            auto it = seg_by_members.find(members);
            if (it == seg_by_members.end()) {
                it = seg_by_members.emplace(members, (int32_t)part.segs.size()).first;
                part.segs.emplace_back();
                part.segs.back().members = members;
            }
            const int32_t sg = it->second;
            const int64_t idx = part.segs[sg].count[kind]++;
            for (size_t q = i; q < j; ++q)
                if (owner[gcell(ents[q].cell)] == rank) {
                    part.gid[kind].push_back(id);
                    part.cell_lid[kind].push_back(cell_l[gcell(ents[q].cell)] * 8 + ents[q].lid);
                    part.seg_of[kind].push_back(sg);
                    part.seg_idx[kind].push_back(idx);
                }
        });
    }
    local.ncut_edge_groups = cut_first(local.edge_ptr, local.edge_ent, part.cell_lid[1]);
    local.ncut_node_groups = cut_first(local.node_ptr, local.node_ent, part.cell_lid[2]);
    {
        // cut faces with both copies on this rank (rehearsal with cut_owner only): first in the pair list -- they are
        // summed with the cut edge / node groups, before the pack, and never ride in the CG r-update
        std::unordered_set<int32_t> cs(part.cell_lid[0].begin(), part.cell_lid[0].end());
        std::vector<int32_t> fp;
        fp.reserve(local.face_pairs.size());
        local.ncut_face_pairs = 0;
        for (int pass = 0; pass < 2; ++pass)
            for (size_t q = 0; q + 2 < local.face_pairs.size(); q += 3) {
                const bool is_cut = cs.count(local.face_pairs[q] * 8 + (local.face_pairs[q + 2] & 15)) != 0;
                if (is_cut != (pass == 0)) continue;
                fp.insert(fp.end(), local.face_pairs.begin() + q, local.face_pairs.begin() + q + 3);
                if (pass == 0) ++local.ncut_face_pairs;
            }
        local.face_pairs.swap(fp);
    }
    std::vector<uint8_t> is_cut_cell(part.cells_g.size(), 0);
    for (int kind = 0; kind < 3; ++kind)
        for (int32_t v : part.cell_lid[kind]) is_cut_cell[v >> 3] = 1;
    local.cells_cut.clear();
    local.cells_inner.clear();
    for (size_t q = 0; q < is_cut_cell.size(); ++q)
        (is_cut_cell[q] ? local.cells_cut : local.cells_inner).push_back((int32_t)q);
}

void build_cell_coefficients(const MeshTables &M, const double *sigma, std::vector<double> &coef)
{
    const int dim = M.dim;
    const int nterm = dim == 3 ? 7 : 4;
    coef.assign((size_t)M.ncells * 8, 0.0);
    for (int64_t c = 0; c < M.ncells; ++c) {
        const double *Ji = &M.jinv[(size_t)c * dim * dim];   // column-major inv(J')
        const double *sg = &sigma[(size_t)c * dim];
        double det = M.detj[c];
        int t = 0;
        for (int a = 0; a < dim; ++a)
            for (int b = a; b < dim; ++b, ++t) {
                // P = Jinv' * (sigma .* Jinv)   (src/apply_local_operators.jl:105)
                double s = 0.0;
                for (int k = 0; k < dim; ++k) s += Ji[k + dim * a] * (sg[k] * Ji[k + dim * b]);
                coef[(size_t)c * 8 + t] = det * s;
            }
        coef[(size_t)c * 8 + nterm - 1] = det;
    }
}

void assemble_coarse_matrix(const MeshTables &M, const double *sigma, double lambda, CoarseMatrix &A)
{
    const int dim = M.dim, N = dim + 1;
    A.node2int.assign(M.nnodes, -1);
    A.interior.clear();
    for (int64_t g = 0; g < M.nnodes; ++g)
        if (!M.node_on_boundary[g] && M.node_first[g] >= 0) {
            A.node2int[g] = (int32_t)A.interior.size();
            A.interior.push_back((int32_t)g);
        }
    A.n = (int64_t)A.interior.size();
    // row by row (rows are independent): the cells around the row's node in ascending order, their 4 x 4 / 3 x 3
    // element matrices' row, columns sorted, duplicates summed in that fixed order
    const double volf = dim == 3 ? 1.0 / 6.0 : 0.5;
    const double massf = dim == 3 ? 1.0 / 20.0 : 1.0 / 12.0;
    std::vector<int32_t> rowlen(A.n, 0);
    struct CV {
        int32_t c;
        double v;
    };
    auto row_entries = [&](int64_t r, std::vector<CV> &buf) {
        buf.clear();
        const int32_t gnode = A.interior[r];
        for (int32_t q = M.node_all_ptr[gnode]; q < M.node_all_ptr[gnode + 1]; ++q) {
            const int64_t c = M.node_all_ent[q] >> 3;
            const int i = M.node_all_ent[q] & 7;
            const double *Ji = &M.jinv[(size_t)c * dim * dim];
            const double *sg = &sigma[(size_t)c * dim];
            const int32_t *el = &M.cells[c * N];
            const double vol = M.detj[c] * volf;
            // gradients = Jinv * refgrads (src/cell_values.jl:117)
            double g[3][4];
            for (int a = 0; a < dim; ++a) {
                double s = 0.0;
                for (int k = 0; k < dim; ++k) {
                    g[a][k + 1] = Ji[a + dim * k];
                    s += Ji[a + dim * k];
                }
                g[a][0] = -s;
            }
            for (int j = 0; j < N; ++j) {
                const int32_t cj = A.node2int[el[j]];
                if (cj < 0) continue;
                double k = 0.0;
                for (int a = 0; a < dim; ++a) k += g[a][i] * sg[a] * g[a][j];
                buf.push_back({cj, vol * (k + lambda * massf * (i == j ? 2.0 : 1.0))});
            }
        }
        std::stable_sort(buf.begin(), buf.end(), [](const CV &x, const CV &y) { return x.c < y.c; });
        size_t w = 0;
        for (size_t q = 0; q < buf.size();) {
            size_t e = q;
            double s = 0.0;
            while (e < buf.size() && buf[e].c == buf[q].c) s += buf[e++].v;
            buf[w++] = {buf[q].c, s};
            q = e;
        }
        buf.resize(w);
    };
    parallel_for(A.n, [&](int64_t r0, int64_t r1) {
        std::vector<CV> buf;
        for (int64_t r = r0; r < r1; ++r) {
            row_entries(r, buf);
            rowlen[r] = (int32_t)buf.size();
        }
    });
    A.rowptr.assign(A.n + 1, 0);
    for (int64_t r = 0; r < A.n; ++r) A.rowptr[r + 1] = A.rowptr[r] + rowlen[r];
    A.colidx.resize(A.rowptr[A.n]);
    A.val.resize(A.rowptr[A.n]);
    A.diag.assign(A.n, 0.0);
    parallel_for(A.n, [&](int64_t r0, int64_t r1) {
        std::vector<CV> buf;
        for (int64_t r = r0; r < r1; ++r) {
            row_entries(r, buf);
            for (size_t q = 0; q < buf.size(); ++q) {
                A.colidx[A.rowptr[r] + q] = buf[q].c;
                A.val[A.rowptr[r] + q] = buf[q].v;
                if (buf[q].c == (int32_t)r) A.diag[r] = buf[q].v;
            }
        }
    });
}

}  // namespace hmg
