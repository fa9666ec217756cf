// Refined reference simplex: hierarchical numbering, entity-major storage order, lattice stencil.
//
// The reference builds each level by red refinement of an explicit mesh and assembles dim^2+1
// CSC matrices per level (src/multilevel_reference.jl:41-61, src/build_local_operators.jl:51-141).
// Here the same object is derived from its closed form: the level-l refinement of the reference
// simplex is the Freudenthal (Kuhn) triangulation of the lattice {i,j,k >= 0, i+j+k <= m},
// m = 2^(l-1), whose edges point in 7 (3D) / 3 (2D) directions.  Only the *numbering* is taken
// from the reference, because it is the API contract of the level vectors:
//   - level 1 nodes: (0,0,0),(1,0,0),(0,1,0),(0,0,1)            src/multilevel_reference.jl:9-13
//   - refinement appends one node per edge, edges enumerated by (from asc, to asc)
//                                                          src/tet/refine.jl:16-21, src/sparse_graph.jl:20-48
//   - faces x3=0, x2=0, x1=0, sum=1; edges (1,2),(1,3),(1,4),(2,3),(2,4),(3,4); entity node lists
//     in ascending node id                                  src/multilevel_reference.jl:63-203
#include "hmg_host.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <stdexcept>

namespace hmg {

std::string &last_error()
{
    static thread_local std::string s;
    return s;
}

static const int DIRS3[15][3] = {
    {0, 0, 0},  {1, 0, 0},  {-1, 0, 0}, {-1, 1, 0}, {1, -1, 0}, {0, 1, 0},   {0, -1, 0}, {0, -1, 1},
    {0, 1, -1}, {-1, 0, 1}, {1, 0, -1}, {0, 0, 1},  {0, 0, -1}, {1, -1, 1}, {-1, 1, -1}};

int stencil_dirs(int dim, const int (**dirs)[3])
{
    *dirs = DIRS3;
    return dim == 3 ? 15 : 7;
}

namespace {

struct Lattice {
    int dim, m;
    int idx(int i, int j, int k) const { return (k * (m + 1) + j) * (m + 1) + i; }
    bool inside(int i, int j, int k) const
    {
        return i >= 0 && j >= 0 && k >= 0 && i + j + k <= m && (dim == 3 || k == 0);
    }
};

// Solve the small dense system for the P1 gradients of a simplex given in lattice coordinates.
// g[a][v] = d phi_v / d x_a  (x = lattice / m), vol = simplex volume in reference coordinates.
void simplex_gradients(int dim, int m, const int X[4][3], double g[3][4], double &vol)
{
    double J[3][3] = {{0}};
    for (int c = 0; c < dim; ++c)
        for (int a = 0; a < dim; ++a) J[a][c] = double(X[c + 1][a] - X[0][a]) / double(m);
    double inv[3][3] = {{0}};
    double det;
    if (dim == 2) {
        det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
        inv[0][0] = J[1][1] / det;
        inv[0][1] = -J[0][1] / det;
        inv[1][0] = -J[1][0] / det;
        inv[1][1] = J[0][0] / det;
        vol = std::fabs(det) / 2.0;
    } else {
        double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
        double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
        double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
        det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
        inv[0][0] = c00 / det;
        inv[1][0] = c01 / det;
        inv[2][0] = c02 / det;
        inv[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
        inv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
        inv[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
        inv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
        inv[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
        inv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
        vol = std::fabs(det) / 6.0;
    }
    // gradients = J^-T * refgrads; refgrads columns: (-1,..,-1), e_1, .., e_dim
    for (int a = 0; a < dim; ++a) {
        double s = 0.0;
        for (int c = 0; c < dim; ++c) {
            g[a][c + 1] = inv[c][a];
            s += inv[c][a];
        }
        g[a][0] = -s;
    }
}

int dir_index(int dim, int di, int dj, int dk)
{
    int nd = dim == 3 ? 15 : 7;
    for (int d = 0; d < nd; ++d)
        if (DIRS3[d][0] == di && DIRS3[d][1] == dj && DIRS3[d][2] == dk) return d;
    return -1;
}

}  // namespace

std::vector<LevelTables> build_level_tables(int dim, int nlevels)
{
    if (dim != 2 && dim != 3) throw std::runtime_error("dim must be 2 or 3");
    // 16-bit lattice addresses and 6/8-bit row indices in the packed tables: m <= 64 (3D), m <= 128 (2D)
    if (nlevels < 1 || nlevels > (dim == 3 ? 7 : 8))
        throw std::runtime_error(dim == 3 ? "nlevels must be in 1..7 for tetrahedra" : "nlevels must be in 1..8 for triangles");
    const int ndir = dim == 3 ? 15 : 7;
    const int ndiff = dim == 3 ? 6 : 3;
    const int nterm = ndiff + 1;
    const int ncorner = dim + 1;
    const int nedge = dim == 3 ? 6 : 3;
    const int nface = dim == 3 ? 4 : 0;

    std::vector<LevelTables> out(nlevels);
    // hierarchical coordinates of the previous / current level, in current-level lattice units
    std::vector<std::array<int, 3>> hc;          // coords of hier id h at the current level
    std::vector<std::array<int32_t, 2>> parents; // parents (hier ids on the previous level) or (-1,-1)

    for (int lev = 1; lev <= nlevels; ++lev) {
        const int m = 1 << (lev - 1);
        Lattice lat{dim, m};
        LevelTables &T = out[lev - 1];
        T.dim = dim;
        T.level = lev;
        T.m = m;

        if (lev == 1) {
            hc.clear();
            hc.push_back({0, 0, 0});
            hc.push_back({1, 0, 0});
            hc.push_back({0, 1, 0});
            if (dim == 3) hc.push_back({0, 0, 1});
            parents.assign(hc.size(), {-1, -1});
        } else {
            // refine: coordinates double, one new node per edge of the previous level,
            // edges enumerated by (from asc, to asc).
            const int mp = m / 2;
            Lattice latp{dim, mp};
            const size_t np = hc.size();
            std::vector<int32_t> id_of((size_t)(mp + 1) * (mp + 1) * (dim == 3 ? mp + 1 : 1), -1);
            for (size_t h = 0; h < np; ++h) id_of[latp.idx(hc[h][0], hc[h][1], hc[h][2])] = (int32_t)h;
            std::vector<std::array<int, 3>> nc(np);
            for (size_t h = 0; h < np; ++h) nc[h] = {2 * hc[h][0], 2 * hc[h][1], 2 * hc[h][2]};
            std::vector<std::array<int32_t, 2>> npar(np, {-1, -1});
            std::vector<int32_t> tos;
            for (size_t from = 0; from < np; ++from) {
                tos.clear();
                for (int d = 1; d < ndir; ++d) {
                    int i = hc[from][0] + DIRS3[d][0], j = hc[from][1] + DIRS3[d][1],
                        k = hc[from][2] + DIRS3[d][2];
                    if (!latp.inside(i, j, k)) continue;
                    int32_t to = id_of[latp.idx(i, j, k)];
                    if (to > (int32_t)from) tos.push_back(to);
                }
                std::sort(tos.begin(), tos.end());
                for (int32_t to : tos) {
                    nc.push_back({hc[from][0] + hc[to][0], hc[from][1] + hc[to][1],
                                  hc[from][2] + hc[to][2]});
                    npar.push_back({(int32_t)from, to});
                }
            }
            hc.swap(nc);
            parents.swap(npar);
        }

        const int nf = (int)hc.size();
        const int expect = dim == 3 ? (m + 1) * (m + 2) * (m + 3) / 6 : (m + 1) * (m + 2) / 2;
        if (nf != expect) throw std::runtime_error("reference lattice: unexpected node count");
        T.nf = nf;
        T.ld = nf;
        T.ncorner = ncorner;
        T.nedge = nedge;
        T.nface = nface;
        T.nei = m - 1;
        T.nfi = dim == 3 ? (m - 1) * (m - 2) / 2 : 0;
        T.nint = dim == 3 ? (m - 1) * (m - 2) * (m - 3) / 6 : (m - 1) * (m - 2) / 2;
        if (T.nint < 0) T.nint = 0;
        if (T.nfi < 0) T.nfi = 0;
        T.off_edge = ncorner;
        T.off_face = T.off_edge + nedge * T.nei;
        T.off_int = T.off_face + nface * T.nfi;
        if (T.off_int + T.nint != nf) throw std::runtime_error("reference lattice: entity count mismatch");

        // ---- entity classification + storage order ---------------------------------------
        T.hier2slot.assign(nf, -1);
        T.slot2hier.assign(nf, -1);
        T.slot_cls.assign(nf, 0);
        T.slot_ijk.assign((size_t)nf * 3, 0);
        // Within an edge / face the DOFs are stored in lattice order (k, then j, then i ascending).
        // That order is canonical across the cells sharing the entity: a cell's node tuple is
        // ascending in the global numbering, so the barycentric coordinates of a fine node with
        // respect to the entity's local vertices (taken in local order) are the same in every
        // adjacent cell, and (k, j, i) lexicographic order equals lexicographic order of those
        // barycentric coordinates on each of the 4 faces / 6 edges of the reference simplex.
        // (The reference lists entity nodes by ascending node id, src/multilevel_reference.jl:125-203;
        // any order shared by all copies gives the same interface sums.)
        std::vector<std::vector<int32_t>> edge_nodes(nedge), face_nodes(std::max(nface, 1));
        std::vector<int32_t> interior;
        auto cls_face = [&](int f) { return 1 + f; };
        auto cls_edge = [&](int e) { return 1 + nface + e; };
        auto cls_corner = [&](int c) { return 1 + nface + nedge + c; };
        for (int h = 0; h < nf; ++h) {
            const int i = hc[h][0], j = hc[h][1], k = hc[h][2];
            if (h < ncorner) {
                T.hier2slot[h] = h;
                T.slot2hier[h] = h;
                T.slot_cls[h] = (uint8_t)cls_corner(h);
            } else if (dim == 3) {
                bool on[4] = {k == 0, j == 0, i == 0, i + j + k == m};
                int cnt = on[0] + on[1] + on[2] + on[3];
                if (cnt == 0)
                    interior.push_back(h);
                else if (cnt == 1)
                    face_nodes[on[0] ? 0 : on[1] ? 1 : on[2] ? 2 : 3].push_back(h);
                else if (cnt == 2)
                    // edges (1,2)=f0&f1 (1,3)=f0&f2 (1,4)=f1&f2 (2,3)=f0&f3 (2,4)=f1&f3 (3,4)=f2&f3
                    edge_nodes[on[0] && on[1] ? 0 : on[0] && on[2] ? 1 : on[1] && on[2] ? 2
                               : on[0] && on[3] ? 3 : on[1] && on[3] ? 4 : 5].push_back(h);
                else
                    throw std::runtime_error("reference lattice: non-corner node on 3 faces");
            } else {
                bool on[3] = {j == 0, i == 0, i + j == m};
                int cnt = on[0] + on[1] + on[2];
                if (cnt == 0)
                    interior.push_back(h);
                else if (cnt == 1)
                    edge_nodes[on[0] ? 0 : on[1] ? 1 : 2].push_back(h);
                else
                    throw std::runtime_error("reference lattice: non-corner node on 2 edges");
            }
        }
        auto lattice_less = [&](int32_t a, int32_t b) {
            if (hc[a][2] != hc[b][2]) return hc[a][2] < hc[b][2];
            if (hc[a][1] != hc[b][1]) return hc[a][1] < hc[b][1];
            return hc[a][0] < hc[b][0];
        };
        for (int e = 0; e < nedge; ++e) {
            if ((int)edge_nodes[e].size() != T.nei) throw std::runtime_error("reference lattice: edge count");
            std::sort(edge_nodes[e].begin(), edge_nodes[e].end(), lattice_less);
            for (int q = 0; q < T.nei; ++q) {
                int h = edge_nodes[e][q], slot = T.off_edge + e * T.nei + q;
                T.hier2slot[h] = slot;
                T.slot2hier[slot] = h;
                T.slot_cls[slot] = (uint8_t)cls_edge(e);
            }
        }
        for (int f = 0; f < nface; ++f) {
            if ((int)face_nodes[f].size() != T.nfi) throw std::runtime_error("reference lattice: face count");
            std::sort(face_nodes[f].begin(), face_nodes[f].end(), lattice_less);
            for (int q = 0; q < T.nfi; ++q) {
                int h = face_nodes[f][q], slot = T.off_face + f * T.nfi + q;
                T.hier2slot[h] = slot;
                T.slot2hier[slot] = h;
                T.slot_cls[slot] = (uint8_t)cls_face(f);
            }
        }
        // interior: lattice order (k, j, i ascending)
        std::sort(interior.begin(), interior.end(), lattice_less);
        if ((int)interior.size() != T.nint) throw std::runtime_error("reference lattice: interior count");
        for (int q = 0; q < T.nint; ++q) {
            int h = interior[q], slot = T.off_int + q;
            T.hier2slot[h] = slot;
            T.slot2hier[slot] = h;
            T.slot_cls[slot] = 0;
        }
        for (int s = 0; s < nf; ++s) {
            int h = T.slot2hier[s];
            T.slot_ijk[3 * s + 0] = hc[h][0];
            T.slot_ijk[3 * s + 1] = hc[h][1];
            T.slot_ijk[3 * s + 2] = hc[h][2];
        }

        // ---- LDS lattice addressing --------------------------------------------------------
        // L(i,j,k) = PO(k) + RO(j; m-k) + i, rows of length m+1-j-k.
        auto tri = [](int n) { return (n + 1) * (n + 2) / 2; };
        std::vector<int> PO(m + 2, 0);
        if (dim == 3)
            for (int k = 0; k <= m; ++k) PO[k + 1] = PO[k] + tri(m - k);
        auto lin = [&](int i, int j, int k) {
            int n = m - k;
            return PO[k] + j * (n + 1) - j * (j - 1) / 2 + i;
        };
        T.meta.assign(nf, 0);
        int amin = 0, amax = nf - 1, amin_interior = 0;
        for (int s = 0; s < nf; ++s) {
            int i = T.slot_ijk[3 * s], j = T.slot_ijk[3 * s + 1], k = T.slot_ijk[3 * s + 2];
            int L = lin(i, j, k);
            int len = m + 1 - j - k;
            int A = 0, B = 0;
            if (dim == 3) {
                A = tri(m - k) - j;
                B = tri(m - k + 1) - j;
            }
            T.meta[s] = pack_meta((uint32_t)L, (uint32_t)len, T.slot_cls[s], (uint32_t)A, (uint32_t)B);
            int addr[15] = {L,           L + 1,           L - 1,       L + len - 1, L - len,
                            L + len,     L - len - 1,     L + A - len, L - B + len + 1,
                            L + A - 1,   L - B + 1,       L + A,       L - B,
                            L + A - len + 1, L - B + len};
            for (int d = 0; d < ndir; ++d) {
                int ni = i + DIRS3[d][0], nj = j + DIRS3[d][1], nk = k + DIRS3[d][2];
                if (lat.inside(ni, nj, nk) && addr[d] != lin(ni, nj, nk))
                    throw std::runtime_error("reference lattice: neighbour addressing mismatch");
                amin = std::min(amin, addr[d]);
                amax = std::max(amax, addr[d]);
                if (T.slot_cls[s] == 0) amin_interior = std::min(amin_interior, addr[d]);
            }
        }
        // Addresses below 0 only occur for zero-weight taps of surface nodes (plane k = 0, row j = 0); the
        // kernels clamp those to 0 instead of paying a front guard zone of ~T(m+2) doubles of LDS per cell.
        if (amin_interior < 0) throw std::runtime_error("reference lattice: an interior node addresses below 0");
        T.lds_g0 = 0;
        T.lds_g1 = std::max(amax - (nf - 1), dim == 3 ? m + 8 : 0);   // (3D: the blocked interior reads whole 7-point plane stars)
        (void)amin;

        // interior sweep: rows (j,k) that contain cell-interior nodes, all positions i = 0..len-1
        T.sweep_meta.clear();
        T.sweep_slot.clear();
        if (T.nint > 0) {
            std::vector<int32_t> slot_of_L(nf, -1);
            for (int s2 = 0; s2 < nf; ++s2) slot_of_L[(int)(T.meta[s2] & 0xffffu)] = s2;
            const int kmax = dim == 3 ? m : 0;
            for (int k = (dim == 3 ? 1 : 0); k <= kmax; ++k)
                for (int j = 1; j <= m - k; ++j) {
                    int len = m + 1 - j - k;
                    if (len < 3) continue;   // no interior node in this row
                    for (int i = 0; i < len; ++i) {
                        int L = lin(i, j, k);
                        int sl = slot_of_L[L];
                        bool active = T.slot_cls[sl] == 0;
                        uint64_t mt = T.meta[sl] & ~((uint64_t)0xff << 24);
                        T.sweep_meta.push_back(mt);
                        T.sweep_slot.push_back(active ? (uint16_t)sl : (uint16_t)0xffff);
                    }
                }
            size_t nact = 0;
            for (uint16_t v : T.sweep_slot) nact += v != 0xffff;
            if ((int)nact != T.nint) throw std::runtime_error("reference lattice: interior sweep does not cover the interior");
        }

        // ---- stencil rows by summing the sub-simplices around every node -------------------
        T.ncls = 1 + nface + nedge + ncorner;
        T.ndir = ndir;
        T.nterm = nterm;
        std::vector<double> rows((size_t)nf * ndir * nterm, 0.0);
        T.dphi.assign((size_t)nf * 3, 0.0);
        T.load.assign((size_t)nf, 0.0);
        std::vector<int32_t> slot_at((size_t)(m + 1) * (m + 1) * (dim == 3 ? m + 1 : 1), -1);
        for (int s = 0; s < nf; ++s)
            slot_at[lat.idx(T.slot_ijk[3 * s], T.slot_ijk[3 * s + 1], T.slot_ijk[3 * s + 2])] = s;
        const int nperm = dim == 3 ? 6 : 2;
        static const int PERM3[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
        static const int PERM2[2][3] = {{0, 1, 0}, {1, 0, 0}};
        // Kuhn coordinates (p,q,r): i = p-q, j = q-r, k = r (3D); i = p-q, j = q (2D).
        for (int p0 = 0; p0 < m; ++p0)
            for (int q0 = 0; q0 < m; ++q0)
                for (int r0 = 0; r0 < (dim == 3 ? m : 1); ++r0)
                    for (int pi = 0; pi < nperm; ++pi) {
                        const int *perm = dim == 3 ? PERM3[pi] : PERM2[pi];
                        int V[4][3];
                        int cur[3] = {p0, q0, r0};
                        bool ok = true;
                        for (int v = 0; v <= dim; ++v) {
                            if (v > 0) cur[perm[v - 1]] += 1;
                            int p = cur[0], q = cur[1], r = dim == 3 ? cur[2] : 0;
                            if (!(p <= m && q <= p && r <= q && r >= 0)) {
                                ok = false;
                                break;
                            }
                            V[v][0] = p - q;
                            V[v][1] = q - r;
                            V[v][2] = r;
                        }
                        if (!ok) continue;
                        double g[3][4], vol;
                        simplex_gradients(dim, m, V, g, vol);
                        int sl[4];
                        for (int v = 0; v <= dim; ++v) sl[v] = slot_at[lat.idx(V[v][0], V[v][1], V[v][2])];
                        for (int v = 0; v <= dim; ++v) T.ref_cells.push_back(T.slot2hier[sl[v]]);
                        for (int u = 0; u <= dim; ++u)
                            for (int a = 0; a < dim; ++a) T.dphi[(size_t)sl[u] * 3 + a] += vol * g[a][u];
                        for (int u = 0; u <= dim; ++u) T.load[sl[u]] += vol / (dim + 1);
                        for (int u = 0; u <= dim; ++u)
                            for (int v = 0; v <= dim; ++v) {
                                int d = dir_index(dim, V[v][0] - V[u][0], V[v][1] - V[u][1], V[v][2] - V[u][2]);
                                if (d < 0) throw std::runtime_error("reference lattice: edge outside direction set");
                                double *row = &rows[((size_t)sl[u] * ndir + d) * nterm];
                                int t = 0;
                                for (int a = 0; a < dim; ++a)
                                    for (int b = a; b < dim; ++b, ++t)
                                        row[t] += a == b ? vol * g[a][u] * g[a][v]
                                                         : vol * (g[a][u] * g[b][v] + g[b][u] * g[a][v]);
                                double mfac = dim == 3 ? vol / 20.0 : vol / 12.0;
                                row[t] += u == v ? 2.0 * mfac : mfac;
                            }
                    }
        T.ctab.assign((size_t)T.ncls * ndir * nterm, 0.0);
        std::vector<int> rep(T.ncls, -1);
        for (int s = 0; s < nf; ++s) {
            int c = T.slot_cls[s];
            const double *row = &rows[(size_t)s * ndir * nterm];
            if (rep[c] < 0) {
                rep[c] = s;
                std::copy(row, row + ndir * nterm, &T.ctab[(size_t)c * ndir * nterm]);
            } else {
                const double *ref = &T.ctab[(size_t)c * ndir * nterm];
                for (int q = 0; q < ndir * nterm; ++q)
                    if (std::fabs(row[q] - ref[q]) > 1e-12 * (1.0 + std::fabs(ref[q])) * m)
                        throw std::runtime_error("reference lattice: stencil not uniform within an entity class");
            }
        }

        // ---- transfer tables (level-1 -> this level) ------------------------------------------
        if (lev > 1) {
            const LevelTables &C = out[lev - 2];
            T.par_a.assign(nf, 0);
            T.par_b.assign(nf, 0);
            std::vector<std::vector<int32_t>> kids(C.nf);   // by coarse hier id: fine hier ids
            for (int h = 0; h < nf; ++h) {
                int s = T.hier2slot[h];
                if (h < C.nf) {
                    T.par_a[s] = T.par_b[s] = C.hier2slot[h];
                    kids[h].push_back(h);
                } else {
                    T.par_a[s] = C.hier2slot[parents[h][0]];
                    T.par_b[s] = C.hier2slot[parents[h][1]];
                    kids[parents[h][0]].push_back(h);
                    kids[parents[h][1]].push_back(h);
                }
            }
            T.rptr.assign(C.nf + 1, 0);
            for (int cs = 0; cs < C.nf; ++cs) T.rptr[cs + 1] = T.rptr[cs] + (int)kids[C.slot2hier[cs]].size();
            T.ridx.resize(T.rptr[C.nf]);
            for (int cs = 0; cs < C.nf; ++cs) {
                auto &kk = kids[C.slot2hier[cs]];
                std::sort(kk.begin(), kk.end());
                for (size_t q = 0; q < kk.size(); ++q) T.ridx[T.rptr[cs] + q] = T.hier2slot[kk[q]];
            }
        }
    }
    return out;
}

}  // namespace hmg
