// Host-side synthesis of the checkerboard problems, threaded: the box of unit cubes split into
// tetrahedra / triangles, the infinity-norm ordering that makes every centred sub-cube a prefix, and the per-cell
// coefficient lookup.  Reference behaviour reproduced (file:line in the reference checkout):
//   mesh      src/tet/generate_grid.jl:6-45, src/tri/generate_grid.jl:6-35   (node ids: last coordinate fastest; corner
//             lookup through a first-index-fastest id table -- for a cube that transposes the geometry, exactly as the
//             reference's loops do)
//   ordering  src/examples/homogenized_coefficients.jl:21-28  (stable sorts by infinity norm: nodes, then cell centres)
//   sigma     src/examples/homogenized_coefficients.jl:494-503 (cell centre + offset, truncated)
#include "../../include/hmg.h"
#include "hmg_host.hpp"

#include <atomic>
#include <cmath>
#include <cstring>
#include <stdexcept>

using namespace hmg;

namespace {

const int CUBE_TETS[6][4] = {{0, 1, 2, 6}, {0, 1, 4, 6}, {1, 3, 2, 6}, {1, 3, 6, 7}, {1, 5, 4, 6}, {1, 5, 6, 7}};
const int SQUARE_TRIS[2][3] = {{0, 1, 2}, {1, 2, 3}};

void sort_small(int64_t *v, int n)
{
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0 && v[j - 1] > v[j]; --j) std::swap(v[j - 1], v[j]);
}

// stable order of the indices 0..n-1 by key (ties: ascending index)
std::vector<int64_t> stable_order(const std::vector<double> &key)
{
    struct KI {
        double k;
        int64_t i;
    };
    std::vector<KI> v(key.size());
    parallel_for((int64_t)key.size(), [&](int64_t a, int64_t b) {
        for (int64_t q = a; q < b; ++q) v[q] = {key[q], q};
    });
    parallel_sort(v, [](const KI &x, const KI &y) { return x.k != y.k ? x.k < y.k : x.i < y.i; });
    std::vector<int64_t> out(key.size());
    parallel_for((int64_t)key.size(), [&](int64_t a, int64_t b) {
        for (int64_t q = a; q < b; ++q) out[q] = v[q].i;
    });
    return out;
}

int fail(const std::exception &e)
{
    last_error() = e.what();
    return 1;
}

}  // namespace

extern "C" {

int hmg_checkerboard_mesh_size(int dim, const int64_t *shape, int64_t *nnodes, int64_t *ncells)
{
    try {
        if ((dim != 2 && dim != 3) || !shape || !nnodes || !ncells) throw std::runtime_error("bad argument");
        int64_t nn = 1, nc = dim == 3 ? 6 : 2;
        for (int a = 0; a < dim; ++a) {
            if (shape[a] < 1) throw std::runtime_error("shape must be positive");
            nn *= shape[a] + 1;
            nc *= shape[a];
        }
        *nnodes = nn;
        *ncells = nc;
    } catch (const std::exception &e) {
        return fail(e);
    }
    return 0;
}

int hmg_checkerboard_mesh(int dim, const int64_t *shape, const double *origin, int transposed_lookup, int ordered,
                          double *coords, int64_t *cells)
{
    try {
        int64_t nn, nc;
        if (hmg_checkerboard_mesh_size(dim, shape, &nn, &nc) != 0) throw std::runtime_error(last_error());
        if (!origin || !coords || !cells) throw std::runtime_error("null argument");
        const int N = dim + 1, ncorner = 1 << dim, per = dim == 3 ? 6 : 2;
        int64_t k[3] = {1, 1, 1}, cstride[3] = {1, 1, 1}, ncubes = 1;
        for (int a = 0; a < dim; ++a) {
            k[a] = shape[a] + 1;
            ncubes *= shape[a];
        }
        if (transposed_lookup)
            for (int a = 1; a < dim; ++a)
                if (shape[a] != shape[0]) throw std::runtime_error("transposed corner lookup needs a cube");
        // node q has the multi-index (q / (k1 k2), (q / k2) % k1, q % k2): last coordinate fastest
        for (int a = dim - 2; a >= 0; --a) cstride[a] = cstride[a + 1] * k[a + 1];
        parallel_for(nn, [&](int64_t a0, int64_t a1) {
            for (int64_t q = a0; q < a1; ++q) {
                int64_t r = q;
                for (int a = 0; a < dim; ++a) {
                    const int64_t ia = r / cstride[a];
                    r -= ia * cstride[a];
                    coords[q * dim + a] = (double)ia + origin[a];
                }
            }
        });
        // corner of cube (c0, c1, c2), offset bit a -> +1 on axis a.  The id table runs with the FIRST index fastest:
        // id = i0 + k0 i1 + k0 k1 i2.  transposed_lookup keeps that id as the node id (the reference's loops; the node
        // then sits at the transposed coordinates), otherwise it is translated to the node with those coordinates.
        int64_t fstride[3] = {1, k[0], k[0] * k[1]};
        parallel_for(ncubes, [&](int64_t a0, int64_t a1) {
            for (int64_t q = a0; q < a1; ++q) {
                int64_t idx[3] = {0, 0, 0}, r = q;      // cube multi-index, first index slowest
                for (int a = dim - 1; a >= 0; --a) {
                    idx[a] = r % shape[a];
                    r /= shape[a];
                }
                int64_t ids[8];
                for (int c = 0; c < ncorner; ++c) {
                    int64_t id = 0, node = 0;
                    for (int a = 0; a < dim; ++a) {
                        const int64_t ia = idx[a] + ((c >> a) & 1);
                        id += ia * fstride[a];
                        node += ia * cstride[a];
                    }
                    ids[c] = transposed_lookup ? id : node;
                }
                for (int t = 0; t < per; ++t) {
                    int64_t *out = cells + (q * per + t) * N;
                    for (int l = 0; l < N; ++l) out[l] = ids[dim == 3 ? CUBE_TETS[t][l] : SQUARE_TRIS[t][l]];
                    sort_small(out, N);
                    for (int l = 0; l < N; ++l) out[l] += 1;      // 1-based, ascending tuples
                }
            }
        });
        if (!ordered) return 0;
        // nodes by infinity norm (stable), cells relabelled and re-sorted per tuple, then cells by the norm of their centre
        std::vector<double> key(nn);
        parallel_for(nn, [&](int64_t a0, int64_t a1) {
            for (int64_t q = a0; q < a1; ++q) {
                double m = 0.0;
                for (int a = 0; a < dim; ++a) m = std::max(m, std::fabs(coords[q * dim + a]));
                key[q] = m;
            }
        });
        const std::vector<int64_t> perm = stable_order(key);
        std::vector<int64_t> inv(nn);
        std::vector<double> nc2((size_t)nn * dim);
        parallel_for(nn, [&](int64_t a0, int64_t a1) {
            for (int64_t q = a0; q < a1; ++q) {
                inv[perm[q]] = q;
                for (int a = 0; a < dim; ++a) nc2[q * dim + a] = coords[perm[q] * dim + a];
            }
        });
        std::memcpy(coords, nc2.data(), sizeof(double) * nc2.size());
        std::vector<double> ckey(nc);
        parallel_for(nc, [&](int64_t a0, int64_t a1) {
            for (int64_t c = a0; c < a1; ++c) {
                int64_t *el = cells + c * N;
                for (int l = 0; l < N; ++l) el[l] = inv[el[l] - 1];
                sort_small(el, N);
                double m = 0.0;
                for (int a = 0; a < dim; ++a) {
                    double acc = coords[el[0] * dim + a];                // mean(get_nodes(mesh, el)): this order of additions
                    for (int l = 1; l < N; ++l) acc += coords[el[l] * dim + a];
                    m = std::max(m, std::fabs(acc / N));
                }
                ckey[c] = m;
                for (int l = 0; l < N; ++l) el[l] += 1;
            }
        });
        const std::vector<int64_t> order = stable_order(ckey);
        std::vector<int64_t> c2((size_t)nc * N);
        parallel_for(nc, [&](int64_t a0, int64_t a1) {
            for (int64_t c = a0; c < a1; ++c)
                for (int l = 0; l < N; ++l) c2[c * N + l] = cells[order[c] * N + l];
        });
        std::memcpy(cells, c2.data(), sizeof(int64_t) * c2.size());
    } catch (const std::exception &e) {
        return fail(e);
    }
    return 0;
}

int hmg_conductivity_per_element(int dim, int64_t nnodes, const double *coords, int64_t ncells, const int64_t *cells,
                                 const int64_t *grid_shape, const double *sigma_grid, const double *offset, double *sigma)
{
    try {
        if ((dim != 2 && dim != 3) || !coords || !cells || !grid_shape || !sigma_grid || !offset || !sigma)
            throw std::runtime_error("bad argument");
        const int N = dim + 1;
        std::atomic<int> bad{0};                 // 1: node id out of range, 2: centre outside the coefficient grid
        parallel_for(ncells, [&](int64_t a0, int64_t a1) {
            for (int64_t c = a0; c < a1; ++c) {
                const int64_t *el = cells + c * N;
                bool ok = true;
                for (int l = 0; l < N; ++l) ok = ok && el[l] >= 1 && el[l] <= nnodes;
                if (!ok) {                       // (checked before any coordinate is read)
                    bad.store(1, std::memory_order_relaxed);
                    continue;
                }
                int64_t lin = 0;
                for (int a = 0; a < dim && ok; ++a) {
                    double acc = coords[(el[0] - 1) * dim + a];
                    for (int l = 1; l < N; ++l) acc += coords[(el[l] - 1) * dim + a];
                    const int64_t ia = (int64_t)std::trunc(acc / N + offset[a]) - 1;
                    if (ia < 0 || ia >= grid_shape[a]) {
                        if (bad.load(std::memory_order_relaxed) == 0) bad.store(2, std::memory_order_relaxed);
                        ok = false;
                    } else
                        lin = lin * grid_shape[a] + ia;
                }
                if (!ok) continue;
                for (int a = 0; a < dim; ++a) sigma[c * dim + a] = sigma_grid[lin * dim + a];
            }
        });
        if (bad.load() == 1) throw std::runtime_error("conductivity_per_element: a cell refers to a node id outside 1..nnodes");
        if (bad.load() == 2) throw std::runtime_error("conductivity_per_element: a cell centre lies outside the coefficient grid");
    } catch (const std::exception &e) {
        return fail(e);
    }
    return 0;
}

int hmg_block_owner(int dim, int64_t nnodes, const double *coords, int64_t ncells, const int64_t *cells,
                    const int64_t *blocks, double width, const double *origin, int32_t *owner)
{
    try {
        if ((dim != 2 && dim != 3) || !coords || !cells || !blocks || !origin || !owner || !(width > 0.0))
            throw std::runtime_error("bad argument");
        const int N = dim + 1;
        for (int a = 0; a < dim; ++a)
            if (blocks[a] < 1) throw std::runtime_error("block_owner: every axis needs at least one block");
        std::atomic<int> bad{0};
        parallel_for(ncells, [&](int64_t a0, int64_t a1) {
            for (int64_t c = a0; c < a1; ++c) {
                const int64_t *el = cells + c * N;
                bool ok = true;
                for (int l = 0; l < N; ++l) ok = ok && el[l] >= 1 && el[l] <= nnodes;
                if (!ok) {
                    bad.store(1, std::memory_order_relaxed);
                    owner[c] = -1;
                    continue;
                }
                int64_t o = 0;
                for (int a = 0; a < dim; ++a) {
                    double acc = coords[(el[0] - 1) * dim + a];
                    for (int l = 1; l < N; ++l) acc += coords[(el[l] - 1) * dim + a];
                    int64_t ia = (int64_t)std::floor((acc / N - origin[a]) / width);
                    ia = std::max<int64_t>(0, std::min<int64_t>(ia, blocks[a] - 1));
                    o = o * blocks[a] + ia;
                }
                owner[c] = (int32_t)o;
            }
        });
        if (bad.load()) throw std::runtime_error("block_owner: a cell refers to a node id outside 1..nnodes");
    } catch (const std::exception &e) {
        return fail(e);
    }
    return 0;
}

}  // extern "C"
