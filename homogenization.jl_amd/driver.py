"""
Host-side driver pieces (level L4 of the reference: src/examples/homogenized_coefficients.jl) that sit
on either side of the hot path: base-mesh synthesis, the infinity-norm ordering that makes a domain
shrink a prefix operation, the checkerboard coefficient field and the `checkerboard_homogenization`
loop itself.  All level-vector work goes to the device through api.py; nothing here touches a level
vector on the host.
"""
from __future__ import annotations

import math

import numpy as np

from . import api
from .api import Mesh, Tet64, Tri64

_CUBE_TETS = ((0, 1, 2, 6), (0, 1, 4, 6), (1, 3, 2, 6), (1, 3, 6, 7), (1, 5, 4, 6), (1, 5, 6, 7))


def hypercube(eltype, n: int, scale=1.0, origin=None) -> Mesh:
    """n^d unit cubes, each split into 6 tetrahedra (3D) / 2 triangles (2D); numbering as the reference's
    `hypercube` (src/tet/generate_grid.jl:6-45, src/tri/generate_grid.jl:6-35): node ids run with the last
    coordinate fastest, cell corners are looked up through a first-index-fastest id table."""
    dim = api._dim_of(eltype)
    origin = np.ones(dim) if origin is None else np.asarray(origin, dtype=np.float64)
    k = n + 1
    # node q (0-based) has multi-index (q // k^(d-1), ..., q % k): last coordinate fastest
    grid = np.indices((k,) * dim).reshape(dim, -1).T.astype(np.float64)
    nodes = scale * grid + origin
    # corner lookup: id(a, b, c) = a + k*b + k^2*c
    strides = k ** np.arange(dim)
    cube = np.indices((n,) * dim).reshape(dim, -1).T                      # first index slowest
    corner = np.array([[(c >> a) & 1 for a in range(dim)] for c in range(2 ** dim)])   # bit a -> +1 in axis a
    ids = (cube[:, None, :] + corner[None, :, :]) @ strides               # (ncubes, 2^dim)
    if dim == 3:
        cells = ids[:, np.array(_CUBE_TETS)].reshape(-1, 4)
    else:
        cells = ids[:, np.array(((0, 1, 2), (1, 2, 3)))].reshape(-1, 3)
    cells = np.sort(cells, axis=1) + 1                                     # 1-based, ascending tuples
    return Mesh(nodes, cells.astype(np.int64))


def box_mesh(eltype, shape, scale=1.0, origin=None) -> Mesh:
    """A box of shape[0] x shape[1] (x shape[2]) unit cubes, each split into 6 tetrahedra / 2 triangles around
    one diagonal like `hypercube` (node ids run with the last coordinate fastest).  Used for multi-GPU weak
    scaling, where the global domain is a brick of per-rank cubes; not a reference entry point."""
    dim = api._dim_of(eltype)
    shape = tuple(int(v) for v in shape)
    assert len(shape) == dim
    origin = np.ones(dim) if origin is None else np.asarray(origin, dtype=np.float64)
    k = np.array(shape) + 1
    grid = np.indices(tuple(k)).reshape(dim, -1).T.astype(np.float64)
    nodes = scale * grid + origin
    strides = np.concatenate([[1], np.cumprod(k[:-1])])
    cube = np.indices(shape).reshape(dim, -1).T
    corner = np.array([[(c >> a) & 1 for a in range(dim)] for c in range(2 ** dim)])
    ids = (cube[:, None, :] + corner[None, :, :]) @ strides
    # node ids run with the LAST coordinate fastest, the lookup table with the FIRST: translate
    lut = np.ravel_multi_index(np.unravel_index(np.arange(int(np.prod(k))), tuple(k), order="F"), tuple(k), order="C")
    ids = lut[ids]
    if dim == 3:
        cells = ids[:, np.array(_CUBE_TETS)].reshape(-1, 4)
    else:
        cells = ids[:, np.array(((0, 1, 2), (1, 2, 3)))].reshape(-1, 3)
    return Mesh(nodes, (np.sort(cells, axis=1) + 1).astype(np.int64))


def _infnorm(a):
    return np.abs(a).max(axis=-1)


def _centers(mesh: Mesh):
    p = mesh.nodes[mesh.elements - 1]
    acc = p[:, 0, :].copy()
    for i in range(1, p.shape[1]):
        acc += p[:, i, :]
    return acc / p.shape[1]


def order_nodes_and_elements_by_magnitude(mesh: Mesh) -> Mesh:
    """Nodes and cells sorted by infinity-norm distance to the origin (stable), so that every centred
    sub-cube is a prefix.  ref: src/examples/homogenized_coefficients.jl:21-28"""
    perm = np.argsort(_infnorm(mesh.nodes), kind="stable")
    inv = np.empty_like(perm)
    inv[perm] = np.arange(perm.size)
    cells = np.sort(inv[mesh.elements - 1], axis=1) + 1
    out = Mesh(mesh.nodes[perm], cells)
    order = np.argsort(_infnorm(_centers(out)), kind="stable")
    out.elements = np.ascontiguousarray(out.elements[order])
    return out


def find_elements_in_radius(mesh: Mesh, radius) -> int:
    return int(np.searchsorted(_infnorm(_centers(mesh)), radius, side="right"))


def find_nodes_in_radius(mesh: Mesh, radius) -> int:
    return int(np.searchsorted(_infnorm(mesh.nodes), radius + 10 * np.finfo(float).eps, side="right"))


def compute_boundary_layer(lam: float, n: int) -> int:
    return int(math.floor(4 * (n + 1) * lam ** -0.5))


def compute_box_radius(k: int, n: int, eps: float = 0.0) -> int:
    return int(math.floor(2.0 ** (n - k * (0.5 - eps))))


def generate_conductivity(dim: int, n: int, seed: int, values=(1.0, 9.0)):
    """One diagonal tensor per unit cube, every entry i.i.d. in `values` (p = 1/2).  The reference draws
    from the unseeded global RNG with values {1, 9} (src/examples/homogenized_coefficients.jl:485-488);
    here the field is seeded and the contrast is a parameter."""
    rng = np.random.default_rng(seed)
    return np.where(rng.random((n,) * dim + (dim,)) < 0.5, values[0], values[1])


def conductivity_per_element(mesh: Mesh, sigma_grid, offset):
    """ref: src/examples/homogenized_coefficients.jl:494-503"""
    idx = np.trunc(_centers(mesh) + np.asarray(offset, dtype=np.float64)).astype(np.int64) - 1
    return np.ascontiguousarray(sigma_grid[tuple(idx[:, a] for a in range(mesh.dim))])


def random_unit_vec(dim):
    v = np.ones(dim)
    return v / np.linalg.norm(v)


def checkerboard_problem(ctx, eltype, width: int, levels: int, seed: int = 0, values=(1.0, 9.0), lam: float = 1.0,
                         origin=None, ordered: bool = True):
    """Base mesh, coefficient field, implicit grid and operator for a width^d checkerboard."""
    dim = api._dim_of(eltype)
    if origin is None:
        origin = (-width / 2.0,) * dim
    base = hypercube(eltype, width, origin=origin)
    if ordered:
        base = order_nodes_and_elements_by_magnitude(base)
    sgrid = generate_conductivity(dim, width, seed, values)
    cond = conductivity_per_element(base, sgrid, tuple(1.0 - o for o in origin))
    implicit = api.ImplicitFineGrid(ctx, base, levels)
    op = api.L2PlusDivAGrad(implicit, lam, cond)
    return base, cond, implicit, op
