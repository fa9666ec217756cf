"""
Host-side driver pieces (level L4 of the reference: src/examples/homogenized_coefficients.jl) that sit
on either side of the hot path: base-mesh synthesis, the infinity-norm ordering that makes a domain
shrink a prefix operation, the checkerboard coefficient field and the `checkerboard_homogenization`
loop itself.  All level-vector work goes to the device through api.py; nothing here touches a level
vector on the host.
"""
from __future__ import annotations

import math
import os
import warnings

import numpy as np

from . import api, vtk
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


def checkerboard_mesh(eltype, shape, origin=None, transposed_lookup=None, ordered: bool = True) -> Mesh:
    """`hypercube` / `box_mesh` (+ `order_nodes_and_elements_by_magnitude` if `ordered`) by the library's threaded host
    code (hmg_checkerboard_mesh): the same arrays as the numpy functions here, which stay as the readable statement
    of the rule and as the test reference.  transposed_lookup: None = like `hypercube` for cubes, like `box_mesh` else."""
    import ctypes
    from . import _lib as L
    lib = L.load()
    dim = api._dim_of(eltype)
    shape = np.ascontiguousarray([int(v) for v in (shape if np.ndim(shape) else (shape,) * dim)], dtype=np.int64)
    assert shape.size == dim
    origin = np.ones(dim) if origin is None else np.ascontiguousarray(origin, dtype=np.float64)
    if transposed_lookup is None:
        transposed_lookup = bool(np.all(shape == shape[0]))
    nn, nc = ctypes.c_int64(), ctypes.c_int64()
    L.check(lib.hmg_checkerboard_mesh_size(dim, shape.ctypes.data_as(L.p_i64), ctypes.byref(nn), ctypes.byref(nc)))
    nodes = np.empty((nn.value, dim), dtype=np.float64)
    cells = np.empty((nc.value, dim + 1), dtype=np.int64)
    L.check(lib.hmg_checkerboard_mesh(dim, shape.ctypes.data_as(L.p_i64), origin.ctypes.data_as(L.p_f64),
                                      1 if transposed_lookup else 0, 1 if ordered else 0,
                                      nodes.ctypes.data_as(L.p_f64), cells.ctypes.data_as(L.p_i64)))
    return Mesh(nodes, cells)


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


def conductivity_per_element(mesh: Mesh, sigma_grid, offset, native: bool = True):
    """ref: src/examples/homogenized_coefficients.jl:494-503 (native: the library's threaded host code; False: numpy)"""
    if native:
        from . import _lib as L
        dim = mesh.dim
        nodes = np.ascontiguousarray(mesh.nodes, dtype=np.float64)
        cells = np.ascontiguousarray(mesh.elements, dtype=np.int64)
        sg = np.ascontiguousarray(sigma_grid, dtype=np.float64)
        assert sg.ndim == dim + 1 and sg.shape[-1] == dim
        gs = np.ascontiguousarray(sg.shape[:dim], dtype=np.int64)
        off = np.ascontiguousarray(offset, dtype=np.float64)
        out = np.empty((cells.shape[0], dim), dtype=np.float64)
        L.check(L.load().hmg_conductivity_per_element(dim, nodes.shape[0], nodes.ctypes.data_as(L.p_f64), cells.shape[0],
                                                      cells.ctypes.data_as(L.p_i64), gs.ctypes.data_as(L.p_i64),
                                                      sg.ctypes.data_as(L.p_f64), off.ctypes.data_as(L.p_f64),
                                                      out.ctypes.data_as(L.p_f64)))
        return out
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
    base = checkerboard_mesh(eltype, width, origin=origin, transposed_lookup=True, ordered=ordered)
    sgrid = generate_conductivity(dim, width, seed, values)
    cond = conductivity_per_element(base, sgrid, tuple(1.0 - o for o in origin))
    implicit = api.ImplicitFineGrid(ctx, base, levels)
    op = api.L2PlusDivAGrad(implicit, lam, cond)
    return base, cond, implicit, op


def checkerboard_homogenization(n: int = 4, eltype=Tri64, refinements: int = 2, smoothing_steps: int = 3,
                                tolerance: float = 1e-4, xi=None, save=None, *, ctx=None, seed: int = 0,
                                values=(1.0, 9.0), sigma_grid=None, x0=None, max_cycles: int = 1000, log=None,
                                timings: dict | None = None, tune_placement: int = 0):
    """checkerboard_homogenization(n, type; refinements, smoothing_steps, tolerance, xi, save) -> sigma
    (src/examples/homogenized_coefficients.jl:174-343) with every level-vector operation on the device.

    Differences to the reference, all explicit: the coefficient field and the initial guess are seeded
    (`seed`, or passed in as `sigma_grid` / `x0`) instead of drawn from the global RNG; `save` (a level, as in the
    reference: checkerboard.vtu + one ahom_k.vtu per outer step, see vtk.py) may also be a (level, directory) pair;
    the level-1 solve is the library's PCG; a domain shrink keeps the level vectors in place
    (their columns are a prefix) instead of copying slices.  Returns (sigma, history) where history holds
    (k, cycle, norm(r), sigma + dsigma, |dsigma - dsigma_prev|) -- the three quantities the reference logs.
    `timings` (a dict) receives wall-clock seconds: "setup_s" (mesh, tables, level vectors, x0, right-hand side -- up to
    the first V-cycle), "solve_s" (everything after), "vcycles", "outer_steps", "cells", and "inexact_vcycles": V-cycles whose
    level-1 solve ran out of its iteration budget (each also raises a warning; 0 in every recorded run).
    `tune_placement` = T > 0: the finest level's five memory blocks are assigned to their roles by measurement
    (api.tune_placement, T candidates; pays off for long runs only -- about 0.1 s per candidate at config 3)."""
    import time
    t_start = time.perf_counter()
    save_dir = "."
    if isinstance(save, tuple):
        save, save_dir = save
    dim = api._dim_of(eltype)
    own_ctx = ctx is None
    if own_ctx:
        ctx = api.Context(0)
    xi = random_unit_vec(dim) if xi is None else np.asarray(xi, dtype=np.float64)
    lam, sigma = 1.0, 0.0
    box_radius = compute_box_radius(0, n)
    boundary_layer = compute_boundary_layer(lam, n)
    total_radius = box_radius + boundary_layer
    width = 2 * total_radius
    base = checkerboard_mesh(eltype, width, origin=(-float(total_radius),) * dim, transposed_lookup=True)
    if sigma_grid is None:
        sigma_grid = generate_conductivity(dim, width, seed, values)
    cond = conductivity_per_element(base, sigma_grid, (total_radius + 1.0,) * dim)
    t_mesh = time.perf_counter()
    total_grids = refinements + 1
    if save is not None:
        if not 1 <= save <= total_grids:
            raise ValueError("save must be a level in 1..refinements+1")
        vtk.export_domain(base, cond, os.path.join(save_dir, "checkerboard"))
    implicit = api.ImplicitFineGrid(ctx, base, total_grids)
    op = api.L2PlusDivAGrad(implicit, lam, cond)
    ops = [op] * total_grids
    t_grid = time.perf_counter()
    states = [api.LevelState(implicit, i + 1) for i in range(total_grids)]
    top = states[-1]
    if tune_placement:
        tuned = api.tune_placement(implicit, ops, states, total_grids, smoothing_steps, trials=int(tune_placement))
        if timings is not None:
            timings["tune_ms"] = tuned
    ctx.sync()
    t_alloc = time.perf_counter()
    if x0 is None:
        top.x.rand(seed + 1)
    else:
        top.x.from_host(x0)
    api.broadcast_interfaces(top.x, implicit, total_grids)
    api.apply_constraint(top.x, total_grids, implicit)
    api.rhs_axi_grad_v(top.b, implicit, xi)
    v_prev = None                                        # allocated at the first domain shrink (10 GB at config 3)
    cur = base
    history = []
    inexact = 0                                          # V-cycles whose budgeted level-1 solve missed coarse_rtol
    ctx.sync()
    t_setup = time.perf_counter()
    for k in range(n + 1):
        base_level = api.BaseLevel(implicit)             # level-1 operator for the current lam / domain
        dsig, dsig_prev = 0.0, 0.0
        for i in range(1, max_cycles + 1):
            if not api.vcycle_tolerant(implicit, base_level, ops, states, total_grids, smoothing_steps):
                # the level-1 solve ran out of its blind iteration budget: this cycle's coarse-grid correction was inexact (a weaker
                # but valid iterate; the library counts the next solve again).  The reference's CHOLMOD solve cannot miss -- say so.
                inexact += 1
                warnings.warn(f"checkerboard_homogenization: V-cycle {i} of outer step {k} used an inexact level-1 solve "
                              f"({inexact} so far)")
            nint = find_elements_in_radius(cur, box_radius)
            area = api.integrate_area(top.x, implicit, nint)
            if k == 0:
                integral = api.integrate_first_term(top.x, implicit, nint, xi, b=top.b)   # b = rhs_a.xi.grad(v) at k = 0
            else:
                integral = api.integrate_terms(top.x, v_prev, implicit, nint)
            dsig = 2.0 ** k * integral / area
            rnorm = api.norm_unique(top.r)
            history.append((k, i, rnorm, sigma + dsig, abs(dsig - dsig_prev)))
            if log:
                log(history[-1])
            if abs(dsig - dsig_prev) < tolerance:
                break
            dsig_prev = dsig
        sigma += dsig
        if save is not None:                             # ref: ...homogenized_coefficients.jl:303
            vtk.export_unknown(implicit, top.x, k, save, os.path.join(save_dir, f"ahom_{k}"))
        lam /= 2
        box_radius = compute_box_radius(k + 1, n)
        boundary_layer = compute_boundary_layer(lam, n)
        if box_radius + boundary_layer > total_radius:
            break
        total_radius = box_radius + boundary_layer
        nn_keep = find_nodes_in_radius(cur, total_radius)
        ne_keep = find_elements_in_radius(cur, total_radius)
        cur = Mesh(cur.nodes[:nn_keep], np.ascontiguousarray(cur.elements[:ne_keep]))
        implicit.shrink(ne_keep, nn_keep)                # new boundary; level vectors keep their storage
        api.apply_constraint(top.x, total_grids, implicit)
        if v_prev is None:
            v_prev = api.DeviceMatrix(top.x.implicit, total_grids)
        v_prev.copyto(top.x)
        op.lam = lam
        api.next_rhs(top.b, top.x, implicit)
    ctx.sync()
    if timings is not None:
        timings.update(setup_s=t_setup - t_start, setup_mesh_s=t_mesh - t_start, setup_tables_s=t_grid - t_mesh,
                       setup_alloc_s=t_alloc - t_grid, setup_init_s=t_setup - t_alloc,
                       solve_s=time.perf_counter() - t_setup, vcycles=len(history),
                       outer_steps=len({h[0] for h in history}), cells=int(base.elements.shape[0]), width=int(width),
                       inexact_vcycles=inexact)
    # the level vectors go back now, not whenever the collector gets to them (71 GB at BASELINE config 3)
    for st in states:
        st.close()
    if v_prev is not None:
        v_prev.close()
    implicit.close()
    if own_ctx:
        ctx.close()
    return sigma, history


def checkerboard_hypercube_multigrid(n: int, eltype=Tet64, refinements: int = 2, max_cycles: int = 5, save=None, *,
                                     ctx=None, seed: int = 1, sigma_grid=None, x0=None):
    """checkerboard_hypercube_multigrid(n, elementtype, refinements, max_cycles, save) -> residual norms
    (src/examples/homogenized_coefficients.jl:509-571): -div(a grad u) = 1 with zero Dirichlet data (lambda = 0),
    `max_cycles` V-cycles with 3 smoothing steps; `refinements` is the number of grids.  Seeded like the other
    driver; `save` = level (or (level, directory)) writes checkerboard_full_<refinements>.vtu with point data "x".
    Returns (rs, state of the finest level, implicit grid)."""
    dim = api._dim_of(eltype)
    own_ctx = ctx is None
    if own_ctx:
        ctx = api.Context(0)
    base = hypercube(eltype, n)
    if sigma_grid is None:
        sigma_grid = generate_conductivity(dim, n, seed)
    cond = conductivity_per_element(base, sigma_grid, (0.0,) * dim)
    implicit = api.ImplicitFineGrid(ctx, base, refinements)
    op = api.L2PlusDivAGrad(implicit, 0.0, cond)
    base_level = api.BaseLevel(implicit)
    states = [api.LevelState(implicit, i + 1) for i in range(refinements)]
    top = states[-1]
    if x0 is None:
        top.x.rand(seed + 1)
    else:
        top.x.from_host(x0)
    api.broadcast_interfaces(top.x, implicit, refinements)
    api.apply_constraint(top.x, refinements, implicit)
    api.local_rhs(top.b, implicit)
    rs = []
    for _ in range(max_cycles):
        api.vcycle(implicit, base_level, [op] * refinements, states, refinements, 3)
        rs.append(api.norm_unique(top.r))
    if save is not None:
        save_dir = "."
        if isinstance(save, tuple):
            save, save_dir = save
        vtk.export_unknown(implicit, top.x, 0, save, os.path.join(save_dir, f"checkerboard_full_{refinements}"),
                           field="x")
    if own_ctx:
        ctx.sync()
    return rs, top, implicit
