"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

numpy/scipy restatement of the *setup* side (meshes, refinement, numbering, interface maps,
operator tables, drivers) of haampie/Homogenization.jl, plus ctypes bindings to the plain-C hot
loops in oracle/hmg_oracle.c. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module. The product (homogenization.jl_amd/) never does.

Every function cites the reference file:line it follows (paths relative to the reference
checkout, e.g. src/multigrid.jl:46-71).  Indices are 0-based here, the reference is 1-based; all
*orders* (node numbering, list orders, loop orders) are the reference's.

PARITY PIN.  The reference is pure Julia and no Julia toolchain exists in the build container,
so the reference cannot be executed here and it ships no golden data files.  The oracle is pinned
by re-stating the reference's own known-answer tests against it (tests/test_oracle_reference_kats.py):
  test/test_operator.jl:9-73            implicit apply + interface sum == assembled matrix, <= 20 eps
  test/interpolation.jl:8-35            affine reproduction under prolongation through 6 levels
  test/refined_reference_element.jl     node counts, base face/edge numbering, per-level counts
  test/implicit_grid.jl:8-92            interface nodes coincide geometrically in every adjacent cell
  test/list_faces.jl:6-27               boundary counts 4/6/4 and 64/96/34
  test/sparse_cell_to_element.jl:4-27   exact `compress` output
  test/generated_grids.jl:4-10          hypercube(Tet, 20): sorted, 21^3 nodes, 6*20^3 cells
Not pinned by any reference test (none exists): smoothing_steps!/vcycle!/coarse solve and the
driver; those are restated from the source only ("parity unpinned" for those rows).
What stands in for a reference-held pin there are statements that share no code path with this file's
cell-local machinery (tests/_global_form.py, tests/_textbook_fem.py): the V-cycle written on global vectors
with matrices assembled on explicitly refined meshes and multiplicity-weighted dots (1e-11), the driver's
right-hand sides and integrals from textbook P1 elements (1e-11), and the converged n = 0 driver against a
direct solve of the boundary value problem (1e-9).  They pin the EXECUTION of the algorithm; which algorithm
the reference runs (its dots, its `steps`, its sigma formula) is pinned by the line citations alone.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i64p = ctypes.POINTER(ctypes.c_int64)
_f64p = ctypes.POINTER(ctypes.c_double)


def _p(a):
    if a.dtype == np.int64:
        return a.ctypes.data_as(_i64p)
    assert a.dtype == np.float64, a.dtype
    return a.ctypes.data_as(_f64p)


def build_lib(force: bool = False) -> str:
    """Compile oracle/hmg_oracle.c -> oracle/libhmg_oracle.so (gcc -O3 -fopenmp)."""
    src = os.path.join(_HERE, "hmg_oracle.c")
    # -march=native code is host specific: key the file name on the CPU's feature flags so a
    # snapshot built elsewhere is rebuilt instead of faulting with an illegal instruction.
    tag = "generic"
    try:
        import hashlib
        with open("/proc/cpuinfo") as f:
            flags = next((l for l in f if l.startswith("flags")), "")
        tag = hashlib.sha1(flags.encode()).hexdigest()[:10]
    except OSError:
        pass
    out = os.path.join(_HERE, f"libhmg_oracle_{tag}.so")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(
            ["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-shared", "-std=c11",
             "-ffp-contract=off", "-o", out, src, "-lm"])
    return out


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build_lib())
        _LIB.orc_dot.restype = ctypes.c_double
        _LIB.orc_max_threads.restype = ctypes.c_int
    return _LIB


# --------------------------------------------------------------------------------------------
# Mesh  (ref: src/grid.jl:19-22)
# --------------------------------------------------------------------------------------------
@dataclass
class Mesh:
    nodes: np.ndarray      # (Nn, dim) float64
    elements: np.ndarray   # (Ne, dim+1) int64, 0-based

    @property
    def dim(self):
        return self.nodes.shape[1]

    def nnodes(self):
        return self.nodes.shape[0]

    def nelements(self):
        return self.elements.shape[0]


def sort_element_nodes(elements):
    """ref: src/sorting_tricks.jl:34-39"""
    return np.sort(elements, axis=1)


def _local_pairs(n):
    return [(i, j) for i in range(n) for j in range(i + 1, n)]


def edge_graph(mesh: Mesh):
    """Sorted unique edges (from < to), ordered by (from asc, to asc).
    ref: src/sparse_graph.jl:20-48.  Returns (edges (E,2), ptr (Nn+1))."""
    els = mesh.elements
    pairs = _local_pairs(els.shape[1])
    a = np.concatenate([els[:, i] for i, _ in pairs])
    b = np.concatenate([els[:, j] for _, j in pairs])
    lo = np.minimum(a, b)
    hi = np.maximum(a, b)
    nn = mesh.nnodes()
    key = np.unique(lo * nn + hi)
    edges = np.stack([key // nn, key % nn], axis=1)
    ptr = np.zeros(nn + 1, dtype=np.int64)
    np.add.at(ptr, edges[:, 0] + 1, 1)
    ptr = np.cumsum(ptr)
    return edges, ptr


def _edge_index(edges, nn, a, b):
    """ref: src/sparse_graph.jl:14-15 (edge_index): natural index of sorted edge (a<b)."""
    keys = edges[:, 0] * nn + edges[:, 1]
    q = np.minimum(a, b) * nn + np.maximum(a, b)
    idx = np.searchsorted(keys, q)
    assert np.all(keys[idx] == q)
    return idx


_TET_CHILDREN = np.array([(1, 5, 6, 7), (5, 2, 8, 9), (6, 8, 3, 10), (7, 9, 10, 4),
                          (5, 6, 7, 9), (5, 6, 8, 9), (6, 7, 9, 10), (6, 8, 9, 10)]) - 1


def refine_uniformly(mesh: Mesh, graph=None, times: int = 1) -> Mesh:
    """Red refinement. ref: src/tet/refine.jl:5-54, src/tri/refine.jl:5-43, src/grid.jl:59-64."""
    if graph is None:
        for _ in range(times):
            mesh = refine_uniformly(mesh, edge_graph(mesh))
        return mesh
    edges, _ = graph
    nn = mesh.nnodes()
    els = mesh.elements
    # Split the edges: new node per edge, in edge order (tet/refine.jl:16-21)
    nodes = np.concatenate([mesh.nodes, (mesh.nodes[edges[:, 0]] + mesh.nodes[edges[:, 1]]) / 2])
    if mesh.dim == 3:
        parts = np.empty((els.shape[0], 10), dtype=np.int64)
        parts[:, :4] = els
        for c, (i, j) in enumerate(_local_pairs(4)):   # (1,2),(1,3),(1,4),(2,3),(2,4),(3,4)
            parts[:, 4 + c] = _edge_index(edges, nn, els[:, i], els[:, j]) + nn
        new = parts[:, _TET_CHILDREN].reshape(-1, 4)   # tet/refine.jl:46-47
    else:
        a = _edge_index(edges, nn, els[:, 0], els[:, 1]) + nn
        b = _edge_index(edges, nn, els[:, 0], els[:, 2]) + nn
        c = _edge_index(edges, nn, els[:, 1], els[:, 2]) + nn
        t1, t2, t3 = els[:, 0], els[:, 1], els[:, 2]
        new = np.stack([np.stack([t1, a, b], 1), np.stack([t2, c, a], 1),
                        np.stack([t3, b, c], 1), np.stack([a, c, b], 1)], axis=1).reshape(-1, 3)
        new = np.sort(new, axis=1)                      # tri/refine.jl:35-38 (sort_bitonic)
    return Mesh(nodes, new)


def hypercube(dim: int, n: int, scale=1.0, origin=None, sorted_: bool = True) -> Mesh:
    """ref: src/tet/generate_grid.jl:6-45, src/tri/generate_grid.jl:6-35.
    NB: node ids are assigned with the LAST loop variable fastest, while `nn` is a column-major
    reshape (first index fastest) -- followed literally."""
    if origin is None:
        origin = (1.0,) * dim
    origin = np.asarray(origin, dtype=np.float64)
    g = np.arange(n + 1, dtype=np.float64)
    if dim == 3:
        X, Y, Z = np.meshgrid(g, g, g, indexing="ij")           # x outer ... z inner
        nodes = np.stack([scale * X.ravel() + origin[0], scale * Y.ravel() + origin[1],
                          scale * Z.ravel() + origin[2]], axis=1)
        nn = np.arange((n + 1) ** 3, dtype=np.int64).reshape((n + 1, n + 1, n + 1), order="F")
        x, y, z = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
        x, y, z = x.ravel(), y.ravel(), z.ravel()               # x outer, z inner loop order
        n1 = nn[x, y, z]; n2 = nn[x + 1, y, z]; n3 = nn[x, y + 1, z]; n4 = nn[x + 1, y + 1, z]
        n5 = nn[x, y, z + 1]; n6 = nn[x + 1, y, z + 1]; n7 = nn[x, y + 1, z + 1]
        n8 = nn[x + 1, y + 1, z + 1]
        els = np.stack([np.stack(t, 1) for t in (
            (n1, n2, n3, n7), (n1, n2, n5, n7), (n2, n4, n3, n7),
            (n2, n4, n7, n8), (n2, n6, n5, n7), (n2, n6, n7, n8))], axis=1).reshape(-1, 4)
    else:
        X, Y = np.meshgrid(g, g, indexing="ij")
        nodes = np.stack([scale * X.ravel() + origin[0], scale * Y.ravel() + origin[1]], axis=1)
        nn = np.arange((n + 1) ** 2, dtype=np.int64).reshape((n + 1, n + 1), order="F")
        x, y = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
        x, y = x.ravel(), y.ravel()
        n1 = nn[x, y]; n2 = nn[x + 1, y]; n3 = nn[x, y + 1]; n4 = nn[x + 1, y + 1]
        els = np.stack([np.stack((n1, n2, n3), 1), np.stack((n2, n3, n4), 1)], axis=1).reshape(-1, 3)
    if sorted_:
        els = sort_element_nodes(els)
    return Mesh(nodes, els.astype(np.int64))


# --------------------------------------------------------------------------------------------
# Cell values (ref: src/cell_values.jl:104-127, src/grid.jl:120-135)
# --------------------------------------------------------------------------------------------
def cell_geometry(mesh: Mesh):
    """Per element: J (affine map), Jinv = inv(J'), detJ = |det J|.
    Returns J (Ne,dim,dim), Jinv (Ne,dim,dim), detJ (Ne,)."""
    p = mesh.nodes[mesh.elements]                    # (Ne, dim+1, dim)
    J = np.transpose(p[:, 1:, :] - p[:, :1, :], (0, 2, 1))   # columns p_k - p_1
    dim = mesh.dim
    if dim == 2:
        a, b, c, d = J[:, 0, 0], J[:, 0, 1], J[:, 1, 0], J[:, 1, 1]
        det = a * d - b * c
        inv = np.empty_like(J)
        inv[:, 0, 0] = d / det; inv[:, 0, 1] = -b / det
        inv[:, 1, 0] = -c / det; inv[:, 1, 1] = a / det
    else:
        x0, x1, x2 = J[:, :, 0], J[:, :, 1], J[:, :, 2]       # columns
        y0 = np.cross(x1, x2)
        det = np.einsum("ij,ij->i", x0, y0)
        y1 = np.cross(x2, x0)
        y2 = np.cross(x0, x1)
        inv = np.stack([y0, y1, y2], axis=1) / det[:, None, None]   # rows y_i / det
    Jinv = np.transpose(inv, (0, 2, 1))              # inv(J') = inv(J)'
    return J, np.ascontiguousarray(Jinv), np.abs(det)


_REF_GRADS = {2: np.array([[-1.0, 1.0, 0.0], [-1.0, 0.0, 1.0]]),
              3: np.array([[-1.0, 1.0, 0.0, 0.0], [-1.0, 0.0, 1.0, 0.0], [-1.0, 0.0, 0.0, 1.0]])}


def quad_rule(dim):
    """ref: src/cell_values.jl:10-28 (TetQuad4, TriQuad3). Returns points (nq,dim), weights."""
    if dim == 3:
        s5 = np.sqrt(5.0)
        a, b = (5.0 + 3.0 * s5) / 20.0, (5.0 - s5) / 20.0
        pts = np.array([[a, b, b], [b, a, b], [b, b, a], [b, b, b]])
        w = np.full(4, 1.0 / 24.0)
    else:
        pts = np.array([[0.0, 0.5], [0.5, 0.0], [0.5, 0.5]])
        w = np.full(3, 1.0 / 6.0)
    return pts, w


def basis_values(dim):
    """phi_i at quad points: (nq, dim+1). ref: src/cell_values.jl:40-51,83"""
    pts, _ = quad_rule(dim)
    return np.concatenate([1.0 - pts.sum(1, keepdims=True), pts], axis=1)


def _sparse_sum(I, J, V, n):
    """Julia `sparse(I,J,V,n,n)` (duplicates summed) followed by dropzeros!"""
    A = sp.coo_matrix((V, (I, J)), shape=(n, n)).tocsc()
    A.sum_duplicates()
    A.eliminate_zeros()
    A.sort_indices()
    return A


def build_local_diffusion_operators(mesh: Mesh):
    """ops[k][l] = int d_k phi_i d_l phi_j on the refined reference element (k,l 0-based), CSC.
    ref: src/build_local_operators.jl:51-105.  Note the reference's A_locals[l,k][i,j] +=
    w * grad_i[k] * grad_j[l] and the final comprehension [... Is[l,k] ... for k, l], so that
    ops[k,l][i,j] = sum_q w * grad_i[k] * grad_j[l] * det."""
    dim = mesh.dim
    N = dim + 1
    _, Jinv, det = cell_geometry(mesh)
    _, w = quad_rule(dim)
    grads = np.einsum("eab,bn->ean", Jinv, _REF_GRADS[dim])   # (Ne, dim, N): gradients = Jinv*ref
    els = mesh.elements
    I = np.repeat(els[:, :, None], N, axis=2).reshape(-1)     # element[i]
    Jc = np.repeat(els[:, None, :], N, axis=1).reshape(-1)    # element[j]
    ops = [[None] * dim for _ in range(dim)]
    for k in range(dim):
        for l in range(dim):
            acc = np.zeros((els.shape[0], N, N))
            for q in range(len(w)):
                acc += w[q] * grads[:, k, :, None] * grads[:, l, None, :]
            V = (acc * det[:, None, None]).reshape(-1)
            ops[k][l] = _sparse_sum(I, Jc, V, mesh.nnodes())
    return ops


def mass_matrix(mesh: Mesh):
    """ref: src/build_local_operators.jl:107-141"""
    dim = mesh.dim
    N = dim + 1
    _, _, det = cell_geometry(mesh)
    _, w = quad_rule(dim)
    phi = basis_values(dim)
    loc = np.zeros((N, N))
    for q in range(len(w)):
        loc += w[q] * np.outer(phi[q], phi[q])
    els = mesh.elements
    I = np.repeat(els[:, :, None], N, axis=2).reshape(-1)
    Jc = np.repeat(els[:, None, :], N, axis=1).reshape(-1)
    V = (loc[None, :, :] * det[:, None, None]).reshape(-1)
    return _sparse_sum(I, Jc, V, mesh.nnodes())


def assemble_matrix_dot(mesh: Mesh):
    """assemble_matrix(mesh, dot): int grad u . grad v.  ref: src/assembly.jl:4-60"""
    dim = mesh.dim
    N = dim + 1
    _, Jinv, det = cell_geometry(mesh)
    _, w = quad_rule(dim)
    grads = np.einsum("eab,bn->ean", Jinv, _REF_GRADS[dim])
    gg = np.einsum("eki,ekj->eij", grads, grads)
    acc = np.zeros_like(gg)
    for q in range(len(w)):
        acc += w[q] * gg
    els = mesh.elements
    I = np.repeat(els[:, :, None], N, axis=2).reshape(-1)
    Jc = np.repeat(els[:, None, :], N, axis=1).reshape(-1)
    V = (acc * det[:, None, None]).reshape(-1)
    A = sp.coo_matrix((V, (I, Jc)), shape=(mesh.nnodes(),) * 2).tocsc()
    A.sum_duplicates()
    return A


def assemble_checkerboard(mesh: Mesh, sigmas, lam=1.0):
    """B[u,v] = int lam u v + sigma grad u . grad v.
    ref: src/examples/homogenized_coefficients.jl:358-402"""
    dim = mesh.dim
    N = dim + 1
    _, Jinv, det = cell_geometry(mesh)
    _, w = quad_rule(dim)
    phi = basis_values(dim)
    grads = np.einsum("eab,bn->ean", Jinv, _REF_GRADS[dim])
    gsg = np.einsum("eki,ek,ekj->eij", grads, sigmas, grads)
    acc = np.zeros_like(gsg)
    for q in range(len(w)):
        acc += w[q] * (lam * np.outer(phi[q], phi[q])[None] + gsg)
    els = mesh.elements
    I = np.repeat(els[:, :, None], N, axis=2).reshape(-1)
    Jc = np.repeat(els[:, None, :], N, axis=1).reshape(-1)
    V = (acc * det[:, None, None]).reshape(-1)
    A = sp.coo_matrix((V, (I, Jc)), shape=(mesh.nnodes(),) * 2).tocsc()
    A.sum_duplicates()
    return A


def assemble_vector(mesh: Mesh, functional=lambda v: v):
    """b[i] = int functional(phi_i), 4-pt / 3-pt quadrature.  ref: src/assembly.jl:121-155"""
    dim = mesh.dim
    _, _, det = cell_geometry(mesh)
    _, w = quad_rule(dim)
    phi = basis_values(dim)
    b_local = np.zeros(dim + 1)
    for q in range(len(w)):                       # qp outer, i inner: the reference's accumulation order
        b_local += w[q] * functional(phi[q])
    b = np.zeros(mesh.nnodes())
    np.add.at(b, mesh.elements.reshape(-1), (b_local[None, :] * det[:, None]).reshape(-1))
    return b


def partial_derivatives_functionals(mesh: Mesh):
    """bs[node, j] = int d phi_node / d x_j. ref: ...homogenized_coefficients.jl:407-442"""
    dim = mesh.dim
    _, Jinv, det = cell_geometry(mesh)
    _, w = quad_rule(dim)
    grads = np.einsum("eab,bn->ean", Jinv, _REF_GRADS[dim])   # (Ne, dim, N)
    loc = np.zeros_like(grads)
    for q in range(len(w)):
        loc += w[q] * grads
    loc = loc * det[:, None, None]
    bs = np.zeros((mesh.nnodes(), dim))
    for i in range(dim + 1):
        np.add.at(bs, mesh.elements[:, i], loc[:, :, i])
    return bs


# --------------------------------------------------------------------------------------------
# Reference element hierarchy  (ref: src/multilevel_reference.jl)
# --------------------------------------------------------------------------------------------
@dataclass
class ReferenceNumbering:
    faces: list
    faces_interior: list
    edges: list
    edges_interior: list
    nodes: np.ndarray


def _is_on_edge(a, b, pts):
    """ref: src/multilevel_reference.jl:83-101 (IsOnEdge)"""
    d = b - a
    unit = d / np.linalg.norm(d)
    vec = pts - a
    proj = vec @ unit
    return np.abs(proj * proj - np.einsum("ij,ij->i", vec, vec)) < 1e-7


def get_local_numbering(m: Mesh) -> ReferenceNumbering:
    """ref: src/multilevel_reference.jl:125-203"""
    eps = np.finfo(np.float64).eps
    x = m.nodes
    if m.dim == 3:
        ref = np.array([[0.0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]])
        faces = [np.flatnonzero(x[:, 2] == 0), np.flatnonzero(x[:, 1] == 0),
                 np.flatnonzero(x[:, 0] == 0), np.flatnonzero(x.sum(1) >= 1 - 10 * eps)]
        src = [(0, 1, 0), (0, 2, 0), (0, 3, 1), (1, 2, 3), (1, 3, 3), (2, 3, 3)]  # (a,b,face)
        edges = [faces[f][_is_on_edge(ref[a], ref[b], x[faces[f]])] for a, b, f in src]
        nodes = np.arange(4)
        fi = [f.copy() for f in faces]
        for f, es in enumerate([(0, 1, 3), (0, 2, 4), (1, 2, 5), (3, 4, 5)]):
            for e in es:
                fi[f] = np.setdiff1d(fi[f], edges[e], assume_unique=True)
        ei = [np.setdiff1d(e, nodes, assume_unique=True) for e in edges]
        return ReferenceNumbering(faces, fi, edges, ei, nodes)
    edges = [np.flatnonzero(x[:, 1] == 0), np.flatnonzero(x[:, 0] == 0),
             np.flatnonzero(x.sum(1) >= 1 - 10 * eps)]
    nodes = np.arange(3)
    ei = [np.setdiff1d(e, nodes, assume_unique=True) for e in edges]
    return ReferenceNumbering([np.zeros(0, np.int64)], [np.zeros(0, np.int64)], edges, ei, nodes)


def interpolation_operator(mesh: Mesh, graph):
    """P (fine x coarse) CSC: identity on old nodes, 1/2,1/2 on edge midpoints.
    ref: src/interpolation.jl:7-50"""
    edges, _ = graph
    nn, ne = mesh.nnodes(), edges.shape[0]
    rows = np.concatenate([np.arange(nn), np.repeat(nn + np.arange(ne), 2)])
    cols = np.concatenate([np.arange(nn), edges.reshape(-1)])
    vals = np.concatenate([np.ones(nn), np.full(2 * ne, 0.5)])
    P = sp.csc_matrix((vals, (rows, cols)), shape=(nn + ne, nn))
    P.sort_indices()
    return P


@dataclass
class MultilevelReference:
    levels: list
    numbering: list
    interops: list


def reference_element(dim) -> Mesh:
    """ref: src/multilevel_reference.jl:3-13"""
    if dim == 2:
        return Mesh(np.array([[0.0, 0], [1, 0], [0, 1]]), np.array([[0, 1, 2]], dtype=np.int64))
    return Mesh(np.array([[0.0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]]),
                np.array([[0, 1, 2, 3]], dtype=np.int64))


def refined_element(n: int, dim: int) -> MultilevelReference:
    """ref: src/multilevel_reference.jl:41-61"""
    levels = [reference_element(dim)]
    numbering = [get_local_numbering(levels[0])]
    interops = []
    for i in range(n - 1):
        graph = edge_graph(levels[i])
        levels.append(refine_uniformly(levels[i], graph))
        numbering.append(get_local_numbering(levels[i + 1]))
        interops.append(interpolation_operator(levels[i], graph))
    for m in levels:
        m.elements = sort_element_nodes(m.elements)
    return MultilevelReference(levels, numbering, interops)


# --------------------------------------------------------------------------------------------
# Interfaces (ref: src/interface.jl)
# --------------------------------------------------------------------------------------------
@dataclass
class SparseCellToElementMap:
    """ref: src/interface.jl:31-35.  offset is 0-based CSR pointer; values split into arrays."""
    offset: np.ndarray
    cells: np.ndarray       # (ncells_ent, N)
    element: np.ndarray
    local_id: np.ndarray

    def __len__(self):
        return self.cells.shape[0]


TET_FACES = ((0, 1, 2), (0, 1, 3), (0, 2, 3), (1, 2, 3))
TET_EDGES = ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))
TRI_EDGES = ((0, 1), (0, 2), (1, 2))


def _list_with_element(mesh: Mesh, local_tuples):
    """ref: src/interface.jl:124-197 (list_*_with_element): element-major listing."""
    els = mesh.elements
    ne = els.shape[0]
    k = len(local_tuples)
    nodes = np.stack([els[:, list(t)] for t in local_tuples], axis=1).reshape(ne * k, -1)
    element = np.repeat(np.arange(ne, dtype=np.int64), k)
    local_id = np.tile(np.arange(k, dtype=np.int64), ne)
    return nodes, element, local_id


def _stable_lexsort(nodes):
    """LSD radix sort on tuples == stable lexicographic sort. ref: src/sorting_tricks.jl:44-74"""
    return np.lexsort(tuple(nodes[:, d] for d in range(nodes.shape[1] - 1, -1, -1)))


def _group_starts(nodes):
    if nodes.shape[0] == 0:
        return np.zeros(0, dtype=bool)
    new = np.ones(nodes.shape[0], dtype=bool)
    new[1:] = np.any(nodes[1:] != nodes[:-1], axis=1)
    return new


def compress(nodes, element, local_id) -> SparseCellToElementMap:
    """ref: src/interface.jl:317-351"""
    if nodes.shape[0] == 0:
        return SparseCellToElementMap(np.zeros(1, np.int64), nodes.reshape(0, max(nodes.shape[1], 1)),
                                      element, local_id)
    new = _group_starts(nodes)
    starts = np.flatnonzero(new)
    offset = np.concatenate([starts, [nodes.shape[0]]]).astype(np.int64)
    return SparseCellToElementMap(offset, nodes[starts], element.copy(), local_id.copy())


def _group_counts(nodes):
    new = _group_starts(nodes)
    gid = np.cumsum(new) - 1
    counts = np.bincount(gid) if nodes.shape[0] else np.zeros(0, np.int64)
    return gid, counts


def _sorted_list(mesh, tuples):
    nodes, el, lid = _list_with_element(mesh, tuples)
    o = _stable_lexsort(nodes)
    return nodes[o], el[o], lid[o]


def _remove_singletons(nodes, el, lid):
    """ref: src/sorting_tricks.jl:130-154"""
    gid, counts = _group_counts(nodes)
    keep = counts[gid] > 1
    return nodes[keep], el[keep], lid[keep]


@dataclass
class Interfaces:
    all_nodes: SparseCellToElementMap
    nodes: SparseCellToElementMap
    edges: SparseCellToElementMap
    faces: SparseCellToElementMap


def interfaces(mesh: Mesh) -> Interfaces:
    """ref: src/interface.jl:65-117"""
    N = mesh.elements.shape[1]
    nl = _sorted_list(mesh, [(i,) for i in range(N)])
    all_nodes = compress(*nl)
    nodes = compress(*_remove_singletons(*nl))
    edges = compress(*_remove_singletons(*_sorted_list(mesh, TET_EDGES if N == 4 else TRI_EDGES)))
    if N == 4:
        faces = compress(*_remove_singletons(*_sorted_list(mesh, TET_FACES)))
    else:
        faces = compress(np.zeros((0, 3), np.int64), np.zeros(0, np.int64), np.zeros(0, np.int64))
    return Interfaces(all_nodes, nodes, edges, faces)


def _intersect(nodes, el, lid, w):
    """Keep entries of the sorted list whose tuple occurs in sorted unique `w`.
    ref: src/interface.jl:291-309"""
    if nodes.shape[0] == 0 or w.shape[0] == 0:
        return nodes[:0], el[:0], lid[:0]
    base = int(max(nodes.max(), w.max())) + 1
    def key(a):
        k = np.zeros(a.shape[0], dtype=np.int64)
        for d in range(a.shape[1]):
            k = k * base + a[:, d]
        return k
    keep = np.isin(key(nodes), key(w))
    return nodes[keep], el[keep], lid[keep]


def list_boundary_nodes_edges_faces(m: Mesh):
    """ref: src/interface.jl:207-284.  Returns (nodes, edges, faces) maps."""
    if m.dim == 3:
        fn, fe, fl = _sorted_list(m, TET_FACES)
        gid, counts = _group_counts(fn)
        keep = counts[gid] == 1                       # remove_repeated_pairs! (faces occur 1x or 2x)
        fn, fe, fl = fn[keep], fe[keep], fl[keep]
        be = np.concatenate([fn[:, [0, 1]], fn[:, [0, 2]], fn[:, [1, 2]]])
        be = np.unique(be, axis=0)
        en, ee, el_ = _intersect(*_sorted_list(m, TET_EDGES), be)
        bn = np.unique(be.reshape(-1))[:, None]
        nn_, ne_, nl_ = _intersect(*_sorted_list(m, [(i,) for i in range(4)]), bn)
        return compress(nn_, ne_, nl_), compress(en, ee, el_), compress(fn, fe, fl)
    en, ee, el_ = _sorted_list(m, TRI_EDGES)
    gid, counts = _group_counts(en)
    keep = counts[gid] == 1
    en, ee, el_ = en[keep], ee[keep], el_[keep]
    bn = np.unique(en.reshape(-1))[:, None]
    nn_, ne_, nl_ = _intersect(*_sorted_list(m, [(i,) for i in range(3)]), bn)
    empty = compress(np.zeros((0, 3), np.int64), np.zeros(0, np.int64), np.zeros(0, np.int64))
    return compress(nn_, ne_, nl_), compress(en, ee, el_), empty


def list_interior_nodes(m: Mesh):
    """ref: src/grid.jl:176-202"""
    tuples = TET_FACES if m.dim == 3 else TRI_EDGES
    fn, _, _ = _sorted_list(m, tuples)
    gid, counts = _group_counts(fn)
    bnd = np.unique(fn[counts[gid] == 1].reshape(-1))
    return np.setdiff1d(np.arange(m.nnodes()), bnd, assume_unique=True)


# --------------------------------------------------------------------------------------------
# Implicit fine grid + level-vector kernels
# --------------------------------------------------------------------------------------------
def _csr_of_lists(lists):
    ptr = np.zeros(len(lists) + 1, dtype=np.int64)
    for i, l in enumerate(lists):
        ptr[i + 1] = ptr[i] + len(l)
    idx = np.concatenate([np.asarray(l, dtype=np.int64) for l in lists]) if lists else np.zeros(0, np.int64)
    return ptr, np.ascontiguousarray(idx.astype(np.int64))


@dataclass
class ImplicitFineGrid:
    """ref: src/implicit_fine_grid.jl:6-18"""
    levels: int
    reference: MultilevelReference
    interfaces: Interfaces
    base: Mesh
    _num_csr: dict = field(default_factory=dict)

    @staticmethod
    def create(base: Mesh, levels: int) -> "ImplicitFineGrid":
        assert np.all(np.diff(base.elements, axis=1) > 0), "base elements must be sorted"
        return ImplicitFineGrid(levels, refined_element(levels, base.dim), interfaces(base), base)

    def nf(self, level):          # level is 1-based like the reference
        return self.reference.levels[level - 1].nnodes()

    def num_csr(self, level, kind):
        key = (level, kind)
        if key not in self._num_csr:
            nb = self.reference.numbering[level - 1]
            if kind == "faces":
                self._num_csr[key] = _csr_of_lists(nb.faces_interior)
            elif kind == "edges":
                self._num_csr[key] = _csr_of_lists(nb.edges_interior)
            else:
                self._num_csr[key] = _csr_of_lists([[n] for n in nb.nodes])
        return self._num_csr[key]

    def construct_full_grid(self, level):
        """ref: src/implicit_fine_grid.jl:41-78 (nodes only)"""
        ref = self.reference.levels[level - 1]
        J, _, _ = cell_geometry(self.base)
        p1 = self.base.nodes[self.base.elements[:, 0]]
        return np.einsum("eab,nb->ena", J, ref.nodes) + p1[:, None, :]   # (Ne, Nf, dim)


def _as_f(x):
    assert x.flags["F_CONTIGUOUS"] and x.dtype == np.float64
    return x


def _class_call(fn, x, implicit, level, kind, smap, *extra):
    ptr, idx = implicit.num_csr(level, kind)
    fn(_p(x), ctypes.c_int64(x.shape[0]), ctypes.c_int64(len(smap)), _p(smap.offset),
       _p(smap.element), _p(smap.local_id), _p(ptr), _p(idx), *extra)


def broadcast_interfaces(x, implicit: ImplicitFineGrid, level: int):
    """ref: src/implicit_fine_grid.jl:209-328"""
    _as_f(x)
    L = lib()
    buf = np.zeros(max(implicit.nf(level), 1))
    if implicit.base.dim == 3:
        _class_call(L.orc_broadcast_class, x, implicit, level, "faces", implicit.interfaces.faces, _p(buf))
    _class_call(L.orc_broadcast_class, x, implicit, level, "edges", implicit.interfaces.edges, _p(buf))
    _class_call(L.orc_broadcast_class, x, implicit, level, "nodes", implicit.interfaces.nodes, _p(buf))
    return x


def zero_out_all_but_one(x, implicit: ImplicitFineGrid, level: int):
    """ref: src/implicit_fine_grid.jl:334-386"""
    _as_f(x)
    L = lib()
    if implicit.base.dim == 3:
        _class_call(L.orc_zero_class, x, implicit, level, "faces", implicit.interfaces.faces, ctypes.c_int(0))
    _class_call(L.orc_zero_class, x, implicit, level, "edges", implicit.interfaces.edges, ctypes.c_int(0))
    _class_call(L.orc_zero_class, x, implicit, level, "nodes", implicit.interfaces.nodes, ctypes.c_int(0))
    return x


@dataclass
class ZeroDirichletConstraint:
    """ref: src/implicit_fine_grid.jl:80-84"""
    nodes: SparseCellToElementMap
    edges: SparseCellToElementMap
    faces: SparseCellToElementMap


def apply_constraint(x, level: int, z: ZeroDirichletConstraint, implicit: ImplicitFineGrid):
    """ref: src/implicit_fine_grid.jl:94-139"""
    _as_f(x)
    L = lib()
    if implicit.base.dim == 3:
        _class_call(L.orc_zero_class, x, implicit, level, "faces", z.faces, ctypes.c_int(1))
    _class_call(L.orc_zero_class, x, implicit, level, "edges", z.edges, ctypes.c_int(1))
    _class_call(L.orc_zero_class, x, implicit, level, "nodes", z.nodes, ctypes.c_int(1))
    return x


def copy_to_base(u, v, implicit: ImplicitFineGrid):
    """ref: src/implicit_fine_grid.jl:148-171"""
    m = implicit.interfaces.all_nodes
    nodes = np.ascontiguousarray(implicit.reference.numbering[0].nodes.astype(np.int64))
    cells = np.ascontiguousarray(m.cells[:, 0])
    lib().orc_copy_to_base(_p(u), _p(_as_f(v)), ctypes.c_int64(v.shape[0]), ctypes.c_int64(len(m)),
                           _p(cells), _p(m.offset), _p(m.element), _p(m.local_id), _p(nodes))


def distribute(v, u, implicit: ImplicitFineGrid):
    """ref: src/implicit_fine_grid.jl:178-202"""
    m = implicit.interfaces.all_nodes
    nodes = np.ascontiguousarray(implicit.reference.numbering[0].nodes.astype(np.int64))
    cells = np.ascontiguousarray(m.cells[:, 0])
    lib().orc_distribute(_p(_as_f(v)), _p(u), ctypes.c_int64(v.shape[0]), ctypes.c_int64(len(m)),
                         _p(cells), _p(m.offset), _p(m.element), _p(m.local_id), _p(nodes))


class _PackedOps:
    """dim*dim CSC matrices packed for the C apply (slot = i + dim*j)."""

    def __init__(self, ops, nf):
        dim = len(ops)
        colptr, base, rowval, nzval = [], [], [], []
        pos = 0
        for j in range(dim):
            for i in range(dim):
                A = ops[i][j]
                assert A.shape == (nf, nf)
                colptr.append(A.indptr.astype(np.int64))
                base.append(pos)
                rowval.append(A.indices.astype(np.int64))
                nzval.append(A.data.astype(np.float64))
                pos += A.nnz
        self.colptr = np.ascontiguousarray(np.concatenate(colptr))
        self.base = np.array(base, dtype=np.int64)
        self.rowval = np.ascontiguousarray(np.concatenate(rowval))
        self.nzval = np.ascontiguousarray(np.concatenate(nzval))


class L2PlusDivAGrad:
    """ref: src/build_local_operators.jl:26-32 (mutable: lam, constraint)"""

    def __init__(self, diffusion_terms, mass, constraint, lam, sigmas):
        self.diffusion_terms = diffusion_terms
        self.mass = mass
        self.constraint = constraint
        self.lam = lam
        self.sigmas = np.ascontiguousarray(sigmas, dtype=np.float64)
        nf = mass.shape[0]
        self._packed = _PackedOps(diffusion_terms, nf)
        self._m = (mass.indptr.astype(np.int64), mass.indices.astype(np.int64),
                   mass.data.astype(np.float64))


class SimpleDiffusion:
    """ref: src/build_local_operators.jl:15-19"""

    def __init__(self, A, bc, a):
        self.A, self.bc, self.a = A, bc, a
        self._packed = _PackedOps(A, A[0][0].shape[0])


_GEOM_CACHE = {}


def _geom(base: Mesh):
    key = id(base)
    if key not in _GEOM_CACHE or _GEOM_CACHE[key][0] is not base:
        _, Jinv, det = cell_geometry(base)
        # C side wants column-major dim x dim per cell
        jinv_cm = np.ascontiguousarray(np.transpose(Jinv, (0, 2, 1)))
        _GEOM_CACHE[key] = (base, jinv_cm, np.ascontiguousarray(det))
    return _GEOM_CACHE[key][1:]


NTHREADS = [0]   # 0 -> every core this process may use


def available_cores():
    """Cores this process can actually run on: scheduler affinity capped by the cgroup CPU quota
    (omp_get_max_threads reports the host's hardware threads even inside a limited container)."""
    n = lib().orc_max_threads()
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(float(parts[0]) / float(parts[1]) + 0.5)))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, int(q / int(g.read()) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def _nthreads():
    return NTHREADS[0] if NTHREADS[0] > 0 else available_cores()


def mul(alpha, base: Mesh, A, x, y):
    """y <- alpha*A*x + y.  ref: src/apply_local_operators.jl:40-46, 85-91"""
    _as_f(x); _as_f(y)
    jinv, det = _geom(base)
    nf, ne = x.shape
    pk = A._packed
    if isinstance(A, L2PlusDivAGrad):
        mc, mr, mv = A._m
        lib().orc_apply_l2divagrad(ctypes.c_double(alpha), ctypes.c_int(base.dim), ctypes.c_int64(ne),
                                   ctypes.c_int64(nf), _p(jinv), _p(det), _p(A.sigmas),
                                   ctypes.c_double(A.lam), _p(pk.colptr), _p(pk.base), _p(pk.rowval),
                                   _p(pk.nzval), _p(mc), _p(mr), _p(mv), _p(x), _p(y),
                                   ctypes.c_int(_nthreads()))
    else:
        lib().orc_apply_simple_diffusion(ctypes.c_double(alpha), ctypes.c_int(base.dim),
                                         ctypes.c_int64(ne), ctypes.c_int64(nf), _p(jinv), _p(det),
                                         ctypes.c_double(A.a), _p(pk.colptr), _p(pk.base),
                                         _p(pk.rowval), _p(pk.nzval), _p(x), _p(y),
                                         ctypes.c_int(_nthreads()))
    return y


@dataclass
class LevelState:
    """ref: src/multigrid.jl:7-25"""
    x: np.ndarray
    b: np.ndarray
    r: np.ndarray
    p: np.ndarray
    Ap: np.ndarray

    @staticmethod
    def create(total_base_elements, total_fine_nodes):
        z = lambda: np.zeros((total_fine_nodes, total_base_elements), order="F")
        return LevelState(z(), z(), z(), z(), z())


def local_residual(implicit, A, curr: LevelState, k: int):
    """ref: src/apply_local_operators.jl:7-27"""
    curr.r[...] = curr.b
    mul(-1.0, implicit.base, A, curr.x, curr.r)
    apply_constraint(curr.r, k, A.constraint if isinstance(A, L2PlusDivAGrad) else A.bc, implicit)


def _dot(a, b):
    return lib().orc_dot(ctypes.c_int64(a.size), _p(a), _p(b), ctypes.c_int(_nthreads()))


def _axpy(alpha, x, y):
    lib().orc_axpy(ctypes.c_int64(x.size), ctypes.c_double(alpha), _p(x), _p(y), ctypes.c_int(_nthreads()))


def smoothing_steps(steps, implicit, ops, curr: LevelState, k: int):
    """`steps` CG iterations. ref: src/multigrid.jl:46-71"""
    local_residual(implicit, ops, curr, k)
    broadcast_interfaces(curr.r, implicit, k)
    curr.p[...] = curr.r
    rsqrprev = _dot(curr.r, curr.r)
    for _ in range(steps):
        curr.Ap.fill(0.0)
        mul(1.0, implicit.base, ops, curr.p, curr.Ap)
        apply_constraint(curr.Ap, k, ops.constraint, implicit)
        broadcast_interfaces(curr.Ap, implicit, k)
        alpha = rsqrprev / _dot(curr.p, curr.Ap)
        _axpy(alpha, curr.p, curr.x)
        _axpy(-alpha, curr.Ap, curr.r)
        rsqr = _dot(curr.r, curr.r)
        lib().orc_xpby(ctypes.c_int64(curr.p.size), _p(curr.r), ctypes.c_double(rsqr / rsqrprev),
                       _p(curr.p), ctypes.c_int(_nthreads()))
        rsqrprev = rsqr


@dataclass
class BaseLevel:
    """ref: src/multigrid.jl:30-41; A_inv is any object with .solve(b)"""
    A_inv: object
    b: np.ndarray
    b_interior: np.ndarray
    interior_nodes: np.ndarray

    @staticmethod
    def create(F, total_nodes, interior_nodes):
        return BaseLevel(F, np.zeros(total_nodes), np.zeros(len(interior_nodes)), interior_nodes)


def restrict_to(y, P, x):
    """ref: src/interpolation.jl:52-62"""
    lib().orc_restrict(_p(_as_f(y)), ctypes.c_int64(y.shape[0]), _p(_as_f(x)), ctypes.c_int64(x.shape[0]),
                       ctypes.c_int64(x.shape[1]), _p(P.indptr.astype(np.int64)),
                       _p(P.indices.astype(np.int64)), _p(P.data), ctypes.c_int(_nthreads()))


def interpolate_and_sum_to(y, P, x):
    """ref: src/interpolation.jl:64-74"""
    lib().orc_interpolate_and_sum(_p(_as_f(y)), ctypes.c_int64(y.shape[0]), _p(_as_f(x)),
                                  ctypes.c_int64(x.shape[0]), ctypes.c_int64(x.shape[1]),
                                  _p(P.indptr.astype(np.int64)), _p(P.indices.astype(np.int64)),
                                  _p(P.data), ctypes.c_int(_nthreads()))


def vcycle(implicit, base: BaseLevel, ops, levels, k: int, steps: int = 2):
    """ref: src/multigrid.jl:73-119 (levels/ops are 0-indexed lists; k is the 1-based level).
    NB :109 -- the recursive call does not forward `steps` (coarser levels use the default 2)."""
    if k == 1:
        broadcast_interfaces(levels[0].b, implicit, 1)
        copy_to_base(base.b, levels[0].b, implicit)
        base.b_interior[...] = base.b[base.interior_nodes]
        tmp = base.A_inv.solve(base.b_interior)
        base.b.fill(0.0)
        base.b[base.interior_nodes] = tmp
        distribute(levels[0].x, base.b, implicit)
    else:
        curr, nxt = levels[k - 1], levels[k - 2]
        P = implicit.reference.interops[k - 2]
        smoothing_steps(steps, implicit, ops[k - 1], curr, k)
        local_residual(implicit, ops[k - 1], curr, k)
        restrict_to(nxt.b, P, curr.r)
        nxt.x.fill(0.0)
        vcycle(implicit, base, ops, levels, k - 1)
        interpolate_and_sum_to(curr.x, P, nxt.x)
        smoothing_steps(steps, implicit, ops[k - 1], curr, k)


# --------------------------------------------------------------------------------------------
# Driver pieces (ref: src/examples/homogenized_coefficients.jl)
# --------------------------------------------------------------------------------------------
def compute_boundary_layer(lam, n):
    """ref: ...homogenized_coefficients.jl:9"""
    return int(np.floor(4 * (n + 1) * lam ** -0.5))


def compute_box_radius(k, n, eps=0.0):
    """ref: ...homogenized_coefficients.jl:10"""
    return int(np.floor(2.0 ** (n - k * (0.5 - eps))))


def infnorm(x):
    return np.max(np.abs(x), axis=-1)


def element_centers(mesh: Mesh):
    """mean(get_nodes(mesh, el)); ref: ...homogenized_coefficients.jl:15"""
    p = mesh.nodes[mesh.elements]
    s = p[:, 0, :].copy()
    for i in range(1, p.shape[1]):
        s = s + p[:, i, :]
    return s / p.shape[1]


def order_nodes_and_elements_by_magnitude(mesh: Mesh) -> Mesh:
    """ref: ...homogenized_coefficients.jl:21-28 (stable sorts)"""
    I = np.argsort(infnorm(mesh.nodes), kind="stable")
    Jp = np.empty_like(I)
    Jp[I] = np.arange(I.size)
    m = Mesh(mesh.nodes[I], np.sort(Jp[mesh.elements], axis=1))
    o = np.argsort(infnorm(element_centers(m)), kind="stable")
    m.elements = np.ascontiguousarray(m.elements[o])
    return m


def find_elements_in_radius(mesh: Mesh, radius):
    """count of leading elements with infnorm(center) <= radius; ref: ...:34-38"""
    return int(np.searchsorted(infnorm(element_centers(mesh)), radius, side="right"))


def find_nodes_in_radius(mesh: Mesh, radius):
    """ref: ...:44-48"""
    return int(np.searchsorted(infnorm(mesh.nodes), radius + 10 * np.finfo(float).eps, side="right"))


def conductivity_per_element(mesh: Mesh, sigma_grid, offset):
    """ref: ...:494-503. sigma_grid has shape (n,)*dim + (dim,); index = trunc(center + offset) (1-based)."""
    c = element_centers(mesh) + np.asarray(offset, dtype=np.float64)
    idx = np.trunc(c).astype(np.int64) - 1
    return np.ascontiguousarray(sigma_grid[tuple(idx[:, d] for d in range(mesh.dim))])


def rhs_axi_grad_v(b, dphis, implicit: ImplicitFineGrid, sigmas, xi):
    """b[i, el] = dot(dphis[i], -detJ * (Jinv' * (sigma .* xi))). ref: ...:449-474"""
    _, Jinv, det = cell_geometry(implicit.base)
    P = -det[:, None] * np.einsum("eki,ek->ei", Jinv, sigmas * np.asarray(xi)[None, :])
    b[...] = dphis @ P.T


def integrate_area(mass, implicit, nsubset):
    """ref: ...:673-689"""
    _, _, det = cell_geometry(implicit.base)
    M_total = mass.sum()
    area = 0.0
    for d in det[:nsubset]:
        area += M_total * d
    return area


def integrate_first_term(v0, dphis, implicit, nsubset, mass, sigmas, xi):
    """ref: ...:592-632"""
    _, Jinv, det = cell_geometry(implicit.base)
    P = -det[:, None] * np.einsum("eki,ek->ei", Jinv, sigmas * np.asarray(xi)[None, :])
    V = v0[:, :nsubset]
    MV = mass @ V
    run = np.einsum("ie,ie->e", V, dphis @ P[:nsubset].T + MV)
    total = 0.0
    for e in range(nsubset):
        total += run[e] * det[e]
    return total


def integrate_terms(vk, vkm1, implicit, nsubset, mass):
    """ref: ...:634-667"""
    _, _, det = cell_geometry(implicit.base)
    V = vk[:, :nsubset]
    MV = mass @ V
    run = np.einsum("ie,ie->e", V + vkm1[:, :nsubset], MV)
    total = 0.0
    for e in range(nsubset):
        total += run[e] * det[e]
    return total


def next_rhs(b, x, implicit, mass, lam):
    """b = lam*|J|*M*x per cell. ref: ...:695-713"""
    _, _, det = cell_geometry(implicit.base)
    b[...] = (mass @ x) * (lam * det)[None, :]


class _SpluSolver:
    """Direct sparse solve standing in for CHOLMOD `cholesky(A) \\ b`
    (ref: ...homogenized_coefficients.jl:260, src/multigrid.jl:84). SuperLU, not CHOLMOD."""

    def __init__(self, A):
        self.lu = spla.splu(sp.csc_matrix(A))

    def solve(self, b):
        return self.lu.solve(b)


def local_rhs(b, implicit: ImplicitFineGrid):
    """b[:, e] = assemble_vector(fine reference mesh, identity) * |det J_e|.  ref: src/implicit_fine_grid.jl:391-409"""
    fine = implicit.reference.levels[-1]
    assert b.shape == (fine.nnodes(), implicit.base.nelements())
    b_ref = assemble_vector(fine)
    _, _, det = cell_geometry(implicit.base)
    b[...] = b_ref[:, None] * det[None, :]
    return b


def make_base_level(base: Mesh, sigmas, lam):
    """ref: ...homogenized_coefficients.jl:259-261"""
    interior = list_interior_nodes(base)
    A = assemble_checkerboard(base, sigmas, lam).tocsr()[interior][:, interior]
    return BaseLevel.create(_SpluSolver(A), base.nnodes(), interior)


def checkerboard_homogenization(n=4, dim=2, refinements=2, smoothing_steps_=3, tolerance=1e-4,
                                xi=None, seed=0, sigma_values=(1.0, 9.0), max_cycles=1000,
                                log=None, x0=None, sigma_grid=None):
    """ref: ...homogenized_coefficients.jl:174-343.  RNG: the reference uses the unseeded global
    RNG (:246,:487); here numpy Generator(seed) draws sigma first, then x0 (inputs can also be
    passed in explicitly so the product driver is fed identical arrays)."""
    rng = np.random.default_rng(seed)
    if xi is None:
        xi = np.ones(dim) / np.linalg.norm(np.ones(dim))
    lam, sigma = 1.0, 0.0
    box_radius = compute_box_radius(0, n)
    boundary_layer = compute_boundary_layer(lam, n)
    total_radius = box_radius + boundary_layer
    width = 2 * total_radius
    base = order_nodes_and_elements_by_magnitude(hypercube(dim, width, origin=(-float(total_radius),) * dim))
    if sigma_grid is None:
        sigma_grid = np.where(rng.random((width,) * dim + (dim,)) < 0.5, sigma_values[0], sigma_values[1])
    cond = conductivity_per_element(base, sigma_grid, (total_radius + 1.0,) * dim)
    total_grids = refinements + 1
    implicit = ImplicitFineGrid.create(base, total_grids)
    constraint = ZeroDirichletConstraint(*list_boundary_nodes_edges_faces(base))
    diff_terms = [build_local_diffusion_operators(m) for m in implicit.reference.levels]
    mass_terms = [mass_matrix(m) for m in implicit.reference.levels]
    ops = [L2PlusDivAGrad(d, m, constraint, lam, cond) for d, m in zip(diff_terms, mass_terms)]
    states = [LevelState.create(base.nelements(), implicit.nf(i + 1)) for i in range(total_grids)]
    top = states[-1]
    if x0 is None:
        x0 = rng.random(top.x.shape)
    top.x[...] = x0
    broadcast_interfaces(top.x, implicit, total_grids)
    apply_constraint(top.x, total_grids, constraint, implicit)
    dphis = partial_derivatives_functionals(implicit.reference.levels[-1])
    rhs_axi_grad_v(top.b, dphis, implicit, cond, xi)
    v_prev = np.zeros_like(top.x)
    history = []
    for k in range(n + 1):
        base_level = make_base_level(base, cond, lam)
        dsig, dsig_prev = 0.0, 0.0
        for i in range(1, max_cycles + 1):
            vcycle(implicit, base_level, ops, states, total_grids, smoothing_steps_)
            nint = find_elements_in_radius(base, box_radius)
            area = integrate_area(mass_terms[-1], implicit, nint)
            if k == 0:
                integral = integrate_first_term(top.x, dphis, implicit, nint, mass_terms[-1], cond, xi)
            else:
                integral = integrate_terms(top.x, v_prev, implicit, nint, mass_terms[-1])
            dsig = 2.0 ** k * integral / area
            zero_out_all_but_one(top.r, implicit, total_grids)
            rnorm = float(np.linalg.norm(top.r))
            history.append((k, i, rnorm, sigma + dsig, abs(dsig - dsig_prev)))
            if log:
                log(history[-1])
            if abs(dsig - dsig_prev) < tolerance:
                break
            dsig_prev = dsig
        sigma += dsig
        lam /= 2
        box_radius = compute_box_radius(k + 1, n)
        boundary_layer = compute_boundary_layer(lam, n)
        if box_radius + boundary_layer > total_radius:
            break
        total_radius = box_radius + boundary_layer
        nn_keep = find_nodes_in_radius(base, total_radius)
        ne_keep = find_elements_in_radius(base, total_radius)
        base = Mesh(base.nodes[:nn_keep], np.ascontiguousarray(base.elements[:ne_keep]))
        cond = np.ascontiguousarray(cond[:ne_keep])
        constraint = ZeroDirichletConstraint(*list_boundary_nodes_edges_faces(base))
        states = [LevelState(*(np.asfortranarray(a[:, :ne_keep]) for a in (s.x, s.b, s.r, s.p, s.Ap)))
                  for s in states]
        top = states[-1]
        implicit = ImplicitFineGrid(total_grids, implicit.reference, interfaces(base), base)
        apply_constraint(top.x, total_grids, constraint, implicit)
        v_prev = top.x.copy(order="F")
        ops = [L2PlusDivAGrad(d, m, constraint, lam, cond) for d, m in zip(diff_terms, mass_terms)]
        next_rhs(top.b, top.x, implicit, mass_terms[-1], lam)
    return sigma, history


def checkerboard_hypercube_multigrid(n, dim=3, refinements=2, max_cycles=5, seed=1, sigma_grid=None, x0=None):
    """Solve -div(a grad u) = 1, u = 0 on the boundary, with `max_cycles` V-cycles (lambda = 0); returns the residual
    norms and the final state.  `refinements` is the number of grids, as in the reference.
    ref: ...homogenized_coefficients.jl:509-571 (VTK output left to the caller; RNG seeded here)."""
    rng = np.random.default_rng(seed)
    base = hypercube(dim, n)
    if sigma_grid is None:
        sigma_grid = np.where(rng.random((n,) * dim + (dim,)) < 0.5, 1.0, 9.0)
    cond = conductivity_per_element(base, sigma_grid, (0.0,) * dim)
    base_level = make_base_level(base, cond, 0.0)
    implicit = ImplicitFineGrid.create(base, refinements)
    constraint = ZeroDirichletConstraint(*list_boundary_nodes_edges_faces(base))
    ops = [L2PlusDivAGrad(build_local_diffusion_operators(m), mass_matrix(m), constraint, 0.0, cond)
           for m in implicit.reference.levels]
    states = [LevelState.create(base.nelements(), implicit.nf(i + 1)) for i in range(refinements)]
    top = states[-1]
    top.x[...] = rng.random(top.x.shape) if x0 is None else x0
    broadcast_interfaces(top.x, implicit, refinements)
    apply_constraint(top.x, refinements, constraint, implicit)
    local_rhs(top.b, implicit)
    rs = []
    for _ in range(max_cycles):
        vcycle(implicit, base_level, ops, states, refinements, 3)
        zero_out_all_but_one(top.r, implicit, refinements)
        rs.append(float(np.linalg.norm(top.r)))
    return rs, top, implicit, base, cond
