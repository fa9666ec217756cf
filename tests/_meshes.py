"""Base meshes for the parity tests that are not split cube lattices."""
import numpy as np


def delaunay_mesh(O, dim, npts, seed):
    """Delaunay triangulation of random points in the unit cube as an oracle Mesh: edges shared by 3..10+ cells,
    nodes by up to dozens, irregular boundary, very different cell shapes.  Near-degenerate hull cells are
    dropped (they would only test conditioning, not indexing)."""
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    pts = rng.random((npts, dim))
    cells = np.sort(Delaunay(pts).simplices.astype(np.int64), axis=1)
    e = pts[cells[:, 1:]] - pts[cells[:, :1]]
    vol = np.abs(np.linalg.det(e))
    cells = np.ascontiguousarray(cells[vol > 1e-4 * vol.max()])
    used = np.unique(cells)
    remap = -np.ones(npts, dtype=np.int64)
    remap[used] = np.arange(used.size)
    return O.Mesh(np.ascontiguousarray(pts[used]), np.sort(remap[cells], axis=1))


CUBE_NODES = np.array([(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0), (0, 0, 1), (1, 0, 1), (0, 1, 1), (1, 1, 1)],
                      dtype=np.float64)
CUBE_ELEMENTS = np.array([(1, 2, 3, 5), (2, 3, 4, 8), (3, 5, 7, 8), (2, 5, 6, 8), (2, 3, 5, 8)], dtype=np.int64) - 1


def five_tet_cube(O, times):
    """The unit cube split into 5 tetrahedra, red-refined `times` times (the base mesh of test/test_operator.jl:9-21)."""
    base = O.refine_uniformly(O.Mesh(CUBE_NODES.copy(), CUBE_ELEMENTS.copy()), times=times)
    base.elements = O.sort_element_nodes(base.elements)
    return base


def match_nodes(total_nodes, rep):
    """Index of every repeated (cell-major) fine node in the explicit fine mesh; coordinates are exact dyadics
    (test/test_operator.jl:35-47)."""
    dim = rep.shape[1]
    w = np.array([1, 1 << 20, 1 << 40][:dim])
    key = lambda p: np.round(p * 4096).astype(np.int64) @ w
    tk = key(total_nodes)
    order = np.argsort(tk)
    pos = np.searchsorted(tk[order], key(rep))
    assert np.all(tk[order][pos] == key(rep))
    return order[pos]
