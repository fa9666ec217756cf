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
