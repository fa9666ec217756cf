"""Textbook P1 finite elements on an explicitly refined simplex mesh -- the expected values for the driver's right-hand sides
and integrals (rhs_a xi grad v!, next_rhs!, integrate_first_term, integrate_terms, integrate_area:
src/examples/homogenized_coefficients.jl:449-474, 592-713) without any of the reference's per-cell reference-element
machinery: per fine cell its volume and the constant gradients of its hat functions, nothing else."""
import math
import numpy as np


class TextbookP1:
    def __init__(self, nodes, elements):
        self.nodes, self.elements = np.asarray(nodes, dtype=np.float64), np.asarray(elements)
        K, nv = self.elements.shape
        d = nv - 1
        V = self.nodes[self.elements]                                  # (K, d+1, dim)
        E = V[:, 1:, :] - V[:, :1, :]                                  # rows = edge vectors from vertex 0
        self.vol = np.abs(np.linalg.det(E)) / math.factorial(d)
        Einv = np.linalg.inv(E)                                        # x - v0 = E^T lambda  =>  grad lambda_i = column i of E^-1
        g = np.empty((K, nv, d))
        g[:, 1:, :] = np.transpose(Einv, (0, 2, 1))
        g[:, 0, :] = -g[:, 1:, :].sum(axis=1)
        self.grad = g
        self.centers = V.mean(axis=1)

    def load(self, a_cell, xi, mask=None):
        """F_i = - int (a xi) . grad phi_i over the (masked) cells; a_cell: (K, dim) diagonal conductivity per cell"""
        w = self.vol if mask is None else self.vol * mask
        contrib = -w[:, None] * np.einsum("kid,kd->ki", self.grad, a_cell * np.asarray(xi)[None, :])
        F = np.zeros(len(self.nodes))
        np.add.at(F, self.elements, contrib)
        return F

    def mass_quadratic(self, u, v, mask=None):
        """u^T M v over the (masked) cells, M_K = |K| / ((d+1)(d+2)) (1 + delta_ij)"""
        nv = self.elements.shape[1]
        w = self.vol if mask is None else self.vol * mask
        U, V = u[self.elements], v[self.elements]                      # (K, d+1)
        q = (U.sum(axis=1) * V.sum(axis=1) + (U * V).sum(axis=1)) / (nv * (nv + 1))
        return float(np.dot(w, q))

    def mass_apply(self, u):
        """M u over all cells"""
        nv = self.elements.shape[1]
        U = u[self.elements]
        loc = (U.sum(axis=1, keepdims=True) + U) * (self.vol / (nv * (nv + 1)))[:, None]
        out = np.zeros(len(self.nodes))
        np.add.at(out, self.elements, loc)
        return out


def driver_setting(O, dim, width, levels, radius, seed=3):
    """Checkerboard on width^dim unit cubes centred at the origin, ordered by magnitude as the driver orders it; the explicitly
    refined mesh with its textbook quantities, the conductivity per fine cell, the indicator of the fine cells inside the box of
    the given radius, and the map repeated node -> node of the refined mesh."""
    rng = np.random.default_rng(seed)
    base = O.order_nodes_and_elements_by_magnitude(O.hypercube(dim, width, origin=(-width / 2.0,) * dim))
    sgrid = np.where(rng.random((width,) * dim + (dim,)) < 0.5, 1.0, 9.0)
    off = (width / 2.0 + 1.0,) * dim
    cond = O.conductivity_per_element(base, sgrid, off)
    implicit = O.ImplicitFineGrid.create(base, levels)
    fine = O.refine_uniformly(base, times=levels - 1)
    T = TextbookP1(fine.nodes, fine.elements)
    a_fine = O.conductivity_per_element(fine, sgrid, off)
    inside = (np.abs(T.centers).max(axis=1) <= radius).astype(np.float64)
    w = np.array([1, 1 << 20, 1 << 40][:dim])
    key = lambda p: np.round((np.asarray(p) + 64.0) * 4096).astype(np.int64) @ w
    tk = key(fine.nodes)
    order = np.argsort(tk)
    rep = implicit.construct_full_grid(levels).reshape(-1, dim)
    pos = np.searchsorted(tk[order], key(rep))
    assert np.all(tk[order][pos] == key(rep))
    mapping = order[pos]
    nint = O.find_elements_in_radius(base, radius)
    assert 0 < nint < base.nelements()
    return base, cond, implicit, T, a_fine, inside, mapping, nint, rng


def converged_first_term(O, dim, sigma_grid, xi, refinements, lam=1.0):
    """What outer step 0 of checkerboard_homogenization (src/examples/homogenized_coefficients.jl:174-343, n = 0: box radius
    1, boundary layer 4, 10^dim unit cubes) converges to, by a sparse direct (large 3D cases: conjugate-gradient, 1e-14) solve of (lam M + K_a) v = F on the explicitly
    refined mesh with v = 0 on the boundary:  sigma_0 = (v . F_box + v^T M_box v) / |box|."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    total_radius, box_radius = 5, 1
    width = 2 * total_radius
    base = O.order_nodes_and_elements_by_magnitude(O.hypercube(dim, width, origin=(-float(total_radius),) * dim))
    fine = O.refine_uniformly(base, times=refinements)
    T = TextbookP1(fine.nodes, fine.elements)
    a = O.conductivity_per_element(fine, sigma_grid, (total_radius + 1.0,) * dim)
    nv = T.elements.shape[1]
    # stiffness with the diagonal conductivity a_K, mass
    ag = T.grad * a[:, None, :]
    Kloc = np.einsum("kid,kjd->kij", ag, T.grad) * T.vol[:, None, None]
    Mloc = (np.ones((nv, nv)) + np.eye(nv))[None, :, :] * (T.vol / (nv * (nv + 1)))[:, None, None]
    rows = np.repeat(T.elements, nv, axis=1).ravel()
    cols = np.tile(T.elements, (1, nv)).ravel()
    n = len(T.nodes)
    A = sp.csr_matrix(((Kloc + lam * Mloc).ravel(), (rows, cols)), shape=(n, n))
    F = T.load(a, xi)
    interior = np.flatnonzero(np.abs(T.nodes).max(axis=1) < total_radius - 1e-9)
    v = np.zeros(n)
    Aii = A[interior][:, interior].tocsr()
    if len(interior) <= 20000:
        v[interior] = spla.spsolve(Aii.tocsc(), F[interior])
    else:                                                             # (3D fill-in: conjugate gradients to 1e-14 instead)
        d = Aii.diagonal()
        prec = spla.LinearOperator(Aii.shape, matvec=lambda r: r / d)
        sol, info = spla.cg(Aii, F[interior], rtol=1e-14, atol=0.0, maxiter=20000, M=prec)
        assert info == 0
        v[interior] = sol
    inside = (np.abs(T.centers).max(axis=1) <= box_radius).astype(np.float64)
    return (float(np.dot(v, T.load(a, xi, inside))) + T.mass_quadratic(v, v, inside)) / float(np.dot(T.vol, inside))
