"""Second, structurally independent statement of smoothing_steps! / vcycle! (src/multigrid.jl:46-119) used by
tests/test_oracle_reference_kats.py (against the oracle) and tests/test_gpu_parity.py (against the device): the same
iteration written on GLOBAL vectors with explicitly assembled matrices -- refine_uniformly + assemble_checkerboard per
level (the recipe of checkerboard_hypercube_full, ...homogenized_coefficients.jl:729-759), interpolation matrices built
from the node coordinates, the reference's duplicate-counting dot products (src/multigrid.jl:54,64,67: a node shared by m
cells counts m times) as multiplicity-weighted sums, a sparse direct solve on level 1, `steps` not forwarded (:109).  None
of the cell-local machinery (lattice tables, interface sums, constraint masks, per-cell transfers, copy_to_base! /
distribute!) takes part."""
import numpy as np


class GlobalForm:
    def __init__(self, O, base, sgrid, lam, implicit, grids, dim):
        import scipy.sparse as sp
        self.O, self.grids, self.dim = O, grids, dim
        w = np.array([1, 1 << 20, 1 << 40][:dim])
        key = lambda p: np.round(np.asarray(p) * 4096).astype(np.int64) @ w       # coordinates are exact dyadics
        self.meshes, self.A, self.inner, self.mapping, self.mult = [], [], [], [], []
        for l in range(grids):
            m = O.refine_uniformly(base, times=l) if l else base
            m.elements = O.sort_element_nodes(m.elements)
            self.meshes.append(m)
            self.A.append(O.assemble_checkerboard(m, O.conductivity_per_element(m, sgrid, (0.0,) * dim), lam).tocsr())
            mask = np.zeros(m.nnodes(), dtype=bool)
            mask[O.list_interior_nodes(m)] = True
            self.inner.append(mask)
            tk = key(m.nodes)
            order = np.argsort(tk)
            rep = implicit.construct_full_grid(l + 1).reshape(-1, dim)           # repeated nodes, cell-major
            pos = np.searchsorted(tk[order], key(rep))
            assert np.all(tk[order][pos] == key(rep))
            self.mapping.append(order[pos])
            self.mult.append(np.bincount(order[pos], minlength=m.nnodes()).astype(np.float64))
        self.interp = [None]
        for l in range(1, grids):                                                 # fine node = coarse node, or midpoint of a coarse edge
            coarse, fine = self.meshes[l - 1], self.meshes[l]
            ck = {int(k): i for i, k in enumerate(key(coarse.nodes))}
            mid = {}
            for el in coarse.elements:
                for a in range(len(el)):
                    for b in range(a + 1, len(el)):
                        mid[int(key(0.5 * (coarse.nodes[el[a]] + coarse.nodes[el[b]])))] = (int(el[a]), int(el[b]))
            rows, cols, vals = [], [], []
            for i, k in enumerate(key(fine.nodes)):
                if int(k) in ck:
                    rows.append(i); cols.append(ck[int(k)]); vals.append(1.0)
                else:
                    a, b = mid[int(k)]
                    rows += [i, i]; cols += [a, b]; vals += [0.5, 0.5]
            self.interp.append(sp.csr_matrix((vals, (rows, cols)), shape=(fine.nnodes(), coarse.nnodes())))

    def gather(self, local, l):
        """one global vector from the consistent copies of a level vector (Nf x Ne)"""
        g = np.zeros(self.meshes[l].nnodes())
        flat = np.asarray(local).reshape(-1, order="F")
        g[self.mapping[l]] = flat
        assert np.array_equal(g[self.mapping[l]], flat)                          # all copies of a node agree
        return g

    def gather_sum(self, local, l):
        """the global load = sum of the local ones"""
        g = np.zeros(self.meshes[l].nnodes())
        np.add.at(g, self.mapping[l], np.asarray(local).reshape(-1, order="F"))
        return g

    def smooth(self, l, x, b, nsteps):
        A, inner, mult = self.A[l], self.inner[l], self.mult[l]
        mdot = lambda u, v: float(np.dot(mult * u, v))
        r = np.where(inner, b - A @ x, 0.0)
        p = r.copy()
        rs = mdot(r, r)
        for _ in range(nsteps):
            Ap = np.where(inner, A @ p, 0.0)
            alpha = rs / mdot(p, Ap)
            x = x + alpha * p
            r = r - alpha * Ap
            rs_new = mdot(r, r)
            p = r + (rs_new / rs) * p
            rs = rs_new
        return x, r

    def vcycle(self, l, x, b, nsteps):
        import scipy.sparse.linalg as spla
        if l == 0:
            x = np.zeros_like(x)
            idx = np.flatnonzero(self.inner[0])
            x[idx] = spla.spsolve(self.A[0][idx][:, idx].tocsc(), b[idx])
            return x, None
        x, _ = self.smooth(l, x, b, nsteps)
        r = np.where(self.inner[l], b - self.A[l] @ x, 0.0)
        xc, _ = self.vcycle(l - 1, np.zeros(self.meshes[l - 1].nnodes()), self.interp[l].T @ r, 2)   # `steps` is not forwarded (:109)
        x = x + self.interp[l] @ xc
        return self.smooth(l, x, b, nsteps)
