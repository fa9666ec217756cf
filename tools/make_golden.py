#!/usr/bin/env python3
"""Generates tests/golden/*.npz: small seeded input/output vectors of every hot-path primitive, produced by the
CPU oracle (oracle/, the restatement of the reference pinned by the reference's own known-answer tests -- the
reference itself is pure Julia and cannot run in the build image, and it ships no data files).

    python tools/make_golden.py          # rewrites tests/golden/

Cases: 3D 2x2x2 cubes (48 tets, perturbed nodes), levels 1..4;  2D 4x4 squares (32 triangles), levels 1..5.
All arrays are in the reference's API layout (Nf x Ne column-major, hierarchical node order, 0-based cells)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def case(dim, n, levels, seed):
    rng = np.random.default_rng(seed)
    m = O.order_nodes_and_elements_by_magnitude(O.hypercube(dim, n, origin=(-n / 2.0,) * dim))
    m.nodes = m.nodes + 0.2 * (rng.random(m.nodes.shape) - 0.5)
    sig = rng.choice([1.0, 9.0], size=(m.nelements(), dim))
    lam = 0.7
    impl = O.ImplicitFineGrid.create(m, levels)
    cons = O.ZeroDirichletConstraint(*O.list_boundary_nodes_edges_faces(m))
    ops = [O.L2PlusDivAGrad(O.build_local_diffusion_operators(l), O.mass_matrix(l), cons, lam, sig)
           for l in impl.reference.levels]
    d = {"dim": dim, "levels": levels, "lam": lam, "nodes": m.nodes, "cells": m.elements, "sigma": sig}
    L = levels
    f = lambda lev: np.asfortranarray(rng.standard_normal((impl.nf(lev), m.nelements())))
    for lev in range(1, L + 1):
        x, y = f(lev), f(lev)
        d[f"apply_x_{lev}"], d[f"apply_y_{lev}"] = x, y
        out = y.copy(order="F"); O.mul(-1.3, m, ops[lev - 1], x, out)
        d[f"apply_out_{lev}"] = out                                   # y + (-1.3) A x
        s = x.copy(order="F"); O.broadcast_interfaces(s, impl, lev); d[f"isum_out_{lev}"] = s
        c = x.copy(order="F"); O.apply_constraint(c, lev, cons, impl); d[f"constraint_out_{lev}"] = c
        z = s.copy(order="F"); O.zero_out_all_but_one(z, impl, lev); d[f"unique_out_{lev}"] = z
        if lev >= 2:
            P = impl.reference.interops[lev - 2]
            xc = f(lev - 1)
            pr = y.copy(order="F"); O.interpolate_and_sum_to(pr, P, xc)
            d[f"prolong_xc_{lev}"], d[f"prolong_out_{lev}"] = xc, pr   # y + P xc
            rs = np.zeros_like(xc, order="F"); O.restrict_to(rs, P, x); d[f"restrict_out_{lev}"] = rs
    # smoother + V-cycles on the top level
    st = [O.LevelState.create(m.nelements(), impl.nf(i + 1)) for i in range(L)]
    st[-1].x[...] = rng.random(st[-1].x.shape)
    O.broadcast_interfaces(st[-1].x, impl, L); O.apply_constraint(st[-1].x, L, cons, impl)
    st[-1].b[...] = f(L)
    d["mg_x0"], d["mg_b"] = st[-1].x.copy(order="F"), st[-1].b.copy(order="F")
    sm = O.LevelState(*(a.copy(order="F") for a in (st[-1].x, st[-1].b, st[-1].r, st[-1].p, st[-1].Ap)))
    O.smoothing_steps(3, impl, ops[-1], sm, L)
    d["smooth_x"], d["smooth_r"], d["smooth_p"] = sm.x, sm.r, sm.p
    base = O.make_base_level(m, sig, lam)
    norms = []
    for cyc in range(3):
        O.vcycle(impl, base, ops, st, L, 3)
        r = st[-1].r.copy(order="F"); O.zero_out_all_but_one(r, impl, L); norms.append(np.linalg.norm(r))
        d[f"vcycle_x_{cyc + 1}"] = st[-1].x.copy(order="F")
    d["vcycle_rnorms"] = np.array(norms)
    return d


def driver_case(dim, n, refinements, tol, seed):
    width = 2 * (O.compute_box_radius(0, n) + O.compute_boundary_layer(1.0, n))
    rng = np.random.default_rng(seed)
    sgrid = np.where(rng.random((width,) * dim + (dim,)) < 0.5, 1.0, 9.0)
    nf = O.refined_element(refinements + 1, dim).levels[-1].nnodes()
    x0 = rng.random((nf, (2 if dim == 2 else 6) * width ** dim))
    sigma, hist = O.checkerboard_homogenization(n=n, dim=dim, refinements=refinements, tolerance=tol, sigma_grid=sgrid, x0=x0)
    return {"dim": dim, "n": n, "refinements": refinements, "tolerance": tol, "sigma_grid": sgrid, "x0": x0,
            "sigma": sigma, "history": np.array(hist)}


if __name__ == "__main__":
    np.savez_compressed(os.path.join(OUT, "tet_2x2x2_L4.npz"), **case(3, 2, 4, 101))
    np.savez_compressed(os.path.join(OUT, "tri_4x4_L5.npz"), **case(2, 4, 5, 202))
    np.savez_compressed(os.path.join(OUT, "driver_tri_n1_r2.npz"), **driver_case(2, 1, 2, 1e-4, 303))
    np.savez_compressed(os.path.join(OUT, "driver_tet_n0_r2.npz"), **driver_case(3, 0, 2, 1e-3, 404))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
