#!/usr/bin/env python3
"""ONE attempt: does the historical smoother of the reference's stale test -- damped Richardson with the
omegas of test/local_operators.jl:45-49 (Tri: 0.2 on every level) -- reproduce the residual contraction the tutorial prints
(docs/src/index.md:296-302: 0.9105 per cycle)?  Oracle only (CPU); everything but the smoother is the tutorial's setup.

RESULT (round 3): no.  With the tutorial's operator (sigma in {1, 9}, lambda = 1) Richardson diverges for omega = 0.2, 0.1 and 0.05
(residual x 5e3, x 2e2, x 6 per cycle): those omegas belong to the SimpleDiffusion operator of the stale test (sigma = 1), whose
spectrum is an order of magnitude smaller.  The published 0.9105 stays unexplained; the attempt is closed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as O
import _docs_example as D


def richardson(steps, omega, implicit, ops, curr, k):
    for _ in range(steps):
        O.local_residual(implicit, ops, curr, k)
        O.broadcast_interfaces(curr.r, implicit, k)
        curr.x += omega * curr.r


def vcycle(implicit, base, ops, levels, k, steps, omega, steps_coarse):
    if k == 1:
        O.vcycle(implicit, base, ops, levels, 1)
        return
    curr, nxt = levels[k - 1], levels[k - 2]
    P = implicit.reference.interops[k - 2]
    richardson(steps, omega, implicit, ops[k - 1], curr, k)
    O.local_residual(implicit, ops[k - 1], curr, k)
    O.restrict_to(nxt.b, P, curr.r)
    nxt.x.fill(0.0)
    vcycle(implicit, base, ops, levels, k - 1, steps_coarse, omega, steps_coarse)
    O.interpolate_and_sum_to(curr.x, P, nxt.x)
    richardson(steps, omega, implicit, ops[k - 1], curr, k)


def run(steps, steps_coarse, omega, cycles=100, n=32, grids=3, lam=1.0, seed=2019):
    base, sigma_grid, x0 = D.inputs(O, n, grids, seed)
    a = O.conductivity_per_element(base, sigma_grid, (0.0, 0.0))
    base_level = O.make_base_level(base, a, lam)
    implicit = O.ImplicitFineGrid.create(base, grids)
    cons = O.ZeroDirichletConstraint(*O.list_boundary_nodes_edges_faces(base))
    ops = [O.L2PlusDivAGrad(O.build_local_diffusion_operators(m), O.mass_matrix(m), cons, lam, a) for m in implicit.reference.levels]
    st = [O.LevelState.create(base.nelements(), implicit.nf(i + 1)) for i in range(grids)]
    top = st[-1]
    O.local_rhs(top.b, implicit)
    top.x[...] = x0
    O.broadcast_interfaces(top.x, implicit, grids)
    O.apply_constraint(top.x, grids, cons, implicit)
    norms = []
    for _ in range(cycles):
        vcycle(implicit, base_level, ops, st, grids, steps, omega, steps_coarse)
        O.local_residual(implicit, ops[-1], top, grids)          # (the Richardson sweep leaves r one update behind)
        O.broadcast_interfaces(top.r, implicit, grids)
        O.zero_out_all_but_one(top.r, implicit, grids)
        norms.append(float(np.linalg.norm(top.r)))
        if not np.isfinite(norms[-1]) or norms[-1] > 1e30:
            break
    return norms


print("published tail:", D.REFERENCE_TAIL, "ratio", D.REFERENCE_TAIL[2] / D.REFERENCE_TAIL[1])
for steps, sc, om in ((1, 1, 0.2), (1, 2, 0.2), (2, 2, 0.2), (1, 1, 0.1), (1, 1, 0.05)):
    nm = run(steps, sc, om)
    tail = nm[-3:]
    print(f"Richardson omega={om} steps={steps}/{sc}: cycles {len(nm)}, last norms {tail}, ratio {tail[-1] / tail[-2] if len(tail) > 1 and tail[-2] else float('nan'):.4f}", flush=True)
