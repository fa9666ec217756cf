"""The multigrid example of the reference's tutorial (docs/src/index.md:162-304), the only V-cycle output the
reference publishes: Tri{Float64} hypercube n = 32 (origin (1,1), unordered), conductivity in {1, 9} per unit square,
lambda = 1, 3 grids, b = local_rhs!, x0 = rand! + broadcast_interfaces! + apply_constraint!, 100 x vcycle! with
smoothing_steps = 1 (2 on the coarser levels: `steps` is not forwarded, src/multigrid.jl:109), norm(r) after
zero_out_all_but_one!.  The reference prints 5.18e-4 / 4.72e-4 / 4.30e-4 after cycles 98 / 99 / 100: a residual
contraction of 0.9105 per cycle (RNG unseeded there, so only the ratio is comparable).

FINDING (round 2): the restatement does not reproduce that number.  With sigma in {1, 9} (the values of
src/examples/homogenized_coefficients.jl:487-488) oracle and device contract by 0.78-0.86 per cycle (alternating, the CG
smoother is nonlinear), for every seed, for n = 16 / 32 / 64, lambda = 0 / 1, with and without forwarding `steps` -- i.e.
FASTER than the printout (residual 6e-7 instead of 4.3e-4 after 100 cycles).  The rate is a strong function of the
contrast: 0.34 for sigma = 1, 0.80 at contrast 9, 0.95-0.99 at contrast 100, so the printed 0.9105 corresponds to a
contrast between 9 and 100.  Which code version and coefficient field produced the printout cannot be established
without running the reference (no Julia here); rows E/F stay "parity unpinned" (DESIGN.md section 6).
One clue in the reference's own text (round 5): the tutorial describes LevelState as "x, b and r" (docs/src/index.md:244-247) where
src/multigrid.jl:7-13 holds five arrays (p and Ap came with the CG smoother), and the reference's stale test/local_operators.jl:60-101
still calls an earlier `vcycle!(..., level_states, ωs, total_levels)` with per-level damping factors -- the tutorial text, and
possibly its printout, predate the CG smoother restated here."""
import numpy as np

REFERENCE_TAIL = (0.0005182895775368055, 0.00047190444233626385, 0.00042970384073489823)   # docs/src/index.md:296-302


def inputs(O, n=32, grids=3, seed=2019, high=9.0):
    rng = np.random.default_rng(seed)
    sigma_grid = np.where(rng.random((n, n, 2)) < 0.5, 1.0, high)
    base = O.hypercube(2, n)
    nf = O.ImplicitFineGrid.create(base, grids).nf(grids)
    x0 = np.asfortranarray(rng.random((nf, base.nelements())))
    return base, sigma_grid, x0


def run_oracle(O, cycles=100, n=32, grids=3, lam=1.0, smoothing_steps=1, seed=2019, high=9.0):
    base, sigma_grid, x0 = inputs(O, n, grids, seed, high)
    a = O.conductivity_per_element(base, sigma_grid, (0.0, 0.0))
    base_level = O.make_base_level(base, a, lam)
    implicit = O.ImplicitFineGrid.create(base, grids)
    assert (base.nnodes(), base.nelements(), implicit.nf(grids)) == ((n + 1) ** 2, 2 * n * n, 15)   # index.md:192-194
    constraint = O.ZeroDirichletConstraint(*O.list_boundary_nodes_edges_faces(base))
    ops = [O.L2PlusDivAGrad(O.build_local_diffusion_operators(m), O.mass_matrix(m), constraint, lam, a)
           for m in implicit.reference.levels]
    states = [O.LevelState.create(base.nelements(), implicit.nf(i + 1)) for i in range(grids)]
    top = states[-1]
    O.local_rhs(top.b, implicit)
    top.x[...] = x0
    O.broadcast_interfaces(top.x, implicit, grids)
    O.apply_constraint(top.x, grids, constraint, implicit)
    norms, xs = [], []
    for _ in range(cycles):
        O.vcycle(implicit, base_level, ops, states, grids, smoothing_steps)
        xs.append(top.x.copy(order="F"))
        O.zero_out_all_but_one(top.r, implicit, grids)
        norms.append(float(np.linalg.norm(top.r)))
    return norms, xs, a


def run_device(hmg, ctx, O, cycles=100, n=32, grids=3, lam=1.0, smoothing_steps=1, seed=2019, high=9.0):
    base, sigma_grid, x0 = inputs(O, n, grids, seed, high)
    a = O.conductivity_per_element(base, sigma_grid, (0.0, 0.0))
    g = hmg.ImplicitFineGrid(ctx, hmg.Mesh(base.nodes, base.elements + 1), grids)
    op = hmg.L2PlusDivAGrad(g, lam, a)
    base_level = hmg.BaseLevel(g)
    states = [hmg.LevelState(g, i + 1) for i in range(grids)]
    top = states[-1]
    hmg.local_rhs(top.b, g)
    top.x.from_host(x0)
    hmg.broadcast_interfaces(top.x, g, grids)
    hmg.apply_constraint(top.x, grids, g)
    norms, xs = [], []
    for _ in range(cycles):
        hmg.vcycle(g, base_level, [op] * grids, states, grids, smoothing_steps)
        xs.append(top.x.to_host())
        norms.append(hmg.norm_unique(top.r))          # zero_out_all_but_one! + norm, r kept
    return norms, xs
