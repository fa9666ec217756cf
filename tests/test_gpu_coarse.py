"""
Level-1 solve of the V-cycle (src/multigrid.jl:74-93; cholesky(...) \\ b of
src/examples/homogenized_coefficients.jl:259-261 replaced by a device PCG -- Jacobi, since round 4 Chebyshev iterates of the Jacobi-scaled operator): how a solve that is enqueued blindly --
no host round trip inside hmg_vcycle -- is policed.  The reference's direct solve cannot fail; here a failure must be
loud, timely and recoverable.
"""
import numpy as np
import pytest

import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver
from homogenization_jl_amd._lib import HmgError

pytestmark = pytest.mark.gpu


def _problem(ctx, w, L, values, seed=5):
    base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, w, L, seed=seed, values=values)
    states = [hmg.LevelState(g, i + 1) for i in range(L)]
    top = states[-1]
    top.x.rand(77)
    hmg.broadcast_interfaces(top.x, g, L)
    hmg.apply_constraint(top.x, L, g)
    hmg.rhs_axi_grad_v(top.b, g, driver.random_unit_vec(3))
    return base, cond, g, op, states


def test_new_operator_gets_a_counted_first_solve():
    """after hmg_grid_set_operator with a harder field the first solve must count its iterations again.  Round 2
    judged the previous matrix's pending probe first, which put the OLD count back into the budget: the first solve on
    the new system was enqueued blindly with it and, if it needed more, the next V-cycle aborted the driver."""
    ctx = hmg.Context(0)
    try:
        L, w = 3, 10
        base, cond, g, op, st = _problem(ctx, w, L, (1.0, 1.0))           # sigma = 1: an easy level-1 system
        bl = hmg.BaseLevel(g)
        for _ in range(3):
            hmg.vcycle(g, bl, [op] * L, st, L, 3)                          # budgeted solves, a probe is pending
        easy = bl.last_iterations()
        hmg.vcycle(g, bl, [op] * L, st, L, 3)                              # ... pending again when the operator changes
        rng = np.random.default_rng(1)
        hard = rng.choice([1.0, 1.0e6], size=cond.shape)
        op2 = hmg.L2PlusDivAGrad(g, 1.0, hard)
        bl2 = hmg.BaseLevel(g)
        for _ in range(3):
            hmg.vcycle(g, bl2, [op2] * L, st, L, 3)
            n = bl2.last_iterations()                                      # raises if the solve was unconverged
            assert n > 0
        assert n > easy, (easy, n)              # (contrast 1e3: 41 -> 66 iterations at this size; past 1.5 x + 16 the old budget fails)
        assert bl2.misses() == 0
        assert np.isfinite(hmg.norm_unique(st[-1].r))
    finally:
        ctx.close()


def test_budget_miss_is_reported_by_the_same_iteration_and_is_recoverable():
    """A budgeted solve that runs out of iterations is an error of the FIRST synchronising call behind its V-cycle (here
    the residual norm the driver takes right after vcycle!, src/examples/homogenized_coefficients.jl:286) -- not of the
    next V-cycle, and never silent for the last one.  The budget is dropped with it: repeating the V-cycle solves the
    slow, checked way and the run goes on.  (Forced here by tightening coarse_rtol under a budget that was counted for
    1e-13: the recurrence residual of CG does fall to 1e-40, but in three times the iterations -- twice the budget.)"""
    ctx = hmg.Context(0)
    try:
        L, w = 3, 10
        base, cond, g, op, st = _problem(ctx, w, L, (1.0, 100.0))
        bl = hmg.BaseLevel(g)
        hmg.vcycle(g, bl, [op] * L, st, L, 3)                              # first solve: counted
        assert bl.last_iterations() > 10        # (plain Jacobi-PCG: 70; with the Chebyshev preconditioner of round 4: 19)
        hmg.vcycle(g, bl, [op] * L, st, L, 3)                              # budgeted, fine
        hmg.norm_unique(st[-1].r)
        ctx.set_option("coarse_rtol", 1e-40)
        hmg.vcycle(g, bl, [op] * L, st, L, 3)                              # budgeted, cannot converge
        with pytest.raises(HmgError, match="did not reach coarse_rtol"):
            hmg.norm_unique(st[-1].r)                                      # same driver iteration
        assert bl.misses() == 1
        ctx.set_option("coarse_rtol", 1e-13)
        hmg.vcycle(g, bl, [op] * L, st, L, 3)                              # recovers: the budget was dropped, counted solve
        assert np.isfinite(hmg.norm_unique(st[-1].r))
        assert bl.last_iterations() > 0 and bl.misses() == 1
        hmg.vcycle(g, bl, [op] * L, st, L, 3)                              # budgeted again
        ctx.set_option("coarse_rtol", 1e-40)
        hmg.vcycle(g, bl, [op] * L, st, L, 3)
        with pytest.raises(HmgError, match="did not reach coarse_rtol"):
            ctx.sync()                                                     # the last V-cycle of a run is judged too
        ctx.set_option("coarse_maxit", 30)                                 # a counted solve that cannot converge fails at once
        with pytest.raises(HmgError, match="coarse_maxit"):
            hmg.vcycle(g, bl, [op] * L, st, L, 3)
    finally:
        ctx.close()


def test_driver_loops_survive_a_budget_miss():
    """vcycle_tolerant (what checkerboard_homogenization and its partitioned form call): a budgeted level-1 solve that runs out of
    iterations makes that V-cycle a weaker iterate, not the end of the run -- False is returned, the budget is dropped, the next
    cycle counts again and converges."""
    ctx = hmg.Context(0)
    try:
        L, w = 3, 10
        base, cond, g, op, st = _problem(ctx, w, L, (1.0, 100.0))
        bl = hmg.BaseLevel(g)
        assert hmg.vcycle_tolerant(g, bl, [op] * L, st, L, 3) is True        # counted
        assert hmg.vcycle_tolerant(g, bl, [op] * L, st, L, 3) is True        # budgeted
        ctx.set_option("coarse_rtol", 1e-40)
        assert hmg.vcycle_tolerant(g, bl, [op] * L, st, L, 3) is False       # budgeted, cannot converge: reported, not raised
        assert bl.misses() == 1
        ctx.set_option("coarse_rtol", 1e-13)
        r0 = hmg.norm_unique(st[-1].r)
        assert hmg.vcycle_tolerant(g, bl, [op] * L, st, L, 3) is True
        assert hmg.norm_unique(st[-1].r) < r0
    finally:
        ctx.close()
