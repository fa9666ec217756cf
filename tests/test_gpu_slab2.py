"""
k_apply_slab2 (csrc/hmg_apply_slab.hip, round 5): the operator apply of cells larger than the LDS (3D level 7) by one persistent
workgroup per CU with loader and evaluator waves, against the kernel it replaces (k_apply_slab, option apply_slab2 = 0):
  * plain apply, residual with source and constraint: the same arithmetic per node in the same order -- equal to the last bit;
  * the fused CG pass (p-update, pending x-updates, r.r in the load phase; p.Ap): vectors to 1e-13 (the per-cell partial sums of
    the two scalars are added in another order, so alpha and beta differ in the last bits);
  * whole V-cycles; fewer workgroups than cells (a workgroup's pipeline runs across its cells), uneven shares, one cell;
  * that the path is taken (hmg_ctx_counter "slab2_launches").
The oracle comparisons of level 7 (tests/test_gpu_parity_l6.py, tests/test_gpu_fullsize_l7.py) run through this kernel as well: it
is the default.  ref: src/apply_local_operators.jl:85-133, src/multigrid.jl:46-71
"""
import numpy as np
import pytest

import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver

pytestmark = pytest.mark.gpu
L = 7


@pytest.fixture(scope="module")
def ctx():
    c = hmg.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module", params=[(3, 0), (3, 7), (3, 16), (2, 0), (1, 0)],
                ids=["162cells", "162cells-7wgs", "162cells-16wgs", "48cells", "6cells"])
def case(request, ctx):
    n, grid = request.param
    base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, n, L, seed=3, values=(1.0, 100.0), lam=0.7)
    ctx.set_option("slab2_grid", grid)
    g.ncubes = n
    yield ctx, g, op
    ctx.set_option("slab2_grid", 0)
    g.close()


def _both(ctx, fn):
    out = []
    for on in (1, 0):
        ctx.set_option("apply_slab2", on)
        try:
            n0 = ctx.counter("slab2_launches")
            out.append(fn())
            assert (ctx.counter("slab2_launches") > n0) == bool(on)
        finally:
            ctx.set_option("apply_slab2", 1)
    return out


def test_plain_apply_and_residual_are_bit_identical(case):
    ctx, g, A = case
    x = hmg.DeviceMatrix(g, L).rand(1)
    b = hmg.DeviceMatrix(g, L).rand(2)

    def run():
        y = hmg.DeviceMatrix(g, L)
        hmg.apply_ex(1.0, g, x, None, y, constrain=True)
        r = hmg.DeviceMatrix(g, L)
        hmg.apply_ex(-1.0, g, x, b, r, constrain=False)
        z = hmg.DeviceMatrix(g, L).copyto(b)
        hmg.mul(0.37, g, A, x, z)                       # y += alpha A x, alpha off the weight-cache values
        res = y.to_host(), r.to_host(), z.to_host()
        for v in (y, r, z):
            v.close()
        return res
    new, old = _both(ctx, run)
    for a, o in zip(new, old):
        assert np.abs(o).max() > 0
        np.testing.assert_array_equal(a, o)
    x.close(); b.close()


@pytest.mark.parametrize("steps", [1, 3])
def test_fused_cg_smoother_matches(case, steps):
    ctx, g, A = case

    def run():
        st = hmg.LevelState(g, L)
        st.x.rand(5); st.b.rand(6)
        hmg.broadcast_interfaces(st.x, g, L)
        hmg.apply_constraint(st.x, L, g)
        hmg.smoothing_steps(steps, g, A, st, L)
        res = [v.to_host() for v in (st.x, st.r, st.p, st.Ap)]
        st.close()
        return res
    new, old = _both(ctx, run)
    for a, o in zip(new, old):
        assert np.isfinite(a).all() and np.abs(a - o).max() <= 1e-13 * np.abs(o).max()


def test_vcycles_match(case):
    ctx, g, A = case
    if g.ncubes == 1:
        pytest.skip("one unit cube has no interior level-1 node: nothing for a V-cycle to solve")

    def run():
        states = [hmg.LevelState(g, i + 1) for i in range(L)]
        top = states[-1]
        top.x.rand(8)
        hmg.broadcast_interfaces(top.x, g, L)
        hmg.apply_constraint(top.x, L, g)
        hmg.rhs_axi_grad_v(top.b, g, driver.random_unit_vec(3))
        bl = hmg.BaseLevel(g)
        for _ in range(2):
            hmg.vcycle(g, bl, [A] * L, states, L, 3)
        res = top.x.to_host(), top.r.to_host()
        for s in states:
            s.close()
        return res
    new, old = _both(ctx, run)
    for a, o in zip(new, old):
        assert np.isfinite(a).all() and np.abs(a - o).max() <= 1e-12 * np.abs(o).max()
    # ... and twice the same bits from the new kernel (fixed assignment of cells, nodes and partial sums)
    again = run()
    np.testing.assert_array_equal(again[0], new[0])
    np.testing.assert_array_equal(again[1], new[1])


def test_restriction_through_the_window_is_bit_identical(case):
    """restrict_to! of level 7 (src/interpolation.jl:52-62) through k_apply_slab2 (eight loader waves, output slots from the
    list) against k_apply_slab: the same 15-point sums in the same order -- the coarse vector is equal to the last bit."""
    ctx, g, A = case
    xf = hmg.DeviceMatrix(g, L).rand(21)
    out = []
    for on in (1, 0):
        ctx.set_option("restrict_slab2", on)
        try:
            n0 = ctx.counter("slab2_launches")
            rc = hmg.DeviceMatrix(g, L - 1)
            hmg.restrict_to(rc, g, xf)
            out.append(rc.to_host())
            rc.close()
            assert (ctx.counter("slab2_launches") > n0) == bool(on)
        finally:
            ctx.set_option("restrict_slab2", 1)
    assert np.abs(out[1]).max() > 0
    np.testing.assert_array_equal(out[0], out[1])
    xf.close()


def test_level6_cells_through_the_window_kernel(monkeypatch):
    """Another geometry for the same kernel: level-6 cells (m = 32, 6 545 nodes) cut into slabs by a 28 KB window (HMG_SLAB_LDS_KB,
    read when the grid is created) and sent through k_apply_slab2 by option slab2_force -- other row counts per slab, other parities
    of the interior run (the paired 16-byte stores), more steps per cell -- against the LDS-resident kernels of level 6, whose
    register-blocked walk adds the same taps in another order (1e-12), and against itself (bit-reproducible)."""
    monkeypatch.setenv("HMG_SLAB_LDS_KB", "28")
    c = hmg.Context(0)
    L6 = 6
    try:
        base, cond, g, A = driver.checkerboard_problem(c, hmg.Tet64, 2, L6, seed=4, values=(1.0, 9.0), lam=0.3)
        x = hmg.DeviceMatrix(g, L6).rand(11)
        b = hmg.DeviceMatrix(g, L6).rand(12)

        def run():
            r = hmg.DeviceMatrix(g, L6)
            hmg.apply_ex(-1.0, g, x, b, r, constrain=True)
            st = hmg.LevelState(g, L6)
            st.x.rand(5); st.b.rand(6)
            hmg.broadcast_interfaces(st.x, g, L6)
            hmg.apply_constraint(st.x, L6, g)
            hmg.smoothing_steps(3, g, A, st, L6)
            res = [v.to_host() for v in (r, st.x, st.r, st.p)]
            r.close(); st.close()
            return res
        out = []
        for force in (1, 1, 0):
            c.set_option("slab2_force", force)
            n0 = c.counter("slab2_launches")
            out.append(run())
            assert (c.counter("slab2_launches") > n0) == bool(force)
        for a, a2, o in zip(*out):
            assert np.isfinite(a).all() and np.abs(o).max() > 0
            np.testing.assert_array_equal(a, a2)
            assert np.abs(a - o).max() <= 1e-12 * np.abs(o).max()
        x.close(); b.close(); g.close()
    finally:
        c.set_option("slab2_force", 0)
        c.close()
