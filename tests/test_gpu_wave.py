"""
GPU parity tests of the one-wave-per-cell apply of level 5 (k_apply_wave, hmg_apply_wave.hip; round 4): 3D, m = 16,
969 nodes per cell -- the cell of BASELINE config 3's second-finest level and of config 2's finest one.

  * against the oracle (1e-11 per primitive, 1e-10 smoother state, 1e-9 / 1e-8 after V-cycles), on a mesh whose cells all
    differ (perturbed geometry: one class-weight row set per cell) and on a checkerboard (a handful of classes);
  * against the 256-thread workgroup kernel it replaces (option apply_wave = 0): every OUTPUT value bit for bit -- the weights
    come from the class cache, formed by the same seven products in the same order, and every node sums its taps in the same
    order (45 edge nodes per cell excepted: one fused multiply-add is contracted the other way round, see the test) -- and the
    CG state to rounding (the per-cell partial sums of p.Ap and r.r are formed over another distribution of the
    nodes over the lanes);
  * the pieces that only exist inside hmg_vcycle (dead tail of the pre-smoother, local residual with two pending x-updates and
    the restriction in its epilogue, coarse-grid correction staged in the image), through hmg_vcycle_down / hmg_vcycle_up;
  * that the path is actually taken (hmg_ctx_counter "wave_launches"), and what falls back to the workgroup kernel.

ref: src/apply_local_operators.jl:85-133, src/multigrid.jl:46-119, src/interpolation.jl:52-74
"""
import numpy as np
import pytest

import homogenization_jl_amd as hmg
from test_gpu_parity import Case, TOL, _oracle_state, relerr

pytestmark = pytest.mark.gpu
LEV = 5
OPTS = ("swap_rp", "fold_x", "fold_prolong", "lazy_dead", "lean_post", "prolong_in_image", "fold_restrict")


@pytest.fixture(scope="module")
def ctx():
    c = hmg.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def case5(oracle, ctx):
    # 3^3 cubes x 6 tets = 162 cells, every one with its own coefficient row (perturbed nodes): 162 classes
    return Case(oracle, ctx, 3, 3, LEV, lam=0.7, perturb=0.2, seed=5)


@pytest.fixture(scope="module")
def board5(oracle, ctx):
    # unperturbed cubes, sigma in {1, 9}^3: at most 48 classes
    return Case(oracle, ctx, 3, 4, LEV, lam=1.0, perturb=0.0, seed=6)


def _launches(ctx):
    return ctx.counter("wave_launches")


@pytest.mark.parametrize("which", ["case5", "board5"])
def test_wave_apply_matches_oracle_and_the_workgroup_kernel(request, ctx, which):
    """mul! with and without a source vector, with and without the constraint: oracle 1e-11, workgroup kernel bit for bit."""
    c = request.getfixturevalue(which)
    x, y = c.rand(LEV), c.rand(LEV)
    dx = c.dev(LEV, x)
    got = {}
    for wave in (1, 0):
        ctx.set_option("apply_wave", wave)
        try:
            n0 = _launches(ctx)
            dy = c.dev(LEV, y)
            hmg.mul(-1.0, c.g, c.A, dx, dy)                      # y += -1 A x  (alpha = -1: the cached weights' other sign)
            dz = hmg.DeviceMatrix(c.g, LEV)
            hmg.apply_ex(1.0, c.g, dx, None, dz, constrain=True)
            dw = hmg.DeviceMatrix(c.g, LEV)
            hmg.apply_ex(1.0, c.g, dx, None, dw, constrain=False)
            got[wave] = (dy.to_host(), dz.to_host(), dw.to_host())
            assert (_launches(ctx) - n0 == 3) == (wave == 1)
        finally:
            ctx.set_option("apply_wave", 1)
    want = y.copy(order="F")
    c.O.mul(-1.0, c.mesh, c.ops[LEV - 1], x, want)
    assert relerr(got[1][0], want) <= TOL
    want2 = np.zeros_like(x, order="F")
    c.O.mul(1.0, c.mesh, c.ops[LEV - 1], x, want2)
    assert relerr(got[1][2], want2) <= TOL
    c.O.apply_constraint(want2, LEV, c.cons, c.impl)
    assert relerr(got[1][1], want2) <= TOL
    # Bit for bit -- except on the 45 nodes of the three edges of the slanted face i+j+k = m, where the two kernels round
    # differently: the workgroup kernel leaves out the absent tap (i+1,j,k) at compile time, the wave kernel multiplies it by its
    # zero weight, and the backend fuses `w0 x0 + w x` into fma(w0, x0, round(w x)) with whichever product comes second in the
    # source -- w2 x2 there, the exact zero here.  Both are the same sum to the last rounding (<= a few ulp of the terms).
    slot = c.g.table_i32("hier2slot", LEV)
    slanted = (slot >= 49) & (slot < 94)
    for a, b in zip(got[1], got[0]):
        np.testing.assert_array_equal(a[~slanted], b[~slanted])
        assert np.abs(a[slanted] - b[slanted]).max() <= 8 * np.finfo(float).eps * np.abs(b).max()


def test_wave_falls_back_where_the_cache_does_not_apply(case5, ctx):
    """alpha other than +-1 (weights are cached for |alpha| = 1), and the mass-only form of next_rhs!: the workgroup kernel
    runs, results against the oracle as before."""
    c = case5
    x, y = c.rand(LEV), c.rand(LEV)
    n0 = _launches(ctx)
    want = y.copy(order="F")
    c.O.mul(-1.3, c.mesh, c.ops[LEV - 1], x, want)
    dy = c.dev(LEV, y)
    hmg.mul(-1.3, c.g, c.A, c.dev(LEV, x), dy)
    assert relerr(dy.to_host(), want) <= TOL
    assert _launches(ctx) == n0


def test_wave_follows_a_new_lambda(oracle, ctx):
    """hmg_grid_set_lambda / a new operator: the cached weights are formed again before the next apply."""
    O = oracle
    c = Case(O, ctx, 3, 2, LEV, lam=0.7, seed=9)
    x = c.rand(LEV)
    for lam in (0.7, 2.5):
        A = hmg.L2PlusDivAGrad(c.g, lam, c.sig)
        ops = O.L2PlusDivAGrad(O.build_local_diffusion_operators(c.impl.reference.levels[LEV - 1]),
                               O.mass_matrix(c.impl.reference.levels[LEV - 1]), c.cons, lam, c.sig)
        want = np.zeros_like(x, order="F")
        O.mul(1.0, c.mesh, ops, x, want)
        n0 = _launches(ctx)
        dz = hmg.DeviceMatrix(c.g, LEV)
        A._bind()
        hmg.apply_ex(1.0, c.g, c.dev(LEV, x), None, dz, constrain=False)
        assert _launches(ctx) == n0 + 1
        assert relerr(dz.to_host(), want) <= TOL


def test_wave_residual_and_constraint(case5):
    c = case5
    O = c.O
    st = O.LevelState.create(c.mesh.nelements(), c.impl.nf(LEV))
    st.x[...] = c.rand(LEV); st.b[...] = c.rand(LEV)
    O.local_residual(c.impl, c.ops[LEV - 1], st, LEV)
    dst = hmg.LevelState(c.g, LEV)
    dst.x.from_host(st.x); dst.b.from_host(st.b)
    hmg.local_residual(c.g, c.A, dst, LEV)
    got = dst.r.to_host()
    assert relerr(got, st.r) <= TOL
    np.testing.assert_array_equal(got == 0.0, st.r == 0.0)


@pytest.mark.parametrize("which", ["case5", "board5"])
@pytest.mark.parametrize("steps", [3, 1])
def test_wave_smoothing_steps(request, ctx, which, steps):
    """smoothing_steps! (fused CG pass) -- oracle 1e-10 on x, r, p, Ap; the workgroup kernel's state to 1e-12."""
    c = request.getfixturevalue(which)
    st = _oracle_state(c, LEV)
    x0, b0 = st.x.copy(order="F"), st.b.copy(order="F")
    c.O.smoothing_steps(steps, c.impl, c.ops[LEV - 1], st, LEV)
    got = {}
    for wave in (1, 0):
        ctx.set_option("apply_wave", wave)
        try:
            n0 = _launches(ctx)
            dst = hmg.LevelState(c.g, LEV)
            dst.x.from_host(x0); dst.b.from_host(b0)
            hmg.smoothing_steps(steps, c.g, c.A, dst, LEV)
            got[wave] = {n: getattr(dst, n).to_host() for n in ("x", "r", "p", "Ap")}
            assert (_launches(ctx) > n0) == (wave == 1)
        finally:
            ctx.set_option("apply_wave", 1)
    for name in ("x", "r", "p", "Ap"):
        assert relerr(got[1][name], getattr(st, name)) <= 1e-10, name
        assert relerr(got[1][name], got[0][name]) <= 1e-12, name


@pytest.mark.parametrize("plain", [0, 1])
@pytest.mark.parametrize("steps", [3, 2, 1])
def test_wave_vcycle_down_leg(case5, ctx, steps, plain):
    """smoothing_steps!, local_residual!, restrict_to!, fill!(next.x, 0) (src/multigrid.jl:100-106) on level 5: plain = 0 is what
    hmg_vcycle runs (dead tail, `x3` mode, restriction in the epilogue).  x, the cell-local residual and the coarse right-hand
    side against the oracle; the epilogue restriction against the stand-alone kernel bit for bit."""
    c = case5
    O = c.O
    st = _oracle_state(c, LEV)
    x0, b0 = st.x.copy(order="F"), st.b.copy(order="F")
    O.smoothing_steps(steps, c.impl, c.ops[LEV - 1], st, LEV)
    O.local_residual(c.impl, c.ops[LEV - 1], st, LEV)
    nb = np.zeros((c.impl.nf(LEV - 1), c.mesh.nelements()), order="F")
    O.restrict_to(nb, c.impl.reference.interops[LEV - 2], st.r)
    for o in OPTS:
        ctx.set_option(o, 0 if plain else 1)
    try:
        states = [None] * LEV
        states[LEV - 2], states[LEV - 1] = hmg.LevelState(c.g, LEV - 1), hmg.LevelState(c.g, LEV)
        states[LEV - 1].x.from_host(x0); states[LEV - 1].b.from_host(b0)
        states[LEV - 2].x.from_host(c.rand(LEV - 1))                     # must come back as zeros
        n0 = _launches(ctx)
        hmg.vcycle_down(c.g, [c.A] * LEV, states, LEV, steps)
        assert _launches(ctx) > n0
        assert relerr(states[LEV - 1].x.to_host(), st.x) <= 1e-10
        assert relerr(states[LEV - 1].r.to_host(), st.r) <= 1e-10
        assert relerr(states[LEV - 2].b.to_host(), nb) <= 1e-10
        assert not states[LEV - 2].x.to_host().any()
        if plain == 0:
            got_b, got_r = states[LEV - 2].b.to_host(), states[LEV - 1].r.to_host()
            ctx.set_option("fold_restrict", 0)
            states[LEV - 1].x.from_host(x0); states[LEV - 1].b.from_host(b0)
            hmg.vcycle_down(c.g, [c.A] * LEV, states, LEV, steps)
            np.testing.assert_array_equal(states[LEV - 2].b.to_host(), got_b)
            np.testing.assert_array_equal(states[LEV - 1].r.to_host(), got_r)
    finally:
        for o in OPTS:
            ctx.set_option(o, 1)


@pytest.mark.parametrize("plain", [0, 1])
def test_wave_vcycle_up_leg(case5, ctx, plain):
    """interpolate_and_sum_to!(curr.x, P, next.x), smoothing_steps! (src/multigrid.jl:112-115): plain = 0 folds the coarse-grid
    correction into the post-smoother's first residual -- the coarse column staged at the even nodes of the wave's image."""
    c = case5
    O, steps = c.O, 3
    st = _oracle_state(c, LEV)
    x0, b0 = st.x.copy(order="F"), st.b.copy(order="F")
    xc = c.rand(LEV - 1)
    O.broadcast_interfaces(xc, c.impl, LEV - 1)
    O.apply_constraint(xc, LEV - 1, c.cons, c.impl)
    O.interpolate_and_sum_to(st.x, c.impl.reference.interops[LEV - 2], xc)
    O.smoothing_steps(steps, c.impl, c.ops[LEV - 1], st, LEV)
    for o in OPTS:
        ctx.set_option(o, 0 if plain == 1 else 1)
    try:
        states = [None] * LEV
        states[LEV - 2], states[LEV - 1] = hmg.LevelState(c.g, LEV - 1), hmg.LevelState(c.g, LEV)
        states[LEV - 1].x.from_host(x0); states[LEV - 1].b.from_host(b0)
        states[LEV - 2].x.from_host(xc)
        hmg.vcycle_up(c.g, [c.A] * LEV, states, LEV, steps)
        assert relerr(states[LEV - 1].x.to_host(), st.x) <= 1e-10
        assert relerr(states[LEV - 1].r.to_host(), st.r) <= 1e-10
        if plain == 1:
            assert relerr(states[LEV - 1].p.to_host(), st.p) <= 1e-10
            assert relerr(states[LEV - 1].Ap.to_host(), st.Ap) <= 1e-10
    finally:
        for o in OPTS:
            ctx.set_option(o, 1)


@pytest.mark.parametrize("which,levels", [("board", 5), ("board", 6), ("perturbed", 5)])
def test_wave_vcycles_match_oracle(oracle, ctx, which, levels):
    """Two V-cycles with level 5 as the finest level and as the second-finest one (BASELINE config 3's shape: level 5 is
    entered with the zero initial guess nobody writes and left through the dead tails): x 1e-9, r 1e-8."""
    O = oracle
    c = Case(O, ctx, 3, 2, levels, lam=1.0, perturb=0.15 if which == "perturbed" else 0.0, seed=21)
    sts = [O.LevelState.create(c.mesh.nelements(), c.impl.nf(i + 1)) for i in range(levels)]
    sts[-1] = _oracle_state(c, levels)
    dsts = [hmg.LevelState(c.g, i + 1) for i in range(levels)]
    dsts[-1].x.from_host(sts[-1].x); dsts[-1].b.from_host(sts[-1].b)
    base, dbase = O.make_base_level(c.mesh, c.sig, c.lam), hmg.BaseLevel(c.g)
    n0 = _launches(ctx)
    for cyc in range(2):
        O.vcycle(c.impl, base, c.ops, sts, levels, 3)
        hmg.vcycle(c.g, dbase, [c.A] * levels, dsts, levels, 3)
        assert relerr(dsts[-1].x.to_host(), sts[-1].x) <= 1e-9, cyc
        assert relerr(dsts[-1].r.to_host(), sts[-1].r) <= 1e-8, cyc
    assert _launches(ctx) - n0 >= 2 * 6


def test_wave_walks_more_cells_than_waves(oracle, ctx):
    """A grid of fewer persistent waves than cells (option wave_grid): every wave walks several cells; results unchanged."""
    c = Case(oracle, ctx, 3, 3, LEV, lam=0.7, perturb=0.2, seed=5)
    x = c.rand(LEV)
    dx = c.dev(LEV, x)
    ref = hmg.DeviceMatrix(c.g, LEV)
    hmg.apply_ex(1.0, c.g, dx, None, ref, constrain=True)
    try:
        ctx.set_option("wave_grid_total", 24)   # 24 waves for 162 cells: 6-7 cells per wave
        out = hmg.DeviceMatrix(c.g, LEV)
        hmg.apply_ex(1.0, c.g, dx, None, out, constrain=True)
        np.testing.assert_array_equal(out.to_host(), ref.to_host())
    finally:
        ctx.set_option("wave_grid", 16)
