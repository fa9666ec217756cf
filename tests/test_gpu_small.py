"""
GPU parity tests of the one-wave kernels of the small 3D levels (hmg_apply_small.hip; round 4): levels 2, 3, 4 = 10, 35, 165
nodes per cell -- k_apply_small (one cell per wave, software-pipelined over the cells a wave walks) and, for level 2,
k_apply_pack (four cells per wave, one per row of 16 lanes).

  * against the oracle (1e-11 per primitive, 1e-10 smoother state, 1e-9 / 1e-8 after V-cycles), on a mesh whose cells all differ
    and on a checkerboard, with a cell count that is not a multiple of four (the last quad of the packed kernel is ragged);
  * the packed kernel against the one-cell-per-wave kernel BIT FOR BIT, outputs and CG state (same arithmetic per node, and the
    per-cell sums are the first four stages of the same cross-lane tree);
  * both against the workgroup kernel they replace (option apply_small = 0) to rounding;
  * inside hmg_vcycle (zero initial guess never written, two pending x-updates, folded prolongation from a 4-node coarse cell);
  * more quads than persistent waves: every wave walks several (the prefetch two quads ahead wraps correctly at the end).

ref: src/apply_local_operators.jl:85-133, src/multigrid.jl:46-119, src/interpolation.jl:52-74
"""
import numpy as np
import pytest

import homogenization_jl_amd as hmg
from test_gpu_parity import Case, TOL, _oracle_state, relerr

pytestmark = pytest.mark.gpu
MODES = {"pack": {}, "one_per_wave": {"apply_pack": 0}, "workgroup": {"apply_small": 0}}


@pytest.fixture(scope="module")
def ctx():
    c = hmg.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def perturbed(oracle, ctx):
    # 3^3 cubes x 6 tets = 162 cells (162 = 4 * 40 + 2: a ragged last quad), every one with its own coefficient row
    return Case(oracle, ctx, 3, 3, 4, lam=0.7, perturb=0.2, seed=5)


@pytest.fixture(scope="module")
def board(oracle, ctx):
    return Case(oracle, ctx, 3, 4, 4, lam=1.0, perturb=0.0, seed=6)


class mode:
    def __init__(self, ctx, name):
        self.ctx, self.opts = ctx, MODES[name]

    def __enter__(self):
        for k, v in self.opts.items():
            self.ctx.set_option(k, v)

    def __exit__(self, *exc):
        for k in self.opts:
            self.ctx.set_option(k, 1)


@pytest.mark.parametrize("lev", [2, 3, 4])
@pytest.mark.parametrize("which", ["perturbed", "board"])
def test_small_apply(request, ctx, which, lev):
    """mul! with alpha = -1 and a source vector, the plain apply with and without the constraint."""
    c = request.getfixturevalue(which)
    x, y = c.rand(lev), c.rand(lev)
    dx = c.dev(lev, x)
    got = {}
    for name in MODES:
        with mode(ctx, name):
            n0 = ctx.counter("small_launches")
            dy = c.dev(lev, y)
            hmg.mul(-1.0, c.g, c.A, dx, dy)
            dz = hmg.DeviceMatrix(c.g, lev)
            hmg.apply_ex(1.0, c.g, dx, None, dz, constrain=True)
            dw = hmg.DeviceMatrix(c.g, lev)
            hmg.apply_ex(1.0, c.g, dx, None, dw, constrain=False)
            got[name] = (dy.to_host(), dz.to_host(), dw.to_host())
            assert (ctx.counter("small_launches") - n0 == 3) == (name != "workgroup"), name
    want = y.copy(order="F")
    c.O.mul(-1.0, c.mesh, c.ops[lev - 1], x, want)
    assert relerr(got["pack"][0], want) <= TOL
    want2 = np.zeros_like(x, order="F")
    c.O.mul(1.0, c.mesh, c.ops[lev - 1], x, want2)
    assert relerr(got["pack"][2], want2) <= TOL
    c.O.apply_constraint(want2, lev, c.cons, c.impl)
    assert relerr(got["pack"][1], want2) <= TOL
    for a, b, w in zip(got["pack"], got["one_per_wave"], got["workgroup"]):
        np.testing.assert_array_equal(a, b)
        assert relerr(a, w) <= 1e-13


@pytest.mark.parametrize("lev", [2, 3, 4])
@pytest.mark.parametrize("which", ["perturbed", "board"])
@pytest.mark.parametrize("steps", [3, 1])
def test_small_smoothing_steps(request, ctx, which, lev, steps):
    """smoothing_steps! (fused CG pass): oracle 1e-10 on x, r, p, Ap; packed against one cell per wave bit for bit."""
    c = request.getfixturevalue(which)
    st = _oracle_state(c, lev)
    x0, b0 = st.x.copy(order="F"), st.b.copy(order="F")
    c.O.smoothing_steps(steps, c.impl, c.ops[lev - 1], st, lev)
    got = {}
    for name in MODES:
        with mode(ctx, name):
            dst = hmg.LevelState(c.g, lev)
            dst.x.from_host(x0); dst.b.from_host(b0)
            hmg.smoothing_steps(steps, c.g, c.A, dst, lev)
            got[name] = {n: getattr(dst, n).to_host() for n in ("x", "r", "p", "Ap")}
    for n in ("x", "r", "p", "Ap"):
        assert relerr(got["pack"][n], getattr(st, n)) <= 1e-10, n
        np.testing.assert_array_equal(got["pack"][n], got["one_per_wave"][n])
        assert relerr(got["pack"][n], got["workgroup"][n]) <= 1e-12, n


@pytest.mark.parametrize("which,levels", [("board", 4), ("perturbed", 3), ("perturbed", 2)])
def test_small_vcycles(oracle, ctx, which, levels):
    """Two V-cycles over levels 1..levels: oracle x 1e-9, r 1e-8; every mode of the small kernels, the packed one bit for bit
    with one cell per wave (level 2 is entered with the zero initial guess nobody writes, left through the dead tails, and its
    coarse-grid correction comes from the 4-node cells of level 1)."""
    O = oracle
    c = Case(O, ctx, 3, 3, levels, lam=1.0, perturb=0.15 if which == "perturbed" else 0.0, seed=21)
    sts = [O.LevelState.create(c.mesh.nelements(), c.impl.nf(i + 1)) for i in range(levels)]
    sts[-1] = _oracle_state(c, levels)
    x0, b0 = sts[-1].x.copy(order="F"), sts[-1].b.copy(order="F")
    base, dbase = O.make_base_level(c.mesh, c.sig, c.lam), hmg.BaseLevel(c.g)
    for cyc in range(2):
        O.vcycle(c.impl, base, c.ops, sts, levels, 3)
    got = {}
    for name in MODES:
        with mode(ctx, name):
            dsts = [hmg.LevelState(c.g, i + 1) for i in range(levels)]
            dsts[-1].x.from_host(x0); dsts[-1].b.from_host(b0)
            for cyc in range(2):
                hmg.vcycle(c.g, dbase, [c.A] * levels, dsts, levels, 3)
            got[name] = (dsts[-1].x.to_host(), dsts[-1].r.to_host())
    assert relerr(got["pack"][0], sts[-1].x) <= 1e-9
    assert relerr(got["pack"][1], sts[-1].r) <= 1e-8
    np.testing.assert_array_equal(got["pack"][0], got["one_per_wave"][0])
    np.testing.assert_array_equal(got["pack"][1], got["one_per_wave"][1])
    assert relerr(got["pack"][0], got["workgroup"][0]) <= 1e-11


def test_small_kernels_walk_more_cells_than_waves(ctx):
    """14^3 cubes = 16 464 cells: 4 116 quads for at most 4 096 persistent waves (and 16 464 cells for 5 120): the tail of the
    software pipeline.  Packed against one cell per wave bit for bit, three levels, two V-cycles."""
    from homogenization_jl_amd import driver
    levels = 3
    base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, 14, levels, seed=3)
    res = {}
    try:
        for name in ("pack", "one_per_wave"):
            with mode(ctx, name):
                st = [hmg.LevelState(g, i + 1) for i in range(levels)]
                st[-1].x.rand(3); st[-1].b.rand(4)
                hmg.broadcast_interfaces(st[-1].x, g, levels)
                hmg.apply_constraint(st[-1].x, levels, g)
                bl = hmg.BaseLevel(g)
                for _ in range(2):
                    hmg.vcycle(g, bl, [op] * levels, st, levels, 3)
                res[name] = (st[-1].x.to_host(), st[-1].r.to_host())
                for s in st:
                    s.close()
    finally:
        g.close()
    assert np.isfinite(res["pack"][0]).all() and np.abs(res["pack"][0]).max() > 0
    np.testing.assert_array_equal(res["pack"][0], res["one_per_wave"][0])
    np.testing.assert_array_equal(res["pack"][1], res["one_per_wave"][1])
