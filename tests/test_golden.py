"""
Committed golden vectors (tests/golden/*.npz, made by tools/make_golden.py from the CPU oracle with fixed
seeds -- the reference is pure Julia, cannot run in the build image and ships no data files).
  CPU: the oracle still reproduces them (pins the checker against drift).
  GPU: the product, through the C ABI, reproduces them.
"""
import os

import numpy as np
import pytest

import homogenization_jl_amd as hmg

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["tet_2x2x2_L4.npz", "tri_4x4_L5.npz"]


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _oracle_setup(O, d):
    m = O.Mesh(d["nodes"], d["cells"])
    L = int(d["levels"])
    impl = O.ImplicitFineGrid.create(m, L)
    cons = O.ZeroDirichletConstraint(*O.list_boundary_nodes_edges_faces(m))
    ops = [O.L2PlusDivAGrad(O.build_local_diffusion_operators(l), O.mass_matrix(l), cons, float(d["lam"]), d["sigma"])
           for l in impl.reference.levels]
    return m, L, impl, cons, ops


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(oracle, name):
    O = oracle
    d = np.load(os.path.join(GOLD, name))
    m, L, impl, cons, ops = _oracle_setup(O, d)
    for lev in range(1, L + 1):
        out = np.asfortranarray(d[f"apply_y_{lev}"].copy())
        O.mul(-1.3, m, ops[lev - 1], np.asfortranarray(d[f"apply_x_{lev}"]), out)
        assert relerr(out, d[f"apply_out_{lev}"]) <= 1e-14
        s = np.asfortranarray(d[f"apply_x_{lev}"].copy()); O.broadcast_interfaces(s, impl, lev)
        np.testing.assert_array_equal(s, d[f"isum_out_{lev}"])
    st = [O.LevelState.create(m.nelements(), impl.nf(i + 1)) for i in range(L)]
    st[-1].x[...] = d["mg_x0"]; st[-1].b[...] = d["mg_b"]
    base = O.make_base_level(m, d["sigma"], float(d["lam"]))
    for cyc in range(3):
        O.vcycle(impl, base, ops, st, L, 3)
        assert relerr(st[-1].x, d[f"vcycle_x_{cyc + 1}"]) <= 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_device_reproduces_golden(name):
    d = np.load(os.path.join(GOLD, name))
    L = int(d["levels"])
    ctx = hmg.Context(0)
    g = hmg.ImplicitFineGrid(ctx, hmg.Mesh(d["nodes"], d["cells"] + 1), L)
    A = hmg.L2PlusDivAGrad(g, float(d["lam"]), d["sigma"])
    dev = lambda lev, a: hmg.DeviceMatrix(g, lev).from_host(a)
    for lev in range(1, L + 1):
        x, y = dev(lev, d[f"apply_x_{lev}"]), dev(lev, d[f"apply_y_{lev}"])
        hmg.mul(-1.3, g, A, x, y)
        assert relerr(y.to_host(), d[f"apply_out_{lev}"]) <= 1e-11
        s = dev(lev, d[f"apply_x_{lev}"]); hmg.broadcast_interfaces(s, g, lev)
        np.testing.assert_array_equal(s.to_host(), d[f"isum_out_{lev}"])
        assert abs(hmg.norm_unique(s) - np.linalg.norm(d[f"unique_out_{lev}"])) <= 1e-12 * np.linalg.norm(d[f"unique_out_{lev}"])
        hmg.zero_out_all_but_one(s, g, lev)
        np.testing.assert_array_equal(s.to_host(), d[f"unique_out_{lev}"])
        c = dev(lev, d[f"apply_x_{lev}"]); hmg.apply_constraint(c, lev, g)
        np.testing.assert_array_equal(c.to_host(), d[f"constraint_out_{lev}"])
        if lev >= 2:
            yf = dev(lev, d[f"apply_y_{lev}"])
            hmg.interpolate_and_sum_to(yf, g, dev(lev - 1, d[f"prolong_xc_{lev}"]))
            np.testing.assert_array_equal(yf.to_host(), d[f"prolong_out_{lev}"])
            bc = hmg.DeviceMatrix(g, lev - 1)
            hmg.restrict_to(bc, g, dev(lev, d[f"apply_x_{lev}"]))
            assert relerr(bc.to_host(), d[f"restrict_out_{lev}"]) <= 1e-14
    st = hmg.LevelState(g, L)
    st.x.from_host(d["mg_x0"]); st.b.from_host(d["mg_b"])
    hmg.smoothing_steps(3, g, A, st, L)
    assert relerr(st.x.to_host(), d["smooth_x"]) <= 1e-10
    assert relerr(st.r.to_host(), d["smooth_r"]) <= 1e-10
    assert relerr(st.p.to_host(), d["smooth_p"]) <= 1e-10
    sts = [hmg.LevelState(g, i + 1) for i in range(L)]
    sts[-1].x.from_host(d["mg_x0"]); sts[-1].b.from_host(d["mg_b"])
    base = hmg.BaseLevel(g)
    for cyc in range(3):
        hmg.vcycle(g, base, [A] * L, sts, L, 3)
        assert relerr(sts[-1].x.to_host(), d[f"vcycle_x_{cyc + 1}"]) <= 1e-9
        assert abs(hmg.norm_unique(sts[-1].r) - d["vcycle_rnorms"][cyc]) <= 1e-8 * d["vcycle_rnorms"][cyc]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["driver_tri_n1_r2.npz", "driver_tet_n0_r2.npz"])
def test_device_driver_reproduces_golden_sigma(name):
    from homogenization_jl_amd import driver
    d = np.load(os.path.join(GOLD, name))
    dim = int(d["dim"])
    ctx = hmg.Context(0)
    sigma, hist = driver.checkerboard_homogenization(int(d["n"]), hmg.Tet64 if dim == 3 else hmg.Tri64,
                                                     refinements=int(d["refinements"]), tolerance=float(d["tolerance"]),
                                                     ctx=ctx, sigma_grid=d["sigma_grid"], x0=d["x0"])
    assert len(hist) == d["history"].shape[0]
    assert abs(sigma - float(d["sigma"])) <= 1e-8
    np.testing.assert_allclose(np.array(hist)[:, 3], d["history"][:, 3], rtol=0, atol=1e-8)
