"""
The BASELINE.json configurations that fit a test run.
  config 1: checkerboard_homogenization(3, Tri64, refinements=3, tolerance=1e-3) -- the reference's own
            CPU-runnable 2D case (plumbing): oracle on CPU, and the product driver on the GPU against it.
  config 2: checkerboard_homogenization(1, Tet64, refinements=4) -- single-GPU correctness vs the CPU oracle.
The reference draws sigma and x0 from an unseeded RNG, so its published sigma values are trend checks only
(BASELINE.md); here both sides get identical seeded arrays.
"""
import numpy as np
import pytest

import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver


def _inputs(dim, n, refinements, seed):
    width = 2 * (driver.compute_box_radius(0, n) + driver.compute_boundary_layer(1.0, n))
    sgrid = driver.generate_conductivity(dim, width, seed)
    nf_tab = {2: [3, 6, 15, 45, 153], 3: [4, 10, 35, 165, 969]}[dim]
    ne = (2 if dim == 2 else 6) * width ** dim
    return sgrid, hmg.host_random((nf_tab[refinements], ne), seed + 100)


@pytest.fixture(scope="module")
def config1_oracle(oracle):
    sgrid, x0 = _inputs(2, 3, 3, 21)
    sigma, hist = oracle.checkerboard_homogenization(n=3, dim=2, refinements=3, tolerance=1e-3, sigma_grid=sgrid, x0=x0)
    return sgrid, x0, sigma, hist


def test_config1_cpu_reference_path(config1_oracle):
    """BASELINE config 1 on the CPU: the multigrid contracts every cycle, the outer loop shrinks the domain and
    sigma is a positive correction of the expected order (published runs: 1.6-1.95 for n=5)."""
    _, _, sigma, hist = config1_oracle
    ks = sorted({h[0] for h in hist})
    for k in ks:
        r = [h[2] for h in hist if h[0] == k]
        assert all(b < a for a, b in zip(r, r[1:]))       # residual norm decreases every V-cycle
    assert 0.5 < sigma < 3.0


@pytest.mark.gpu
def test_config1_device_driver_matches_cpu(config1_oracle):
    sgrid, x0, want, hist_o = config1_oracle
    ctx = hmg.Context(0)
    got, hist_d = driver.checkerboard_homogenization(3, hmg.Tri64, refinements=3, tolerance=1e-3, ctx=ctx,
                                                     sigma_grid=sgrid, x0=x0)
    assert len(hist_d) == len(hist_o)
    assert abs(got - want) <= 1e-8


@pytest.mark.gpu
def test_driver_with_domain_shrink_matches_cpu(oracle):
    """n = 5 is the smallest size whose outer loop actually shrinks the domain (56 -> 55 -> ...): exercises
    hmg_grid_shrink, the new Dirichlet boundary, v_prev, next_rhs! and integrate_terms inside the driver."""
    sgrid, x0 = _inputs(2, 5, 2, 31)
    want, hist_o = oracle.checkerboard_homogenization(n=5, dim=2, refinements=2, tolerance=1e-3, sigma_grid=sgrid, x0=x0)
    assert len({h[0] for h in hist_o}) >= 2               # at least one shrink happened
    ctx = hmg.Context(0)
    got, hist_d = driver.checkerboard_homogenization(5, hmg.Tri64, refinements=2, tolerance=1e-3, ctx=ctx,
                                                     sigma_grid=sgrid, x0=x0)
    assert [h[:2] for h in hist_d] == [h[:2] for h in hist_o]
    assert abs(got - want) <= 1e-8
    for a, b in zip(hist_o, hist_d):
        assert abs(a[3] - b[3]) <= 1e-8


@pytest.mark.gpu
def test_config2_single_gpu_vs_cpu(oracle):
    """checkerboard_homogenization(1, Tet64, refinements=4): 20^3 cubes, 48 000 cells, L=5, 4.65e7 fine DOFs."""
    sgrid, x0 = _inputs(3, 1, 4, 5)
    want, hist_o = oracle.checkerboard_homogenization(n=1, dim=3, refinements=4, tolerance=1e-3, sigma_grid=sgrid, x0=x0)
    ctx = hmg.Context(0)
    got, hist_d = driver.checkerboard_homogenization(1, hmg.Tet64, refinements=4, tolerance=1e-3, ctx=ctx,
                                                     sigma_grid=sgrid, x0=x0)
    assert len(hist_d) == len(hist_o)
    assert abs(got - want) <= 1e-8
    for a, b in zip(hist_o, hist_d):
        assert abs(a[2] - b[2]) <= 1e-7 * a[2]
