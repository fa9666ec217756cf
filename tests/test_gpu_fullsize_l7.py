"""
BASELINE config 5's per-GPU share at full size: 16^3 unit cubes x 6 tetrahedra = 24 576 cells, L = 7 (47 905 nodes per cell: the
cell does not fit the LDS, k_apply_slab walks it in slabs of k-planes), conductivities in {1, 100} -- 1.18e9 fine DOFs, 9.4 GB per
level vector.  Round 4 held this kernel path against the oracle on 384 cells only (tests/test_gpu_parity_l6.py); here, at the size
the 8-GPU run gives every rank:
  * operator apply, restriction and prolongation against the oracle on sampled cells (they are cell-local), 1e-11 / 1e-14 / exact;
  * interface sum, Dirichlet constraint and first-copy mask against the oracle on all 1.18e9 entries, bit for bit;
  * hmg_vcycle with every exact saving on and off: x and r bit-identical after two V-cycles, no allocation inside them;
  * bit-reproducibility on fresh memory, contraction of the residual, a bounded level-1 iteration count.
ref: src/apply_local_operators.jl:85-133, src/implicit_fine_grid.jl:94-386, src/interpolation.jl:52-74, src/multigrid.jl:73-119
"""
import numpy as np
import pytest

import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver

pytestmark = pytest.mark.gpu
W, L = 16, 7
EXACT = ("lean_post", "lazy_post", "lazy_top", "lazy_dead", "fold_x", "swap_rp", "fold_prolong", "prolong_in_image", "fold_faces",
         "fold_restrict", "zero_entry", "cell_order")


def _device_too_small():
    try:
        import torch
        if not torch.cuda.is_available():
            return True
        free, _ = torch.cuda.mem_get_info(0)
        return free < 150e9               # two sets of level states (2 x 56 GB) + the spare vector + tables
    except Exception:
        return True


@pytest.fixture(scope="module")
def prob():
    if _device_too_small():
        pytest.skip("needs 150 GB of free device memory")
    ctx = hmg.Context(0)
    base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, W, L, seed=5, values=(1.0, 100.0), lam=0.5)
    op.base_mesh, op.cond = base, cond
    yield ctx, g, op
    g.close()
    ctx.close()


def _start(g, states, seed):
    top = states[-1]
    top.x.rand(seed)
    hmg.broadcast_interfaces(top.x, g, L)
    hmg.apply_constraint(top.x, L, g)
    hmg.rhs_axi_grad_v(top.b, g, np.array([1.0, 1.0, 1.0]) / np.sqrt(3.0))
    return top


def test_level7_apply_and_transfers_match_oracle_on_sampled_cells(prob, oracle):
    ctx, g, A = prob
    O = oracle
    ne = g.ncells()
    assert ne == 6 * W ** 3 and g.nf(L) == 47905
    rng = np.random.default_rng(4)
    cells = np.unique(np.concatenate([[0, 1, ne - 2, ne - 1], rng.integers(0, ne, 8)]))
    nf, nfc = g.nf(L), g.nf(L - 1)
    x = hmg.DeviceMatrix(g, L).rand(77)
    b = hmg.DeviceMatrix(g, L).rand(76)
    y = hmg.DeviceMatrix(g, L)
    hmg.apply_ex(-1.0, g, x, b, y, constrain=False)          # y = b - A x: the local residual's arithmetic
    got = y.to_host()[:, cells]
    xs = np.asfortranarray(np.concatenate([hmg.host_random((nf, 1), 77, cell_offset=int(c)) for c in cells], axis=1))
    bs = np.asfortranarray(np.concatenate([hmg.host_random((nf, 1), 76, cell_offset=int(c)) for c in cells], axis=1))
    sub = O.Mesh(np.asarray(A.base_mesh.nodes), np.ascontiguousarray(A.base_mesh.elements[cells] - 1))
    impl1 = O.ImplicitFineGrid.create(O.hypercube(3, 1), L)
    ref = impl1.reference.levels[L - 1]
    Ao = O.L2PlusDivAGrad(O.build_local_diffusion_operators(ref), O.mass_matrix(ref), None, A.lam,
                          np.ascontiguousarray(A.cond[cells]))
    want = bs.copy(order="F")
    O.mul(-1.0, sub, Ao, xs, want)
    assert np.abs(got - want).max() <= 1e-11 * np.abs(want).max()
    # level transfers between levels 7 and 6 (k_restrict through the slab window, k_prolong_add_big)
    P = impl1.reference.interops[L - 2]
    xc = hmg.DeviceMatrix(g, L - 1).rand(79)
    rc = hmg.DeviceMatrix(g, L - 1)
    hmg.restrict_to(rc, g, x)
    hmg.interpolate_and_sum_to(x, g, xc)
    got_r, got_p = rc.to_host()[:, cells], x.to_host()[:, cells]
    for v in (x, b, y, xc, rc):
        v.close()
    c0 = np.asfortranarray(np.concatenate([hmg.host_random((nfc, 1), 79, cell_offset=int(c)) for c in cells], axis=1))
    want_r = np.zeros_like(c0, order="F")
    O.restrict_to(want_r, P, xs)
    assert np.abs(got_r - want_r).max() <= 1e-14 * np.abs(want_r).max()
    O.interpolate_and_sum_to(xs, P, c0)
    np.testing.assert_array_equal(got_p, xs)


def test_level7_interface_sum_constraint_first_copies_match_oracle_bit_for_bit(prob, oracle):
    ctx, g, A = prob
    O = oracle
    mesh = O.Mesh(np.asarray(A.base_mesh.nodes), np.ascontiguousarray(A.base_mesh.elements - 1))
    impl = O.ImplicitFineGrid.create(mesh, L)
    cons = O.ZeroDirichletConstraint(*O.list_boundary_nodes_edges_faces(mesh))
    x = hmg.DeviceMatrix(g, L).rand(91)
    want = x.to_host()
    hmg.broadcast_interfaces(x, g, L)
    O.broadcast_interfaces(want, impl, L)
    assert np.array_equal(x.to_host(), want)
    hmg.apply_constraint(x, L, g)
    hmg.zero_out_all_but_one(x, g, L)
    O.apply_constraint(want, L, cons, impl)
    O.zero_out_all_but_one(want, impl, L)
    assert np.array_equal(x.to_host(), want)
    x.close()


def test_level7_exact_savings_are_exact_and_a_vcycle_allocates_nothing(prob):
    ctx, g, A = prob
    base = hmg.BaseLevel(g)
    res = []
    for on in (1, 0):
        for o in EXACT:
            ctx.set_option(o, on)
        ctx.set_option("lazy_top", 2 * on)
        try:
            states = [hmg.LevelState(g, i + 1) for i in range(L)]
            top = _start(g, states, 21)
            allocs = ctx.counter("device_allocs")
            for _ in range(2):
                hmg.vcycle(g, base, [A] * L, states, L, 3)
            assert ctx.counter("device_allocs") == allocs
            assert ctx.counter("lazy_top_form") == 2 * on
            res.append(states)
        finally:
            for o in EXACT:
                ctx.set_option(o, 1)
            ctx.set_option("lazy_top", 2)
    a, b = res[0][-1], res[1][-1]
    assert hmg.dot(a.x, a.x) > 0.0 and base.last_iterations() < 120
    hmg.axpy(-1.0, a.x, b.x)
    hmg.axpy(-1.0, a.r, b.r)
    assert hmg.dot(b.x, b.x) == 0.0 and hmg.dot(b.r, b.r) == 0.0
    for states in res:
        for st in states:
            st.close()


def test_level7_vcycle_is_reproducible_and_contracts(prob):
    ctx, g, A = prob
    base = hmg.BaseLevel(g)
    keep, norms = None, []
    for run in range(2):
        states = [hmg.LevelState(g, i + 1) for i in range(L)]
        if run == 1:
            pad = hmg.DeviceMatrix(g, L - 1)      # other memory blocks for the second run's vectors
        top = _start(g, states, 33)
        for _ in range(3):
            hmg.vcycle(g, base, [A] * L, states, L, 3)
            if run == 0:
                norms.append(hmg.norm_unique(top.r))
        if keep is None:
            keep = (top.x.copy(), top.r.copy())
        else:
            hmg.axpy(-1.0, keep[0], top.x)
            hmg.axpy(-1.0, keep[1], top.r)
            assert hmg.dot(top.x, top.x) == 0.0 and hmg.dot(top.r, top.r) == 0.0
            pad.close()
        for st in states:
            st.close()
    for v in keep:
        v.close()
    # (under contrast 100 the algorithm's own rate is slow: 0.43, 0.73, 0.80 per cycle at 12^3 cubes, tests/test_gpu_parity_l6.py)
    assert all(np.isfinite(norms)) and all(b < a for a, b in zip(norms, norms[1:])) and norms[-1] < 0.7 * norms[0], norms
