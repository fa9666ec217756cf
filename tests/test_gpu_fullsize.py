"""
BASELINE config 3 at full size (32^3 cubes x 6 tets, L = 6, 1.29e9 fine DOFs, 10.3 GB per vector).
  * against the oracle at full size: one whole V-cycle (the oracle needs ~66 s on 16 host cores and ~90 GB of host
    memory), interface sum / Dirichlet constraint / first-copy mask on all 1.29e9 entries (bit for bit), the operator
    apply and the level transfers on sampled cells (they are cell-local);
  * the optimised V-cycle against the plain sequence of the same library, bit for bit;
  * size-independent properties of the operators: linearity and symmetry of the local operator, constants in the
    kernel of the diffusion part, total mass, interface-sum consistency, restriction = prolongation^T, multigrid
    contraction.
"""
import numpy as np
import pytest

import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver

pytestmark = pytest.mark.gpu
W, L = 32, 6


@pytest.fixture(scope="module")
def prob():
    ctx = hmg.Context(0)
    base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, W, L, seed=0, lam=0.75)
    op.base_mesh, op.cond = base, cond
    yield ctx, g, op
    g.close()
    ctx.close()


def test_apply_matches_oracle_on_sampled_cells_full_size(prob, oracle):
    """The operator apply is cell-local (src/apply_local_operators.jl:85-133): at BASELINE config 3's full size the columns
    of 16 sampled cells (first, last, random) of y = alpha A x are compared with the oracle's 10 CSC passes on exactly
    those cells -- same geometry, same sigma, same x (the hashed fill has a numpy twin) -- at the tolerance of the small
    cases, 1e-11; likewise restrict_to! and interpolate_and_sum_to! between levels 6 and 5."""
    ctx, g, A = prob
    O = oracle
    ne = g.ncells()
    rng = np.random.default_rng(3)
    cells = np.unique(np.concatenate([[0, 1, ne - 2, ne - 1], rng.integers(0, ne, 12)]))
    nf = g.nf(L)
    x = hmg.DeviceMatrix(g, L).rand(77)
    y = hmg.DeviceMatrix(g, L).fill(0.0)
    hmg.mul(-1.3, g, A, x, y)
    got = y.to_host()[:, cells]
    y.close(); x.close()
    xs = np.asfortranarray(np.concatenate([hmg.host_random((nf, 1), 77, cell_offset=int(c)) for c in cells], axis=1))
    sub = O.Mesh(np.asarray(A.base_mesh.nodes), np.ascontiguousarray(A.base_mesh.elements[cells] - 1))
    ref = O.ImplicitFineGrid.create(O.hypercube(3, 1), L).reference.levels[L - 1]
    Ao = O.L2PlusDivAGrad(O.build_local_diffusion_operators(ref), O.mass_matrix(ref), None, A.lam,
                          np.ascontiguousarray(A.cond[cells]))
    want = np.zeros_like(xs, order="F")
    O.mul(-1.3, sub, Ao, xs, want)
    assert np.abs(got - want).max() <= 1e-11 * np.abs(want).max()
    # the level transfers are cell-local too (src/interpolation.jl:52-74): restriction 1e-14, prolongation bit for bit
    P = O.ImplicitFineGrid.create(O.hypercube(3, 1), L).reference.interops[L - 2]
    nfc = g.nf(L - 1)
    xf = hmg.DeviceMatrix(g, L).rand(78)
    xc = hmg.DeviceMatrix(g, L - 1).rand(79)
    rc = hmg.DeviceMatrix(g, L - 1)
    hmg.restrict_to(rc, g, xf)
    hmg.interpolate_and_sum_to(xf, g, xc)
    got_r, got_p = rc.to_host()[:, cells], xf.to_host()[:, cells]
    for v in (xf, xc, rc):
        v.close()
    f0 = np.asfortranarray(np.concatenate([hmg.host_random((nf, 1), 78, cell_offset=int(c)) for c in cells], axis=1))
    c0 = np.asfortranarray(np.concatenate([hmg.host_random((nfc, 1), 79, cell_offset=int(c)) for c in cells], axis=1))
    want_r = np.zeros_like(c0, order="F")
    O.restrict_to(want_r, P, f0)
    assert np.abs(got_r - want_r).max() <= 1e-14 * np.abs(want_r).max()
    O.interpolate_and_sum_to(f0, P, c0)
    np.testing.assert_array_equal(got_p, f0)


def test_operator_properties_full_size(prob):
    ctx, g, A = prob
    x = hmg.DeviceMatrix(g, L).rand(1)
    y = hmg.DeviceMatrix(g, L).rand(2)
    hmg.broadcast_interfaces(x, g, L)            # consistent (all copies equal) vectors
    hmg.broadcast_interfaces(y, g, L)
    ax = hmg.DeviceMatrix(g, L).fill(0.0)
    ay = hmg.DeviceMatrix(g, L).fill(0.0)
    hmg.mul(1.0, g, A, x, ax)
    hmg.mul(1.0, g, A, y, ay)
    # symmetry: for consistent x, y  <y, S A x>_unique = sum_raw y (A_local x) = sum_raw x (A_local y)
    s1, s2 = hmg.dot(y, ax), hmg.dot(x, ay)
    assert abs(s1 - s2) <= 1e-11 * abs(s1)
    # positive definiteness of lam M + K
    assert hmg.dot(x, ax) > 0.0
    # linearity: A(2x - 3y) = 2 Ax - 3 Ay
    z = hmg.DeviceMatrix(g, L).copyto(x)
    hmg.xpby(y, -2.0 / 3.0, z)                   # z = y - (2/3) z = y - (2/3) x  ->  -3 z... keep it simple below
    az = hmg.DeviceMatrix(g, L).fill(0.0)
    hmg.mul(1.0, g, A, z, az)
    hmg.axpy(2.0 / 3.0, ax, az)                  # az + (2/3) Ax - Ay should vanish
    hmg.axpy(-1.0, ay, az)
    assert np.sqrt(hmg.dot(az, az)) <= 1e-12 * np.sqrt(hmg.dot(ay, ay))


def test_workgroup_shapes_agree_full_size(prob):
    """The two level-6 instantiations (three 512-thread workgroups per CU / two of 1024 threads, option apply_wg512)
    evaluate every node with the same operands in the same order: at full size the outputs of the plain apply and of a
    residual with a source vector are equal to the last bit."""
    ctx, g, A = prob
    x = hmg.DeviceMatrix(g, L).rand(11)
    b = hmg.DeviceMatrix(g, L).rand(12)
    outs = []
    for wg in (1, 0):
        ctx.set_option("apply_wg512", wg)
        try:
            y = hmg.DeviceMatrix(g, L)
            hmg.apply_ex(1.0, g, x, None, y, constrain=True)
            r = hmg.DeviceMatrix(g, L)
            hmg.apply_ex(-1.0, g, x, b, r, constrain=True)
            outs.append((y, r))
        finally:
            ctx.set_option("apply_wg512", 1)
    (y1, r1), (y0, r0) = outs
    hmg.axpy(-1.0, y1, y0)                       # y0 - y1
    hmg.axpy(-1.0, r1, r0)
    assert hmg.dot(y0, y0) == 0.0 and hmg.dot(r0, r0) == 0.0
    assert hmg.dot(y1, y1) > 0.0
    for v in (y0, y1, r0, r1, x, b):
        v.close()


def test_constants_and_mass_full_size(prob):
    ctx, g, A = prob
    one = hmg.DeviceMatrix(g, L).fill(1.0)
    out = hmg.DeviceMatrix(g, L).fill(0.0)
    hmg.mul(1.0, g, A, one, out)
    # K 1 = 0 in every cell, so A 1 = lam M 1 and 1'(A 1) = lam * |domain| = lam * W^3
    total = hmg.dot(one, out)
    assert abs(total - A.lam * W ** 3) <= 1e-9 * W ** 3
    A.lam = 0.0
    try:
        out.fill(0.0)
        hmg.mul(1.0, g, A, one, out)
        assert np.sqrt(hmg.dot(out, out)) <= 1e-9          # pure diffusion annihilates constants
    finally:
        A.lam = 0.75


def test_interface_sum_and_transfer_full_size(prob):
    ctx, g, A = prob
    x = hmg.DeviceMatrix(g, L).rand(3)
    n_before = hmg.norm_unique(x)
    hmg.broadcast_interfaces(x, g, L)
    y = hmg.DeviceMatrix(g, L).copyto(x)
    hmg.broadcast_interfaces(y, g, L)            # summing a consistent vector multiplies shared DOFs by their multiplicity
    hmg.zero_out_all_but_one(y, g, L)
    z = hmg.DeviceMatrix(g, L).copyto(x)
    hmg.zero_out_all_but_one(z, g, L)
    # first copies: y = mult * z  =>  <z, y> >= <z, z> with equality only without sharing
    assert hmg.dot(z, y) > hmg.dot(z, z) > 0.0 and n_before > 0.0
    # restriction is the transpose of prolongation (per column, hence for the raw dot product)
    xf = hmg.DeviceMatrix(g, L).rand(4)
    xc = hmg.DeviceMatrix(g, L - 1).rand(5)
    pf = hmg.DeviceMatrix(g, L).fill(0.0)
    hmg.interpolate_and_sum_to(pf, g, xc)
    rc = hmg.DeviceMatrix(g, L - 1)
    hmg.restrict_to(rc, g, xf)
    a, b = hmg.dot(pf, xf), hmg.dot(xc, rc)
    assert abs(a - b) <= 1e-12 * abs(a)


def test_vcycle_contracts_full_size(prob):
    ctx, g, A = prob
    states = [hmg.LevelState(g, i + 1) for i in range(L)]
    top = states[-1]
    top.x.rand(6)
    hmg.broadcast_interfaces(top.x, g, L)
    hmg.apply_constraint(top.x, L, g)
    hmg.rhs_axi_grad_v(top.b, g, driver.random_unit_vec(3))
    base = hmg.BaseLevel(g)
    norms = []
    for _ in range(4):
        hmg.vcycle(g, base, [A] * L, states, L, 3)
        norms.append(hmg.norm_unique(top.r))
    assert all(np.isfinite(norms))
    assert all(b < 0.7 * a for a, b in zip(norms, norms[1:])), norms


EXACT = ("lean_post", "lazy_post", "lazy_top", "lazy_dead", "fold_x", "swap_rp", "fold_prolong", "prolong_in_image", "fold_faces", "fold_restrict", "zero_entry", "cell_order")


def test_exact_savings_leave_x_and_r_untouched_full_size(prob):
    """hmg_vcycle with and without every exact saving of the library -- the dead tails of the smoothers (lean_post,
    lazy_dead, fold_x: work whose results the reference's own control flow overwrites before reading), r taken as p by
    exchanging handles (swap_rp), the prolongation folded into the post-smoother's first residual (fold_prolong,
    prolong_in_image), the face sums of A p formed inside the r-update (fold_faces), the restriction in the local residual's epilogue
    (fold_restrict), coarse levels whose zero initial guess is never written (zero_entry), the XCD-aware order in which the apply
    launches walk the cells (cell_order): x and r of the finest level after two V-cycles are equal to the last bit."""
    ctx, g, A = prob
    base = hmg.BaseLevel(g)
    res = []
    for on in (1, 0):
        for o in EXACT:
            ctx.set_option(o, on)
        ctx.set_option("lazy_top", 2 * on)           # (2, the default: the three-update form with its spare vector)
        try:
            states = [hmg.LevelState(g, i + 1) for i in range(L)]
            top = states[-1]
            top.x.rand(21)
            hmg.broadcast_interfaces(top.x, g, L)
            hmg.apply_constraint(top.x, L, g)
            hmg.rhs_axi_grad_v(top.b, g, driver.random_unit_vec(3))
            for _ in range(2):
                hmg.vcycle(g, base, [A] * L, states, L, 3)
            res.append(states)
        finally:
            for o in EXACT:
                ctx.set_option(o, 1)
            ctx.set_option("lazy_top", 2)
    a, b = res[0][-1], res[1][-1]
    assert hmg.dot(a.x, a.x) > 0.0
    hmg.axpy(-1.0, a.x, b.x)
    hmg.axpy(-1.0, a.r, b.r)
    assert hmg.dot(b.x, b.x) == 0.0 and hmg.dot(b.r, b.r) == 0.0
    for states in res:
        for st in states:
            st.close()


def test_vcycle_is_reproducible_bit_for_bit_full_size(prob):
    """Race freedom by construction (one work-item group per shared entity, copies summed in ascending cell order, reductions
    in fixed trees): three V-cycles from the same x0 and b, run three times -- on fresh level vectors each time (other memory
    blocks), once with the blocks' roles re-assigned by hmg_level_tune_placement -- leave x and r equal to the last bit on all
    1.29e9 entries."""
    ctx, g, A = prob
    base = hmg.BaseLevel(g)
    xi = driver.random_unit_vec(3)
    keep = None
    for run in range(3):
        states = [hmg.LevelState(g, i + 1) for i in range(L)]
        if run == 2:
            before, after = hmg.tune_placement(g, [A] * L, states, L, 3, trials=3, extra=1)
            assert after <= before
        top = states[-1]
        top.x.rand(33)
        hmg.broadcast_interfaces(top.x, g, L)
        hmg.apply_constraint(top.x, L, g)
        hmg.rhs_axi_grad_v(top.b, g, xi)
        for _ in range(3):
            hmg.vcycle(g, base, [A] * L, states, L, 3)
        if keep is None:
            keep = (top.x.copy(), top.r.copy())
            assert hmg.dot(keep[0], keep[0]) > 0.0
        else:
            hmg.axpy(-1.0, keep[0], top.x)
            hmg.axpy(-1.0, keep[1], top.r)
            assert hmg.dot(top.x, top.x) == 0.0 and hmg.dot(top.r, top.r) == 0.0, run
        for st in states:
            st.close()
    for v in keep:
        v.close()


def test_interface_sum_constraint_duplicates_match_oracle_full_size(prob, oracle):
    """broadcast_interfaces!, apply_constraint! and zero_out_all_but_one! (src/implicit_fine_grid.jl:94-386) on all
    1.29e9 entries of a config-3 level-6 vector against the oracle run on the same 10 GB array -- integer / index work plus
    sums in the reference's copy order: equal to the last bit."""
    ctx, g, A = prob
    O = oracle
    mesh = O.Mesh(np.asarray(A.base_mesh.nodes), np.ascontiguousarray(A.base_mesh.elements - 1))
    impl = O.ImplicitFineGrid.create(mesh, L)
    cons = O.ZeroDirichletConstraint(*O.list_boundary_nodes_edges_faces(mesh))
    x = hmg.DeviceMatrix(g, L).rand(91)
    want = x.to_host()
    hmg.broadcast_interfaces(x, g, L)
    O.broadcast_interfaces(want, impl, L)
    got = x.to_host()
    assert np.array_equal(got, want)
    hmg.apply_constraint(x, L, g)
    O.apply_constraint(want, L, cons, impl)
    got = x.to_host()
    assert np.array_equal(got, want)
    hmg.zero_out_all_but_one(x, g, L)
    O.zero_out_all_but_one(want, impl, L)
    got = x.to_host()
    assert np.array_equal(got, want)
    x.close()


def _host_cannot_hold_the_oracle():
    import os
    if os.environ.get("HMG_SKIP_FULLSIZE_ORACLE") == "1":
        return True
    try:
        import psutil
        return psutil.virtual_memory().available < 120e9      # oracle level vectors 61 GB + two downloads
    except Exception:
        return False


@pytest.mark.skipif(_host_cannot_hold_the_oracle(),
                    reason="needs ~90 GB of host memory and about a minute of CPU time (HMG_SKIP_FULLSIZE_ORACLE=1 skips it)")
def test_vcycle_matches_oracle_full_size(prob, oracle):
    """One whole V-cycle (3 / 2 CG smoothing steps, 6 levels, coarse solve) on BASELINE config 3 -- 196 608 cells,
    1.29e9 fine DOFs -- by the device and by the oracle on the host's cores, same x0 and b: x 1e-9, r 1e-8 as in the
    small cases (measured on MI355X + 16 host cores: x 1.6e-12, r 8.6e-13; oracle 66 s, device 0.14 s).
    ref: src/multigrid.jl:73-119"""
    import time
    ctx, g, A = prob
    O = oracle
    mesh = O.Mesh(np.asarray(A.base_mesh.nodes), np.ascontiguousarray(A.base_mesh.elements - 1))
    impl = O.ImplicitFineGrid.create(mesh, L)
    cons = O.ZeroDirichletConstraint(*O.list_boundary_nodes_edges_faces(mesh))
    sig = np.ascontiguousarray(A.cond)
    ops = [O.L2PlusDivAGrad(O.build_local_diffusion_operators(l), O.mass_matrix(l), cons, A.lam, sig)
           for l in impl.reference.levels]
    states = [hmg.LevelState(g, i + 1) for i in range(L)]
    top = states[-1]
    top.x.rand(31)
    hmg.broadcast_interfaces(top.x, g, L)
    hmg.apply_constraint(top.x, L, g)
    hmg.rhs_axi_grad_v(top.b, g, driver.random_unit_vec(3))
    sts = [O.LevelState.create(mesh.nelements(), impl.nf(i + 1)) for i in range(L)]
    sts[-1].x[...] = top.x.to_host()
    sts[-1].b[...] = top.b.to_host()
    hmg.vcycle(g, hmg.BaseLevel(g), [A] * L, states, L, 3)
    t0 = time.perf_counter()
    O.vcycle(impl, O.make_base_level(mesh, sig, A.lam), ops, sts, L, 3)
    print(f"oracle V-cycle at full size: {time.perf_counter() - t0:.1f} s on {O.available_cores()} cores")
    gx, gr = top.x.to_host(), top.r.to_host()
    ex = np.abs(gx - sts[-1].x).max() / np.abs(sts[-1].x).max()
    er = np.abs(gr - sts[-1].r).max() / np.abs(sts[-1].r).max()
    print(f"full-size V-cycle, device vs oracle: rel max err x {ex:.3e}, r {er:.3e}")
    for st in states:
        st.close()
    assert ex <= 1e-9 and er <= 1e-8
