"""
GPU parity tests: every hot-path primitive of libhmg_hip.so (through the C ABI / Python host mirror)
against the CPU oracle on identical seeded inputs.  FP64 tolerances (stated per test):
  per primitive   rel max-norm error <= 1e-11   
  interface sum / masks / transfer of exact data: bit-exact where the operation order is the
  reference's (copies summed in ascending cell order).
"""
import numpy as np
import pytest

import homogenization_jl_amd as hmg

pytestmark = pytest.mark.gpu

TOL = 1e-11


@pytest.fixture(scope="module")
def ctx():
    c = hmg.Context(0)
    yield c
    c.close()


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


class Case:
    """A small configuration shared by oracle and device."""

    def __init__(self, O, ctx, dim, n, levels, lam=0.7, perturb=0.0, seed=0, ordered=True, mesh=None,
                 values=(1.0, 9.0)):
        self.O, self.dim, self.levels, self.lam = O, dim, levels, lam
        m = mesh if mesh is not None else O.hypercube(dim, n, origin=(-n / 2.0,) * dim)
        if ordered and mesh is None:
            m = O.order_nodes_and_elements_by_magnitude(m)
        rng = np.random.default_rng(seed)
        if perturb:
            m.nodes = m.nodes + perturb * (rng.random(m.nodes.shape) - 0.5)
        self.mesh = m
        self.sig = rng.choice(list(values), size=(m.nelements(), dim))
        self.impl = O.ImplicitFineGrid.create(m, levels)
        self.cons = O.ZeroDirichletConstraint(*O.list_boundary_nodes_edges_faces(m))
        self.ops = [O.L2PlusDivAGrad(O.build_local_diffusion_operators(l), O.mass_matrix(l), self.cons, lam, self.sig)
                    for l in self.impl.reference.levels]
        self.g = hmg.ImplicitFineGrid(ctx, hmg.Mesh(m.nodes, m.elements + 1), levels)
        self.A = hmg.L2PlusDivAGrad(self.g, lam, self.sig)
        self.rng = rng

    def rand(self, level):
        return np.asfortranarray(self.rng.standard_normal((self.impl.nf(level), self.mesh.nelements())))

    def dev(self, level, a):
        return hmg.DeviceMatrix(self.g, level).from_host(a)


@pytest.fixture(scope="module")
def case3(oracle, ctx):
    return Case(oracle, ctx, 3, 4, 5, perturb=0.2, seed=1)


@pytest.fixture(scope="module")
def case2(oracle, ctx):
    return Case(oracle, ctx, 2, 6, 6, perturb=0.2, seed=2)


def test_upload_download_roundtrip(case3):
    c = case3
    for lev in range(1, c.levels + 1):
        a = c.rand(lev)
        np.testing.assert_array_equal(c.dev(lev, a).to_host(), a)


def test_device_rand_matches_host_twin(case3):
    c = case3
    d = hmg.DeviceMatrix(c.g, c.levels).rand(1234, cell_offset=5)
    np.testing.assert_array_equal(d.to_host(), hmg.host_random(d.shape, 1234, 5))


@pytest.mark.parametrize("threads", [0, 256, 512, 640])
@pytest.mark.parametrize("which", ["case3", "case2"])
def test_apply_matches_oracle(request, ctx, which, threads):
    """mul!(alpha, base, A, x, y) -- ref: src/apply_local_operators.jl:85-133 (every workgroup-size variant)"""
    c = request.getfixturevalue(which)
    ctx.set_option("apply_threads", threads)
    try:
        for lev in range(1, c.levels + 1):
            x, y = c.rand(lev), c.rand(lev)
            want = y.copy(order="F")
            c.O.mul(-1.3, c.mesh, c.ops[lev - 1], x, want)
            dx, dy = c.dev(lev, x), c.dev(lev, y)
            hmg.mul(-1.3, c.g, c.A, dx, dy)
            assert relerr(dy.to_host(), want) <= TOL, (which, lev)
    finally:
        ctx.set_option("apply_threads", 0)


@pytest.mark.parametrize("which", ["case3", "case2"])
def test_residual_and_constraint(request, which):
    """local_residual! / apply_constraint! -- ref: src/apply_local_operators.jl:18-27, src/implicit_fine_grid.jl:94-139"""
    c = request.getfixturevalue(which)
    O = c.O
    for lev in range(2, c.levels + 1):
        st = O.LevelState.create(c.mesh.nelements(), c.impl.nf(lev))
        st.x[...] = c.rand(lev); st.b[...] = c.rand(lev)
        O.local_residual(c.impl, c.ops[lev - 1], st, lev)
        dst = hmg.LevelState(c.g, lev)
        dst.x.from_host(st.x); dst.b.from_host(st.b)
        hmg.local_residual(c.g, c.A, dst, lev)
        got = dst.r.to_host()
        assert relerr(got, st.r) <= TOL
        np.testing.assert_array_equal(got == 0.0, st.r == 0.0)          # same Dirichlet zero pattern
        a = c.rand(lev)
        want = a.copy(order="F"); O.apply_constraint(want, lev, c.cons, c.impl)
        d = c.dev(lev, a); hmg.apply_constraint(d, lev, c.g)
        np.testing.assert_array_equal(d.to_host(), want)


@pytest.mark.parametrize("which", ["case3", "case2"])
def test_interface_sum_and_duplicates_bit_exact(request, which):
    """broadcast_interfaces! / zero_out_all_but_one! -- ref: src/implicit_fine_grid.jl:209-386"""
    c = request.getfixturevalue(which)
    O = c.O
    for lev in range(1, c.levels + 1):
        a = c.rand(lev)
        want = a.copy(order="F"); O.broadcast_interfaces(want, c.impl, lev)
        d = c.dev(lev, a); hmg.broadcast_interfaces(d, c.g, lev)
        np.testing.assert_array_equal(d.to_host(), want)                # same copy order => same bits
        want2 = want.copy(order="F"); O.zero_out_all_but_one(want2, c.impl, lev)
        n_unique = hmg.norm_unique(d)
        assert abs(n_unique - np.linalg.norm(want2)) <= 1e-13 * np.linalg.norm(want2)
        hmg.zero_out_all_but_one(d, c.g, lev)
        np.testing.assert_array_equal(d.to_host(), want2)


@pytest.mark.parametrize("which", ["case3", "case2"])
def test_transfer(request, which):
    """restrict_to! / interpolate_and_sum_to! -- ref: src/interpolation.jl:52-74"""
    c = request.getfixturevalue(which)
    O = c.O
    for lev in range(2, c.levels + 1):
        P = c.impl.reference.interops[lev - 2]
        xf, xc = c.rand(lev), c.rand(lev - 1)
        want = xf.copy(order="F"); O.interpolate_and_sum_to(want, P, xc)
        df, dc = c.dev(lev, xf), c.dev(lev - 1, xc)
        hmg.interpolate_and_sum_to(df, c.g, dc)
        np.testing.assert_array_equal(df.to_host(), want)               # same operation order
        wantb = np.zeros_like(xc, order="F"); O.restrict_to(wantb, P, xf)
        db = hmg.DeviceMatrix(c.g, lev - 1)
        hmg.restrict_to(db, c.g, c.dev(lev, xf))
        assert relerr(db.to_host(), wantb) <= 1e-15 * 16


def test_affine_reproduction_under_prolongation(oracle, ctx):
    """ref: test/interpolation.jl:8-35, on the device path"""
    O = oracle
    c = Case(O, ctx, 3, 2, 6, lam=1.0, seed=3, ordered=False)
    direction = np.array([0.3, -1.1, 0.7])
    u = 10.0 + c.mesh.nodes @ direction
    v = hmg.DeviceMatrix(c.g, 1)
    hmg.distribute(v, u, c.g)
    np.testing.assert_array_equal(hmg.copy_to_base(c.g, v), u)
    for lev in range(2, 7):
        nxt = hmg.DeviceMatrix(c.g, lev).fill(0.0)
        hmg.interpolate_and_sum_to(nxt, c.g, v)
        v = nxt
        full = c.impl.construct_full_grid(lev)
        np.testing.assert_allclose(v.to_host().T, 10.0 + full @ direction, rtol=1e-13)


def test_blas1(case3):
    c = case3
    lev = c.levels
    a, b = c.rand(lev), c.rand(lev)
    da, db = c.dev(lev, a), c.dev(lev, b)
    assert abs(hmg.dot(da, db) - float(np.vdot(a, b))) <= 1e-12 * np.linalg.norm(a) * np.linalg.norm(b)
    hmg.axpy(0.37, da, db)
    np.testing.assert_allclose(db.to_host(), b + 0.37 * a, rtol=0, atol=1e-15 * 8)
    hmg.xpby(da, -1.7, db)
    np.testing.assert_allclose(db.to_host(), a - 1.7 * (b + 0.37 * a), rtol=0, atol=1e-14)
    assert abs(hmg.norm(da) - np.linalg.norm(a)) <= 1e-13 * np.linalg.norm(a)
    dc = da.copy()                                                       # copy / similar / fill!
    da.fill(0.0)
    np.testing.assert_array_equal(dc.to_host(), a)
    assert dc.similar().shape == a.shape and not da.to_host().any()


@pytest.mark.parametrize("which", ["case3", "case2"])
def test_gather_scatter_base(request, which):
    """copy_to_base! / distribute! -- ref: src/implicit_fine_grid.jl:148-202"""
    c = request.getfixturevalue(which)
    O = c.O
    a = c.rand(1)
    u = np.zeros(c.mesh.nnodes()); O.copy_to_base(u, a, c.impl)
    np.testing.assert_array_equal(hmg.copy_to_base(c.g, c.dev(1, a)), u)
    w = c.rng.standard_normal(c.mesh.nnodes())
    want = np.zeros_like(a, order="F"); O.distribute(want, w, c.impl)
    d = hmg.DeviceMatrix(c.g, 1); hmg.distribute(d, w, c.g)
    np.testing.assert_array_equal(d.to_host(), want)


def _oracle_state(c, lev):
    O = c.O
    st = O.LevelState.create(c.mesh.nelements(), c.impl.nf(lev))
    st.x[...] = c.rand(lev)
    O.broadcast_interfaces(st.x, c.impl, lev)
    O.apply_constraint(st.x, lev, c.cons, c.impl)
    st.b[...] = c.rand(lev)
    return st


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("which", ["case3", "case2"])
def test_smoothing_steps(request, oracle, ctx, which, fused):
    """smoothing_steps! (CG, duplicate-counting dots) -- ref: src/multigrid.jl:46-71. tol 1e-10 on x, r, p.
    fused=1: p-update / x-update / dot products inside the apply pass; fused=0: one kernel per statement."""
    if fused:
        c = request.getfixturevalue(which)
    else:
        ctx.set_option("fuse_cg", 0)              # read when a grid is created
        try:
            c = Case(oracle, ctx, 3 if which == "case3" else 2, 4, 4, perturb=0.2, seed=17)
        finally:
            ctx.set_option("fuse_cg", 1)
    lev = c.levels
    st = _oracle_state(c, lev)
    dst = hmg.LevelState(c.g, lev)
    dst.x.from_host(st.x); dst.b.from_host(st.b)
    c.O.smoothing_steps(3, c.impl, c.ops[lev - 1], st, lev)
    hmg.smoothing_steps(3, c.g, c.A, dst, lev)
    assert relerr(dst.x.to_host(), st.x) <= 1e-10
    assert relerr(dst.r.to_host(), st.r) <= 1e-10
    assert relerr(dst.p.to_host(), st.p) <= 1e-10


@pytest.mark.parametrize("which,dim,n,levels", [("3d", 3, 4, 4), ("2d", 2, 8, 5)])
def test_vcycle_matches_oracle(oracle, ctx, which, dim, n, levels):
    """vcycle! incl. coarse solve -- ref: src/multigrid.jl:73-119.  x after one and three V-cycles: rel 1e-9."""
    O = oracle
    c = Case(O, ctx, dim, n, levels, lam=1.0, seed=11)
    sts = [O.LevelState.create(c.mesh.nelements(), c.impl.nf(i + 1)) for i in range(levels)]
    top = _oracle_state(c, levels)
    sts[-1] = top
    dsts = [hmg.LevelState(c.g, i + 1) for i in range(levels)]
    dsts[-1].x.from_host(top.x); dsts[-1].b.from_host(top.b)
    base = O.make_base_level(c.mesh, c.sig, 1.0)
    dbase = hmg.BaseLevel(c.g)
    norms = []
    for cyc in range(3):
        O.vcycle(c.impl, base, c.ops, sts, levels, 3)
        hmg.vcycle(c.g, dbase, [c.A] * levels, dsts, levels, 3)
        assert relerr(dsts[-1].x.to_host(), sts[-1].x) <= 1e-9, (which, cyc)
        r = sts[-1].r.copy(order="F"); O.zero_out_all_but_one(r, c.impl, levels)
        norms.append((np.linalg.norm(r), hmg.norm_unique(dsts[-1].r)))
    for a, b in norms:
        assert abs(a - b) <= 1e-8 * a
    assert norms[-1][0] < 0.2 * norms[0][0]                              # multigrid contracts
    assert dbase.last_iterations() > 0


@pytest.mark.parametrize("dim,n,grids,steps", [(2, 4, 4, 3), (3, 2, 3, 3), (3, 2, 4, 1), (3, 3, 5, 3), (3, 2, 6, 3)])
def test_vcycle_matches_its_global_matrix_form(oracle, ctx, dim, n, grids, steps):
    """The device against the SECOND statement of smoothing_steps! / vcycle! (tests/_global_form.py: global vectors, matrices
    assembled on explicitly refined meshes, multiplicity-weighted dots, direct solve on level 1) -- no cell-local code of the
    oracle takes part in the expected values; the oracle supplies only the mesh, the assembly and the node numbering.
    x after each of two V-cycles: rel 1e-9 (the level-1 solve is PCG to 1e-13 here, a direct solve there); r: 1e-8."""
    from _global_form import GlobalForm
    O = oracle
    lam = 1.0
    rng = np.random.default_rng(23)
    base = O.hypercube(dim, n)
    sgrid = np.where(rng.random((n,) * dim + (dim,)) < 0.5, 1.0, 9.0)
    cond = O.conductivity_per_element(base, sgrid, (0.0,) * dim)
    implicit = O.ImplicitFineGrid.create(base, grids)                    # (node coordinates of the repeated nodes only)
    G = GlobalForm(O, base, sgrid, lam, implicit, grids, dim)
    g = hmg.ImplicitFineGrid(ctx, hmg.Mesh(base.nodes, base.elements + 1), grids)
    A = hmg.L2PlusDivAGrad(g, lam, cond)
    sts = [hmg.LevelState(g, i + 1) for i in range(grids)]
    top = sts[-1]
    top.x.rand(5)
    hmg.broadcast_interfaces(top.x, g, grids)
    hmg.apply_constraint(top.x, grids, g)
    hmg.local_rhs(top.b, g)
    gx = G.gather(top.x.to_host(), grids - 1)
    gb = G.gather_sum(top.b.to_host(), grids - 1)
    bl = hmg.BaseLevel(g)
    for cycle in range(2):
        hmg.vcycle(g, bl, [A] * grids, sts, grids, steps)
        gx, gr = G.vcycle(grids - 1, gx, gb, steps)
        dx, dr = G.gather(top.x.to_host(), grids - 1), G.gather(top.r.to_host(), grids - 1)
        assert np.abs(dx - gx).max() <= 1e-9 * np.abs(gx).max(), (cycle, np.abs(dx - gx).max())
        assert np.abs(dr - gr).max() <= 1e-8 * np.abs(gr).max(), (cycle, np.abs(dr - gr).max())
    for s in sts:
        s.close()
    g.close()


@pytest.mark.parametrize("dim,width,levels,radius", [(2, 6, 4, 2), (3, 4, 3, 1), (3, 4, 5, 1)])
def test_driver_integrals_equal_textbook_fem(oracle, ctx, dim, width, levels, radius):
    """The device's right-hand sides and integrals (hmg_rhs_axi_grad, hmg_next_rhs, hmg_integrate) against textbook P1 elements
    on the explicitly refined mesh (tests/_textbook_fem.py): no reference-element table of the oracle takes part in the
    expected values.  1e-11."""
    from _textbook_fem import driver_setting
    O = oracle
    base, cond, implicit, T, a_fine, inside, mapping, nint, rng = driver_setting(O, dim, width, levels, radius)
    xi = rng.standard_normal(dim)
    nf, ne = implicit.nf(levels), base.nelements()
    lam = 0.37
    g = hmg.ImplicitFineGrid(ctx, hmg.Mesh(base.nodes, base.elements + 1), levels)
    A = hmg.L2PlusDivAGrad(g, lam, cond)
    A._bind()
    gather_sum = lambda loc: np.bincount(mapping, weights=loc.reshape(-1, order="F"), minlength=len(T.nodes))

    def consistent(seed):
        v = np.asfortranarray(np.random.default_rng(seed).standard_normal((nf, ne)))
        O.broadcast_interfaces(v, implicit, levels)
        gl = np.zeros(len(T.nodes))
        gl[mapping] = v.reshape(-1, order="F")
        return hmg.DeviceMatrix(g, levels).from_host(v), gl

    b = hmg.DeviceMatrix(g, levels)
    hmg.rhs_axi_grad_v(b, g, xi)
    F = T.load(a_fine, xi)
    assert np.abs(gather_sum(b.to_host()) - F).max() <= 1e-11 * np.abs(F).max()
    v0, g0 = consistent(1)
    v1, g1 = consistent(2)
    want = float(np.dot(g0, T.load(a_fine, xi, inside))) + T.mass_quadratic(g0, g0, inside)
    assert abs(hmg.integrate_first_term(v0, g, nint, xi, b=b) - want) <= 1e-11 * abs(want)
    want = T.mass_quadratic(g0 + g1, g0, inside)
    assert abs(hmg.integrate_terms(v0, v1, g, nint) - want) <= 1e-11 * abs(want)
    assert abs(hmg.integrate_area(v0, g, nint) - float(np.dot(T.vol, inside))) <= 1e-12 * nint
    hmg.next_rhs(b, v0, g)
    want = lam * T.mass_apply(g0)
    assert np.abs(gather_sum(b.to_host()) - want).max() <= 1e-11 * np.abs(want).max()
    for v in (b, v0, v1):
        v.close()
    g.close()


@pytest.mark.parametrize("dim,refinements", [(2, 3), (3, 1), (3, 2)])
def test_driver_converges_to_the_direct_fem_answer(oracle, ctx, dim, refinements):
    """driver.checkerboard_homogenization with n = 0 (one outer step) on the device, run to a tight tolerance, against a sparse
    direct solve of the same boundary value problem on the explicitly refined mesh (tests/_textbook_fem.py): mesh ordering,
    conductivity lookup, right-hand side, V-cycle, integrals and the sigma formula in one number.  1e-8."""
    from _textbook_fem import converged_first_term
    from homogenization_jl_amd import driver
    rng = np.random.default_rng(8)
    sgrid = np.where(rng.random((10,) * dim + (dim,)) < 0.5, 1.0, 9.0)
    xi = rng.standard_normal(dim)
    xi /= np.linalg.norm(xi)
    sigma, hist = driver.checkerboard_homogenization(0, hmg.Tet64 if dim == 3 else hmg.Tri64, refinements=refinements, tolerance=1e-12,
                                                     xi=xi, sigma_grid=sgrid, ctx=ctx, max_cycles=60)
    want = converged_first_term(oracle, dim, sgrid, xi, refinements)
    assert abs(sigma - want) <= 1e-8 * abs(want), (sigma, want, len(hist))


@pytest.mark.parametrize("dim,npts,levels", [(3, 60, 4), (2, 80, 5)])
def test_unstructured_delaunay_mesh(oracle, ctx, dim, npts, levels):
    """A base mesh that is not a split cube lattice: Delaunay triangulation of random points (edges shared by
    3..10+ cells, nodes by up to dozens, irregular boundary, cells of very different shape).  Operator apply,
    interface sum, constraint, smoother and V-cycle against the oracle; same tolerances as the lattice cases."""
    from _meshes import delaunay_mesh
    O = oracle
    mesh = delaunay_mesh(O, dim, npts, 100 + dim)
    c = Case(O, ctx, dim, 0, levels, lam=1.0, seed=21, mesh=mesh)
    x = c.rand(levels)
    y = np.zeros_like(x)
    O.mul(1.0, mesh, c.ops[-1], x, y)
    O.apply_constraint(y, levels, c.cons, c.impl)
    O.broadcast_interfaces(y, c.impl, levels)
    dx, dy = c.dev(levels, x), c.dev(levels, np.zeros_like(x))
    hmg.mul(1.0, c.g, c.A, dx, dy)
    hmg.apply_constraint(dy, levels, c.g)
    hmg.broadcast_interfaces(dy, c.g, levels)
    assert relerr(dy.to_host(), y) <= TOL
    sts = [O.LevelState.create(mesh.nelements(), c.impl.nf(i + 1)) for i in range(levels)]
    sts[-1] = _oracle_state(c, levels)
    dsts = [hmg.LevelState(c.g, i + 1) for i in range(levels)]
    dsts[-1].x.from_host(sts[-1].x); dsts[-1].b.from_host(sts[-1].b)
    base, dbase = O.make_base_level(mesh, c.sig, 1.0), hmg.BaseLevel(c.g)
    for cyc in range(2):
        O.vcycle(c.impl, base, c.ops, sts, levels, 3)
        hmg.vcycle(c.g, dbase, [c.A] * levels, dsts, levels, 3)
        assert relerr(dsts[-1].x.to_host(), sts[-1].x) <= 1e-9, cyc
        assert relerr(dsts[-1].r.to_host(), sts[-1].r) <= 1e-8, cyc


def test_simple_diffusion_matches_assembled_matrix(oracle, ctx):
    """The reference's own operator test on the device (test/test_operator.jl:9-73): SimpleDiffusion mul! +
    broadcast_interfaces! on the implicit grid == assembled stiffness matrix times the vector on the explicitly
    refined mesh (5-tet cube, refined once, 5 levels).  The reference asserts <= 20 eps for its CSC passes; the
    stencil form sums in another order: 1e-13 relative here."""
    from _meshes import five_tet_cube, match_nodes
    O = oracle
    levels = 5
    base = five_tet_cube(O, 1)
    implicit = O.ImplicitFineGrid.create(base, levels)
    g = hmg.ImplicitFineGrid(ctx, hmg.Mesh(base.nodes, base.elements + 1), levels)
    A = hmg.SimpleDiffusion(g, 1.0)
    rng = np.random.default_rng(1)
    local_x = np.asfortranarray(rng.random((implicit.nf(levels), base.nelements())))
    O.broadcast_interfaces(local_x, implicit, levels)
    total_fine = O.refine_uniformly(base, times=levels - 1)
    mapping = match_nodes(total_fine.nodes, implicit.construct_full_grid(levels).reshape(-1, 3))
    total_x = np.zeros(total_fine.nnodes())
    total_x[mapping] = local_x.reshape(-1, order="F")
    total_y = O.assemble_matrix_dot(total_fine) @ total_x
    dx = hmg.DeviceMatrix(g, levels).from_host(local_x)
    dy = hmg.DeviceMatrix(g, levels)
    hmg.mul(1.0, g, A, dx, dy)
    hmg.broadcast_interfaces(dy, g, levels)
    got = dy.to_host().reshape(-1, order="F")
    assert np.abs(got - total_y[mapping]).max() <= 1e-13 * np.abs(total_y).max()
    # a second operator object on the same grid re-binds itself when used
    B = hmg.L2PlusDivAGrad(g, 0.5, np.full((base.nelements(), 3), 2.0))
    dz = hmg.DeviceMatrix(g, levels)
    hmg.mul(1.0, g, B, dx, dz)
    dy.fill(0.0)
    hmg.mul(1.0, g, A, dx, dy)
    hmg.broadcast_interfaces(dy, g, levels)
    assert np.abs(dy.to_host().reshape(-1, order="F") - total_y[mapping]).max() <= 1e-13 * np.abs(total_y).max()


def test_shrink_then_vcycle(oracle, ctx):
    """Domain shrink (prefix of cells/nodes + new boundary) then a V-cycle -- ref: ...homogenized_coefficients.jl:309-336"""
    O = oracle
    c = Case(O, ctx, 3, 6, 3, lam=0.5, seed=5)
    ne = O.find_elements_in_radius(c.mesh, 2)
    nn = O.find_nodes_in_radius(c.mesh, 2)
    sub = O.Mesh(c.mesh.nodes[:nn], np.ascontiguousarray(c.mesh.elements[:ne]))
    sig = np.ascontiguousarray(c.sig[:ne])
    impl = O.ImplicitFineGrid.create(sub, 3)
    cons = O.ZeroDirichletConstraint(*O.list_boundary_nodes_edges_faces(sub))
    ops = [O.L2PlusDivAGrad(O.build_local_diffusion_operators(l), O.mass_matrix(l), cons, 0.5, sig)
           for l in impl.reference.levels]
    x_full = c.rand(3); b_full = c.rand(3)
    dsts = [hmg.LevelState(c.g, i + 1) for i in range(3)]
    dsts[-1].x.from_host(x_full); dsts[-1].b.from_host(b_full)
    c.g.shrink(ne, nn)
    hmg.broadcast_interfaces(dsts[-1].x, c.g, 3)
    hmg.apply_constraint(dsts[-1].x, 3, c.g)
    sts = [O.LevelState.create(ne, impl.nf(i + 1)) for i in range(3)]
    sts[-1].x[...] = x_full[:, :ne]; sts[-1].b[...] = b_full[:, :ne]
    O.broadcast_interfaces(sts[-1].x, impl, 3)
    O.apply_constraint(sts[-1].x, 3, cons, impl)
    O.vcycle(impl, O.make_base_level(sub, sig, 0.5), ops, sts, 3, 3)
    hmg.vcycle(c.g, hmg.BaseLevel(c.g), [c.A] * 3, dsts, 3, 3)
    assert relerr(dsts[-1].x.to_host(), sts[-1].x) <= 1e-9


def test_errors_are_reported_not_thrown(case3):
    c = case3
    a = hmg.DeviceMatrix(c.g, 2)
    b = hmg.DeviceMatrix(c.g, 3)
    with pytest.raises(hmg._lib.HmgError):
        hmg.mul(1.0, c.g, c.A, a, b)          # level mismatch
    with pytest.raises(hmg._lib.HmgError):
        hmg.mul(1.0, c.g, c.A, b, b)          # aliasing


def test_driver_right_hand_sides(case3):
    """rhs_a xi grad v! and next_rhs! -- ref: src/examples/homogenized_coefficients.jl:449-474, 695-713"""
    c = case3
    O = c.O
    lev = c.levels
    xi = np.ones(3) / np.sqrt(3.0)
    dphis = O.partial_derivatives_functionals(c.impl.reference.levels[-1])
    want = np.zeros((c.impl.nf(lev), c.mesh.nelements()), order="F")
    O.rhs_axi_grad_v(want, dphis, c.impl, c.sig, xi)
    b = hmg.DeviceMatrix(c.g, lev)
    hmg.rhs_axi_grad_v(b, c.g, xi)
    assert relerr(b.to_host(), want) <= 1e-12
    x = c.rand(lev)
    want2 = np.zeros_like(x, order="F")
    O.next_rhs(want2, x, c.impl, O.mass_matrix(c.impl.reference.levels[-1]), c.lam)
    hmg.next_rhs(b, c.dev(lev, x), c.g)
    assert relerr(b.to_host(), want2) <= 1e-12


def test_driver_integrals(case3):
    """integrate_first_term / integrate_terms / integrate_area -- ref: ...homogenized_coefficients.jl:592-689"""
    c = case3
    O = c.O
    lev = c.levels
    xi = np.array([0.2, -0.5, 0.84])
    mass = O.mass_matrix(c.impl.reference.levels[-1])
    dphis = O.partial_derivatives_functionals(c.impl.reference.levels[-1])
    v, w = c.rand(lev), c.rand(lev)
    dv, dw = c.dev(lev, v), c.dev(lev, w)
    db = hmg.DeviceMatrix(c.g, lev)
    hmg.rhs_axi_grad_v(db, c.g, xi)                                    # dot(dphi_i, P): the first term's linear part
    for nsub in (0, 7, c.mesh.nelements()):
        a = O.integrate_first_term(v, dphis, c.impl, nsub, mass, c.sig, xi)
        assert abs(hmg.integrate_first_term(dv, c.g, nsub, xi, b=db) - a) <= 1e-11 * max(abs(a), 1.0)
        assert abs(hmg.integrate_first_term(dv, c.g, nsub, xi) - a) <= 1e-11 * max(abs(a), 1.0)   # temporary b
        b = O.integrate_terms(v, w, c.impl, nsub, mass)
        assert abs(hmg.integrate_terms(dv, dw, c.g, nsub) - b) <= 1e-11 * max(abs(b), 1.0)
        ar = O.integrate_area(mass, c.impl, nsub)
        assert abs(hmg.integrate_area(dv, c.g, nsub) - ar) <= 1e-12 * max(ar, 1.0)


@pytest.mark.parametrize("dim,n,refinements,tol", [(2, 1, 2, 1e-4), (3, 0, 2, 1e-3)])
def test_checkerboard_homogenization_matches_oracle_driver(oracle, ctx, dim, n, refinements, tol):
    """The whole driver (outer shrink loop, V-cycles to tolerance, integrals, next rhs) on the device vs the
    oracle's restatement of src/examples/homogenized_coefficients.jl:174-343, identical sigma field and x0.
    |delta sigma| <= 1e-8, same number of V-cycles."""
    from homogenization_jl_amd import driver
    O = oracle
    tag = hmg.Tet64 if dim == 3 else hmg.Tri64
    lam = 1.0
    width = 2 * (driver.compute_box_radius(0, n) + driver.compute_boundary_layer(lam, n))
    sgrid = driver.generate_conductivity(dim, width, 11)
    nf = {2: [3, 6, 15, 45], 3: [4, 10, 35, 165]}[dim][refinements]
    ne = (2 if dim == 2 else 6) * width ** dim
    x0 = hmg.host_random((nf, ne), 99)
    want, hist_o = O.checkerboard_homogenization(n=n, dim=dim, refinements=refinements, tolerance=tol,
                                                 sigma_grid=sgrid, x0=x0)
    got, hist_d = driver.checkerboard_homogenization(n, tag, refinements=refinements, tolerance=tol, ctx=ctx,
                                                     sigma_grid=sgrid, x0=x0)
    assert len(hist_o) == len(hist_d)
    assert abs(got - want) <= 1e-8
    for a, b in zip(hist_o, hist_d):
        assert a[:2] == b[:2]
        assert abs(a[2] - b[2]) <= 1e-7 * max(a[2], 1e-12)          # residual norm per cycle
        assert abs(a[3] - b[3]) <= 1e-8                              # sigma + dsigma per cycle


@pytest.mark.parametrize("dim,n,grids", [(3, 3, 3), (2, 5, 4)])
def test_hypercube_multigrid_driver_matches_oracle(oracle, ctx, tmp_path, dim, n, grids):
    """checkerboard_hypercube_multigrid (...homogenized_coefficients.jl:509-571): local_rhs!, lambda = 0, residual norm
    per V-cycle and the final x against the oracle driver; the VTK file of the finest level holds x."""
    from homogenization_jl_amd import driver, vtk
    O = oracle
    tag = hmg.Tet64 if dim == 3 else hmg.Tri64
    sgrid = driver.generate_conductivity(dim, n, 7)
    nf = O.ImplicitFineGrid.create(O.hypercube(dim, 1), grids).nf(grids)
    x0 = hmg.host_random((nf, (6 if dim == 3 else 2) * n ** dim), 5)
    rs_o, top_o, impl_o, base_o, _ = O.checkerboard_hypercube_multigrid(n, dim=dim, refinements=grids, max_cycles=4,
                                                                       sigma_grid=sgrid, x0=x0)
    b_o = top_o.b.copy()
    rs_d, top_d, g = driver.checkerboard_hypercube_multigrid(n, tag, grids, 4, save=(grids, str(tmp_path)), ctx=ctx,
                                                             sigma_grid=sgrid, x0=x0)
    assert relerr(top_d.b.to_host(), b_o) <= TOL                        # local_rhs!
    for a, b in zip(rs_o, rs_d):
        assert abs(a - b) <= 1e-8 * a
    assert relerr(top_d.x.to_host(), top_o.x) <= 1e-9
    out = vtk.read_vtu(str(tmp_path / f"checkerboard_full_{grids}.vtu"))
    np.testing.assert_array_equal(out["point_data"]["x"], top_d.x.to_host().T.ravel())
    np.testing.assert_allclose(out["points"][:, :dim], impl_o.construct_full_grid(grids).reshape(-1, dim), atol=1e-14)


def test_driver_save_writes_vtk(ctx, tmp_path):
    """`save = level` (src/examples/homogenized_coefficients.jl:219,303): checkerboard.vtu + one ahom_k.vtu per outer
    step on the full grid of that level; sigma is unaffected."""
    from homogenization_jl_amd import driver, vtk
    sig0, hist0 = driver.checkerboard_homogenization(1, hmg.Tri64, refinements=2, tolerance=1e-3, ctx=ctx)
    sig1, hist1 = driver.checkerboard_homogenization(1, hmg.Tri64, refinements=2, tolerance=1e-3, ctx=ctx,
                                                     save=(2, str(tmp_path)))
    assert sig0 == sig1 and len(hist0) == len(hist1)
    dom = vtk.read_vtu(str(tmp_path / "checkerboard.vtu"))
    ne = dom["offsets"].size
    assert dom["cell_data"]["a"].shape == (ne, 2)
    outer = sorted({h[0] for h in hist1})
    for k in outer:
        out = vtk.read_vtu(str(tmp_path / f"ahom_{k}.vtu"))
        assert out["point_data"]["v"].size == out["points"].shape[0]
        assert out["points"].shape[0] % 6 == 0                          # Nf(level 2) = 6 nodes per coarse triangle
        assert out["points"].shape[0] // 6 <= ne
        assert np.isfinite(out["point_data"]["v"]).all()


@pytest.mark.parametrize("fused", [1, 0])
def test_level7_cells_larger_than_lds(oracle, ctx, fused):
    """refinements = 6: Nf = 47 905 (374 KiB per cell) does not fit the 160 KiB LDS; the slab-wise apply (rolling
    window of k-planes, fused CG pass included) takes over.  ref: src/apply_local_operators.jl:85-133,
    src/multigrid.jl:46-119"""
    O = oracle
    ctx.set_option("fuse_cg", fused)
    try:
        c = Case(O, ctx, 3, 1, 7, lam=0.9, perturb=0.1, seed=13, ordered=False)
    finally:
        ctx.set_option("fuse_cg", 1)
    lev = 7
    x, y = c.rand(lev), c.rand(lev)
    want = y.copy(order="F")
    O.mul(0.7, c.mesh, c.ops[lev - 1], x, want)
    dx, dy = c.dev(lev, x), c.dev(lev, y)
    hmg.mul(0.7, c.g, c.A, dx, dy)
    assert relerr(dy.to_host(), want) <= TOL
    st = _oracle_state(c, lev)
    dst = hmg.LevelState(c.g, lev)
    dst.x.from_host(st.x); dst.b.from_host(st.b)
    O.smoothing_steps(3, c.impl, c.ops[lev - 1], st, lev)
    hmg.smoothing_steps(3, c.g, c.A, dst, lev)
    assert relerr(dst.x.to_host(), st.x) <= 1e-10
    assert relerr(dst.r.to_host(), st.r) <= 1e-10
    assert relerr(dst.p.to_host(), st.p) <= 1e-10
    # transfer to / from level 6
    P = c.impl.reference.interops[lev - 2]
    wantb = np.zeros((c.impl.nf(lev - 1), c.mesh.nelements()), order="F")
    O.restrict_to(wantb, P, st.r)
    db = hmg.DeviceMatrix(c.g, lev - 1)
    hmg.restrict_to(db, c.g, dst.r)
    assert relerr(db.to_host(), wantb) <= 1e-10
    if fused:                                                            # one V-cycle through all 7 levels
        c = Case(O, ctx, 3, 2, 7, lam=0.9, perturb=0.1, seed=14)         # (2^3 cubes: level 1 has an interior node)
        sts = [O.LevelState.create(c.mesh.nelements(), c.impl.nf(i + 1)) for i in range(lev)]
        sts[-1] = _oracle_state(c, lev)
        dsts = [hmg.LevelState(c.g, i + 1) for i in range(lev)]
        dsts[-1].x.from_host(sts[-1].x); dsts[-1].b.from_host(sts[-1].b)
        base, dbase = O.make_base_level(c.mesh, c.sig, 0.9), hmg.BaseLevel(c.g)
        O.vcycle(c.impl, base, c.ops, sts, lev, 3)
        hmg.vcycle(c.g, dbase, [c.A] * lev, dsts, lev, 3)
        assert relerr(dsts[-1].x.to_host(), sts[-1].x) <= 1e-9
        assert relerr(dsts[-1].r.to_host(), sts[-1].r) <= 1e-8


def test_level8_triangles_maximum_size(oracle, ctx):
    """Largest supported 2D level (refinements = 7, m = 128, Nf = 8385, 66 KiB lattice image): apply, smoother
    and a V-cycle through all 8 levels against the oracle."""
    O = oracle
    lev = 8
    c = Case(O, ctx, 2, 2, lev, lam=1.0, perturb=0.1, seed=17)
    x, y = c.rand(lev), c.rand(lev)
    want = y.copy(order="F")
    O.mul(-0.4, c.mesh, c.ops[lev - 1], x, want)
    dx, dy = c.dev(lev, x), c.dev(lev, y)
    hmg.mul(-0.4, c.g, c.A, dx, dy)
    assert relerr(dy.to_host(), want) <= TOL
    sts = [O.LevelState.create(c.mesh.nelements(), c.impl.nf(i + 1)) for i in range(lev)]
    sts[-1] = _oracle_state(c, lev)
    dsts = [hmg.LevelState(c.g, i + 1) for i in range(lev)]
    dsts[-1].x.from_host(sts[-1].x); dsts[-1].b.from_host(sts[-1].b)
    base, dbase = O.make_base_level(c.mesh, c.sig, 1.0), hmg.BaseLevel(c.g)
    O.vcycle(c.impl, base, c.ops, sts, lev, 3)
    hmg.vcycle(c.g, dbase, [c.A] * lev, dsts, lev, 3)
    assert relerr(dsts[-1].x.to_host(), sts[-1].x) <= 1e-9
    assert relerr(dsts[-1].r.to_host(), sts[-1].r) <= 1e-8


def test_single_cell_and_all_dirichlet(oracle, ctx):
    """Edge case: one tetrahedron.  No interfaces, the whole surface is Dirichlet, the level-1 system is empty
    (x1 = 0); apply / smoother / V-cycle must still agree with the oracle."""
    O = oracle
    nodes = np.array([(0, 0, 0), (1, 0, 0), (0.2, 1, 0), (0.1, 0.3, 1)], dtype=np.float64)
    m = O.Mesh(nodes, np.array([[0, 1, 2, 3]], dtype=np.int64))
    levels = 5
    sig = np.array([[1.0, 9.0, 9.0]])
    impl = O.ImplicitFineGrid.create(m, levels)
    cons = O.ZeroDirichletConstraint(*O.list_boundary_nodes_edges_faces(m))
    ops = [O.L2PlusDivAGrad(O.build_local_diffusion_operators(l), O.mass_matrix(l), cons, 0.5, sig)
           for l in impl.reference.levels]
    g = hmg.ImplicitFineGrid(ctx, hmg.Mesh(m.nodes, m.elements + 1), levels)
    A = hmg.L2PlusDivAGrad(g, 0.5, sig)
    rng = np.random.default_rng(0)
    sts = [O.LevelState.create(1, impl.nf(i + 1)) for i in range(levels)]
    sts[-1].x[...] = rng.random(sts[-1].x.shape); sts[-1].b[...] = rng.standard_normal(sts[-1].x.shape)
    O.apply_constraint(sts[-1].x, levels, cons, impl)
    dsts = [hmg.LevelState(g, i + 1) for i in range(levels)]
    dsts[-1].x.from_host(sts[-1].x); dsts[-1].b.from_host(sts[-1].b)
    before = dsts[-1].x.to_host()
    hmg.broadcast_interfaces(dsts[-1].x, g, levels)
    np.testing.assert_array_equal(dsts[-1].x.to_host(), before)          # nothing shared: a no-op
    assert O.list_interior_nodes(m).size == 0
    # (A V-cycle is not defined here even in the reference: on levels 2-3 every DOF is Dirichlet, r = 0 and the
    #  CG step divides 0/0.)  The finest level has interior DOFs: the smoother must agree.
    O.smoothing_steps(3, impl, ops[-1], sts[-1], levels)
    hmg.smoothing_steps(3, g, A, dsts[-1], levels)
    assert relerr(dsts[-1].x.to_host(), sts[-1].x) <= 1e-10
    for lev in range(1, levels + 1):
        x = np.asfortranarray(rng.standard_normal((impl.nf(lev), 1)))
        want = np.zeros_like(x, order="F"); O.mul(1.0, m, ops[lev - 1], x, want)
        dy = hmg.DeviceMatrix(g, lev).fill(0.0)
        hmg.mul(1.0, g, A, hmg.DeviceMatrix(g, lev).from_host(x), dy)
        assert relerr(dy.to_host(), want) <= TOL


def test_wrapped_torch_memory(case3):
    """hmg_vec_wrap: level vectors living in caller-owned device memory (a torch tensor), as the multi-GPU layer
    and a torch-based host would use them."""
    import torch
    c = case3
    lev = c.levels
    n = c.g.ld(lev) * c.g.ncells()
    tx = torch.zeros(n, dtype=torch.float64, device="cuda")
    ty = torch.zeros(n, dtype=torch.float64, device="cuda")
    x, y = c.rand(lev), c.rand(lev)
    dx = hmg.DeviceMatrix(c.g, lev, device_ptr=tx.data_ptr()).from_host(x)
    dy = hmg.DeviceMatrix(c.g, lev, device_ptr=ty.data_ptr()).from_host(y)
    want = y.copy(order="F")
    c.O.mul(0.5, c.mesh, c.ops[lev - 1], x, want)
    hmg.mul(0.5, c.g, c.A, dx, dy)
    assert relerr(dy.to_host(), want) <= TOL
    c.g.ctx.sync()
    # the library wrote into the tensor's storage: raw-storage sums agree
    assert abs(float((ty * ty).sum().item()) - hmg.dot(dy, dy)) <= 1e-9 * hmg.dot(dy, dy)
    assert dx.device_ptr() == tx.data_ptr()


def test_vcycle_leaves_wrapped_buffers_consistent(oracle, ctx):
    """Inside a V-cycle the smoother exchanges the device pointers of r and p at CG step 0 (no copy of r into p);
    pre- and post-smoother make it an even number of exchanges, so caller-owned buffers wrapped as r and p must
    hold r and p afterwards, and the result must equal the plain path (swap_rp = fold_x = fold_prolong = 0) bit for bit
    where the arithmetic is the same (r, p) and to rounding in x."""
    import torch
    O = oracle
    levels = 4
    c = Case(O, ctx, 3, 3, levels, lam=1.0, seed=31)
    n = c.g.ld(levels) * c.g.ncells()
    tr = torch.zeros(n, dtype=torch.float64, device="cuda")
    tp = torch.zeros(n, dtype=torch.float64, device="cuda")
    st0 = _oracle_state(c, levels)
    results = []
    for opts in ((1, 1), (0, 0)):
        ctx.set_option("swap_rp", opts[0]); ctx.set_option("fold_x", opts[1]); ctx.set_option("fold_prolong", opts[1]); ctx.set_option("lazy_dead", opts[1]); ctx.set_option("lean_post", opts[1])
        try:
            dsts = [hmg.LevelState(c.g, i + 1) for i in range(levels)]
            top = dsts[-1]
            top.r = hmg.DeviceMatrix(c.g, levels, device_ptr=tr.data_ptr())
            top.p = hmg.DeviceMatrix(c.g, levels, device_ptr=tp.data_ptr())
            top.x.from_host(st0.x); top.b.from_host(st0.b)
            hmg.vcycle(c.g, hmg.BaseLevel(c.g), [c.A] * levels, dsts, levels, 3)
            ctx.sync()
            assert top.r.device_ptr() == tr.data_ptr() and top.p.device_ptr() == tp.data_ptr()
            r_host, p_host = top.r.to_host(), top.p.to_host()
            # (the raw buffers are in storage order, to_host() in the reference's hierarchical order: per column
            #  the same values, permuted)
            raw_r = tr.cpu().numpy().reshape(r_host.shape, order="F")
            raw_p = tp.cpu().numpy().reshape(p_host.shape, order="F")
            np.testing.assert_array_equal(np.sort(raw_r, axis=0), np.sort(r_host, axis=0))
            np.testing.assert_array_equal(np.sort(raw_p, axis=0), np.sort(p_host, axis=0))
            assert not np.array_equal(np.sort(raw_r, axis=0), np.sort(raw_p, axis=0))
            results.append((top.x.to_host(), r_host, p_host))
        finally:
            ctx.set_option("swap_rp", 1); ctx.set_option("fold_x", 1); ctx.set_option("fold_prolong", 1); ctx.set_option("lazy_dead", 1); ctx.set_option("lean_post", 1)
    # (lean_post: p and Ap are scratch after hmg_vcycle -- the reference's last p-update is dead; the plain run keeps it)
    (xa, ra, pa), (xb, rb, pb) = results
    assert relerr(xa, xb) <= 1e-13 and relerr(ra, rb) <= 1e-12
    sts = [O.LevelState.create(c.mesh.nelements(), c.impl.nf(i + 1)) for i in range(levels)]
    sts[-1] = st0
    O.vcycle(c.impl, O.make_base_level(c.mesh, c.sig, 1.0), c.ops, sts, levels, 3)
    assert relerr(xa, sts[-1].x) <= 1e-9 and relerr(pb, sts[-1].p) <= 1e-8 and relerr(ra, sts[-1].r) <= 1e-8


def test_handles_may_be_destroyed_in_any_order(oracle):
    """include/hmg.h: a vector keeps its grid alive, a grid its context.  Destroying the context, then the grid, then the
    vectors (what the finalizers of a garbage-collected host may do) leaves the vectors usable and frees everything
    with the last one; a destroyed vector's block is handed to the next vector of the same size (option vec_pool),
    zero-filled again."""
    O = oracle
    m = O.order_nodes_and_elements_by_magnitude(O.hypercube(3, 2, origin=(-1.0, -1.0, -1.0)))
    ctx = hmg.Context(0)
    g = hmg.ImplicitFineGrid(ctx, hmg.Mesh(m.nodes, m.elements + 1), 3)
    a = hmg.DeviceMatrix(g, 3).fill(2.0)
    b = hmg.DeviceMatrix(g, 3).fill(3.0)
    ptr_b = b.device_ptr()
    b.close()
    c = hmg.DeviceMatrix(g, 3)
    assert c.device_ptr() == ptr_b and not c.to_host().any()
    c.fill(3.0)
    d = hmg.DeviceMatrix(g, 2)
    d.close()
    ctx.release_memory()                          # hmg_ctx_release_memory: the pooled block of d goes back to the device
    d = hmg.DeviceMatrix(g, 2)
    assert not d.to_host().any()
    d.close()
    n = a.shape[0] * a.shape[1]
    ctx.close()                                   # the caller's reference only
    assert g.ncells() == m.nelements()            # the grid still answers
    g.close()                                     # hmg_grid_destroy while two of its vectors live
    assert hmg.dot(a, c) == 6.0 * n
    a.close()
    assert hmg.dot(c, c) == 9.0 * n
    c.close()                                     # last reference: grid and context go now


EXACT_OPTIONS = ("lean_post", "lazy_post", "lazy_top", "lazy_dead", "fold_x", "swap_rp", "fold_prolong", "prolong_in_image", "fold_faces", "fold_restrict",
                 "zero_entry", "cell_order")


@pytest.mark.parametrize("steps", [1, 2, 4, 5])
def test_exact_savings_are_exact_for_other_step_counts(ctx, steps):
    """The same with 1, 2, 4 and 5 smoothing steps on the finest level (the deferred x-updates of its post-smoother take another
    form for each: none, two updates in the last r-update, three with the spare direction vector behind a regular step)."""
    from homogenization_jl_amd import driver
    levels = 4
    base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, 4, levels, seed=12)
    res = []
    try:
        for on in (1, 0, "lazy_top=1"):
            for o in EXACT_OPTIONS:
                ctx.set_option(o, 1 if on else 0)
            ctx.set_option("lazy_top", {1: 2, 0: 0}.get(on, 1))
            st = [hmg.LevelState(g, i + 1) for i in range(levels)]
            st[-1].x.rand(3); st[-1].b.rand(4)
            hmg.broadcast_interfaces(st[-1].x, g, levels)
            hmg.apply_constraint(st[-1].x, levels, g)
            bl = hmg.BaseLevel(g)
            for _ in range(2):
                hmg.vcycle(g, bl, [op] * levels, st, levels, steps)
            res.append((st[-1].x.to_host(), st[-1].r.to_host()))
            for s in st:
                s.close()
    finally:
        for o in EXACT_OPTIONS:
            ctx.set_option(o, 1)
        ctx.set_option("lazy_top", 2)
        g.close()
    assert np.isfinite(res[0][0]).all() and np.abs(res[0][0]).max() > 0
    for other in res[1:]:
        np.testing.assert_array_equal(res[0][0], other[0])
        np.testing.assert_array_equal(res[0][1], other[1])


@pytest.mark.parametrize("dim,n,levels", [(3, 4, 5), (3, 2, 6), (2, 8, 5)])
def test_exact_savings_are_exact(ctx, dim, n, levels):
    """Every saving hmg_vcycle takes that the reference's own control flow makes exact -- dead tails of both smoothers, r taken
    as p by a handle exchange, prolongation folded into a residual (coarse column staged in the image on level 6), face sums
    inside the r-update, restriction in the local residual's epilogue (level 6), coarse levels whose zero initial guess is
    never written -- on and off: x and r of the finest level after three V-cycles are the same to the last bit
    (src/multigrid.jl:46-119 is the plain sequence)."""
    from homogenization_jl_amd import driver
    tag = hmg.Tet64 if dim == 3 else hmg.Tri64
    base, cond, g, op = driver.checkerboard_problem(ctx, tag, n, levels, seed=11)
    res = []
    try:
        for on in (1, 0, "lazy_top=1"):
            for o in EXACT_OPTIONS:
                ctx.set_option(o, 1 if on else 0)
            ctx.set_option("lazy_top", {1: 2, 0: 0}.get(on, 1))      # (2, the default: three x-updates wait; 1: two)
            st = [hmg.LevelState(g, i + 1) for i in range(levels)]
            st[-1].x.rand(3); st[-1].b.rand(4)
            hmg.broadcast_interfaces(st[-1].x, g, levels)
            hmg.apply_constraint(st[-1].x, levels, g)
            bl = hmg.BaseLevel(g)
            allocs = ctx.counter("device_allocs")
            for _ in range(3):
                hmg.vcycle(g, bl, [op] * levels, st, levels, 3)
            # setup is over once grid, operator, level-1 system and level vectors exist: not even the FIRST V-cycle allocates
            # (the spare direction vector of lazy_top = 2 was reserved with the finest level's first vector)
            assert ctx.counter("device_allocs") == allocs
            if dim == 3:
                assert ctx.counter("lazy_top_form") == {1: 2, 0: 0}.get(on, 1)
                assert ctx.counter("spare_bytes") >= 8 * g.ld(levels) * g.ncells()
            res.append((st[-1].x.to_host(), st[-1].r.to_host(), hmg.norm_unique(st[-1].r)))
            for s in st:
                s.close()
    finally:
        for o in EXACT_OPTIONS:
            ctx.set_option(o, 1)
        ctx.set_option("lazy_top", 2)
        g.close()
    for other in res[1:]:
        np.testing.assert_array_equal(res[0][0], other[0])
        np.testing.assert_array_equal(res[0][1], other[1])
        assert res[0][2] == other[2] and np.isfinite(res[0][2])


def test_spare_vector_is_setup_memory_with_a_reported_state(ctx):
    """hmg_grid_reserve_spare (round 5): the sixth finest-level vector of the three-update form is reserved at setup, released on
    request -- V-cycles then take the two-update form and say so -- and x, r are the same bits either way."""
    from homogenization_jl_amd import driver
    levels = 4
    base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, 4, levels, seed=5)
    res = []
    try:
        held = ctx.counter("spare_bytes")              # (other grids of this context that are still alive)
        for spare in (True, False, True):
            g.reserve_spare(spare)
            assert ctx.counter("spare_bytes") - held == (8 * g.ld(levels) * g.ncells() if spare else 0)
            st = [hmg.LevelState(g, i + 1) for i in range(levels)]
            assert ctx.counter("spare_bytes") - held == (8 * g.ld(levels) * g.ncells() if spare else 0)   # (released stays released)
            st[-1].x.rand(3); st[-1].b.rand(4)
            hmg.broadcast_interfaces(st[-1].x, g, levels)
            hmg.apply_constraint(st[-1].x, levels, g)
            bl = hmg.BaseLevel(g)
            allocs = ctx.counter("device_allocs")
            for _ in range(2):
                hmg.vcycle(g, bl, [op] * levels, st, levels, 3)
            assert ctx.counter("device_allocs") == allocs
            assert ctx.counter("lazy_top_form") == (2 if spare else 1)
            res.append((st[-1].x.to_host(), st[-1].r.to_host()))
            for s_ in st:
                s_.close()
    finally:
        g.close()
    assert ctx.counter("spare_bytes") == held             # the grid took its spare vector with it
    for other in res[1:]:
        np.testing.assert_array_equal(res[0][0], other[0])
        np.testing.assert_array_equal(res[0][1], other[1])
