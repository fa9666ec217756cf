"""
Pins the CPU oracle against the reference's own known-answer tests (the reference ships no
golden data files and cannot be executed here -- no Julia toolchain).  Each test restates one
file of /root/reference/test (cited), against oracle/oracle.py + oracle/hmg_oracle.c.
"""
import numpy as np
import pytest

CUBE_NODES = np.array([(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0), (0, 0, 1), (1, 0, 1), (0, 1, 1), (1, 1, 1)],
                      dtype=np.float64)
CUBE_ELEMENTS = np.array([(1, 2, 3, 5), (2, 3, 4, 8), (3, 5, 7, 8), (2, 5, 6, 8), (2, 3, 5, 8)], dtype=np.int64) - 1


def five_tet_cube(O, times):
    base = O.refine_uniformly(O.Mesh(CUBE_NODES.copy(), CUBE_ELEMENTS.copy()), times=times)
    base.elements = O.sort_element_nodes(base.elements)
    return base


def test_operator_matches_assembled_matrix(oracle):
    """ref: test/test_operator.jl:9-73 -- implicit mul! + broadcast_interfaces! == assembled A*x, <= 20 eps."""
    O = oracle
    levels = 5
    base = five_tet_cube(O, 1)
    implicit = O.ImplicitFineGrid.create(base, levels)
    nf = implicit.nf(levels)
    rng = np.random.default_rng(1)
    local_x = np.asfortranarray(rng.random((nf, base.nelements())))
    O.broadcast_interfaces(local_x, implicit, levels)
    local_y = np.zeros_like(local_x, order="F")
    constraint = O.ZeroDirichletConstraint(*O.list_boundary_nodes_edges_faces(base))
    A = O.SimpleDiffusion(O.build_local_diffusion_operators(implicit.reference.levels[levels - 1]), constraint, 1.0)

    rep = implicit.construct_full_grid(levels).reshape(-1, 3)          # repeated nodes, cell-major
    total_fine = O.refine_uniformly(base, times=levels - 1)
    total_A = O.assemble_matrix_dot(total_fine)

    # geometric node matching (test_operator.jl:35-47); coordinates are exact dyadics
    key = lambda p: np.round(p * 4096).astype(np.int64) @ np.array([1, 1 << 20, 1 << 40])
    tk = key(total_fine.nodes)
    order = np.argsort(tk)
    pos = np.searchsorted(tk[order], key(rep))
    assert np.all(tk[order][pos] == key(rep))                           # every node matches another
    mapping = order[pos]
    assert np.max(np.abs(total_fine.nodes[mapping] - rep)) < 1e-4

    total_x = np.zeros(total_fine.nnodes())
    total_x[mapping] = local_x.reshape(-1, order="F")
    O.mul(1.0, base, A, local_x, local_y)
    O.broadcast_interfaces(local_y, implicit, levels)
    total_y = total_A @ total_x
    err = np.max(np.abs(total_y[mapping] - local_y.reshape(-1, order="F")))
    assert err <= 20 * np.finfo(float).eps


def test_interpolation_reproduces_affine(oracle):
    """ref: test/interpolation.jl:8-35"""
    O = oracle
    total_levels = 6
    base = five_tet_cube(O, 0)
    implicit = O.ImplicitFineGrid.create(base, total_levels)
    direction = np.random.default_rng(2).standard_normal(3)
    xs = 10.0 + base.nodes @ direction
    ys = np.zeros((implicit.nf(1), base.nelements()), order="F")
    O.distribute(ys, xs, implicit)
    for level in range(2, total_levels + 1):
        P = implicit.reference.interops[level - 2]
        # y <- 0 + P*x through the oracle's transfer kernel
        new = np.zeros((implicit.nf(level), base.nelements()), order="F")
        O.interpolate_and_sum_to(new, P, ys)
        ys = new
        full = implicit.construct_full_grid(level)                     # (Ne, Nf, 3)
        expect = 10.0 + full @ direction
        np.testing.assert_allclose(ys.T, expect, rtol=1e-8)


def test_refined_reference_element(oracle):
    """ref: test/refined_reference_element.jl:5-37"""
    O = oracle
    N = 8
    tets = O.refined_element(N, 3)
    assert tets.levels[0].nnodes() == 4
    assert tets.levels[1].nnodes() == 10
    nb = tets.numbering[0]
    assert [list(f + 1) for f in nb.faces] == [[1, 2, 3], [1, 2, 4], [1, 3, 4], [2, 3, 4]]
    assert [list(e + 1) for e in nb.edges] == [[1, 2], [1, 3], [1, 4], [2, 3], [2, 4], [3, 4]]
    for i in range(1, N + 1):
        for f in tets.numbering[i - 1].faces:
            assert len(f) == sum(range(1, 2 ** (i - 1) + 2))
        for e in tets.numbering[i - 1].edges:
            assert len(e) == 2 ** (i - 1) + 1


def test_interface_nodes_coincide(oracle):
    """ref: test/implicit_grid.jl:8-92 -- ordering inside faces_interior/edges_interior is consistent across cells."""
    O = oracle
    refs = 5
    base = five_tet_cube(O, 3)
    implicit = O.ImplicitFineGrid.create(base, refs)
    inter = implicit.interfaces
    for level in range(1, refs + 1):
        nb = implicit.reference.numbering[level - 1]
        full = implicit.construct_full_grid(level)                     # (Ne, Nf, 3)
        for smap, lists in ((inter.nodes, [[n] for n in nb.nodes]), (inter.edges, nb.edges_interior),
                            (inter.faces, nb.faces_interior)):
            per = len(lists[0])
            if per == 0:
                continue
            idx = np.array([np.asarray(l) for l in lists])              # (nlocal, per)
            xs = full[smap.element[:, None], idx[smap.local_id]]        # (nvals, per, 3)
            first = np.repeat(smap.offset[:-1], np.diff(smap.offset))
            np.testing.assert_allclose(xs, xs[first], rtol=0, atol=1e-12)


def test_list_faces(oracle):
    """ref: test/list_faces.jl:6-27"""
    O = oracle
    nodes = np.array([(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)], dtype=np.float64)
    mesh = O.Mesh(nodes, np.array([[0, 1, 2, 3]], dtype=np.int64))
    n, e, f = O.list_boundary_nodes_edges_faces(mesh)
    assert (len(f), len(e), len(n)) == (4, 6, 4)
    mesh2 = O.refine_uniformly(mesh, times=2)
    mesh2.elements = O.sort_element_nodes(mesh2.elements)
    n, e, f = O.list_boundary_nodes_edges_faces(mesh2)
    assert len(f) == 4 * 16
    assert len(e) == 2 * 16 * 3
    assert len(n) == sum(range(1, 6)) * 4 - 6 * 3 - 2 * 4
    assert len(O.list_interior_nodes(mesh2)) == mesh2.nnodes() - len(n)


def test_sparse_cell_to_element(oracle):
    """ref: test/sparse_cell_to_element.jl:4-27 (1-based expected values)"""
    O = oracle
    m = O.compress(np.array([(1, 2), (1, 2), (2, 3), (2, 3)]), np.array([1, 2, 3, 5]), np.array([2, 3, 4, 6]))
    assert list(m.offset + 1) == [1, 3, 5]
    assert [tuple(c) for c in m.cells] == [(1, 2), (2, 3)]
    assert list(zip(m.element, m.local_id)) == [(1, 2), (2, 3), (3, 4), (5, 6)]
    m = O.compress(np.array([(1, 2), (2, 3)]), np.array([1, 3]), np.array([2, 4]))
    assert list(m.offset + 1) == [1, 2, 3]
    assert [tuple(c) for c in m.cells] == [(1, 2), (2, 3)]
    assert list(zip(m.element, m.local_id)) == [(1, 2), (3, 4)]


def test_generated_grids(oracle):
    """ref: test/generated_grids.jl:4-10"""
    O = oracle
    mesh = O.hypercube(3, 20)
    assert np.all(np.diff(mesh.elements, axis=1) > 0)
    assert mesh.nnodes() == 21 ** 3
    assert mesh.nelements() == 6 * 20 ** 3
    # the 6 tets of every cube tile it: total volume == 20^3
    _, _, det = O.cell_geometry(mesh)
    assert abs(det.sum() / 6 - 20 ** 3) < 1e-9


def test_sorted_set_helpers(oracle):
    """ref: test/tricks.jl, test/counting_sort.jl -- helpers as used by the interface builders."""
    O = oracle
    nodes = np.array([(3, 1), (1, 2), (3, 1), (1, 1), (2, 9)])
    o = O._stable_lexsort(nodes)
    assert [tuple(r) for r in nodes[o]] == [(1, 1), (1, 2), (2, 9), (3, 1), (3, 1)]
    assert list(o[-2:]) == [0, 2]                                       # stable
    n, e, l = O._remove_singletons(nodes[o], np.arange(5), np.arange(5))
    assert [tuple(r) for r in n] == [(3, 1), (3, 1)]


def test_documented_grid_size_example(oracle):
    """ref: docs/src/index.md:192-194 -- Tri n=32, level 3: 1089 nodes, 2048 cells, 15 nodes/cell."""
    O = oracle
    base = O.hypercube(2, 32)
    implicit = O.ImplicitFineGrid.create(base, 3)
    assert (base.nnodes(), base.nelements(), implicit.nf(3)) == (1089, 2048, 15)
    assert implicit.nf(3) * base.nelements() == 30720


@pytest.mark.parametrize("dim,n,grids", [(2, 3, 4), (3, 2, 3)])
def test_multigrid_converges_to_direct_solution(oracle, dim, n, grids):
    """Independent second oracle (recipe of checkerboard_hypercube_full, ...homogenized_coefficients.jl:729-759):
    explicit refine_uniformly + assemble_checkerboard + sparse direct solve of -div(a grad u) = 1; the V-cycle
    iteration of checkerboard_hypercube_multigrid (:509-571: local_rhs!, lambda = 0) must converge to it."""
    import scipy.sparse.linalg as spla
    O = oracle
    rs, top, implicit, base, cond = O.checkerboard_hypercube_multigrid(n, dim=dim, refinements=grids, max_cycles=40, seed=4)
    assert rs[-1] < 1e-9 * rs[0]
    fine = O.refine_uniformly(base, times=grids - 1)
    fine.elements = O.sort_element_nodes(fine.elements)
    # conductivity of a fine cell = that of the unit cube containing its centre
    rng = np.random.default_rng(4)
    sgrid = np.where(rng.random((n,) * dim + (dim,)) < 0.5, 1.0, 9.0)
    np.testing.assert_array_equal(O.conductivity_per_element(base, sgrid, (0.0,) * dim), cond)
    A = O.assemble_checkerboard(fine, O.conductivity_per_element(fine, sgrid, (0.0,) * dim), 0.0).tocsr()
    b = O.assemble_vector(fine)
    interior = O.list_interior_nodes(fine)
    u = np.zeros(fine.nnodes())
    u[interior] = spla.spsolve(A[interior][:, interior].tocsc(), b[interior])
    rep = implicit.construct_full_grid(grids).reshape(-1, dim)
    scale = 1 << 12
    w = np.array([1, 1 << 20, 1 << 40][:dim])
    key = lambda p: np.round(p * scale).astype(np.int64) @ w
    tk = key(fine.nodes)
    order = np.argsort(tk)
    pos = np.searchsorted(tk[order], key(rep))
    assert np.all(tk[order][pos] == key(rep))
    want = u[order[pos]].reshape(top.x.shape, order="F")
    assert np.abs(top.x - want).max() <= 1e-9 * np.abs(want).max()


@pytest.mark.parametrize("dim,n,grids,steps", [(2, 4, 3, 3), (2, 3, 4, 1), (3, 2, 3, 3)])
def test_vcycle_equals_its_global_matrix_form(oracle, dim, n, grids, steps):
    """Second independent statement of smoothing_steps! / vcycle! (src/multigrid.jl:46-119): tests/_global_form.py -- the same
    iteration on GLOBAL vectors with explicitly assembled matrices per level and multiplicity-weighted dot products.  None of
    the cell-local machinery takes part, so agreement to rounding pins all of it at once."""
    from _global_form import GlobalForm
    O = oracle
    lam = 1.0
    rng = np.random.default_rng(11)
    base = O.hypercube(dim, n)
    sgrid = np.where(rng.random((n,) * dim + (dim,)) < 0.5, 1.0, 9.0)
    cond = O.conductivity_per_element(base, sgrid, (0.0,) * dim)
    base_level = O.make_base_level(base, cond, lam)
    implicit = O.ImplicitFineGrid.create(base, grids)
    constraint = O.ZeroDirichletConstraint(*O.list_boundary_nodes_edges_faces(base))
    ops = [O.L2PlusDivAGrad(O.build_local_diffusion_operators(m), O.mass_matrix(m), constraint, lam, cond)
           for m in implicit.reference.levels]
    states = [O.LevelState.create(base.nelements(), implicit.nf(i + 1)) for i in range(grids)]
    top = states[-1]
    top.x[...] = rng.random(top.x.shape)
    O.broadcast_interfaces(top.x, implicit, grids)
    O.apply_constraint(top.x, grids, constraint, implicit)
    O.local_rhs(top.b, implicit)
    G = GlobalForm(O, base, sgrid, lam, implicit, grids, dim)
    gx = G.gather(top.x, grids - 1)
    gb = G.gather_sum(top.b, grids - 1)
    for cycle in range(2):
        O.vcycle(implicit, base_level, ops, states, grids, steps)
        gx, gr = G.vcycle(grids - 1, gx, gb, steps)
        ox, orr = G.gather(top.x, grids - 1), G.gather(top.r, grids - 1)
        assert np.abs(ox - gx).max() <= 1e-11 * np.abs(gx).max(), (cycle, np.abs(ox - gx).max())
        assert np.abs(orr - gr).max() <= 1e-10 * max(np.abs(gr).max(), 1e-300), (cycle, np.abs(orr - gr).max())


@pytest.mark.parametrize("dim,width,levels,radius", [(2, 6, 4, 2), (3, 4, 3, 1)])
def test_driver_integrals_equal_textbook_fem(oracle, dim, width, levels, radius):
    """rhs_a xi grad v!, next_rhs!, integrate_first_term / _terms / _area (src/examples/homogenized_coefficients.jl:449-474,
    592-713) against textbook P1 elements on the explicitly refined mesh (tests/_textbook_fem.py: cell volumes and hat-function
    gradients only).  Unit cubes (|J| = 1), as in every use the reference makes of them."""
    O = oracle
    from _textbook_fem import driver_setting
    base, cond, implicit, T, a_fine, inside, mapping, nint, rng = driver_setting(O, dim, width, levels, radius)
    xi = rng.standard_normal(dim)
    nf, ne = implicit.nf(levels), base.nelements()
    fine_ref = implicit.reference.levels[-1]
    dphis, mass = O.partial_derivatives_functionals(fine_ref), O.mass_matrix(fine_ref)
    gather_sum = lambda loc: np.bincount(mapping, weights=loc.reshape(-1, order="F"), minlength=len(T.nodes))

    def consistent(seed):
        v = np.asfortranarray(np.random.default_rng(seed).standard_normal((nf, ne)))
        O.broadcast_interfaces(v, implicit, levels)
        g = np.zeros(len(T.nodes))
        g[mapping] = v.reshape(-1, order="F")
        return v, g

    b = np.zeros((nf, ne), order="F")
    O.rhs_axi_grad_v(b, dphis, implicit, cond, xi)
    F = T.load(a_fine, xi)
    assert np.abs(gather_sum(b) - F).max() <= 1e-12 * np.abs(F).max()
    v0, g0 = consistent(1)
    v1, g1 = consistent(2)
    want = float(np.dot(g0, T.load(a_fine, xi, inside))) + T.mass_quadratic(g0, g0, inside)
    got = O.integrate_first_term(v0, dphis, implicit, nint, mass, cond, xi)
    assert abs(got - want) <= 1e-11 * abs(want)
    want = T.mass_quadratic(g0 + g1, g0, inside)
    got = O.integrate_terms(v0, v1, implicit, nint, mass)
    assert abs(got - want) <= 1e-11 * abs(want)
    assert abs(O.integrate_area(mass, implicit, nint) - float(np.dot(T.vol, inside))) <= 1e-12 * nint
    lam = 0.37
    O.next_rhs(b, v0, implicit, mass, lam)
    want = lam * T.mass_apply(g0)
    assert np.abs(gather_sum(b) - want).max() <= 1e-12 * np.abs(want).max()


@pytest.mark.parametrize("dim,refinements", [(2, 3), (3, 1)])
def test_driver_converges_to_the_direct_fem_answer(oracle, dim, refinements):
    """checkerboard_homogenization with n = 0 (one outer step; src/examples/homogenized_coefficients.jl:174-343) run to a tight
    tolerance must land on what a sparse direct solve of the same boundary value problem on the explicitly refined mesh gives
    (tests/_textbook_fem.py) -- mesh ordering, conductivity lookup, right-hand side, V-cycle, integrals and the sigma formula
    in one number."""
    from _textbook_fem import converged_first_term
    O = oracle
    rng = np.random.default_rng(8)
    sgrid = np.where(rng.random((10,) * dim + (dim,)) < 0.5, 1.0, 9.0)
    xi = rng.standard_normal(dim)
    xi /= np.linalg.norm(xi)
    sigma, hist = O.checkerboard_homogenization(n=0, dim=dim, refinements=refinements, tolerance=1e-12, xi=xi, sigma_grid=sgrid,
                                                max_cycles=60)
    want = converged_first_term(O, dim, sgrid, xi, refinements)
    assert abs(sigma - want) <= 1e-9 * abs(want), (sigma, want, len(hist))
