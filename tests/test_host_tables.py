"""
CPU tests of the product's host-side tables (C++ in libhmg_hip.so, reached through the C ABI with a
NULL context -- no device, no compute) against the oracle's literal restatement of the reference:
hierarchical numbering, entity lists and orders, transfer tables, stencil tables, mesh masks,
coarse matrix.  The stencil is evaluated here in numpy from the exported tables (test-side
arithmetic, not a product code path).
"""
import numpy as np
import pytest

import homogenization_jl_amd as hmg

DIRS3 = np.array([(0, 0, 0), (1, 0, 0), (-1, 0, 0), (-1, 1, 0), (1, -1, 0), (0, 1, 0), (0, -1, 0), (0, -1, 1),
                  (0, 1, -1), (-1, 0, 1), (1, 0, -1), (0, 0, 1), (0, 0, -1), (1, -1, 1), (-1, 1, -1)])


def small_mesh(O, dim, n=2, perturb=0.0, seed=0):
    m = O.hypercube(dim, n)
    if perturb:
        rng = np.random.default_rng(seed)
        m.nodes = m.nodes + perturb * (rng.random(m.nodes.shape) - 0.5)
    return m


def host_grid(m, levels):
    return hmg.ImplicitFineGrid(None, hmg.Mesh(m.nodes, m.elements + 1), levels)


@pytest.mark.parametrize("dim,levels", [(3, 6), (2, 7)])
def test_numbering_and_entities(oracle, dim, levels):
    O = oracle
    ref = O.refined_element(levels, dim)
    g = host_grid(small_mesh(O, dim, 1), levels)
    for lev in range(1, levels + 1):
        m = 2 ** (lev - 1)
        nf = ref.levels[lev - 1].nnodes()
        assert g.nf(lev) == nf
        h2s = g.table_i32("hier2slot", lev)
        ijk = g.table_i32("slot_ijk", lev).reshape(-1, 3)
        lay = g.table_i32("layout", lev)
        nfx, ld, ncorner, nedge, nface, nei, nfi, nint, off_edge, off_face, off_int = lay[:11]
        assert sorted(h2s) == list(range(nf))
        # lattice coordinates of every hierarchical node match the literally refined mesh
        want = np.round(ref.levels[lev - 1].nodes * m).astype(int)
        np.testing.assert_array_equal(ijk[h2s][:, :dim], want)
        nb = ref.numbering[lev - 1]
        # entity-major storage: every entity of the reference's numbering is one contiguous run
        # (inside a run the library uses lattice order, see test_entity_order_is_canonical)
        np.testing.assert_array_equal(h2s[nb.nodes], np.arange(dim + 1))
        for e, lst in enumerate(nb.edges_interior):
            np.testing.assert_array_equal(np.sort(h2s[lst]), off_edge + e * nei + np.arange(len(lst)))
            assert len(lst) == nei
        if dim == 3:
            for f, lst in enumerate(nb.faces_interior):
                np.testing.assert_array_equal(np.sort(h2s[lst]), off_face + f * nfi + np.arange(len(lst)))
                assert len(lst) == nfi


@pytest.mark.parametrize("dim,levels,times", [(3, 5, 2), (2, 6, 2)])
def test_entity_order_is_canonical(oracle, dim, levels, times):
    """The k-th stored DOF of a shared face / edge is the same physical point in every adjacent cell
    (the property test/implicit_grid.jl:8-92 checks for the reference's ascending-id lists), here for
    the library's lattice order, on a refined 5-tet cube / a perturbed triangle mesh."""
    O = oracle
    if dim == 3:
        nodes = np.array([(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0), (0, 0, 1), (1, 0, 1), (0, 1, 1), (1, 1, 1)], float)
        els = np.array([(1, 2, 3, 5), (2, 3, 4, 8), (3, 5, 7, 8), (2, 5, 6, 8), (2, 3, 5, 8)]) - 1
        m = O.refine_uniformly(O.Mesh(nodes, els), times=times)
        m.elements = O.sort_element_nodes(m.elements)
    else:
        m = small_mesh(O, 2, 4, perturb=0.3, seed=9)
    g = host_grid(m, levels)
    impl = O.ImplicitFineGrid.create(m, levels)
    for lev in range(2, levels + 1):
        lay = g.table_i32("layout", lev)
        nei, nfi, off_edge, off_face = lay[5], lay[6], lay[8], lay[9]
        h2s = g.table_i32("hier2slot", lev)
        s2h = np.argsort(h2s)
        full = impl.construct_full_grid(lev)                              # (Ne, Nf(hier), dim)
        if dim == 3 and nfi:
            fp = g.table_i32("face_pairs").reshape(-1, 3)
            for ca, cb, lf in fp:
                a = full[ca, s2h[off_face + (lf & 15) * nfi + np.arange(nfi)]]
                b = full[cb, s2h[off_face + (lf >> 4) * nfi + np.arange(nfi)]]
                np.testing.assert_allclose(a, b, rtol=0, atol=1e-12)
        if nei:
            ptr, ent = g.table_i32("edge_ptr"), g.table_i32("edge_ent")
            for e in range(len(ptr) - 1):
                pts = [full[v >> 3, s2h[off_edge + (v & 7) * nei + np.arange(nei)]] for v in ent[ptr[e]:ptr[e + 1]]]
                for p in pts[1:]:
                    np.testing.assert_allclose(p, pts[0], rtol=0, atol=1e-12)


@pytest.mark.parametrize("dim,levels", [(3, 5), (2, 6)])
def test_transfer_tables(oracle, dim, levels):
    O = oracle
    ref = O.refined_element(levels, dim)
    g = host_grid(small_mesh(O, dim, 1), levels)
    rng = np.random.default_rng(0)
    for lev in range(2, levels + 1):
        P = ref.interops[lev - 2]
        h2s_f, h2s_c = g.table_i32("hier2slot", lev), g.table_i32("hier2slot", lev - 1)
        pa, pb = g.table_i32("par_a", lev), g.table_i32("par_b", lev)
        xc = rng.random(P.shape[1])
        xc_s = np.zeros_like(xc); xc_s[h2s_c] = xc
        yf_s = np.where(pa == pb, xc_s[pa], 0.5 * xc_s[pa] + 0.5 * xc_s[pb])
        np.testing.assert_allclose(yf_s[h2s_f], P @ xc, rtol=0, atol=1e-15)
        # parent order = CSC column order (smaller hierarchical id first)
        s2h_c = np.argsort(h2s_c)
        assert np.all(s2h_c[pa] <= s2h_c[pb])
        rptr, ridx = g.table_i32("rptr", lev), g.table_i32("ridx", lev)
        xf = rng.random(P.shape[0])
        xf_s = np.zeros_like(xf); xf_s[h2s_f] = xf
        bc_s = np.array([xf_s[ridx[rptr[c]]] + 0.5 * xf_s[ridx[rptr[c] + 1:rptr[c + 1]]].sum()
                         for c in range(P.shape[1])])
        np.testing.assert_allclose(bc_s[h2s_c], P.T @ xf, rtol=0, atol=1e-14)


def eval_stencil(g, lev, dim, coef, alpha, lam, x_slots):
    """numpy evaluation of the exported class stencil for every cell: y (nf, ne) in slot order."""
    ndir = 15 if dim == 3 else 7
    nterm = 7 if dim == 3 else 4
    ijk = g.table_i32("slot_ijk", lev).reshape(-1, 3)
    cls = g.table_i32("slot_cls", lev)
    ctab = g.table_f64("ctab", lev).reshape(-1, ndir, nterm)
    m = g.table_i32("layout", lev)[16]
    nf, ne = x_slots.shape
    at = -np.ones((m + 3, m + 3, m + 3), dtype=int)
    at[ijk[:, 0] + 1, ijk[:, 1] + 1, ijk[:, 2] + 1] = np.arange(nf)
    y = np.zeros_like(x_slots)
    s = coef.reshape(-1, 8)[:, :nterm].copy() * alpha
    s[:, nterm - 1] *= lam
    W = np.einsum("cdt,et->ecd", ctab, s)                      # (ne, ncls, ndir)
    for d in range(ndir):
        nb = at[ijk[:, 0] + 1 + DIRS3[d, 0], ijk[:, 1] + 1 + DIRS3[d, 1], ijk[:, 2] + 1 + DIRS3[d, 2]]
        ok = nb >= 0
        w = W[:, cls, d].T                                       # (nf, ne)
        assert np.all(w[~ok] == 0.0)                             # out-of-cell taps carry zero weight
        y[ok] += w[ok] * x_slots[nb[ok]]
    return y


@pytest.mark.parametrize("dim,levels,n", [(3, 4, 2), (3, 5, 1), (2, 5, 3)])
def test_stencil_tables_match_reference_apply(oracle, dim, levels, n):
    """class-form lattice stencil == the reference's dim^2+1 CSC scatter passes, on a perturbed mesh."""
    O = oracle
    m = small_mesh(O, dim, n, perturb=0.3, seed=4)
    rng = np.random.default_rng(5)
    sig = rng.choice([1.0, 9.0], size=(m.nelements(), dim))
    lam, alpha = 0.7, -1.3
    g = host_grid(m, levels)
    g.set_operator(sig, lam)
    coef = g.table_f64("coef")
    impl = O.ImplicitFineGrid.create(m, levels)
    for lev in range(1, levels + 1):
        lvl = impl.reference.levels[lev - 1]
        A = O.L2PlusDivAGrad(O.build_local_diffusion_operators(lvl), O.mass_matrix(lvl), None, lam, sig)
        nf = impl.nf(lev)
        x = np.asfortranarray(rng.standard_normal((nf, m.nelements())))
        y = np.zeros_like(x, order="F")
        O.mul(alpha, m, A, x, y)
        h2s = g.table_i32("hier2slot", lev)
        xs = np.zeros_like(x); xs[h2s] = x
        ys = eval_stencil(g, lev, dim, coef, alpha, lam, xs)
        scale = np.abs(y).max()
        assert np.abs(ys[h2s] - y).max() <= 1e-13 * scale


@pytest.mark.parametrize("kind", ["lattice", "delaunay"])
def test_mesh_masks_and_lists(oracle, kind):
    O = oracle
    if kind == "lattice":
        m = O.order_nodes_and_elements_by_magnitude(O.hypercube(3, 4, origin=(-2.0, -2.0, -2.0)))
    else:
        from _meshes import delaunay_mesh
        m = delaunay_mesh(O, 3, 80, 7)
    levels = 3
    g = host_grid(m, levels)
    impl = O.ImplicitFineGrid.create(m, levels)
    cn, ce, cf = O.list_boundary_nodes_edges_faces(m)
    nface, nedge = 4, 6
    want = np.zeros(m.nelements(), dtype=np.int64)
    np.bitwise_or.at(want, cf.element, 1 << cf.local_id)
    np.bitwise_or.at(want, ce.element, 1 << (nface + ce.local_id))
    np.bitwise_or.at(want, cn.element, 1 << (nface + nedge + cn.local_id))
    np.testing.assert_array_equal(g.table_i32("dmask"), want)
    # duplicate mask: copies 2..n of each shared entity
    inter = impl.interfaces
    dup = np.zeros(m.nelements(), dtype=np.int64)
    for smap, shift in ((inter.faces, 0), (inter.edges, nface), (inter.nodes, nface + nedge)):
        first = np.zeros(len(smap.element), dtype=bool)
        first[smap.offset[:-1]] = True
        np.bitwise_or.at(dup, smap.element[~first], 1 << (shift + smap.local_id[~first]))
    np.testing.assert_array_equal(g.table_i32("dupmask"), dup)
    # shared lists keep the reference's copy order (ascending cell)
    # (faces: the two copies of a face keep the reference's order; the list itself is sorted by its first cell, then
    #  by that cell's local face -- the faces are independent of one another, the r-update fetches them per cell)
    fp = g.table_i32("face_pairs").reshape(-1, 3)
    assert fp.shape[0] == len(inter.faces)
    o0 = inter.faces.offset[:-1]
    ref = np.stack([inter.faces.element[o0], inter.faces.element[o0 + 1],
                    inter.faces.local_id[o0], inter.faces.local_id[o0 + 1]], axis=1)
    ref = ref[np.lexsort((ref[:, 2], ref[:, 0]))]
    assert np.all(np.diff(fp[:, 0]) >= 0)
    got = np.stack([fp[:, 0], fp[:, 1], fp[:, 2] & 15, fp[:, 2] >> 4], axis=1)
    got = got[np.lexsort((got[:, 2], got[:, 0]))]
    np.testing.assert_array_equal(got, ref)
    np.testing.assert_array_equal(g.table_i32("edge_ptr"), inter.edges.offset)
    np.testing.assert_array_equal(g.table_i32("edge_ent"), inter.edges.element * 8 + inter.edges.local_id)
    np.testing.assert_array_equal(g.table_i32("node_ptr"), inter.nodes.offset)
    np.testing.assert_array_equal(g.table_i32("node_ent"), inter.nodes.element * 8 + inter.nodes.local_id)
    an = inter.all_nodes
    np.testing.assert_array_equal(g.table_i32("node_first"),
                                  an.element[an.offset[:-1]] * 8 + an.local_id[an.offset[:-1]])
    np.testing.assert_array_equal(g.interior_nodes(), O.list_interior_nodes(m))


@pytest.mark.parametrize("dim", [2, 3])
def test_coarse_matrix(oracle, dim):
    import scipy.sparse as sp
    O = oracle
    m = small_mesh(O, dim, 3, perturb=0.2, seed=1)
    rng = np.random.default_rng(2)
    sig = rng.choice([1.0, 9.0], size=(m.nelements(), dim))
    g = host_grid(m, 2)
    g.set_operator(sig, 0.35)
    g.coarse_setup()
    interior = O.list_interior_nodes(m)
    want = O.assemble_checkerboard(m, sig, 0.35).tocsr()[interior][:, interior]
    got = sp.csr_matrix((g.table_f64("coarse_val"), g.table_i32("coarse_colidx"), g.table_i32("coarse_rowptr")),
                        shape=want.shape)
    assert abs(got - want).max() <= 1e-13 * abs(want).max()


def test_shrink_rebuilds_boundary(oracle):
    O = oracle
    m = O.order_nodes_and_elements_by_magnitude(O.hypercube(3, 6, origin=(-3.0, -3.0, -3.0)))
    g = host_grid(m, 2)
    ne = O.find_elements_in_radius(m, 2)
    nn = O.find_nodes_in_radius(m, 2)
    g.shrink(ne, nn)
    sub = O.Mesh(m.nodes[:nn], m.elements[:ne])
    assert g.ncells() == ne and g.nnodes_base() == nn
    np.testing.assert_array_equal(g.interior_nodes(), O.list_interior_nodes(sub))


def test_library_exports_every_declared_symbol():
    import re, os
    from importlib import import_module
    lib = hmg._lib.load()
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "hmg.h")).read()
    declared = set(re.findall(r"\b(hmg_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"hmg_exchange_fn"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/hmg.h but not exported"
        assert name in hmg._lib.SIGNATURES, f"{name} has no ctypes signature"


def test_no_cpu_compute_path():
    """Without a device context there is no compute: the library refuses, it does not fall back."""
    import ctypes
    lib = hmg._lib.load()
    h = ctypes.c_void_p()
    rc = lib.hmg_ctx_create(0, None, ctypes.byref(h))
    import torch
    if not torch.cuda.is_available():
        assert rc != 0 and b"no CPU fallback" in lib.hmg_last_error()


@pytest.mark.parametrize("dim,n", [(3, 5), (2, 7)])
def test_driver_mesh_matches_reference_generator(oracle, dim, n):
    """Product host mesh synthesis == literal restatement of hypercube + order_nodes_and_elements_by_magnitude
    (ref: src/tet/generate_grid.jl, src/examples/homogenized_coefficients.jl:21-28, :494-503)."""
    from homogenization_jl_amd import driver
    O = oracle
    tag = hmg.Tet64 if dim == 3 else hmg.Tri64
    origin = (-n / 2.0,) * dim
    a = driver.hypercube(tag, n, origin=origin)
    b = O.hypercube(dim, n, origin=origin)
    np.testing.assert_array_equal(a.nodes, b.nodes)
    np.testing.assert_array_equal(a.elements - 1, b.elements)
    a2 = driver.order_nodes_and_elements_by_magnitude(a)
    b2 = O.order_nodes_and_elements_by_magnitude(b)
    np.testing.assert_array_equal(a2.nodes, b2.nodes)
    np.testing.assert_array_equal(a2.elements - 1, b2.elements)
    grid = driver.generate_conductivity(dim, n, 3)
    off = tuple(1.0 - o for o in origin)
    np.testing.assert_array_equal(driver.conductivity_per_element(a2, grid, off),
                                  O.conductivity_per_element(b2, grid, off))
    assert driver.find_elements_in_radius(a2, 1) == O.find_elements_in_radius(b2, 1)
    assert driver.find_nodes_in_radius(a2, 1) == O.find_nodes_in_radius(b2, 1)
    for lam, nn in ((1.0, 2), (0.5, 3), (0.25, 1)):
        assert driver.compute_boundary_layer(lam, nn) == O.compute_boundary_layer(lam, nn)
        assert driver.compute_box_radius(2, nn) == O.compute_box_radius(2, nn)


def test_input_validation_reports_errors():
    """Bad meshes are refused with a message (status code + hmg_last_error), never a crash."""
    nodes = np.array([(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)], dtype=np.float64)
    with pytest.raises(hmg._lib.HmgError, match="ascending"):
        hmg.ImplicitFineGrid(None, hmg.Mesh(nodes, np.array([[2, 1, 3, 4]])), 2)
    with pytest.raises(hmg._lib.HmgError, match="out of range"):
        hmg.ImplicitFineGrid(None, hmg.Mesh(nodes, np.array([[1, 2, 3, 5]])), 2)
    with pytest.raises(hmg._lib.HmgError, match="degenerate"):
        flat = nodes.copy(); flat[3] = (1, 1, 0)
        hmg.ImplicitFineGrid(None, hmg.Mesh(flat, np.array([[1, 2, 3, 4]])), 2)
    with pytest.raises(hmg._lib.HmgError, match="nlevels"):
        hmg.ImplicitFineGrid(None, hmg.Mesh(nodes, np.array([[1, 2, 3, 4]])), 8)
    g = hmg.ImplicitFineGrid(None, hmg.Mesh(nodes, np.array([[1, 2, 3, 4]])), 3)
    with pytest.raises(hmg._lib.HmgError, match="no compute path|without a device"):
        hmg.DeviceMatrix(g, 2)                              # host-only grid: no vectors, no compute


def test_single_cell_mesh_tables():
    """One tetrahedron: nothing is shared, everything on the surface is Dirichlet, no interior base node."""
    nodes = np.array([(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)], dtype=np.float64)
    g = hmg.ImplicitFineGrid(None, hmg.Mesh(nodes, np.array([[1, 2, 3, 4]])), 4)
    assert g.table_i32("face_pairs").size == 0
    assert list(g.table_i32("edge_ptr")) == [0] and list(g.table_i32("node_ptr")) == [0]
    assert g.table_i32("dmask")[0] == (1 << 14) - 1 and g.table_i32("dupmask")[0] == 0
    assert g.interior_nodes().size == 0
    assert set(g.table_i32("mult")[:14]) == {1}


@pytest.mark.parametrize("dim,level", [(2, 3), (3, 3)])
def test_vtk_export_of_a_level(oracle, tmp_path, dim, level):
    """`save`: construct_full_grid (src/implicit_fine_grid.jl:41-78) and the .vtu files of export_domain /
    export_unknown (src/examples/homogenized_coefficients.jl:69-87), read back with a minimal parser."""
    from homogenization_jl_amd import vtk
    O = oracle
    m = small_mesh(O, dim, 2, perturb=0.2, seed=3)
    g = host_grid(m, 4)
    nodes, cells = vtk.construct_full_grid(g, level)
    impl = O.ImplicitFineGrid.create(m, 4)
    ref = impl.reference.levels[level - 1]
    nf, ne = ref.nodes.shape[0], m.nelements()
    assert nodes.shape == (nf * ne, dim) and cells.shape == (ref.elements.shape[0] * ne, dim + 1)
    for e in (0, ne - 1):
        X = m.nodes[m.elements[e]]
        want = X[0] + ref.nodes @ (X[1:] - X[0])
        np.testing.assert_allclose(nodes[e * nf:(e + 1) * nf], want, atol=1e-14)
        got = {tuple(sorted(c)) for c in (cells[e * ref.elements.shape[0]:(e + 1) * ref.elements.shape[0]] - e * nf).tolist()}
        assert got == {tuple(sorted(c)) for c in ref.elements.tolist()}
    rng = np.random.default_rng(0)
    x = rng.standard_normal((impl.nf(4), ne))
    path = vtk.export_unknown(g, x, 2, level, str(tmp_path / "ahom_2"))
    back = vtk.read_vtu(path)
    np.testing.assert_array_equal(back["points"][:, :dim], nodes)
    np.testing.assert_array_equal(back["connectivity"].reshape(-1, dim + 1), cells)
    np.testing.assert_array_equal(back["offsets"], (dim + 1) * np.arange(1, cells.shape[0] + 1))
    assert set(back["types"].tolist()) == {5 if dim == 2 else 10}
    np.testing.assert_array_equal(back["point_data"]["v"], x[:nf, :].T.ravel())
    sig = rng.choice([1.0, 9.0], size=(ne, dim))
    back = vtk.read_vtu(vtk.export_domain(g.base, sig, str(tmp_path / "checkerboard")))
    np.testing.assert_array_equal(back["cell_data"]["a"], sig)
    assert back["connectivity"].reshape(-1, dim + 1).tolist() == m.elements.tolist()


@pytest.mark.parametrize("dim,levels", [(3, 5), (2, 6)])
def test_unit_load_table(oracle, dim, levels):
    """int phi over the refined reference cell == assemble_vector(refined_mesh, identity) (src/assembly.jl:121-155)."""
    O = oracle
    g = host_grid(small_mesh(O, dim, 1), levels)
    impl = O.ImplicitFineGrid.create(small_mesh(O, dim, 1), levels)
    for lev in range(1, levels + 1):
        want = O.assemble_vector(impl.reference.levels[lev - 1])
        got = g.table_f64("load", lev)[g.table_i32("hier2slot", lev)]
        np.testing.assert_allclose(got, want, rtol=1e-13)


def test_partitioned_shrink_host_tables(oracle):
    """hmg_grid_shrink on a partitioned grid (host tables only): the ranks' cells are exactly the prefix, every
    rank keeps a prefix of its own cells, and the Dirichlet masks are those of the SHRUNK global mesh."""
    from homogenization_jl_amd import driver, dist as hdist
    O = oracle
    world, dim, width = 4, 2, 8
    origin = (-width / 2.0,) * dim
    base = driver.order_nodes_and_elements_by_magnitude(driver.hypercube(hmg.Tri64, width, origin=origin))
    owner = hdist.block_owner(base, hdist.block_shape(world, dim), width / 2.0, origin)
    ne_keep = driver.find_elements_in_radius(base, 3)
    nn_keep = driver.find_nodes_in_radius(base, 3)
    assert 0 < ne_keep < base.elements.shape[0]
    small = O.Mesh(base.nodes[:nn_keep], base.elements[:ne_keep] - 1)
    cn, ce, cf = O.list_boundary_nodes_edges_faces(small)
    nedge = 3
    want = np.zeros(ne_keep, dtype=np.int64)                     # 2D: edges are the "faces" of the mask layout
    np.bitwise_or.at(want, ce.element, 1 << ce.local_id)
    np.bitwise_or.at(want, cn.element, 1 << (nedge + cn.local_id))
    seen = []
    for rank in range(world):
        g = hdist.PartitionedGrid(None, base, 3, owner, rank, world)
        before = g.local_cells.copy()
        g.shrink(ne_keep, nn_keep)
        after = g.local_cells
        np.testing.assert_array_equal(after, before[before < ne_keep])          # a prefix of the rank's own cells
        assert g.ncells() == after.size
        np.testing.assert_array_equal(g.table_i32("dmask"), want[after])
        seen.append(after)
    np.testing.assert_array_equal(np.sort(np.concatenate(seen)), np.arange(ne_keep))


@pytest.mark.parametrize("dim,shape,origin,transposed", [(3, (6, 6, 6), (-3.0, -3.0, -3.0), True), (2, (9, 9), (-4.5, -4.5), True),
                                                         (3, (4, 6, 2), (-2.0, -3.0, -1.0), False), (2, (5, 3), (1.0, 1.0), False),
                                                         (3, (5, 5, 5), (1.0, 1.0, 1.0), True)])
@pytest.mark.parametrize("ordered", [True, False])
def test_native_checkerboard_synthesis_equals_numpy_statement(dim, shape, origin, transposed, ordered):
    """hmg_checkerboard_mesh / hmg_conductivity_per_element (threaded host C++) against the numpy restatement of
    hypercube / order_nodes_and_elements_by_magnitude / conductivity_per_element
    (src/tet/generate_grid.jl:6-45, src/examples/homogenized_coefficients.jl:21-28, 494-503): identical arrays."""
    import homogenization_jl_amd as hmg
    from homogenization_jl_amd import driver
    tag = hmg.Tet64 if dim == 3 else hmg.Tri64
    ref = driver.hypercube(tag, shape[0], origin=origin) if transposed else driver.box_mesh(tag, shape, origin=origin)
    if ordered:
        ref = driver.order_nodes_and_elements_by_magnitude(ref)
    got = driver.checkerboard_mesh(tag, shape, origin=origin, transposed_lookup=transposed, ordered=ordered)
    np.testing.assert_array_equal(got.nodes, ref.nodes)
    np.testing.assert_array_equal(got.elements, ref.elements)
    rng = np.random.default_rng(1)
    sg = rng.choice([1.0, 9.0], size=tuple(shape) + (dim,))
    off = tuple(1.0 - o for o in origin)
    np.testing.assert_array_equal(driver.conductivity_per_element(got, sg, off, native=True),
                                  driver.conductivity_per_element(ref, sg, off, native=False))


def test_setup_threads_do_not_change_the_tables(monkeypatch):
    """The threaded sorts of the table builders are total orders: interface lists, masks and the level-1 matrix pattern
    are the same whatever the thread count (values of the matrix: summed in a fixed order)."""
    import os, subprocess, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, json, hashlib
sys.path.insert(0, %r)
import numpy as np
import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver
base = driver.checkerboard_mesh(hmg.Tet64, 24, origin=(-12.0,) * 3, transposed_lookup=True)
sg = driver.generate_conductivity(3, 24, 0)
cond = driver.conductivity_per_element(base, sg, (13.0,) * 3)
g = hmg.ImplicitFineGrid(None, base, 3)
g.set_operator(cond, 1.0)
g.coarse_setup()
h = hashlib.sha256()
for name in ("face_pairs", "edge_ptr", "edge_ent", "node_ptr", "node_ent", "node_first", "dmask", "dupmask", "mult",
             "coarse_rowptr", "coarse_colidx"):
    h.update(g.table_i32(name).tobytes())
h.update(g.table_f64("coarse_val").tobytes())
h.update(base.nodes.tobytes()); h.update(base.elements.tobytes())
print(h.hexdigest())
''' % ROOT
    out = []
    for t in ("1", "8"):
        env = dict(os.environ, HMG_SETUP_THREADS=t)
        out.append(subprocess.check_output([sys.executable, "-c", code], env=env).decode().strip().splitlines()[-1])
    assert out[0] == out[1]
