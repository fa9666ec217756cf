"""
Multi-rank path on CPU (gloo, world_size 2 and 4): the library's partition analysis (C++ host code:
local meshes, global cut ids, global multiplicities and masks, node ownership for the replicated coarse
solve) driven through the same cut tables the device pack/unpack kernels use.  The per-rank compute is
done with the oracle here (test infrastructure), the exchange is a real torch.distributed all_reduce.
Checked against the serial oracle on the global mesh.
"""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, dim, shape, levels, q, analysis="global"):
    try:
        sys.path.insert(0, ROOT)
        import torch
        import torch.distributed as dist
        import homogenization_jl_amd as hmg
        from homogenization_jl_amd import driver, dist as hdist
        from homogenization_jl_amd import _lib as HL
        from oracle import oracle as O
        O.NTHREADS[0] = 1
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        tag = hmg.Tet64 if dim == 3 else hmg.Tri64
        if shape == "delaunay":       # unstructured mesh, ragged partition (ownership by a hash of the cell id)
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from _meshes import delaunay_mesh
            dm = delaunay_mesh(O, dim, 70, 13)
            base = hmg.Mesh(dm.nodes, dm.elements + 1)
            owner = ((np.arange(dm.nelements()) * 2654435761 >> 7) % world).astype(np.int32)
        else:
            origin = tuple(-s / 2.0 for s in shape)
            base = driver.order_nodes_and_elements_by_magnitude(driver.box_mesh(tag, shape, origin=origin))
            blocks = hdist.block_shape(world, dim)
            width = shape[0] // blocks[0]
            owner = hdist.block_owner(base, blocks, width, origin)
        os.environ["HMG_PARTITION_ANALYSIS"] = analysis          # "global": cut ids every rank agrees on (all-reduce form too)
        g = hdist.PartitionedGrid(None, base, levels, owner, rank, world)
        gm = O.Mesh(base.nodes, base.elements - 1)
        lm = O.Mesh(g.base.nodes, g.base.elements - 1)
        rng = np.random.default_rng(7)
        sig = rng.choice([1.0, 9.0], size=(gm.nelements(), dim))
        lam, L = 0.8, levels
        gi = O.ImplicitFineGrid.create(gm, levels)
        li = O.ImplicitFineGrid(levels, gi.reference, O.interfaces(lm), lm)
        lvl = gi.reference.levels[-1]
        diff, mass = O.build_local_diffusion_operators(lvl), O.mass_matrix(lvl)
        gA = O.L2PlusDivAGrad(diff, mass, None, lam, sig)
        lA = O.L2PlusDivAGrad(diff, mass, None, lam, sig[g.local_cells])
        nf = gi.nf(L)
        x = np.asfortranarray(rng.standard_normal((nf, gm.nelements())))
        O.broadcast_interfaces(x, gi, L)                                  # consistent input
        # serial reference
        y = np.zeros_like(x, order="F"); O.mul(1.0, gm, gA, x, y)
        cons = O.ZeroDirichletConstraint(*O.list_boundary_nodes_edges_faces(gm))
        O.apply_constraint(y, L, cons, gi)
        ysum = y.copy(order="F"); O.broadcast_interfaces(ysum, gi, L)
        # this rank
        xl = np.asfortranarray(x[:, g.local_cells])
        yl = np.zeros_like(xl, order="F"); O.mul(1.0, lm, lA, xl, yl)
        lay = g.table_i32("layout", L)
        nface, nedge = lay[4], lay[3]
        nei, nfi, off_edge, off_face = lay[5], lay[6], lay[8], lay[9]
        s2h = np.argsort(g.table_i32("hier2slot", L))
        dmask = g.table_i32("dmask")
        ent = []                                                          # (bit, slots) of every entity
        for f in range(nface):
            ent.append(off_face + f * nfi + np.arange(nfi))
        for e in range(nedge):
            ent.append(off_edge + e * nei + np.arange(nei))
        for c in range(dim + 1):
            ent.append(np.array([c]))
        for bit, slots in enumerate(ent):                                 # Dirichlet mask from the GLOBAL boundary
            cells = np.flatnonzero((dmask >> bit) & 1)
            yl[np.ix_(s2h[slots], cells)] = 0.0
        np.testing.assert_array_equal(yl == 0.0, y[:, g.local_cells] == 0.0)
        # fused-CG identity with GLOBAL multiplicities:  sum mult*x*y_local == dot(x, S y)
        mult = g.table_i32("mult").reshape(-1, 16)
        w = np.ones_like(xl)
        for bit, slots in enumerate(ent):
            w[s2h[slots], :] = mult[:, bit][None, :]
        t = torch.tensor([float((w * xl * yl).sum()), float(np.vdot(xl, xl))], dtype=torch.float64)
        dist.all_reduce(t)
        assert abs(t[0].item() - float(np.vdot(x, ysum))) <= 1e-10 * abs(float(np.vdot(x, ysum)))
        assert abs(t[1].item() - float(np.vdot(x, x))) <= 1e-12 * float(np.vdot(x, x))
        # local interface sum, then the cut exchange exactly as the library packs / unpacks it
        O.broadcast_interfaces(yl, li, L)
        counts = g.table_i32("cut_counts")
        per = {"faces": nfi, "edges": nei, "nodes": 1}
        global_ids = analysis == "global"
        offs = {"faces": off_face, "edges": off_edge, "nodes": 0}
        yl_local = yl.copy(order="F")                                     # (kept for the second form of the exchange below)
        tot = counts[0] * nfi + counts[1] * nei + counts[2]
        if global_ids:                                                    # (a halo-only analysis has no ids the ranks agree on)
            assert tot <= g.exchange_doubles()
            buf = torch.zeros(max(tot, 1), dtype=torch.float64)
            base_off = {"faces": 0, "edges": counts[0] * nfi, "nodes": counts[0] * nfi + counts[1] * nei}
            views = {}
            for kind in ("faces", "edges", "nodes"):
                gid = g.table_i32("cut_gid_" + kind).astype(np.int64)
                ce = g.table_i32("cut_ent_" + kind).astype(np.int64)
                if per[kind] == 0 or gid.size == 0:
                    continue
                k = np.arange(per[kind])
                slots = offs[kind] + (ce & 7)[:, None] * per[kind] + k[None, :]
                rows, cols = s2h[slots], (ce >> 3)[:, None] + 0 * k[None, :]
                dst = base_off[kind] + gid[:, None] * per[kind] + k[None, :]
                first = np.zeros(gid.size, dtype=bool)
                first[np.unique(gid, return_index=True)[1]] = True
                buf[torch.from_numpy(dst[first].ravel())] = torch.from_numpy(yl[rows[first], cols[first]].ravel())
                views[kind] = (rows, cols, dst)
            dist.all_reduce(buf)
            for kind, (rows, cols, dst) in views.items():
                yl[rows, cols] = buf[torch.from_numpy(dst.ravel())].numpy().reshape(dst.shape)
            err = np.abs(yl - ysum[:, g.local_cells]).max() / np.abs(ysum).max()
            assert err <= 1e-13, err
        # ---- the same exchange among the SHARERS only (hmg_grid_set_exchange_p2p): segments = cut entities grouped by the
        # set of ranks that share them; per segment and peer one message each way, then the members' partials are added in
        # ascending rank order.  Layout, message list and order come from the library's tables, the transport is gloo.
        HL.check(HL.load().hmg_grid_set_exchange_p2p(g.h, 1, HL.P2P_FN(0), HL.P2P_FN(0), None, 0))
        sptr, smem = g.table_i32("seg_ptr"), g.table_i32("seg_members")
        scnt = g.table_i32("seg_counts").reshape(-1, 3).astype(np.int64)
        nseg = scnt.shape[0]
        pers = np.array([nfi, nei, 1], dtype=np.int64)
        size = scnt @ pers
        soff = np.concatenate([[0], np.cumsum(size)])
        assert int(HL.load().hmg_grid_cut_buffer_doubles(g.h, levels)) == soff[-1]
        nmsg = ctypes.c_int64()
        HL.check(HL.load().hmg_grid_exchange_messages(g.h, levels, None, 0, ctypes.byref(nmsg)))
        msgs = np.zeros(nmsg.value, dtype=np.int64)
        HL.check(HL.load().hmg_grid_exchange_messages(g.h, levels, msgs.ctypes.data_as(HL.p_i64), msgs.size, ctypes.byref(nmsg)))
        msgs = msgs.reshape(-1, 4)
        want_msgs = []
        stage_off = 0
        for sq in range(nseg):
            members = smem[sptr[sq]:sptr[sq + 1]]
            assert rank in members and list(members) == sorted(set(members.tolist()))
            for m in members:
                if m != rank:
                    if size[sq] > 0:
                        want_msgs.append((m, soff[sq], size[sq], stage_off))
                    stage_off += size[sq]
        assert [tuple(r) for r in msgs.tolist()] == [tuple(int(v) for v in r) for r in want_msgs]
        assert int(HL.load().hmg_grid_cut_stage_doubles(g.h)) >= stage_off
        buf2 = torch.zeros(max(int(soff[-1]), 1), dtype=torch.float64)
        stage = torch.zeros(max(stage_off, 1), dtype=torch.float64)
        views2 = {}
        kbase = {"faces": np.zeros(nseg, np.int64), "edges": scnt[:, 0] * nfi, "nodes": scnt[:, 0] * nfi + scnt[:, 1] * nei}
        for kind in ("faces", "edges", "nodes"):
            sg = g.table_i32("cut_seg_" + kind).astype(np.int64)
            sx = g.table_i32("cut_sidx_" + kind).astype(np.int64)
            gid = g.table_i32("cut_gid_" + kind).astype(np.int64)
            ce = g.table_i32("cut_ent_" + kind).astype(np.int64)
            if per[kind] == 0 or sg.size == 0:
                continue
            k = np.arange(per[kind])
            slots = offs[kind] + (ce & 7)[:, None] * per[kind] + k[None, :]
            rows, cols = s2h[slots], (ce >> 3)[:, None] + 0 * k[None, :]
            dst = (soff[sg] + kbase[kind][sg] + sx * per[kind])[:, None] + k[None, :]
            first = np.zeros(gid.size, dtype=bool)
            first[np.unique(gid, return_index=True)[1]] = True
            buf2[torch.from_numpy(dst[first].ravel())] = torch.from_numpy(yl_local[rows[first], cols[first]].ravel())
            views2[kind] = (rows, cols, dst)
        ops = []
        for peer, off, cnt, so in msgs:
            ops.append(dist.P2POp(dist.isend, buf2[off:off + cnt].clone(), int(peer)))
            ops.append(dist.P2POp(dist.irecv, stage[so:so + cnt], int(peer)))
        if ops:
            for wk in dist.batch_isend_irecv(ops):
                wk.wait()
        so = 0
        for sq in range(nseg):                                              # (what k_seg_sum does)
            members = smem[sptr[sq]:sptr[sq + 1]]
            acc = None
            for m in members:
                if m == rank:
                    v = buf2[soff[sq]:soff[sq + 1]].clone()
                else:
                    v = stage[so:so + size[sq]]
                    so += size[sq]
                acc = v.clone() if acc is None else acc + v
            buf2[soff[sq]:soff[sq + 1]] = acc
        yl2 = yl_local
        for kind, (rows, cols, dst) in views2.items():
            yl2[rows, cols] = buf2[torch.from_numpy(dst.ravel())].numpy().reshape(dst.shape)
        err = np.abs(yl2 - ysum[:, g.local_cells]).max() / np.abs(ysum).max()
        assert err <= 1e-13, err
        if global_ids:
            # The two forms against each other: the sharers-only form adds the members' partials in ascending rank
            # order; a collective adds them in the order of its algorithm.  Two ranks: one commutative addition per shared DOF
            # -- the same bits.  More sharers: the same to rounding only (bench.py's preflight reports
            # `other_form_bit_identical`, never requires it).
            if world == 2:
                np.testing.assert_array_equal(yl2, yl)
            else:
                assert np.abs(yl2 - yl).max() <= 8 * world * np.finfo(float).eps * np.abs(ysum).max()
        # what it saves: doubles this rank sends vs the all-reduce buffer every rank pushes through the ring
        sent = torch.tensor([float(msgs[:, 2].sum()) if len(msgs) else 0.0, float(tot)], dtype=torch.float64)
        if world == 8 and global_ids:
            assert sent[0].item() * 2 <= sent[1].item(), sent          # octants: at least a factor 2
        HL.check(HL.load().hmg_grid_set_exchange_p2p(g.h, 0, HL.P2P_FN(0), HL.P2P_FN(0), None, 0))
        # first-copy masks give each DOF exactly once across ranks
        dup = g.table_i32("dupmask")
        uniq = np.ones_like(xl)
        for bit, slots in enumerate(ent):
            uniq[np.ix_(s2h[slots], np.flatnonzero((dup >> bit) & 1))] = 0.0
        t = torch.tensor([float((uniq * ysum[:, g.local_cells] ** 2).sum())], dtype=torch.float64)
        dist.all_reduce(t)
        ref = ysum.copy(order="F"); O.zero_out_all_but_one(ref, gi, L)
        assert abs(t.item() - float(np.vdot(ref, ref))) <= 1e-11 * float(np.vdot(ref, ref))
        # node ownership for the replicated coarse solve: every global node owned exactly once
        own = torch.zeros(gm.nnodes(), dtype=torch.float64)
        own[torch.from_numpy(g.local_nodes[g.table_i32("part_owned") == 1])] = 1.0
        dist.all_reduce(own)
        assert bool((own == 1.0).all())
        # global coarse matrix is available on every rank
        g.set_operator(sig, lam)
        g.coarse_setup()
        import scipy.sparse as sp
        interior = O.list_interior_nodes(gm)
        want = O.assemble_checkerboard(gm, sig, lam).tocsr()[interior][:, interior]
        got = sp.csr_matrix((g.table_f64("coarse_val"), g.table_i32("coarse_colidx"), g.table_i32("coarse_rowptr")),
                            shape=want.shape)
        assert abs(got - want).max() <= 1e-13 * abs(want).max()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:                                                # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))


@pytest.mark.parametrize("world,dim,shape,levels", [(2, 3, (4, 2, 2), 4), (4, 3, (4, 4, 2), 3), (2, 2, (6, 3), 4),
                                                      (3, 3, "delaunay", 3), (2, 2, "delaunay", 4),
                                                      (8, 3, (4, 4, 4), 3)])   # octants: centre node shared by 8 ranks
@pytest.mark.parametrize("analysis", ["global", "halo"])
def test_partitioned_interface_sum_matches_serial(world, dim, shape, levels, analysis):
    """analysis = "halo" (the default of hmg_grid_create_partition): every rank analyses its own cells and their one-cell halo
    only; "global": the whole mesh on every rank (needed by the all-reduce form of the exchange, which is checked there too)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, dim, shape, levels, q, analysis)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in sorted(res):
        assert msg == "ok", f"rank {rank}: {msg}"


def test_synthetic_cut_tables_match_the_eight_rank_partition(monkeypatch):
    """Rehearsal partition (hmg_grid_create_partition_rehearsal): one rank holds every cell, cut_owner = the octant of a
    cell.  The cut entities are those of the real 8-rank partition (same global counts, same ids); every copy is local;
    the cut face pairs lead the pair list; edge / node groups and cells are split cut-first as in a real partition."""
    sys.path.insert(0, ROOT)
    import homogenization_jl_amd as hmg
    from homogenization_jl_amd import dist as hdist, driver
    monkeypatch.setenv("HMG_PARTITION_ANALYSIS", "global")           # (the comparison is by global cut id)
    w, L = 4, 3
    base = driver.checkerboard_mesh(hmg.Tet64, (w, w, w), origin=(-w / 2.0,) * 3, transposed_lookup=False)
    octant = hdist.block_owner(base, (2, 2, 2), w // 2, (-w / 2.0,) * 3)
    zero = np.zeros_like(octant)
    syn = hdist.PartitionedGrid(None, base, L, zero, 0, 1, cut_owner=octant)
    cs = syn.table_i32("cut_counts")
    assert syn.local_cells.size == base.elements.shape[0]
    real = [hdist.PartitionedGrid(None, base, L, octant, r, 8) for r in range(8)]
    cr = [g.table_i32("cut_counts") for g in real]
    for k in range(3):
        assert all(c[k] == cs[k] for c in cr)                      # global cut counts agree
        assert cs[3 + k] == sum(c[3 + k] for c in cr)              # the one rank holds every copy
    # a real partition has no cut face with two local copies; the rehearsal has all of them, leading the pair list
    assert all(c[6] == 0 for c in cr) and cs[6] == cs[0]
    fp = syn.table_i32("face_pairs").reshape(-1, 3)
    cut_faces = set(syn.table_i32("cut_ent_faces").tolist())
    lead = {int(a) * 8 + (int(lf) & 15) for a, b, lf in fp[:cs[6]]} | {int(b) * 8 + (int(lf) >> 4) for a, b, lf in fp[:cs[6]]}
    assert lead == cut_faces
    rest = {int(a) * 8 + (int(lf) & 15) for a, b, lf in fp[cs[6]:]}
    assert not (rest & cut_faces)
    # same ids for the same entity: the copies (global cell, local entity) behind every cut id agree with the real ranks'
    for kind in ("faces", "edges", "nodes"):
        want = {}
        for g in real:
            gid, ent = g.table_i32("cut_gid_" + kind), g.table_i32("cut_ent_" + kind)
            for i, e in zip(gid, ent):
                want.setdefault(int(i), set()).add((int(g.local_cells[e >> 3]), int(e & 7)))
        got = {}
        for i, e in zip(syn.table_i32("cut_gid_" + kind), syn.table_i32("cut_ent_" + kind)):
            got.setdefault(int(i), set()).add((int(syn.local_cells[e >> 3]), int(e & 7)))
        assert got == want
    assert cs[9] + cs[10] == base.elements.shape[0] and cs[9] > 0 and cs[10] > 0   # cut cells first, inner cells after
    # without cut_owner a one-rank partition has no cut at all
    plain = hdist.PartitionedGrid(None, base, L, zero, 0, 1)
    assert not plain.table_i32("cut_counts").any() or plain.table_i32("cut_counts")[10] == base.elements.shape[0]


@pytest.mark.parametrize("world,dim,shape", [(2, 3, (4, 2, 2)), (8, 3, (4, 4, 4)), (4, 2, (6, 6)), (3, 3, "delaunay"), (2, 2, "delaunay")])
def test_halo_analysis_equals_global_analysis(world, dim, shape):
    """Partition analysis on a rank's cells plus their one-cell halo (hmg_grid_create_partition's default) against the analysis of
    the whole mesh: local mesh, Dirichlet / first-copy masks, multiplicities, node ownership, the cut copies, the segments (members,
    counts, order), every copy's place inside its segment, the message list and the replicated level-1 matrix are identical; only the
    global cut ids -- which the halo analysis cannot know and the exchange among the sharers does not need -- are gone."""
    sys.path.insert(0, ROOT)
    import ctypes
    import homogenization_jl_amd as hmg
    from homogenization_jl_amd import driver, dist as hdist, _lib as HL
    levels = 3
    tag = hmg.Tet64 if dim == 3 else hmg.Tri64
    if shape == "delaunay":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from _meshes import delaunay_mesh
        from oracle import oracle as O
        dm = delaunay_mesh(O, dim, 70, 13)
        base = hmg.Mesh(dm.nodes, dm.elements + 1)
        owner = ((np.arange(dm.nelements()) * 2654435761 >> 7) % world).astype(np.int32)
    else:
        origin = tuple(-s / 2.0 for s in shape)
        base = driver.order_nodes_and_elements_by_magnitude(driver.box_mesh(tag, shape, origin=origin))
        blocks = hdist.block_shape(world, dim)
        owner = hdist.block_owner(base, blocks, shape[0] // blocks[0], origin)
    rng = np.random.default_rng(5)
    sig = rng.choice([1.0, 9.0], size=(base.elements.shape[0], dim))
    lib = HL.load()
    old = os.environ.get("HMG_PARTITION_ANALYSIS")
    try:
        for rank in range(world):
            grids = {}
            for mode in ("global", "halo"):
                os.environ["HMG_PARTITION_ANALYSIS"] = mode
                g = hdist.PartitionedGrid(None, base, levels, owner, rank, world)
                HL.check(lib.hmg_grid_set_exchange_p2p(g.h, 1, HL.P2P_FN(0), HL.P2P_FN(0), None, 0))
                g.set_operator(sig, 0.7)
                g.coarse_setup()
                grids[mode] = g
            a, b = grids["global"], grids["halo"]
            for name in ("part_cells", "part_nodes", "part_owned", "dmask", "dupmask", "mult", "face_pairs", "edge_ptr", "edge_ent",
                         "node_ptr", "node_ent", "node_first", "seg_ptr", "seg_members", "seg_counts", "cut_ent_faces", "cut_ent_edges",
                         "cut_ent_nodes", "cut_seg_faces", "cut_seg_edges", "cut_seg_nodes", "cut_sidx_faces", "cut_sidx_edges",
                         "cut_sidx_nodes", "coarse_rowptr", "coarse_colidx"):
                np.testing.assert_array_equal(a.table_i32(name), b.table_i32(name), err_msg=f"rank {rank}: {name}")
            np.testing.assert_array_equal(a.table_i32("cut_counts")[3:], b.table_i32("cut_counts")[3:])
            np.testing.assert_array_equal(a.table_f64("coarse_val"), b.table_f64("coarse_val"))
            np.testing.assert_array_equal(a.table_f64("coef"), b.table_f64("coef"))
            for lev in range(1, levels + 1):
                msgs = []
                for g in (a, b):
                    n = ctypes.c_int64()
                    HL.check(lib.hmg_grid_exchange_messages(g.h, lev, None, 0, ctypes.byref(n)))
                    m = np.zeros(n.value, dtype=np.int64)
                    HL.check(lib.hmg_grid_exchange_messages(g.h, lev, m.ctypes.data_as(HL.p_i64), m.size, ctypes.byref(n)))
                    msgs.append(m)
                np.testing.assert_array_equal(msgs[0], msgs[1])
            # the all-reduce form is refused on the halo grid instead of summing apples and oranges
            HL.check(lib.hmg_grid_set_exchange_p2p(b.h, 0, HL.P2P_FN(0), HL.P2P_FN(0), None, 0))
            assert lib.hmg_grid_cut_buffer_doubles(b.h, levels) == -1 and b"halo" in lib.hmg_last_error()
    finally:
        if old is None:
            os.environ.pop("HMG_PARTITION_ANALYSIS", None)
        else:
            os.environ["HMG_PARTITION_ANALYSIS"] = old
