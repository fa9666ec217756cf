"""
One process per GPU: partition of the base mesh by coarse-cell ownership and the exchange of the
interface DOFs that lie on partition cuts, over torch.distributed (backend "nccl" is RCCL on ROCm; any
backend that all-reduces device tensors works, e.g. gloo for single-GPU rehearsals).

The library (libhmg_hip.so) does the partition analysis, packs one value per cut DOF into an exchange
buffer after its local interface sum and unpacks the reduced values into every local copy; this module
only owns the communicator-side of that step: an in-place sum over ranks of a slice of a torch tensor,
issued on the same HIP stream as the kernels.  The only other communication is the sum of the CG
scalars and the gather of level-1 nodal values for the replicated coarse solve (same callback).
"""
from __future__ import annotations

import ctypes
import warnings
import weakref

import numpy as np

try:            # torch first: its bundled HIP runtime must be the one the process loads (see _lib.load)
    import torch  # noqa: F401
except ImportError:   # single-GPU use does not need torch
    torch = None

from . import _lib as L
from . import api, driver


class PartitionedGrid(api.ImplicitFineGrid):
    """ImplicitFineGrid over the cells `owner == rank` of a global base mesh."""

    def __init__(self, ctx, base: api.Mesh, levels: int, owner, rank: int, nranks: int, cut_owner=None):
        """cut_owner (rehearsals only, see hmg_grid_create_partition_rehearsal): which entities count as cut is decided by
        cut_owner[] while owner[] still decides which cells are local."""
        self._lib = L.load()
        self.ctx = ctx
        self.global_base = base
        self.levels = levels
        self.rank, self.nranks = rank, nranks
        nodes = np.ascontiguousarray(base.nodes, dtype=np.float64)
        cells = np.ascontiguousarray(base.elements, dtype=np.int64)
        own = np.ascontiguousarray(owner, dtype=np.int32)
        assert own.shape == (cells.shape[0],)
        h = ctypes.c_void_p()
        if cut_owner is None:
            L.check(self._lib.hmg_grid_create_partition(ctx.h if ctx else None, base.dim, levels, nodes.shape[0],
                                                        nodes.ctypes.data_as(L.p_f64), cells.shape[0],
                                                        cells.ctypes.data_as(L.p_i64), own.ctypes.data_as(L.p_i32),
                                                        rank, nranks, ctypes.byref(h)))
        else:
            cown = np.ascontiguousarray(cut_owner, dtype=np.int32)
            assert cown.shape == own.shape
            L.check(self._lib.hmg_grid_create_partition_rehearsal(
                ctx.h if ctx else None, base.dim, levels, nodes.shape[0], nodes.ctypes.data_as(L.p_f64), cells.shape[0],
                cells.ctypes.data_as(L.p_i64), own.ctypes.data_as(L.p_i32), cown.ctypes.data_as(L.p_i32), rank, nranks,
                ctypes.byref(h)))
        self.h = h
        self._fin = weakref.finalize(self, self._lib.hmg_grid_destroy, h)   # (as api.ImplicitFineGrid: vectors keep the grid alive)
        self.local_cells = self.table_i32("part_cells").astype(np.int64)     # global ids, ascending
        self.local_nodes = self.table_i32("part_nodes").astype(np.int64)
        inv = -np.ones(nodes.shape[0], dtype=np.int64)
        inv[self.local_nodes] = np.arange(self.local_nodes.size)
        self.base = api.Mesh(nodes[self.local_nodes], inv[cells[self.local_cells] - 1] + 1)
        self._exchange = None

    def set_operator(self, sigmas_global, lam):
        s = np.ascontiguousarray(sigmas_global, dtype=np.float64)
        assert s.shape == (self.global_base.elements.shape[0], self.global_base.dim)
        L.check(self._lib.hmg_grid_set_operator(self.h, s.ctypes.data_as(L.p_f64), float(lam)))

    def exchange_doubles(self):
        n = int(self._lib.hmg_grid_cut_buffer_doubles(self.h, 0))
        if n < 0:
            raise L.HmgError(self._lib.hmg_last_error().decode())
        return n

    def _refresh_local(self):
        nodes = self.global_base.nodes
        cells = self.global_base.elements
        self.local_cells = self.table_i32("part_cells").astype(np.int64)     # global ids, ascending
        self.local_nodes = self.table_i32("part_nodes").astype(np.int64)
        inv = -np.ones(nodes.shape[0], dtype=np.int64)
        inv[self.local_nodes] = np.arange(self.local_nodes.size)
        self.base = api.Mesh(nodes[self.local_nodes], inv[cells[self.local_cells] - 1] + 1)

    def shrink(self, ncells_prefix_global, nnodes_prefix_global):
        """Domain shrink to a prefix of the GLOBAL cells / nodes: every rank keeps its cells below the prefix
        length (a prefix of its own columns); cut entities, masks and ownership are re-derived by the library."""
        L.check(self._lib.hmg_grid_shrink(self.h, int(ncells_prefix_global), int(nnodes_prefix_global)))
        self._refresh_local()

    def local_count_below(self, ncells_prefix_global):
        """Number of this rank's cells with a global id below the given prefix length."""
        return int(np.searchsorted(self.local_cells, ncells_prefix_global, side="left"))


class Exchange:
    """The sum over ranks behind one partitioned grid.

    backend "rccl" (default whenever the process group is NCCL = RCCL): the library's own communicator
    (hmg_comm_init / hmg_grid_use_comm, include/hmg.h) -- ncclAllReduce on the context's HIP streams, no Python between
    two kernels of a V-cycle; torch.distributed is used once, to hand rank 0's unique id to the other ranks.
    backend "torch": callbacks into torch.distributed (any backend that reduces device tensors, e.g. gloo for
    single-GPU rehearsals); the collectives are issued on the CONTEXT's stream, whatever torch's current stream is."""

    def __init__(self, ctx: api.Context, grid: PartitionedGrid, group=None, backend: str | None = None,
                 comm_ranks: tuple | None = None, sharers: bool | None = None):
        """comm_ranks = (rank, nranks) of the communicator when it differs from the grid's partition (rehearsal of one
        rank's share of a larger partition, context option comm_rehearsal).
        sharers (default: on, HMG_EXCHANGE=allreduce switches it off): the cut DOFs are exchanged among the ranks that
        share them only -- point-to-point messages per segment (include/hmg.h, hmg_grid_set_exchange_p2p) -- instead of one
        all-reduce over the global cut buffer."""
        import os as _os
        if sharers is None:
            sharers = _os.environ.get("HMG_EXCHANGE", "") != "allreduce"
        self.sharers = sharers
        import torch
        import torch.distributed as dist
        self.dist, self.group, self.ctx = dist, group, ctx
        crank, cn = comm_ranks if comm_ranks is not None else (grid.rank, grid.nranks)
        if backend is None:
            backend = "rccl" if dist.is_initialized() and dist.get_backend(group) == "nccl" else "torch"
        self.backend = backend
        self.calls = 0
        self.seconds = 0.0          # host time spent inside the collectives (blocking backends only, e.g. gloo)
        self.doubles = 0
        dev = torch.device("cuda", ctx.device)
        lib = L.load()
        if backend == "rccl":
            if not getattr(ctx, "_comm_ready", False):
                uid = [api.Context.comm_unique_id() if crank == 0 else None]
                if cn > 1:
                    dist.broadcast_object_list(uid, src=0, group=group)
                ctx.comm_init(cn, crank, uid[0])
                ctx._comm_ready = True
            if comm_ranks is not None and (crank, cn) != (grid.rank, grid.nranks):
                ctx.set_option("comm_rehearsal", 1)
            L.check(lib.hmg_grid_use_comm(grid.h))
            grid._exchange = self
            return
        if sharers:      # (the segment layout decides the buffer sizes: switch it on before asking for them)
            L.check(lib.hmg_grid_set_exchange_p2p(grid.h, 1, L.P2P_FN(0), L.P2P_FN(0), None, 0))
        n = max(grid.exchange_doubles(), 1)
        self.buf = torch.zeros(n, dtype=torch.float64, device=dev)
        self.scal = torch.zeros(16, dtype=torch.float64, device=dev)
        # the context outlives every grid: it keeps the memory the library now points into
        ctx._keepalive += [self.buf, self.scal]
        L.check(lib.hmg_ctx_set_scalar_bank(ctx.h, ctypes.c_void_p(self.scal.data_ptr())))
        import time as _time
        sh = ctx.stream_handle()
        self._stream = torch.cuda.ExternalStream(sh, device=dev) if sh else torch.cuda.default_stream(dev)

        def _sum(tensor, ptr, count):
            off = (ptr - tensor.data_ptr()) // 8
            t0 = _time.perf_counter()
            with torch.cuda.stream(self._stream):      # behind the pack kernels, before the unpack kernels
                self.dist.all_reduce(tensor[off:off + count], group=self.group)
            self.seconds += _time.perf_counter() - t0
            self.calls += 1
            self.doubles += count
            return 0

        def _exchange(user, ptr, count):
            try:
                return _sum(self.buf, ptr, count)
            except Exception as e:          # never let an exception cross the C boundary
                print("hmg exchange failed:", e, flush=True)
                return 1

        def _scalar(user, ptr, count):
            try:
                return _sum(self.scal, ptr, count)
            except Exception as e:
                print("hmg scalar sum failed:", e, flush=True)
                return 1

        self._work = None

        def _begin(user, ptr, count):
            try:
                off = (ptr - self.buf.data_ptr()) // 8
                t0 = _time.perf_counter()
                with torch.cuda.stream(self._stream):
                    self._work = self.dist.all_reduce(self.buf[off:off + count], group=self.group, async_op=True)
                self.seconds += _time.perf_counter() - t0
                self.calls += 1
                self.doubles += count
                return 0
            except Exception as e:
                print("hmg exchange (begin) failed:", e, flush=True)
                return 1

        def _end(user):
            try:
                if self._work is not None:
                    t0 = _time.perf_counter()
                    with torch.cuda.stream(self._stream):
                        self._work.wait()          # NCCL: the context's stream waits; gloo: the host waits
                    self.seconds += _time.perf_counter() - t0
                    self._work = None
                return 0
            except Exception as e:
                print("hmg exchange (end) failed:", e, flush=True)
                return 1

        def _p2p(user, ptr, sptr, nmsgs, msgs):
            # one exchange among the sharers: every message = my partial segment to the peer, the peer's into the staging
            # area.  gloo moves host tensors only: staged through the host (rehearsals; RCCL takes the in-library path).
            try:
                t0 = _time.perf_counter()
                m = np.ctypeslib.as_array(msgs, shape=(int(nmsgs) * 4,)).reshape(-1, 4).copy() if nmsgs else np.zeros((0, 4), np.int64)
                with torch.cuda.stream(self._stream):
                    off0 = (ptr - self.buf.data_ptr()) // 8
                    soff0 = (sptr - self.stage.data_ptr()) // 8 if nmsgs else 0
                    host = self.dist.get_backend(self.group) != "nccl"
                    ops, recvs = [], []
                    for peer, off, cnt, soff in m:
                        src = self.buf[off0 + off:off0 + off + cnt]
                        dst = self.stage[soff0 + soff:soff0 + soff + cnt]
                        if host:
                            src = src.cpu()
                            tmp = torch.empty(int(cnt), dtype=torch.float64)
                            recvs.append((dst, tmp))
                            dst = tmp
                        ops.append(self.dist.P2POp(self.dist.isend, src, int(peer), self.group))
                        ops.append(self.dist.P2POp(self.dist.irecv, dst, int(peer), self.group))
                    if ops:
                        for w in self.dist.batch_isend_irecv(ops):
                            w.wait()
                    for dst, tmp in recvs:
                        dst.copy_(tmp)
                self.seconds += _time.perf_counter() - t0
                self.calls += 1
                self.doubles += int(m[:, 2].sum()) if len(m) else 0
                return 0
            except Exception as e:
                print("hmg p2p exchange failed:", e, flush=True)
                return 1

        self._cb = (L.EXCHANGE_FN(_exchange), L.EXCHANGE_FN(_scalar), L.EXCHANGE_FN(_begin),
                    L.EXCHANGE_END_FN(_end), L.P2P_FN(_p2p))
        ctx._keepalive.append(self._cb)                                     # the grid calls them as long as it lives
        L.check(lib.hmg_grid_set_exchange(grid.h, self._cb[0], self._cb[1], None,
                                          ctypes.c_void_p(self.buf.data_ptr()), n))
        L.check(lib.hmg_grid_set_exchange_async(grid.h, self._cb[2], self._cb[3]))
        if sharers:
            # (the blocking transport serves as `begin` as well: by the time it returns the messages have landed, and the
            #  `end` callback above finds no work to wait for)
            ns = max(int(lib.hmg_grid_cut_stage_doubles(grid.h)), 1)
            self.stage = torch.zeros(ns, dtype=torch.float64, device=dev)
            ctx._keepalive.append(self.stage)
            L.check(lib.hmg_grid_set_exchange_p2p(grid.h, 1, self._cb[4], self._cb[4],
                                                  ctypes.c_void_p(self.stage.data_ptr()), ns))
        grid._exchange = self

    def set_overlap(self, grid, enabled: bool):
        L.check(L.load().hmg_grid_set_overlap(grid.h, 1 if enabled else 0))

    def stats(self):
        """(collectives, doubles moved) so far."""
        if self.backend == "rccl":
            return self.ctx.comm_stats()
        return self.calls, self.doubles

    def rank_sum(self, *vals):
        """Sum of a few host doubles over the ranks (the driver's per-cycle integrals)."""
        if self.backend == "rccl":
            return self.ctx.comm_sum_host(*vals)
        import torch
        t = torch.tensor(vals, dtype=torch.float64, device=torch.device("cuda", self.ctx.device))
        self.dist.all_reduce(t, group=self.group)
        return [float(v) for v in t.tolist()]


def block_shape(world: int, dim: int = 3):
    """Number of per-rank blocks along each axis: powers of two are spread over the axes (1,2,4,8 ->
    1x1x1, 2x1x1, 2x2x1, 2x2x2 = octants), anything else becomes slabs."""
    shape = [1] * dim
    w = world
    a = 0
    while w % 2 == 0 and w > 1:
        shape[a % dim] *= 2
        w //= 2
        a += 1
    shape[0] *= w
    return tuple(shape)


def block_owner(base: api.Mesh, blocks, width, origin, native: bool = True):
    """owner[c] = index of the width^d block that contains the centre of cell c (native: the library's threaded host
    code, hmg_block_owner; False: the numpy statement of the same rule)."""
    if native:
        dim = base.dim
        nodes = np.ascontiguousarray(base.nodes, dtype=np.float64)
        cells = np.ascontiguousarray(base.elements, dtype=np.int64)
        bl = np.ascontiguousarray(blocks, dtype=np.int64)
        org = np.ascontiguousarray(origin, dtype=np.float64)
        owner = np.empty(cells.shape[0], dtype=np.int32)
        L.check(L.load().hmg_block_owner(dim, nodes.shape[0], nodes.ctypes.data_as(L.p_f64), cells.shape[0],
                                         cells.ctypes.data_as(L.p_i64), bl.ctypes.data_as(L.p_i64), float(width),
                                         org.ctypes.data_as(L.p_f64), owner.ctypes.data_as(L.p_i32)))
        return owner
    c = driver._centers(base) - np.asarray(origin, dtype=np.float64)
    idx = np.minimum((c // width).astype(np.int64), np.array(blocks) - 1)
    owner = np.zeros(c.shape[0], dtype=np.int64)
    for a in range(len(blocks)):
        owner = owner * blocks[a] + idx[:, a]
    return owner.astype(np.int32)


class PartitionedProblem:
    pass


def partitioned_checkerboard(ctx, width: int, levels: int, world: int, rank: int, seed: int = 0, values=(1.0, 9.0),
                             lam: float = 1.0, group=None, backend=None, synthetic_cut: bool = False,
                             rehearse_world: int | None = None):
    """Weak-scaling checkerboard: a brick of `world` blocks of width^3 unit cubes, one block per rank.

    Two single-GPU rehearsals of the partitioned code path (world = 1, a 1-rank communicator):
      synthetic_cut      the one block is cut by its three mid-planes (cut_owner = octant of a cell): every cut entity
                         has all its copies on this rank, the sum over ranks is the identity, the results equal the
                         unpartitioned grid's bit for bit -- cut-first cell lists, pack / unpack, events and the
                         all-reduce of an 8-rank partition run at full size;
      rehearse_world=N   this process holds rank 0's block of the N-rank brick (config 4: N = 8, 64^3 cubes) and its cut
                         tables, replicated level-1 system included; the neighbours' contributions are missing from the
                         sums, so the numbers mean nothing, but per-rank work, message sizes and stream choreography are
                         those of the real run."""
    layout = rehearse_world if rehearse_world else world
    blocks = block_shape(layout, 3)
    shape = tuple(width * b for b in blocks)
    origin = tuple(-s / 2.0 for s in shape)
    base = driver.checkerboard_mesh(api.Tet64, shape, origin=origin, transposed_lookup=False)
    rng = np.random.default_rng(seed)
    sgrid = np.where(rng.random(shape + (3,)) < 0.5, values[0], values[1])
    cond = driver.conductivity_per_element(base, sgrid, tuple(1.0 - o for o in origin))
    owner = block_owner(base, blocks, width, origin)
    cut_owner = None
    if synthetic_cut:
        assert layout == 1 and width % 2 == 0, "the synthetic cut splits one block of even width at its mid-planes"
        cut_owner = block_owner(base, (2, 2, 2), width // 2, origin)
    if rehearse_world:
        grid = PartitionedGrid(ctx, base, levels, owner, rank, rehearse_world)
        ex = Exchange(ctx, grid, group, backend, comm_ranks=(rank, world))
    else:
        grid = PartitionedGrid(ctx, base, levels, owner, rank, world, cut_owner=cut_owner)
        ex = Exchange(ctx, grid, group, backend)
    op = api.L2PlusDivAGrad(grid, lam, cond)
    p = PartitionedProblem()
    p.base, p.cond, p.implicit, p.op, p.exchange = grid.base, cond, grid, op, ex
    p.global_base, p.owner = base, owner
    p.global_shape = "x".join(str(s) for s in shape)
    p.base_level = lambda: api.BaseLevel(grid)
    return p


def partitioned_checkerboard_homogenization(ctx, n: int, eltype, world: int, rank: int, refinements: int = 2,
                                            smoothing_steps: int = 3, tolerance: float = 1e-4, xi=None, seed: int = 0,
                                            values=(1.0, 9.0), sigma_grid=None, x0=None, max_cycles: int = 1000,
                                            group=None, log=None, backend=None, stats: dict | None = None):
    """driver.checkerboard_homogenization over `world` ranks (one GPU each): the base mesh is split into blocks about
    the origin (halves / quadrants / octants for 2, 4, 8 ranks), so that the centred sub-domains the outer loop
    shrinks to stay balanced.  Every rank runs the same host loop; the per-cycle integrals are local sums
    added over the ranks, everything else goes through the partitioned V-cycle.  Returns (sigma, history) like the
    single-GPU driver, identical on every rank.  `stats` (a dict) receives "inexact_vcycles": V-cycles whose budgeted level-1 solve
    missed coarse_rtol (every rank sees the same replicated solve, hence the same count)."""
    dim = api._dim_of(eltype)
    # the blocks are halves per axis about the origin: 1, 2, 4 (and 8 in 3D) ranks.  Checked on every rank before any
    # collective, so that an unsupported size fails everywhere instead of hanging the ranks that do own cells
    if world not in ((1, 2, 4) if dim == 2 else (1, 2, 4, 8)):
        raise ValueError(f"partitioned_checkerboard_homogenization: {world} ranks are not supported in {dim}D "
                         "(blocks are halves per axis: 1, 2, 4" + (", 8" if dim == 3 else "") + ")")
    xi = driver.random_unit_vec(dim) if xi is None else np.asarray(xi, dtype=np.float64)
    lam, sigma = 1.0, 0.0
    box_radius = driver.compute_box_radius(0, n)
    boundary_layer = driver.compute_boundary_layer(lam, n)
    total_radius = box_radius + boundary_layer
    width = 2 * total_radius
    origin = (-float(total_radius),) * dim
    base = driver.checkerboard_mesh(eltype, width, origin=origin, transposed_lookup=True)
    if sigma_grid is None:
        sigma_grid = driver.generate_conductivity(dim, width, seed, values)
    cond = driver.conductivity_per_element(base, sigma_grid, (total_radius + 1.0,) * dim)
    blocks = block_shape(world, dim)
    owner = block_owner(base, blocks, width / 2.0, origin)          # halves per axis; axes with one block clamp to 0
    total_grids = refinements + 1
    grid = PartitionedGrid(ctx, base, total_grids, owner, rank, world)
    ex = Exchange(ctx, grid, group, backend)
    op = api.L2PlusDivAGrad(grid, lam, cond)
    states = [api.LevelState(grid, i + 1) for i in range(total_grids)]
    top = states[-1]
    nf = grid.nf(total_grids)
    if x0 is None:
        x0 = api.host_random((nf, base.elements.shape[0]), seed + 1)       # hashed by GLOBAL cell id
    top.x.from_host(np.asfortranarray(x0[:, grid.local_cells]))
    api.broadcast_interfaces(top.x, grid, total_grids)
    api.apply_constraint(top.x, total_grids, grid)
    api.rhs_axi_grad_v(top.b, grid, xi)
    v_prev = None                                        # allocated at the first domain shrink
    rank_sum = ex.rank_sum
    cur = base
    history = []
    inexact = 0
    for k in range(n + 1):
        base_level = api.BaseLevel(grid)
        dsig, dsig_prev = 0.0, 0.0
        for i in range(1, max_cycles + 1):
            if not api.vcycle_tolerant(grid, base_level, [op] * total_grids, states, total_grids, smoothing_steps):
                inexact += 1                             # (as in driver.checkerboard_homogenization: counted and said)
                if rank == 0:
                    warnings.warn(f"partitioned_checkerboard_homogenization: V-cycle {i} of outer step {k} used an inexact "
                                  f"level-1 solve ({inexact} so far)")
            nint = grid.local_count_below(driver.find_elements_in_radius(cur, box_radius))
            area = api.integrate_area(top.x, grid, nint)
            if k == 0:
                integral = api.integrate_first_term(top.x, grid, nint, xi, b=top.b)
            else:
                integral = api.integrate_terms(top.x, v_prev, grid, nint)
            area, integral = rank_sum(area, integral)
            dsig = 2.0 ** k * integral / area
            rnorm = api.norm_unique(top.r)
            history.append((k, i, rnorm, sigma + dsig, abs(dsig - dsig_prev)))
            if log and rank == 0:
                log(history[-1])
            if abs(dsig - dsig_prev) < tolerance:
                break
            dsig_prev = dsig
        sigma += dsig
        lam /= 2
        box_radius = driver.compute_box_radius(k + 1, n)
        boundary_layer = driver.compute_boundary_layer(lam, n)
        if box_radius + boundary_layer > total_radius:
            break
        total_radius = box_radius + boundary_layer
        nn_keep = driver.find_nodes_in_radius(cur, total_radius)
        ne_keep = driver.find_elements_in_radius(cur, total_radius)
        cur = api.Mesh(cur.nodes[:nn_keep], np.ascontiguousarray(cur.elements[:ne_keep]))
        grid.shrink(ne_keep, nn_keep)
        api.apply_constraint(top.x, total_grids, grid)
        if v_prev is None:
            v_prev = api.DeviceMatrix(top.x.implicit, total_grids)
        v_prev.copyto(top.x)
        op.lam = lam
        api.next_rhs(top.b, top.x, grid)
    if stats is not None:
        stats["inexact_vcycles"] = inexact
    return sigma, history
