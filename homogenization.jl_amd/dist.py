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

import numpy as np

try:            # torch first: its bundled HIP runtime must be the one the process loads (see _lib.load)
    import torch  # noqa: F401
except ImportError:   # single-GPU use does not need torch
    torch = None

from . import _lib as L
from . import api, driver


class PartitionedGrid(api.ImplicitFineGrid):
    """ImplicitFineGrid over the cells `owner == rank` of a global base mesh."""

    def __init__(self, ctx, base: api.Mesh, levels: int, owner, rank: int, nranks: int):
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
        L.check(self._lib.hmg_grid_create_partition(ctx.h if ctx else None, base.dim, levels, nodes.shape[0],
                                                    nodes.ctypes.data_as(L.p_f64), cells.shape[0],
                                                    cells.ctypes.data_as(L.p_i64), own.ctypes.data_as(L.p_i32),
                                                    rank, nranks, ctypes.byref(h)))
        self.h = h
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
        return int(self._lib.hmg_grid_cut_buffer_doubles(self.h, 0))


class Exchange:
    """Sum-over-ranks callbacks for one grid, on torch.distributed."""

    def __init__(self, ctx: api.Context, grid: PartitionedGrid, group=None):
        import torch
        import torch.distributed as dist
        self.dist, self.group = dist, group
        dev = torch.device("cuda", ctx.device)
        n = max(grid.exchange_doubles(), 1)
        self.buf = torch.zeros(n, dtype=torch.float64, device=dev)
        self.scal = torch.zeros(16, dtype=torch.float64, device=dev)
        lib = L.load()
        L.check(lib.hmg_ctx_set_scalar_bank(ctx.h, ctypes.c_void_p(self.scal.data_ptr())))
        self.calls = 0
        self.seconds = 0.0          # host time spent inside the collectives (blocking backends only, e.g. gloo)
        self.doubles = 0
        import time as _time

        def _sum(tensor, ptr, count):
            off = (ptr - tensor.data_ptr()) // 8
            t0 = _time.perf_counter()
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
                    self._work.wait()          # NCCL: the current stream waits; gloo: the host waits
                    self.seconds += _time.perf_counter() - t0
                    self._work = None
                return 0
            except Exception as e:
                print("hmg exchange (end) failed:", e, flush=True)
                return 1

        self._cb = (L.EXCHANGE_FN(_exchange), L.EXCHANGE_FN(_scalar), L.EXCHANGE_FN(_begin),
                    L.EXCHANGE_END_FN(_end))                                # keep alive
        L.check(lib.hmg_grid_set_exchange(grid.h, self._cb[0], self._cb[1], None,
                                          ctypes.c_void_p(self.buf.data_ptr()), n))
        L.check(lib.hmg_grid_set_exchange_async(grid.h, self._cb[2], self._cb[3]))
        grid._exchange = self

    def set_overlap(self, grid, enabled: bool):
        L.check(L.load().hmg_grid_set_overlap(grid.h, 1 if enabled else 0))


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


def block_owner(base: api.Mesh, blocks, width, origin):
    """owner[c] = index of the width^d block that contains the centre of cell c."""
    c = driver._centers(base) - np.asarray(origin, dtype=np.float64)
    idx = np.minimum((c // width).astype(np.int64), np.array(blocks) - 1)
    owner = np.zeros(c.shape[0], dtype=np.int64)
    for a in range(len(blocks)):
        owner = owner * blocks[a] + idx[:, a]
    return owner.astype(np.int32)


class PartitionedProblem:
    pass


def partitioned_checkerboard(ctx, width: int, levels: int, world: int, rank: int, seed: int = 0, values=(1.0, 9.0),
                             lam: float = 1.0, group=None):
    """Weak-scaling checkerboard: a brick of `world` blocks of width^3 unit cubes, one block per rank."""
    blocks = block_shape(world, 3)
    shape = tuple(width * b for b in blocks)
    origin = tuple(-s / 2.0 for s in shape)
    base = driver.order_nodes_and_elements_by_magnitude(driver.box_mesh(api.Tet64, shape, origin=origin))
    rng = np.random.default_rng(seed)
    sgrid = np.where(rng.random(shape + (3,)) < 0.5, values[0], values[1])
    cond = driver.conductivity_per_element(base, sgrid, tuple(1.0 - o for o in origin))
    owner = block_owner(base, blocks, width, origin)
    grid = PartitionedGrid(ctx, base, levels, owner, rank, world)
    ex = Exchange(ctx, grid, group)
    op = api.L2PlusDivAGrad(grid, lam, cond)
    p = PartitionedProblem()
    p.base, p.cond, p.implicit, p.op, p.exchange = grid.base, cond, grid, op, ex
    p.global_base, p.owner = base, owner
    p.global_shape = "x".join(str(s) for s in shape)
    p.base_level = lambda: api.BaseLevel(grid)
    return p
