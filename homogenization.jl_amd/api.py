"""
Host-side mirror of the reference's level-3 interface (the part of Homogenization.jl that operates on
level vectors), driving libhmg_hip.so through its C ABI.  Names, argument order and meaning follow the
reference (bang-less: `mul!` -> `mul`), so parity tests read like the reference's own tests:

    ImplicitFineGrid(base, levels)            src/implicit_fine_grid.jl:13-18
    ZeroDirichletConstraint                   src/implicit_fine_grid.jl:80-84 (derived by the library)
    L2PlusDivAGrad(..., lam, sigmas)          src/build_local_operators.jl:26-32
    LevelState(ncells, nnodes)                src/multigrid.jl:7-25
    mul / local_residual                      src/apply_local_operators.jl:7-27, 85-133
    apply_constraint / broadcast_interfaces / zero_out_all_but_one / copy_to_base / distribute
                                              src/implicit_fine_grid.jl:94-386
    restrict_to / interpolate_and_sum_to      src/interpolation.jl:52-74
    smoothing_steps / vcycle / BaseLevel      src/multigrid.jl:30-119

Level vectors live in HBM (FP64, entity-major storage order); `DeviceMatrix` is the
`AbstractMatrix` stand-in: `.to_host()` / `.from_host()` speak the reference's Nf x Ne column-major
hierarchical layout.  There is no CPU implementation behind these calls.
"""
from __future__ import annotations

import ctypes
import os
import weakref
from dataclasses import dataclass

import numpy as np

from . import _lib as L

Tri64 = "Tri64"   # element-type tags (src/grid.jl:29-37)
Tet64 = "Tet64"


def _dim_of(tag):
    return {Tri64: 2, Tet64: 3, 2: 2, 3: 3}[tag]


@dataclass
class Mesh:
    """Base mesh (src/grid.jl:19-22). nodes: (Nn, dim) float64; elements: (Ne, dim+1) int64, 1-based,
    each row ascending (src/implicit_fine_grid.jl:14)."""
    nodes: np.ndarray
    elements: np.ndarray

    @property
    def dim(self):
        return self.nodes.shape[1]


class Context:
    """One GPU + one HIP stream. stream: a raw hipStream_t (int) such as
    torch.cuda.current_stream().cuda_stream, or None for a library-owned stream."""

    def __init__(self, device: int = 0, stream=None):
        self._lib = L.load()
        h = ctypes.c_void_p()
        if stream is None:
            L.check(self._lib.hmg_ctx_create(device, None, ctypes.byref(h)))
        else:   # a raw handle; 0 is torch's default stream
            L.check(self._lib.hmg_ctx_create_on_stream(device, ctypes.c_void_p(int(stream)), ctypes.byref(h)))
        self.h = h
        self.device = device
        self._keepalive = []      # caller-side memory the library points into (e.g. a scalar bank tensor)
        self._fin = weakref.finalize(self, self._lib.hmg_ctx_destroy, h)
        for kv in filter(None, os.environ.get("HMG_OPTIONS", "").split(",")):   # dev knobs for A/B runs, e.g. HMG_OPTIONS=fold_x=0
            name, val = kv.split("=")
            self.set_option(name, int(val))

    def stream_handle(self) -> int:
        """The context's hipStream_t as an integer (0 = the null stream): foreign collectives must be issued on it."""
        return int(self._lib.hmg_ctx_stream(self.h) or 0)

    # -- in-library communicator (RCCL), see include/hmg.h ---------------------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        L.check(L.load().hmg_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, nranks: int, rank: int, unique_id: bytes):
        assert len(unique_id) == 128
        L.check(self._lib.hmg_comm_init(self.h, nranks, rank, ctypes.c_char_p(unique_id)))

    def comm_stats(self):
        n = ctypes.c_int64()
        d = ctypes.c_int64()
        L.check(self._lib.hmg_comm_stats(self.h, ctypes.byref(n), ctypes.byref(d)))
        return n.value, d.value

    def comm_sum_host(self, *vals):
        """Sum of a few host doubles over the ranks of the in-library communicator (blocking)."""
        a = np.array(vals, dtype=np.float64)
        L.check(self._lib.hmg_comm_sum_host(self.h, a.ctypes.data_as(L.p_f64), a.size))
        return [float(v) for v in a]

    def sync(self):
        L.check(self._lib.hmg_ctx_sync(self.h))

    def release_memory(self):
        """Hand the pooled blocks of destroyed level vectors back to the device (hmg_ctx_release_memory)."""
        L.check(self._lib.hmg_ctx_release_memory(self.h))

    def set_option(self, name: str, value):
        if isinstance(value, float):
            L.check(self._lib.hmg_ctx_set_option_f64(self.h, name.encode(), value))
        else:
            L.check(self._lib.hmg_ctx_set_option(self.h, name.encode(), int(value)))

    def apply_timing_level(self, level: int):
        """(launches, total_ms, algorithmic_bytes) of the timed operator applies of one level (option "time_apply" = 1 times
        every level)."""
        n = ctypes.c_int64()
        ms = ctypes.c_double()
        by = ctypes.c_double()
        L.check(self._lib.hmg_ctx_apply_timing_level(self.h, int(level), ctypes.byref(n), ctypes.byref(ms), ctypes.byref(by)))
        return n.value, ms.value, by.value

    def counter(self, name: str) -> int:
        """Diagnostic counters of the library ("wave_launches": launches of the one-wave-per-cell level-5 apply)."""
        return int(self._lib.hmg_ctx_counter(self.h, name.encode()))

    def apply_timing(self):
        """(launches, total_ms, algorithmic_bytes) of the operator applies timed since option
        "time_apply" was last set."""
        n = ctypes.c_int64()
        ms = ctypes.c_double()
        by = ctypes.c_double()
        L.check(self._lib.hmg_ctx_apply_timing(self.h, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(by)))
        return n.value, ms.value, by.value

    def close(self):
        if self.h:
            self._fin()           # hmg_ctx_destroy, once
            self.h = None


class ImplicitFineGrid:
    """ImplicitFineGrid(base, levels) + the ZeroDirichletConstraint of the base mesh boundary.
    ctx=None builds host tables only (no compute)."""

    def __init__(self, ctx: Context | None, base: Mesh, levels: int):
        self._lib = L.load()
        self.ctx = ctx
        self.base = base
        self.levels = levels
        nodes = np.ascontiguousarray(base.nodes, dtype=np.float64)
        cells = np.ascontiguousarray(base.elements, dtype=np.int64)
        h = ctypes.c_void_p()
        L.check(self._lib.hmg_grid_create(ctx.h if ctx else None, base.dim, levels, nodes.shape[0],
                                          nodes.ctypes.data_as(L.p_f64), cells.shape[0],
                                          cells.ctypes.data_as(L.p_i64), ctypes.byref(h)))
        self.h = h
        self._fin = weakref.finalize(self, self._lib.hmg_grid_destroy, h)   # (vectors keep the grid alive, the grid the ctx)

    # -- queries -----------------------------------------------------------------------------
    def nlevels(self):
        return self.levels

    def ncells(self):
        return int(self._lib.hmg_grid_ncells(self.h))

    def nnodes_base(self):
        return int(self._lib.hmg_grid_nnodes(self.h))

    def nf(self, level):
        return int(self._lib.hmg_grid_nf(self.h, level))

    def ld(self, level):
        return int(self._lib.hmg_grid_ld(self.h, level))

    def table_i32(self, which, level=1):
        n = ctypes.c_int64()
        L.check(self._lib.hmg_grid_table_i32(self.h, level, which.encode(), None, 0, ctypes.byref(n)))
        out = np.zeros(n.value, dtype=np.int32)
        L.check(self._lib.hmg_grid_table_i32(self.h, level, which.encode(), out.ctypes.data_as(L.p_i32), n.value,
                                             ctypes.byref(n)))
        return out

    def table_f64(self, which, level=1):
        n = ctypes.c_int64()
        L.check(self._lib.hmg_grid_table_f64(self.h, level, which.encode(), None, 0, ctypes.byref(n)))
        out = np.zeros(n.value, dtype=np.float64)
        L.check(self._lib.hmg_grid_table_f64(self.h, level, which.encode(), out.ctypes.data_as(L.p_f64), n.value,
                                             ctypes.byref(n)))
        return out

    def interior_nodes(self):
        """list_interior_nodes(base), 0-based (src/grid.jl:176-202)"""
        return self.table_i32("interior_nodes")

    # -- operator / domain -------------------------------------------------------------------
    def set_operator(self, sigmas, lam):
        s = np.ascontiguousarray(sigmas, dtype=np.float64)
        assert s.shape == (self.base.elements.shape[0], self.base.dim)
        L.check(self._lib.hmg_grid_set_operator(self.h, s.ctypes.data_as(L.p_f64), float(lam)))

    def set_lambda(self, lam):
        L.check(self._lib.hmg_grid_set_lambda(self.h, float(lam)))

    def shrink(self, ncells_prefix, nnodes_prefix):
        L.check(self._lib.hmg_grid_shrink(self.h, ncells_prefix, nnodes_prefix))

    def reserve_spare(self, enable=True):
        """The sixth finest-level vector of the default V-cycle form (include/hmg.h: hmg_grid_reserve_spare): True reserves it now
        and raises if the memory is not there, False releases it (the reference's five vectors per level)."""
        L.check(self._lib.hmg_grid_reserve_spare(self.h, 1 if enable else 0))

    def coarse_setup(self):
        L.check(self._lib.hmg_coarse_setup(self.h))

    def close(self):
        if self.h:
            self._fin()
            self.h = None


class DeviceMatrix:
    """A level vector: the reference's `Nf x Ne` matrix, resident in HBM."""

    def __init__(self, implicit: ImplicitFineGrid, level: int, device_ptr=None):
        self._lib = L.load()
        self.implicit = implicit
        self.level = level
        h = ctypes.c_void_p()
        if device_ptr is None:
            L.check(self._lib.hmg_vec_create(implicit.h, level, ctypes.byref(h)))
        else:
            L.check(self._lib.hmg_vec_wrap(implicit.h, level, ctypes.c_void_p(device_ptr), ctypes.byref(h)))
        self.h = h
        # HBM goes back when the object is collected (or close()d); `implicit` is referenced above, so the grid and its
        # context outlive every vector
        self._fin = weakref.finalize(self, self._lib.hmg_vec_destroy, h)

    @property
    def shape(self):
        return (self.implicit.nf(self.level), self.implicit.ncells())

    def from_host(self, a):
        a = np.asfortranarray(a, dtype=np.float64)
        assert a.shape == self.shape, (a.shape, self.shape)
        L.check(self._lib.hmg_vec_upload(self.h, a.ctypes.data_as(L.p_f64)))
        return self

    def to_host(self):
        out = np.zeros(self.shape, dtype=np.float64, order="F")
        L.check(self._lib.hmg_vec_download(self.h, out.ctypes.data_as(L.p_f64)))
        return out

    def fill(self, v):                       # fill!
        L.check(self._lib.hmg_vec_fill(self.h, float(v)))
        return self

    def rand(self, seed, cell_offset=0):     # rand! (seeded, layout independent)
        L.check(self._lib.hmg_vec_fill_random(self.h, int(seed), int(cell_offset)))
        return self

    def copyto(self, src):                   # copyto!(self, src)
        L.check(self._lib.hmg_vec_copy(self.h, src.h))
        return self

    def similar(self):                       # similar(x)
        return DeviceMatrix(self.implicit, self.level)

    def copy(self):                          # copy(x)
        return self.similar().copyto(self)

    def device_ptr(self):
        return int(self._lib.hmg_vec_device_ptr(self.h) or 0)

    def close(self):
        if self.h:
            self._fin()
            self.h = None


def host_random(shape, seed, cell_offset=0):
    """numpy twin of hmg_vec_fill_random (hash of (seed, cell, hierarchical node id))."""
    nf, ne = shape
    with np.errstate(over="ignore"):
        idx = ((np.arange(ne, dtype=np.uint64) + np.uint64(cell_offset)) << np.uint64(20))[None, :] | \
            np.arange(nf, dtype=np.uint64)[:, None]
        z = np.uint64(seed) + (idx + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    return np.asfortranarray((z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0))


# BLAS-1 generics used on state vectors (src/multigrid.jl:54,64-68)
def dot(x: DeviceMatrix, y: DeviceMatrix) -> float:
    out = ctypes.c_double()
    L.check(L.load().hmg_vec_dot(x.h, y.h, ctypes.byref(out)))
    return out.value


def norm(x: DeviceMatrix) -> float:
    """norm(x) over the raw storage: shared DOFs counted once per copy, like BLAS on the reference's Matrix."""
    return float(np.sqrt(dot(x, x)))


def axpy(alpha, x: DeviceMatrix, y: DeviceMatrix):
    L.check(L.load().hmg_vec_axpy(float(alpha), x.h, y.h))


def xpby(r: DeviceMatrix, beta, p: DeviceMatrix):
    """p .= r .+ beta .* p"""
    L.check(L.load().hmg_vec_xpby(r.h, float(beta), p.h))


def norm_unique(r: DeviceMatrix) -> float:
    """norm(r) after zero_out_all_but_one!(r) -- without destroying r."""
    out = ctypes.c_double()
    L.check(L.load().hmg_vec_norm_unique(r.h, ctypes.byref(out)))
    return out.value


class _Operator:
    """The library keeps ONE operator (per-cell tensors + lambda) per grid; operator objects re-bind themselves
    when they are used after another one (the reference passes the operator to every call instead)."""

    def _bind(self):
        g = self.implicit
        if getattr(g, "_bound_op", None) is not self:
            g.set_operator(self.sigmas, self._lam)
            g._bound_op = self


class L2PlusDivAGrad(_Operator):
    """lam*I - div(sigma grad), sigma constant per coarse cell; carries the Dirichlet constraint
    (held by the grid).  Mutable lam like the reference's struct.  ref: src/build_local_operators.jl:26-32"""

    def __init__(self, implicit: ImplicitFineGrid, lam: float, sigmas):
        self.implicit = implicit
        self._lam = float(lam)
        self.sigmas = np.ascontiguousarray(sigmas, dtype=np.float64)
        self._bind()

    @property
    def lam(self):
        return self._lam

    @lam.setter
    def lam(self, v):
        self._lam = float(v)
        if getattr(self.implicit, "_bound_op", None) is self:
            self.implicit.set_lambda(self._lam)


class SimpleDiffusion(_Operator):
    """-a * Laplacian with a scalar coefficient, no mass term (src/build_local_operators.jl:14-24,
    src/apply_local_operators.jl:40-72): the same kernels with sigma = (a, ..., a) in every cell and lambda = 0."""

    def __init__(self, implicit: ImplicitFineGrid, a: float = 1.0):
        self.implicit = implicit
        self.a = float(a)
        self._lam = 0.0
        self.sigmas = np.full((implicit.base.elements.shape[0], implicit.base.dim), self.a)
        self._bind()


class LevelState:
    """x, b, r, p, Ap of one level (src/multigrid.jl:7-25)."""

    def __init__(self, implicit: ImplicitFineGrid, level: int):
        self.level = level
        self.x = DeviceMatrix(implicit, level)
        self.b = DeviceMatrix(implicit, level)
        self.r = DeviceMatrix(implicit, level)
        self.p = DeviceMatrix(implicit, level)
        self.Ap = DeviceMatrix(implicit, level)

    def handles(self):
        return [self.x.h, self.b.h, self.r.h, self.p.h, self.Ap.h]

    def close(self):
        for v in (self.x, self.b, self.r, self.p, self.Ap):
            v.close()


def mul(alpha, implicit: ImplicitFineGrid, A, x: DeviceMatrix, y: DeviceMatrix):
    """y <- alpha*A*x + y   (mul!(alpha, base, A, x, y); A: L2PlusDivAGrad or SimpleDiffusion)"""
    A._bind()
    L.check(L.load().hmg_apply(implicit.h, x.level, float(alpha), x.h, y.h))


def apply_ex(alpha, implicit: ImplicitFineGrid, x: DeviceMatrix, src, out: DeviceMatrix, constrain=False):
    """out = (src or 0) + alpha*A*x, optionally followed by the Dirichlet constraint (one fused kernel)."""
    L.check(L.load().hmg_apply_ex(implicit.h, x.level, float(alpha), x.h, src.h if src is not None else None, out.h,
                                  1 if constrain else 0))


def local_residual(implicit, A, curr: LevelState, k: int):
    A._bind()
    L.check(L.load().hmg_residual(implicit.h, k, curr.x.h, curr.b.h, curr.r.h))


def apply_constraint(x: DeviceMatrix, level: int, implicit: ImplicitFineGrid):
    L.check(L.load().hmg_constraint(implicit.h, level, x.h))


def broadcast_interfaces(x: DeviceMatrix, implicit: ImplicitFineGrid, level: int):
    L.check(L.load().hmg_interface_sum(implicit.h, level, x.h))


def zero_out_all_but_one(x: DeviceMatrix, implicit: ImplicitFineGrid, level: int):
    L.check(L.load().hmg_zero_duplicates(implicit.h, level, x.h))


def restrict_to(y_coarse: DeviceMatrix, implicit: ImplicitFineGrid, x_fine: DeviceMatrix):
    """restrict_to!(y, P, x) with P = interops[x.level - 1]"""
    L.check(L.load().hmg_restrict(implicit.h, x_fine.level, x_fine.h, y_coarse.h))


def interpolate_and_sum_to(y_fine: DeviceMatrix, implicit: ImplicitFineGrid, x_coarse: DeviceMatrix):
    L.check(L.load().hmg_prolong_add(implicit.h, y_fine.level, x_coarse.h, y_fine.h))


def copy_to_base(implicit: ImplicitFineGrid, v1: DeviceMatrix):
    u = np.zeros(implicit.nnodes_base())
    L.check(L.load().hmg_gather_base(implicit.h, v1.h, u.ctypes.data_as(L.p_f64)))
    return u


def distribute(v1: DeviceMatrix, u, implicit: ImplicitFineGrid):
    u = np.ascontiguousarray(u, dtype=np.float64)
    L.check(L.load().hmg_scatter_base(implicit.h, u.ctypes.data_as(L.p_f64), v1.h))


def rhs_axi_grad_v(b: DeviceMatrix, implicit: ImplicitFineGrid, xi):
    """rhs_a xi grad v!(b, dphis, implicit, sigmas, xi): b[i, el] = dot(dphi_i, -|J| J^-1 (sigma .* xi))
    (src/examples/homogenized_coefficients.jl:449-474)"""
    xi = np.ascontiguousarray(xi, dtype=np.float64)
    L.check(L.load().hmg_rhs_axi_grad(implicit.h, xi.ctypes.data_as(L.p_f64), b.h))


def local_rhs(b: DeviceMatrix, implicit: ImplicitFineGrid):
    """local_rhs!(b, implicit): unit load, b[:, e] = |det J_e| * int phi (src/implicit_fine_grid.jl:391-409)"""
    L.check(L.load().hmg_local_rhs(implicit.h, b.h))


def next_rhs(b: DeviceMatrix, x: DeviceMatrix, implicit: ImplicitFineGrid):
    """next_rhs!: b = lam*|J|*M*x  (src/examples/homogenized_coefficients.jl:695-713)"""
    L.check(L.load().hmg_next_rhs(implicit.h, x.h, b.h))


def _integrate(implicit, mode, v, vprev, nsub, xi):
    out = ctypes.c_double()
    xp = None
    if xi is not None:
        xi = np.ascontiguousarray(xi, dtype=np.float64)
        xp = xi.ctypes.data_as(L.p_f64)
    L.check(L.load().hmg_integrate(implicit.h, mode, v.h if v is not None else None,
                                   vprev.h if vprev is not None else None, int(nsub), xp, ctypes.byref(out)))
    return out.value


def integrate_first_term(v0: DeviceMatrix, implicit: ImplicitFineGrid, nsubset: int, xi, b: DeviceMatrix = None) -> float:
    """sum over the first `nsubset` cells of |J| * v0.(dphi.P + M v0)  (src/examples/homogenized_coefficients.jl:592-632).
    dphi.P is the entry of rhs_a xi grad v! for the same xi: pass that vector as `b` (the driver's right-hand side of
    outer step 0); without it a temporary is filled."""
    if b is None:
        b = v0.similar()
        rhs_axi_grad_v(b, implicit, xi)
        try:
            return _integrate(implicit, 0, v0, b, nsubset, xi)
        finally:
            b.close()
    return _integrate(implicit, 0, v0, b, nsubset, xi)


def integrate_terms(vk: DeviceMatrix, vkm1: DeviceMatrix, implicit: ImplicitFineGrid, nsubset: int) -> float:
    """sum |J| * (vk + vkm1).(M vk)  (src/examples/homogenized_coefficients.jl:634-667)"""
    return _integrate(implicit, 1, vk, vkm1, nsubset, None)


def integrate_area(v: DeviceMatrix, implicit: ImplicitFineGrid, nsubset: int) -> float:
    """1' M 1 under the selected cells  (src/examples/homogenized_coefficients.jl:673-689)"""
    return _integrate(implicit, 2, v, None, nsubset, None)


def smoothing_steps(steps, implicit, ops, curr: LevelState, k: int):
    ops._bind()
    L.check(L.load().hmg_smooth(implicit.h, k, steps, curr.x.h, curr.b.h, curr.r.h, curr.p.h, curr.Ap.h))


class BaseLevel:
    """Coarse-level solver handle: the library's device-resident CG on the assembled level-1 operator, preconditioned by
    four Chebyshev iterates of the Jacobi-scaled operator (27 iterations at config 3), replaces
    `cholesky(assemble_checkerboard(...)[interior, interior])`."""

    def __init__(self, implicit: ImplicitFineGrid):
        self.implicit = implicit
        implicit.coarse_setup()

    def last_iterations(self):
        """Iterations of the last level-1 solve (waits for it).  Raises if that solve did not converge."""
        n = int(L.load().hmg_coarse_last_iterations(self.implicit.h))
        if n < 0:
            raise L.HmgError(L.load().hmg_last_error().decode())
        return n

    def misses(self):
        """Budgeted level-1 solves that ran out of iterations so far (each was reported as an error, or dropped with
        the matrix it belonged to)."""
        return int(L.load().hmg_coarse_misses(self.implicit.h))


def vcycle(implicit: ImplicitFineGrid, base: BaseLevel, ops, levels, k: int, steps: int = 2, steps_coarse: int = 2):
    """vcycle!(implicit, base, ops, levels, k, steps).  steps_coarse = 2 reproduces the reference,
    which does not forward `steps` to the recursive call (src/multigrid.jl:109)."""
    ops[k - 1]._bind()
    L.check(L.load().hmg_vcycle(implicit.h, k, steps, steps_coarse, _state_handles(levels)))


def vcycle_tolerant(implicit: ImplicitFineGrid, base: BaseLevel, ops, levels, k: int, steps: int = 2, steps_coarse: int = 2):
    """vcycle! for driver loops: a budgeted level-1 solve that ran out of its blind iteration budget (the library reports it at the
    first synchronising call behind the V-cycle, include/hmg.h) does not end the run -- the V-cycle it belongs to used an inexact
    coarse-grid correction, which makes it a weaker but valid iterate of the outer iteration; the library has dropped the budget
    and counts the next solve again.  Returns False for such a cycle.  (The reference's CHOLMOD solve cannot miss,
    src/multigrid.jl:84; any other error is raised.)"""
    vcycle(implicit, base, ops, levels, k, steps, steps_coarse)
    try:
        implicit.ctx.sync()
    except L.HmgError as e:
        if "did not reach coarse_rtol" not in str(e):
            raise
        return False
    return True


def _state_handles(levels):
    arr = (ctypes.c_void_p * (5 * len(levels)))()
    for i, st in enumerate(levels):
        if st is None:
            continue
        for q, h in enumerate(st.handles()):
            arr[5 * i + q] = h
    return arr


def tune_placement(implicit: ImplicitFineGrid, ops, levels, k: int, steps: int = 3, trials: int = 8, extra: int = 2):
    """hmg_level_tune_placement: which memory block plays x, b, r, p, Ap of levels[k-1] is chosen by timing level k's share
    of a V-cycle (down + up half, levels[k-2] as the coarse side) per candidate assignment.  Call it on fresh states: the
    vectors of both levels come back zero-filled.  Returns (ms before, ms after)."""
    ops[k - 1]._bind()
    ms = (ctypes.c_double * 2)()
    L.check(L.load().hmg_level_tune_placement(implicit.h, k, int(steps), _state_handles(levels), int(extra), int(trials), ms))
    return ms[0], ms[1]


def vcycle_down(implicit: ImplicitFineGrid, ops, levels, k: int, steps: int = 2):
    """First half of one level of vcycle! (src/multigrid.jl:100-106): smoothing_steps!, local_residual!,
    restrict_to!(next.b, P, curr.r), fill!(next.x, 0).  Only levels[k-1] and levels[k-2] are touched (other entries
    may be None); p and Ap of level k are scratch afterwards."""
    ops[k - 1]._bind()
    L.check(L.load().hmg_vcycle_down(implicit.h, k, steps, _state_handles(levels)))


def vcycle_up(implicit: ImplicitFineGrid, ops, levels, k: int, steps: int = 2):
    """Second half (src/multigrid.jl:112-115): interpolate_and_sum_to!(curr.x, P, next.x), smoothing_steps!."""
    ops[k - 1]._bind()
    L.check(L.load().hmg_vcycle_up(implicit.h, k, steps, _state_handles(levels)))
