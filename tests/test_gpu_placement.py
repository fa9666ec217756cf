"""
hmg_level_tune_placement (include/hmg.h): which memory block plays x, b, r, p, Ap of a LevelState (src/multigrid.jl:7-25)
is chosen by measurement at setup.  Whatever it chooses, nothing numerical may change: the handles stay valid, come back
zero-filled, and a V-cycle on a tuned state is bit-identical to one on an untouched state.
"""
import numpy as np
import pytest

import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver
from homogenization_jl_amd._lib import HmgError

pytestmark = pytest.mark.gpu


def _fill(st, g, L):
    top = st[-1]
    top.x.rand(77)
    hmg.broadcast_interfaces(top.x, g, L)
    hmg.apply_constraint(top.x, L, g)
    hmg.rhs_axi_grad_v(top.b, g, driver.random_unit_vec(3))


@pytest.mark.parametrize("extra,trials", [(0, 5), (2, 7)])
def test_tuned_state_gives_the_same_vcycle(extra, trials):
    ctx = hmg.Context(0)
    try:
        L, w = 4, 6
        base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, w, L, seed=3)
        ref = [hmg.LevelState(g, i + 1) for i in range(L)]
        tun = [hmg.LevelState(g, i + 1) for i in range(L)]
        before, after = hmg.tune_placement(g, [op] * L, tun, L, 3, trials=trials, extra=extra)
        assert before > 0.0 and 0.0 < after <= before
        for st in tun[-2:]:
            for v in (st.x, st.b, st.r, st.p, st.Ap):
                assert not v.to_host().any()                    # zero-filled, like fresh states
        ptrs = {v.device_ptr() for v in (tun[-1].x, tun[-1].b, tun[-1].r, tun[-1].p, tun[-1].Ap)}
        assert len(ptrs) == 5 and 0 not in ptrs
        _fill(ref, g, L)
        _fill(tun, g, L)
        bl = hmg.BaseLevel(g)
        for _ in range(2):
            hmg.vcycle(g, bl, [op] * L, ref, L, 3)
            hmg.vcycle(g, bl, [op] * L, tun, L, 3)
        assert np.array_equal(ref[-1].x.to_host(), tun[-1].x.to_host())
        assert np.array_equal(ref[-1].r.to_host(), tun[-1].r.to_host())
        # a lower level can be tuned as well
        more = [hmg.LevelState(g, i + 1) for i in range(L)]
        b2, a2 = hmg.tune_placement(g, [op] * L, more, 2, 2, trials=3, extra=1)
        assert a2 <= b2
    finally:
        ctx.close()


def test_tuning_refuses_vectors_it_does_not_own():
    ctx = hmg.Context(0)
    try:
        import torch
        L, w = 3, 4
        base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, w, L, seed=3)
        st = [hmg.LevelState(g, i + 1) for i in range(L)]
        buf = torch.zeros(g.ld(L) * g.ncells(), dtype=torch.float64, device="cuda:0")
        st[-1].p = hmg.DeviceMatrix(g, L, device_ptr=buf.data_ptr())     # caller-owned memory: its block cannot change hands
        with pytest.raises(HmgError):
            hmg.tune_placement(g, [op] * L, st, L, 3, trials=2, extra=0)
        two = [hmg.LevelState(g, i + 1) for i in range(L)]
        two[-1].Ap = two[-1].p
        with pytest.raises(HmgError):
            hmg.tune_placement(g, [op] * L, two, L, 3, trials=2, extra=0)
        with pytest.raises(HmgError):
            hmg.tune_placement(g, [op] * L, two, 1, 3, trials=2, extra=0)      # level 1 has no smoother
    finally:
        ctx.close()
