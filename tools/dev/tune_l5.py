#!/usr/bin/env python3
"""Does tuning the placement of level 5's vectors (1.5 GB each) pay as well?  Level 6 tuned first; V-cycles timed before and after level 5's tuning."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver
L = 6
ctx = hmg.Context(0)
base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, 32, L, seed=0)
st = [hmg.LevelState(g, i + 1) for i in range(L)]
print("level 6:", hmg.tune_placement(g, [op] * L, st, L, 3, trials=8, extra=2))


def fill():
    st[-1].x.rand(1); hmg.broadcast_interfaces(st[-1].x, g, L); hmg.apply_constraint(st[-1].x, L, g)
    hmg.rhs_axi_grad_v(st[-1].b, g, driver.random_unit_vec(3))


def bench(n=10):
    hmg.vcycle(g, bl, [op] * L, st, L, 3); ctx.sync()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); hmg.vcycle(g, bl, [op] * L, st, L, 3); ctx.sync(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


fill()
bl = hmg.BaseLevel(g)
print("V-cycle before:", bench())
for v in (st[-1].x, st[-1].b, st[-1].r, st[-1].p, st[-1].Ap):
    pass
# level 5 (its tuning zero-fills levels 5 and 4 only)
print("level 5:", hmg.tune_placement(g, [op] * L, st, L - 1, 2, trials=12, extra=3))
print("V-cycle after :", bench())
print("V-cycle again :", bench())
