#!/usr/bin/env python3
"""Where does a V-cycle spend its time? Times vcycle(k) for k = L..1 (each includes everything below)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver
L = 6
ctx = hmg.Context(0)
base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, 32, L, seed=0)
st = [hmg.LevelState(g, i + 1) for i in range(L)]
st[-1].x.rand(1); hmg.broadcast_interfaces(st[-1].x, g, L); hmg.apply_constraint(st[-1].x, L, g)
hmg.rhs_axi_grad_v(st[-1].b, g, driver.random_unit_vec(3))
bl = hmg.BaseLevel(g)
hmg.vcycle(g, bl, [op] * L, st, L, 3); ctx.sync()
prev = None
for k in range(L, 0, -1):
    steps = 3 if k == L else 2
    hmg.vcycle(g, bl, [op] * L, st, k, steps); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        hmg.vcycle(g, bl, [op] * L, st, k, steps)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 3 * 1e3
    print(json.dumps({"vcycle_from_level": k, "ms": dt, "pcg_iters": bl.last_iterations()}), flush=True)
