#!/usr/bin/env python3
"""Cost of the (replicated) level-1 solve at BASELINE config 4's size: 64^3 unit cubes, 274 625 nodes."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver
for w, hi in ((32, 9.0), (64, 9.0), (32, 100.0), (64, 100.0)):
    L = 2
    ctx = hmg.Context(0)
    t0 = time.perf_counter()
    base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, w, L, seed=0, values=(1.0, hi))
    st = [hmg.LevelState(g, i + 1) for i in range(L)]
    st[-1].x.rand(1); hmg.broadcast_interfaces(st[-1].x, g, L); hmg.apply_constraint(st[-1].x, L, g)
    hmg.rhs_axi_grad_v(st[-1].b, g, driver.random_unit_vec(3))
    bl = hmg.BaseLevel(g)
    ctx.sync()
    t_setup = time.perf_counter() - t0
    for _ in range(3):
        hmg.vcycle(g, bl, [op] * L, st, L, 2)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        hmg.vcycle(g, bl, [op] * L, st, 1, 2)          # level-1 branch only: interface sum, gather, PCG, scatter
    ctx.sync()
    print(json.dumps({"width": w, "sigma_high": hi, "setup_s": round(t_setup, 3), "coarse_solve_ms": round((time.perf_counter() - t0) * 100, 3),
                      "pcg_iterations": bl.last_iterations()}), flush=True)
    for s in st:
        s.close()
    g.close(); ctx.close()
