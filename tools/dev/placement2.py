#!/usr/bin/env python3
"""Is the SECOND 61 GB a process allocates faster than the first (tools/ab_options.py A/A: side B always wins)?
  python tools/dev/placement2.py <n_dummy_sets>   -- n sets of level vectors are created (and kept) before the measured one"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver
n_dummy = int(sys.argv[1])
L = 6
ctx = hmg.Context(0)
base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, 32, L, seed=0)
keep = [[hmg.LevelState(g, i + 1) for i in range(L)] for _ in range(n_dummy)]
st = [hmg.LevelState(g, i + 1) for i in range(L)]
st[-1].x.rand(1234); hmg.broadcast_interfaces(st[-1].x, g, L); hmg.apply_constraint(st[-1].x, L, g)
hmg.rhs_axi_grad_v(st[-1].b, g, driver.random_unit_vec(3))
bl = hmg.BaseLevel(g)
for _ in range(3):
    hmg.vcycle(g, bl, [op] * L, st, L, 3)
ctx.sync()
t0 = time.perf_counter()
for _ in range(10):
    hmg.vcycle(g, bl, [op] * L, st, L, 3)
ctx.sync()
print(f"{n_dummy} set(s) allocated before: {(time.perf_counter() - t0) * 100:.2f} ms per V-cycle, x at {st[-1].x.device_ptr():#x}", flush=True)
