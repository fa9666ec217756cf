#!/usr/bin/env python3
"""Kernel launches of one V-cycle started from level 4 (config 3's mesh), by kernel: run under rocprofv3 --kernel-trace, the marker
kernels (k_fill_random, which no V-cycle launches) bracket the counted cycle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver
L = 6
ctx = hmg.Context(0)
base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, 32, L, seed=0)
st = [hmg.LevelState(g, i + 1) for i in range(L)]
st[3].x.rand(1); st[3].b.rand(2)
bl = hmg.BaseLevel(g)
for _ in range(3):
    hmg.vcycle(g, bl, [op] * L, st, 4, 2)
ctx.sync()
mark = hmg.DeviceMatrix(g, 1)
mark.rand(1); ctx.sync()
hmg.vcycle(g, bl, [op] * L, st, 4, 2)
ctx.sync()
mark.rand(2); ctx.sync()
