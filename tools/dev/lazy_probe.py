#!/usr/bin/env python3
"""lazy_post on/off: per-level difference of x after one V-cycle (dev probe)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver
levels = 5
ctx = hmg.Context(0)
base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, 4, levels, seed=11)
def run(ncyc):
    st = [hmg.LevelState(g, i + 1) for i in range(levels)]
    st[-1].x.rand(3); st[-1].b.rand(4)
    hmg.broadcast_interfaces(st[-1].x, g, levels)
    hmg.apply_constraint(st[-1].x, levels, g)
    bl = hmg.BaseLevel(g)
    for _ in range(ncyc):
        hmg.vcycle(g, bl, [op] * levels, st, levels, 3)
    out = [s.x.to_host() for s in st]
    for s in st:
        s.close()
    return out
for ncyc in (1, 2):
    ctx.set_option("lazy_post", 1); a = run(ncyc)
    ctx.set_option("lazy_post", 0); b = run(ncyc)
    for l in range(levels):
        d = np.abs(a[l] - b[l])
        print(f"cycles {ncyc} level {l + 1}: x differs in {(a[l] != b[l]).sum()} of {a[l].size}, max abs {d.max():.3e}, max |x| {np.abs(b[l]).max():.3e}")
