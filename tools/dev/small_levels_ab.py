#!/usr/bin/env python3
"""V-cycle from level 4 / 3 / 2 down at config 3's mesh, for several values of option "persistent_waves" (one-wave apply
workgroups per CU that loop over the cells; 0 = one workgroup per cell), alternating in one context."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver
L = 4
vals = [int(v) for v in (sys.argv[1:] or ["0", "32", "16", "64", "8"])]
ctx = hmg.Context(0)
base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, 32, L, seed=0)
st = [hmg.LevelState(g, i + 1) for i in range(L)]
st[-1].x.rand(1); hmg.broadcast_interfaces(st[-1].x, g, L); hmg.apply_constraint(st[-1].x, L, g)
hmg.rhs_axi_grad_v(st[-1].b, g, driver.random_unit_vec(3))
bl = hmg.BaseLevel(g)
for v in vals:
    ctx.set_option("persistent_waves", v)
    hmg.vcycle(g, bl, [op] * L, st, L, 2)
ctx.sync()
res = {(v, k): [] for v in vals for k in (4, 3, 2)}
for rnd in range(6):
    for v in (vals if rnd % 2 == 0 else vals[::-1]):
        ctx.set_option("persistent_waves", v)
        for k in (4, 3, 2):
            hmg.vcycle(g, bl, [op] * L, st, k, 2); ctx.sync()
            t0 = time.perf_counter()
            for _ in range(5):
                hmg.vcycle(g, bl, [op] * L, st, k, 2)
            ctx.sync()
            res[(v, k)].append((time.perf_counter() - t0) / 5 * 1e3)
for v in vals:
    print(f"persistent_waves per CU = {v:3d}:  " + "   ".join(f"from level {k}: {np.median(res[(v, k)]):6.3f} ms" for k in (4, 3, 2)), flush=True)
print("residual norm", hmg.norm_unique(st[-1].r))
