#!/usr/bin/env python3
"""vcycle(k) for k = 4, 3, 2 with different workgroup sizes of the small-cell apply (option apply_threads)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for nt in [int(a) for a in sys.argv[1:]] or [0, 64, 128, 192, 256]:
    ctx.set_option("apply_threads", nt)
    out = {"apply_threads": nt}
    for k in (5, 4, 3, 2):
        hmg.vcycle(g, bl, [op] * L, st, k, 2); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            hmg.vcycle(g, bl, [op] * L, st, k, 2)
        ctx.sync()
        out[f"vcycle_from_{k}_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
    print(json.dumps(out), flush=True)
