#!/usr/bin/env python3
"""Which exact saving is not exact? Each option of EXACT_OPTIONS switched off alone, x and r after 3 V-cycles vs all on."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver
OPTS = ("lean_post", "lazy_post", "lazy_top", "lazy_dead", "fold_x", "swap_rp", "fold_prolong", "prolong_in_image", "fold_faces", "fold_restrict",
        "zero_entry", "cell_order")
dim, n, levels = 3, 4, int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx = hmg.Context(0)
base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, n, levels, seed=11)
def run():
    st = [hmg.LevelState(g, i + 1) for i in range(levels)]
    st[-1].x.rand(3); st[-1].b.rand(4)
    hmg.broadcast_interfaces(st[-1].x, g, levels)
    hmg.apply_constraint(st[-1].x, levels, g)
    bl = hmg.BaseLevel(g)
    for _ in range(3):
        hmg.vcycle(g, bl, [op] * levels, st, levels, 3)
    out = (st[-1].x.to_host(), st[-1].r.to_host())
    for s in st:
        s.close()
    return out
for wave in (1, 0):
    ctx.set_option("apply_wave", wave)
    ref = run()
    for o in OPTS:
        ctx.set_option(o, 0)
        got = run()
        ctx.set_option(o, 1)
        print(f"apply_wave={wave} {o:18s} off: x differs in {(got[0] != ref[0]).sum():8d}, r in {(got[1] != ref[1]).sum():8d}", flush=True)
