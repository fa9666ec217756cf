#!/usr/bin/env python3
"""Same-process A/B of two option sets on BASELINE config 3: two contexts (one per option set), each with its own grid and
level vectors, V-cycles alternated in short bursts so that clock / thermal drift of the box hits both alike (separate
bench.py runs scatter by 2-3 ms on one box, more than most of the effects worth measuring).
  python tools/ab_options.py "apply_wg512=0" "apply_wg512=1" [--rounds 8] [--burst 3] [--levels 6] [--width 32]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver

ap = argparse.ArgumentParser()
ap.add_argument("a")
ap.add_argument("b")
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--burst", type=int, default=3)
ap.add_argument("--levels", type=int, default=6)
ap.add_argument("--width", type=int, default=32)
args = ap.parse_args()


def setup(opts):
    ctx = hmg.Context(0)
    for kv in filter(None, opts.split(",")):
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    L = args.levels
    base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, args.width, L, seed=0)
    st = [hmg.LevelState(g, i + 1) for i in range(L)]
    st[-1].x.rand(1234)
    hmg.broadcast_interfaces(st[-1].x, g, L)
    hmg.apply_constraint(st[-1].x, L, g)
    hmg.rhs_axi_grad_v(st[-1].b, g, driver.random_unit_vec(3))
    bl = hmg.BaseLevel(g)
    for _ in range(2):
        hmg.vcycle(g, bl, [op] * L, st, L, 3)
    ctx.sync()
    return ctx, g, op, st, bl


sides = [setup(args.a), setup(args.b)]
t = [[], []]
for r in range(args.rounds):
    for s in (0, 1) if r % 2 == 0 else (1, 0):
        ctx, g, op, st, bl = sides[s]
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(args.burst):
            hmg.vcycle(g, bl, [op] * args.levels, st, args.levels, 3)
        ctx.sync()
        t[s].append((time.perf_counter() - t0) / args.burst * 1e3)
for s, name in ((0, args.a), (1, args.b)):
    a = np.array(t[s])
    print(f"{name:40s} median {np.median(a):8.3f} ms   mean {a.mean():8.3f}   min {a.min():8.3f}   max {a.max():8.3f}")
d = np.array(t[1]) - np.array(t[0])
print(f"B - A: median {np.median(d):+.3f} ms, mean {d.mean():+.3f} +- {d.std() / np.sqrt(len(d)):.3f}")
print("residual norms:", hmg.norm_unique(sides[0][3][-1].r), hmg.norm_unique(sides[1][3][-1].r))
