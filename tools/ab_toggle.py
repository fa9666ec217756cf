#!/usr/bin/env python3
"""A/B of context options on ONE set of level vectors: the options of hmg_ctx_set_option are run-time switches, so the same
context, grid and vectors run bursts of V-cycles with the option sets alternating -- the physical placement of the 61 GB,
which moves a V-cycle by 1-7 ms between two allocations on one box, is the same on both sides.
  python tools/ab_toggle.py "zero_entry=0" "zero_entry=1" [--rounds 10] [--burst 3] [--levels 6] [--width 32]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver

ap = argparse.ArgumentParser()
ap.add_argument("sets", nargs="+")
ap.add_argument("--rounds", type=int, default=10)
ap.add_argument("--burst", type=int, default=3)
ap.add_argument("--levels", type=int, default=6)
ap.add_argument("--width", type=int, default=32)
args = ap.parse_args()
keys = [sorted(kv.split("=")[0] for kv in o.split(",") if kv) for o in args.sets]
assert all(k == keys[0] for k in keys), "every option set must assign the same keys (the switches persist from one burst to the next)"
ctx = hmg.Context(0)
L = args.levels
base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, args.width, L, seed=0)
st = [hmg.LevelState(g, i + 1) for i in range(L)]
st[-1].x.rand(1234)
hmg.broadcast_interfaces(st[-1].x, g, L)
hmg.apply_constraint(st[-1].x, L, g)
hmg.rhs_axi_grad_v(st[-1].b, g, driver.random_unit_vec(3))
bl = hmg.BaseLevel(g)


def apply(opts):
    for kv in filter(None, opts.split(",")):
        k, v = kv.split("=")
        ctx.set_option(k, int(v))


for o in args.sets:
    apply(o)
    for _ in range(2):
        hmg.vcycle(g, bl, [op] * L, st, L, 3)
ctx.sync()
t = [[] for _ in args.sets]
n = len(args.sets)
for r in range(args.rounds):
    order = list(range(n)) if r % 2 == 0 else list(range(n - 1, -1, -1))
    for s in order:
        apply(args.sets[s])
        hmg.vcycle(g, bl, [op] * L, st, L, 3)          # (first cycle after a switch not timed)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(args.burst):
            hmg.vcycle(g, bl, [op] * L, st, L, 3)
        ctx.sync()
        t[s].append((time.perf_counter() - t0) / args.burst * 1e3)
for s, name in enumerate(args.sets):
    a = np.array(t[s])
    print(f"{name:44s} median {np.median(a):8.3f} ms   mean {a.mean():8.3f}   min {a.min():8.3f}   max {a.max():8.3f}")
for s in range(1, n):
    d = np.array(t[s]) - np.array(t[0])
    print(f"[{args.sets[s]}] - [{args.sets[0]}]: median {np.median(d):+.3f} ms, mean {d.mean():+.3f} +- {d.std() / np.sqrt(len(d)):.3f}")
