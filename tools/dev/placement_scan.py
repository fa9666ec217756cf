#!/usr/bin/env python3
"""Placement of the finest level's five vectors ("Measurement note"): the same binary runs the V-cycle in
a fast and a slow mode depending on where x, b, r, p, Ap lie relative to each other.  This scan moves ONE vector at a time
by a single-bit shift inside its own allocation (the vectors are torch buffers wrapped with hmg_vec_wrap, so nothing in the
library changes) and times three smoothing steps and one V-cycle; it then descends greedily (best shift per vector in turn),
which is what a setup-time tuner would do.
  python tools/dev/placement_scan.py [--width 32] [--levels 6] [--bits 12-21] [--arena]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=32)
ap.add_argument("--levels", type=int, default=6)
ap.add_argument("--bits", default="12-21")
ap.add_argument("--arena", action="store_true", help="carve the five vectors out of ONE allocation")
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
lo, hi = (int(v) for v in args.bits.split("-"))
SLACK = 2 << hi

ctx = hmg.Context(0)
L = args.levels
base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, args.width, L, seed=0)
nbytes = 8 * g.ld(L) * g.ncells()
NAMES = ["x", "b", "r", "p", "Ap"]
if args.arena:
    span = (nbytes + SLACK + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    arena = torch.empty(5 * span, dtype=torch.uint8, device="cuda:0")
    bases = [arena.data_ptr() + i * span for i in range(5)]
else:
    bufs = [torch.empty(nbytes + SLACK, dtype=torch.uint8, device="cuda:0") for _ in range(5)]
    bases = [b.data_ptr() for b in bufs]
torch.cuda.synchronize()
print("bases", [hex(b) for b in bases], "bytes per vector", nbytes, flush=True)
lower = [hmg.LevelState(g, i + 1) for i in range(L - 1)]
bl = hmg.BaseLevel(g)
xi = driver.random_unit_vec(3)


def state(shifts):
    s = hmg.LevelState.__new__(hmg.LevelState)
    s.level = L
    for n, b0, sh in zip(NAMES, bases, shifts):
        setattr(s, n, hmg.DeviceMatrix(g, L, device_ptr=b0 + sh))
    for n in ("r", "p", "Ap"):
        getattr(s, n).fill(0.0)
    s.x.rand(1234)
    hmg.broadcast_interfaces(s.x, g, L)
    hmg.apply_constraint(s.x, L, g)
    hmg.rhs_axi_grad_v(s.b, g, xi)
    return s


def measure(shifts):
    s = state(shifts)
    st = lower + [s]
    hmg.smoothing_steps(3, g, op, s, L)
    ctx.sync()
    ts = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        hmg.smoothing_steps(3, g, op, s, L)
        ctx.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    hmg.vcycle(g, bl, [op] * L, st, L, 3)
    ctx.sync()
    tv = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        hmg.vcycle(g, bl, [op] * L, st, L, 3)
        ctx.sync()
        tv.append((time.perf_counter() - t0) * 1e3)
    s.close()
    return float(np.median(ts)), float(np.median(tv))


def show(tag, shifts, r):
    print(f"{tag:28s} shifts(KB) {[sh >> 10 for sh in shifts]}  smooth3 {r[0]:7.2f} ms   vcycle {r[1]:7.2f} ms", flush=True)


zero = [0] * 5
r0 = measure(zero)
show("baseline", zero, r0)
show("baseline again", zero, measure(zero))
table = {}
for v in range(5):
    for k in range(lo, hi + 1):
        sh = list(zero)
        sh[v] = 1 << k
        r = measure(sh)
        table[(v, k)] = r
        show(f"{NAMES[v]} + 2^{k}", sh, r)
print("\nsmooth3 (ms) by vector (rows) and shift bit (columns %d..%d)" % (lo, hi))
for v in range(5):
    print(f"{NAMES[v]:3s}", " ".join(f"{table[(v, k)][0]:6.2f}" for k in range(lo, hi + 1)))
print("vcycle (ms)")
for v in range(5):
    print(f"{NAMES[v]:3s}", " ".join(f"{table[(v, k)][1]:6.2f}" for k in range(lo, hi + 1)))

# greedy descent on the V-cycle time: what a tuner at setup would do
cur, best = list(zero), r0
for sweep in range(2):
    for v in range(5):
        for k in [None] + list(range(lo, hi + 1)):
            sh = list(cur)
            sh[v] = 0 if k is None else 1 << k
            if sh == cur:
                continue
            r = measure(sh)
            if r[1] < best[1] - 0.15:
                cur, best = sh, r
                show(f"  descent: {NAMES[v]} -> {0 if k is None else 1 << k}", cur, best)
show("descent result", cur, best)
show("descent result again", cur, measure(cur))
show("baseline again", zero, measure(zero))
# all vectors moved together: relative offsets unchanged, so nothing should move
for k in (12, 16, 20):
    sh = [1 << k] * 5
    show(f"all + 2^{k}", sh, measure(sh))
