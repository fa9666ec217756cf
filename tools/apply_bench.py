#!/usr/bin/env python3
"""Micro-benchmark of the finest-level operator apply (used under rocprofv3 for the roofline numbers).
  python tools/apply_bench.py [--width 32] [--levels 6] [--reps 10] [--threads T] [--mode ap|mul|res]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=32)
ap.add_argument("--levels", type=int, default=6)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--threads", type=int, default=None)
ap.add_argument("--mode", default="ap")
ap.add_argument("--unblocked", action="store_true", help="node-per-thread interior sweep (round-1 kernel) for A/B runs")
ap.add_argument("--others", action="store_true", help="also time interface sum / vector kernels / transfer")
a = ap.parse_args()
ctx = hmg.Context(0)
if a.threads is not None:
    ctx.set_option("apply_threads", a.threads)
if a.unblocked:
    ctx.set_option("apply_unblocked", 1)
L = a.levels
base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, a.width, L, seed=0)
x = hmg.DeviceMatrix(g, L).rand(1)
y = hmg.DeviceMatrix(g, L).rand(2)
z = hmg.DeviceMatrix(g, L).rand(3)
ndof = g.nf(L) * g.ncells()


def run(fn, reps, label, bytes_per_dof):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"kernel": label, "ms": dt * 1e3, "GBps_algorithmic": bytes_per_dof * ndof / dt / 1e9,
                      "bytes_per_dof": bytes_per_dof}), flush=True)


if a.mode == "ap":
    run(lambda: hmg.apply_ex(1.0, g, x, None, y, True), a.reps, "apply out=A*x (16 B/DOF)", 16)
elif a.mode == "mul":
    run(lambda: hmg.mul(1.0, g, op, x, y), a.reps, "apply y+=A*x (24 B/DOF)", 24)
else:
    run(lambda: hmg.apply_ex(-1.0, g, x, z, y, True), a.reps, "residual r=b-A*x (24 B/DOF)", 24)
if a.others:
    run(lambda: hmg.broadcast_interfaces(y, g, L), a.reps, "interface sum (16 B x 31% of DOFs)", 16 * 0.313)
    run(lambda: hmg.dot(x, y), a.reps, "dot (16 B/DOF, incl. host sync)", 16)
    run(lambda: hmg.axpy(0.5, x, y), a.reps, "axpy (24 B/DOF)", 24)
    c = hmg.DeviceMatrix(g, L - 1)
    run(lambda: hmg.restrict_to(c, g, x), a.reps, "restrict (8 B/fine DOF)", 8)
    run(lambda: hmg.interpolate_and_sum_to(y, g, c), a.reps, "prolong-add (16 B/fine DOF)", 16)
