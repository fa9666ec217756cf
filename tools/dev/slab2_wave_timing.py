#!/usr/bin/env python3
"""Per-wave step times of k_apply_slab2 (level 7): how long every one of the sixteen waves of a workgroup works between two step
barriers -- mean, spread over the steps, and the mean over the steps of the SLOWEST wave (what a barrier waits for).
Needs the dev library: make -C homogenization.jl_amd/csrc phase-timing
  python tools/dev/slab2_wave_timing.py [--mode ap|res|cg0|cg1] [--width 16]
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import homogenization_jl_amd as hmg
from homogenization_jl_amd import _lib, driver

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=16)
ap.add_argument("--mode", default="ap")
a = ap.parse_args()
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libhmg_hip_phase_timing.so")
ctx = hmg.Context(0)
L = 7
base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, a.width, L, seed=0)
x = hmg.DeviceMatrix(g, L).rand(1)
y = hmg.DeviceMatrix(g, L).rand(2)
z = hmg.DeviceMatrix(g, L).rand(3)
if a.mode in ("cg0", "cg1"):
    stt = hmg.LevelState(g, L)
    stt.x.rand(1); stt.b.rand(2)
for rep in range(3):
    if a.mode == "res":
        hmg.apply_ex(-1.0, g, x, y, z, constrain=True)
    elif a.mode == "cg0":
        hmg.smoothing_steps(1, g, op, stt, L)
    elif a.mode == "cg1":
        hmg.smoothing_steps(2, g, op, stt, L)
    else:
        hmg.apply_ex(1.0, g, x, None, z, constrain=True)
ctx.sync()
raw = g.table_f64("phase_stamps")
nwg = min(g.ncells(), 256)
off = 2 * g.ncells() + 20 * nwg
w = raw[off: off + 64 * nwg].reshape(nwg, 16, 4)
tick = 0.01  # us
n = w[:, :, 2]
mean = w[:, :, 0] / n * tick
std = np.sqrt(np.maximum(w[:, :, 1] / n - (w[:, :, 0] / n) ** 2, 0.0)) * tick
mx = w[:, :, 3] * tick
print(f"mode {a.mode}, {4} loader waves; steps per workgroup {n[0, 0]:.0f}; step time of a wave = barrier exit -> next arrival, us")
print("wave  role       mean over workgroups of: mean step   std over steps   longest step")
for v in range(16):
    role = "loader" if v < 4 else "evaluator"
    print(f"{v:4d}  {role:9s}  {mean[:, v].mean():8.2f}  {std[:, v].mean():8.2f}  {mx[:, v].mean():8.2f}")
ld, ev = mean[:, :4], mean[:, 4:]
print(f"loaders: mean {ld.mean():.2f}, slowest wave of a workgroup {ld.max(axis=1).mean():.2f}; evaluators: mean {ev.mean():.2f}, slowest wave {ev.max(axis=1).mean():.2f}")
