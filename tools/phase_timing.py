#!/usr/bin/env python3
"""Where does a k_apply workgroup spend its life?  Runs the finest-level operator apply from the dev build
`make -C homogenization.jl_amd/csrc phase-timing` (libhmg_hip_phase_timing.so: thread 0 of every workgroup stamps
wall_clock64(), 100 MHz, at 6 points) and prints the mean duration of each phase.
  python tools/phase_timing.py [--mode ap|res|cg0|cg1] [--width 32] [--levels 6]
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import homogenization_jl_amd as hmg
from homogenization_jl_amd import _lib, driver

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=32)
ap.add_argument("--levels", type=int, default=6)
ap.add_argument("--wg512", type=int, default=1, help="option apply_wg512: 1 = three 512-thread workgroups per CU, 0 = two of 1024 threads")
ap.add_argument("--nres", type=int, default=0, help="resident workgroups on the chip (0: 768 / 512 by --wg512; level 5: 2048)")
ap.add_argument("--mode", default="ap", help="ap | res | cg0 (fused CG step 0: p = r stored, 24 B/DOF) | cg1 (fused CG step 1, 48 B/DOF)")
a = ap.parse_args()
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libhmg_hip_phase_timing.so")
ctx = hmg.Context(0)
ctx.set_option("apply_wg512", a.wg512)
if a.levels < 7:
    ctx.set_option("cell_order", 0)     # (the stamps of k_apply are indexed by cell and written by full-grid launches without a cell list)
L = a.levels
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
        hmg.smoothing_steps(1, g, op, stt, L)      # residual, then CG step 0 (stamps of the last apply launch survive)
    elif a.mode == "cg1":
        hmg.smoothing_steps(2, g, op, stt, L)      # ..., step 1 = full fused pass
    else:
        hmg.apply_ex(1.0, g, x, None, z, constrain=True)
ctx.sync()
raw = g.table_f64("phase_stamps")
st = raw[2 * g.ncells():].reshape(-1, 8)[: g.ncells(), :7]
tick = 10.0  # ns per wall_clock64 tick (100 MHz)
if g.nf(L) > 20000 and ctx.counter("slab2_launches") > 0:
    # k_apply_slab2 (hmg_apply_slab.hip): one persistent workgroup per CU; thread 0 (a loader) and thread 512 (an evaluator)
    # add up their work and their barrier waits over all the (cell, slab) steps of the workgroup
    nwg = min(g.ncells(), 256)
    rec = raw[2 * g.ncells(): 2 * g.ncells() + 16 * nwg].reshape(nwg, 16)
    ld, ev = rec[:, :8], rec[:, 8:]
    life = (ld[:, 5] - ld[:, 0]) * tick / 1e3
    print(f"slab2 kernel, mode {a.mode}: workgroups {nwg}, {g.ncells() / nwg:.1f} cells each; kernel span "
          f"{(max(ld[:, 5].max(), ev[:, 5].max()) - min(ld[:, 0].min(), ev[:, 0].min())) * tick / 1e6:.3f} ms; workgroup lifetime mean {life.mean():.1f} us")
    for name, col, arr in (("loaders: filling windows", 1, ld), ("loaders: waiting at the step barrier", 2, ld),
                           ("evaluators: cell interior", 1, ev), ("evaluators: surface entities", 2, ev),
                           ("evaluators: waiting at the step barrier", 3, ev), ("evaluators: first window", 4, ev),
                           ("evaluators: waiting for the requested words / values", 6, ev), ("evaluators: issuing the previous step's stores", 7, ev)):
        d = arr[:, col] * tick / 1e3
        print(f"  {name:44s} mean {d.mean():9.1f} us  ({100 * d.mean() / life.mean():4.1f} %)   per cell {d.mean() / (g.ncells() / nwg):7.2f} us")
    cy = raw[2 * g.ncells() + 16 * nwg: 2 * g.ncells() + 20 * nwg].reshape(nwg, 4)
    n = cy[:, 3].sum()
    if n > 0:
        print(f"  one interior node evaluation (first evaluator wave, all its nodes), shader cycles: decode + addresses {cy[:, 0].sum() / n:6.0f}   "
              f"15 LDS reads + wait {cy[:, 1].sum() / n:6.0f}   17 FP64 ops {cy[:, 2].sum() / n:6.0f}   ({n:.0f} samples)")
    sys.exit(0)
if g.nf(L) > 20000:
    # k_apply_slab (cells larger than the LDS): thread 0's time per phase summed over the cell's slabs, barrier waits included
    names = ["weight table", "window move + zero fill (+ wait for the previous slab)", "HBM -> LDS (to its barrier)",
             "surface entities", "cell interior"]
    life = (st[:, 6] - st[:, 0]) * tick / 1e3
    print(f"slab kernel, mode {a.mode}: workgroups {st.shape[0]}; kernel span {(st[:, 6].max() - st[:, 0].min()) * tick / 1e6:.3f} ms; "
          f"workgroup lifetime mean {life.mean():.1f} us (p10 {np.percentile(life, 10):.1f}, p90 {np.percentile(life, 90):.1f})")
    for i, n in enumerate(names):
        d = st[:, 1 + i] * tick / 1e3
        print(f"  {n:56s} mean {d.mean():7.2f} us  ({100 * d.mean() / life.mean():4.1f} %)   p10 {np.percentile(d, 10):7.2f}   p90 {np.percentile(d, 90):7.2f}")
    s0 = np.sort(st[:, 0])
    nres = a.nres or 512
    print(f"  per-slot period: {(s0[-1] - s0[0]) * tick / 1e3 / (len(s0) / nres):.2f} us for {nres} resident workgroups")
    sys.exit(0)
d = np.diff(st, axis=1) * tick / 1e3
names = ["W table + column load -> LDS", "table prefetch", "barrier wait", "surface loop", "interior loop",
         "epilogue (reductions)"]
print(f"workgroups {st.shape[0]}; kernel span {(st[:, 6].max() - st[:, 0].min()) * tick / 1e6:.3f} ms")
for i, n in enumerate(names):
    print(f"  {n:32s} mean {d[:, i].mean():7.2f} us   p10 {np.percentile(d[:, i], 10):7.2f}   p90 {np.percentile(d[:, i], 90):7.2f}")
life = (st[:, 6] - st[:, 0]) * tick / 1e3
print(f"  {'workgroup lifetime (stamped)':32s} mean {life.mean():7.2f} us")
# slot turnaround: sort by start time; with 512 resident workgroups, start[i + 512] - end-ish
s0 = np.sort(st[:, 0]); e5 = np.sort(st[:, 6])
nres = a.nres or (768 if a.wg512 else 512)
gap = (s0[nres:] - e5[:-nres]) * tick / 1e3
print(f"  start of workgroup i+{nres} minus end of i (sorted): mean {gap.mean():7.2f} us  (launch + drain overhead per slot)")
print(f"  per-slot period: {(s0[-1] - s0[0]) * tick / 1e3 / (len(s0) / nres):.2f} us")
