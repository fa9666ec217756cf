#!/usr/bin/env python3
"""Where do the one-wave kernel and the workgroup kernel differ? (dev probe)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import homogenization_jl_amd as hmg
from oracle import oracle as O
from test_gpu_parity import Case
ctx = hmg.Context(0)
c = Case(O, ctx, 3, 3, 5, lam=0.7, perturb=0.2, seed=5)
h2s = c.g.table_i32("hier2slot", 5)
x, y = c.rand(5), c.rand(5)
dx = c.dev(5, x)
for alpha, src in ((-1.0, True), (1.0, False), (-1.0, False), (1.0, True)):
    res = {}
    for wave in (1, 0):
        ctx.set_option("apply_wave", wave)
        out = hmg.DeviceMatrix(c.g, 5)
        hmg.apply_ex(alpha, c.g, dx, c.dev(5, y) if src else None, out, constrain=False)
        res[wave] = out.to_host()
    bad = np.argwhere(res[1] != res[0])
    slots = h2s[bad[:, 0]]
    kinds = np.where(slots < 4, "corner", np.where(slots < 94, "edge", np.where(slots < 514, "face", "interior")))
    u, n = np.unique(kinds, return_counts=True)
    print(f"alpha {alpha} src {src}: {len(bad)} differ:", dict(zip(u, n)), "cells", len(np.unique(bad[:, 1])),
          "max rel", (np.abs(res[1] - res[0]) / np.maximum(np.abs(res[0]), 1e-300)).max())
    if len(bad):
        print("   first slots:", sorted(set(slots.tolist()))[:40])
# which is closer to the oracle on the slots that differ?
ctx.set_option("apply_wave", 1)
want = np.zeros_like(x, order="F")
O.mul(1.0, c.mesh, c.ops[4], x, want)
for wave in (1, 0):
    ctx.set_option("apply_wave", wave)
    out = hmg.DeviceMatrix(c.g, 5)
    hmg.apply_ex(1.0, c.g, dx, None, out, constrain=False)
    got = out.to_host()
    s2h = np.argsort(h2s)
    for name, sl in (("edges0-2", slice(4, 49)), ("edges3-5", slice(49, 94)), ("faces", slice(94, 514)), ("interior", slice(514, 969))):
        rows = s2h[sl]
        print(f"wave={wave} {name}: max abs err vs oracle {np.abs(got[rows] - want[rows]).max():.3e} (|want| max {np.abs(want[rows]).max():.2e})")
ct = c.g.table_f64("ctab", 5).reshape(15, 15, 7)
for cls in (5, 8, 9, 10):
    print("class", cls, "max |term| per tap:", " ".join(f"{np.abs(ct[cls, d]).max():.1e}" for d in range(15)))
