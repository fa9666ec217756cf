#!/usr/bin/env python3
"""The context's vector pool under memory pressure: 24 level-6 vectors (247 GB) are destroyed into the pool, then vectors of
another size are created -- their hipMalloc fails until the pool has been handed back (vec_alloc / DevBuf::alloc retry)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver
ctx = hmg.Context(0)
base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, 32, 6, seed=0)
vs = [hmg.DeviceMatrix(g, 6) for _ in range(24)]
print("24 vectors of 10.3 GB created", flush=True)
for v in vs:
    v.close()
print("destroyed: the pool holds them", flush=True)
base2, cond2, g2, op2 = driver.checkerboard_problem(ctx, hmg.Tet64, 30, 6, seed=0)     # other sizes: tables first, then vectors
ws = [hmg.DeviceMatrix(g2, 6).fill(1.0) for _ in range(12)]
ctx.sync()
print("12 vectors of", ws[0].shape, "created behind it; dot =", hmg.dot(ws[0], ws[1]), flush=True)
