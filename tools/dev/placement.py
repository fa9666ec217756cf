#!/usr/bin/env python3
"""Does it matter WHERE in HBM the level vectors sit?  One V-cycle timing with N GB of other memory allocated first.
  python tools/dev/placement.py <dummy_GB> [free]     (free: hand the dummy back before the level vectors are created)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver
gb = int(sys.argv[1]); free_it = len(sys.argv) > 2
dummy = [torch.zeros(1 << 27, dtype=torch.float64, device="cuda") for _ in range(gb)] if gb else None   # 1 GiB each
torch.cuda.synchronize()
if free_it:
    dummy = None
    torch.cuda.empty_cache()
L = 6
ctx = hmg.Context(0)
base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, 32, L, seed=0)
st = [hmg.LevelState(g, i + 1) for i in range(L)]
st[-1].x.rand(1234); hmg.broadcast_interfaces(st[-1].x, g, L); hmg.apply_constraint(st[-1].x, L, g)
hmg.rhs_axi_grad_v(st[-1].b, g, driver.random_unit_vec(3))
bl = hmg.BaseLevel(g)
for _ in range(3):
    hmg.vcycle(g, bl, [op] * L, st, L, 3)
ctx.sync()
t0 = time.perf_counter()
for _ in range(10):
    hmg.vcycle(g, bl, [op] * L, st, L, 3)
ctx.sync()
print(f"dummy {gb} GB{' (freed)' if free_it else ''}: {(time.perf_counter() - t0) * 100:.2f} ms per V-cycle, x at {st[-1].x.device_ptr():#x}", flush=True)
