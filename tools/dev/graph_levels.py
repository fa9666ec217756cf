#!/usr/bin/env python3
"""Does a hipGraph help the launch-bound part of the V-cycle?  Captures vcycle(k) for k = 4 (levels 1-4 + the level-1
PCG, ~600 launches) with torch.cuda.CUDAGraph on the stream the library launches on, and times eager vs replay."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver
L = 6
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ctx = hmg.Context(0, stream=s.cuda_stream)
    base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, 32, L, seed=0)
    st = [hmg.LevelState(g, i + 1) for i in range(L)]
    st[-1].x.rand(1); hmg.broadcast_interfaces(st[-1].x, g, L); hmg.apply_constraint(st[-1].x, L, g)
    hmg.rhs_axi_grad_v(st[-1].b, g, driver.random_unit_vec(3))
    bl = hmg.BaseLevel(g)
    hmg.vcycle(g, bl, [op] * L, st, L, 3); ctx.sync()
    for k in (4, 5):
        hmg.vcycle(g, bl, [op] * L, st, k, 2); bl.last_iterations(); ctx.sync()     # budget known, probe judged
        t0 = time.perf_counter()
        for _ in range(5):
            hmg.vcycle(g, bl, [op] * L, st, k, 2)
        ctx.sync()
        eager = (time.perf_counter() - t0) / 5 * 1e3
        bl.last_iterations()
        ctx.set_option("coarse_probe", 0)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            hmg.vcycle(g, bl, [op] * L, st, k, 2)
        gr.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            gr.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / 5 * 1e3
        ctx.set_option("coarse_probe", 1)
        print(json.dumps({"vcycle_from_level": k, "eager_ms": eager, "hipgraph_replay_ms": graph}), flush=True)
