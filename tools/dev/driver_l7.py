#!/usr/bin/env python3
"""checkerboard_homogenization(1, Tet64, refinements=6, tolerance=1e-5) on one GPU: 20^3 cubes, 48 000 cells, L = 7,
2.3e9 fine DOFs (18.4 GB per level-7 vector) -- BASELINE config 5's refinement depth and contrast at the size one GPU holds."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver
ctx = hmg.Context(0)
for hi in (9.0, 100.0):
    tm = {}
    t0 = time.perf_counter()
    sigma, hist = driver.checkerboard_homogenization(1, hmg.Tet64, refinements=6, tolerance=1e-5, ctx=ctx, seed=0,
                                                     values=(1.0, hi), timings=tm)
    print(json.dumps({"sigma_high": hi, "seconds": round(time.perf_counter() - t0, 3), "setup_s": round(tm["setup_s"], 3),
                      "solve_s": round(tm["solve_s"], 3), "vcycles": tm["vcycles"], "sigma": sigma,
                      "residual_norms": [round(h[2], 6) for h in hist][:3] + ["..."] + [round(h[2], 9) for h in hist][-2:]}), flush=True)
