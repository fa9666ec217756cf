#!/usr/bin/env python3
"""Wall-clock of the whole driver to a tolerance: device path vs the CPU oracle, identical inputs.
  python tools/driver_bench.py --n 1 --dim 3 --refinements 4 --tolerance 1e-5 [--no-cpu]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1)
ap.add_argument("--dim", type=int, default=3)
ap.add_argument("--refinements", type=int, default=4)
ap.add_argument("--tolerance", type=float, default=1e-5)
ap.add_argument("--no-cpu", action="store_true")
a = ap.parse_args()
width = 2 * (driver.compute_box_radius(0, a.n) + driver.compute_boundary_layer(1.0, a.n))
sgrid = driver.generate_conductivity(a.dim, width, 5)
tag = hmg.Tet64 if a.dim == 3 else hmg.Tri64
ctx = hmg.Context(0)
t0 = time.perf_counter()
sig, hist = driver.checkerboard_homogenization(a.n, tag, refinements=a.refinements, tolerance=a.tolerance, ctx=ctx,
                                               sigma_grid=sgrid, seed=7)
ctx.sync()
t_gpu = time.perf_counter() - t0
out = {"config": f"checkerboard_homogenization({a.n}, {tag}, refinements={a.refinements}, tolerance={a.tolerance})",
       "width": width, "sigma_gpu": sig, "vcycles": len(hist), "wall_s_gpu_incl_setup": t_gpu}
if not a.no_cpu:
    from oracle import oracle as O
    nf = hist and None
    impl_nf = {2: [3, 6, 15, 45, 153, 561], 3: [4, 10, 35, 165, 969, 6545]}[a.dim][a.refinements]
    ne = (2 if a.dim == 2 else 6) * width ** a.dim
    x0 = hmg.host_random((impl_nf, ne), 8)               # the device path's rand(seed + 1)
    t0 = time.perf_counter()
    sig_c, hist_c = O.checkerboard_homogenization(n=a.n, dim=a.dim, refinements=a.refinements, tolerance=a.tolerance,
                                                  sigma_grid=sgrid, x0=x0)
    out.update({"sigma_cpu": sig_c, "vcycles_cpu": len(hist_c), "wall_s_cpu": time.perf_counter() - t0,
                "cpu_threads": O.available_cores(), "abs_diff_sigma": abs(sig - sig_c)})
print(json.dumps(out))
