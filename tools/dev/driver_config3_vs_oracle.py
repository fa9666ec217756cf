#!/usr/bin/env python3
"""BASELINE config 3 through the whole driver, checkerboard_homogenization(2, Tet64, refinements=5, tolerance=1e-5): the device
run to the end, and the CPU oracle on the host's cores for as many V-cycles as a time budget allows -- same sigma field, same x0
(generated on the device, handed to both).  The per-cycle estimates sigma + dsigma and residual norms of the two are compared.
  python tools/dev/driver_config3_vs_oracle.py [cpu_seconds]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
n, refinements, tol = 2, 5, 1e-5
width = 2 * (driver.compute_box_radius(0, n) + driver.compute_boundary_layer(1.0, n))
sgrid = driver.generate_conductivity(3, width, 0)
ctx = hmg.Context(0)
base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, width, refinements + 1, seed=0)
x0 = hmg.DeviceMatrix(g, refinements + 1).rand(1).to_host()
g.close()
t0 = time.perf_counter()
sig_d, hist_d = driver.checkerboard_homogenization(n, hmg.Tet64, refinements=refinements, tolerance=tol, ctx=ctx,
                                                   sigma_grid=sgrid, x0=x0)
t_gpu = time.perf_counter() - t0
print(json.dumps({"device": {"seconds_incl_upload_of_x0": round(t_gpu, 3), "vcycles": len(hist_d), "sigma": sig_d}}), flush=True)
hist_o = []
t0 = time.perf_counter()


class Stop(Exception):
    pass


def log(h):
    hist_o.append(h)
    print(f"  oracle cycle {len(hist_o)}: {time.perf_counter() - t0:7.1f} s  sigma {h[3]:.12f}  |r| {h[2]:.6e}", flush=True)
    if time.perf_counter() - t0 > budget:
        raise Stop()


try:
    O.checkerboard_homogenization(n=n, dim=3, refinements=refinements, tolerance=tol, sigma_grid=sgrid, x0=x0, log=log)
except Stop:
    pass
t_cpu = time.perf_counter() - t0
m = len(hist_o)
ds = max(abs(a[3] - b[3]) for a, b in zip(hist_o, hist_d))
dr = max(abs(a[2] - b[2]) / a[2] for a, b in zip(hist_o, hist_d))
print(json.dumps({"oracle": {"cycles_done": m, "seconds": round(t_cpu, 1), "seconds_per_cycle": round(t_cpu / max(m, 1), 1),
                             "cores": int(O.available_cores()),
                             "extrapolated_seconds_for_all_cycles": round(t_cpu / max(m, 1) * len(hist_d), 0)},
                  "max_abs_diff_sigma_estimate_over_common_cycles": ds, "max_rel_diff_residual_norm": dr}), flush=True)
