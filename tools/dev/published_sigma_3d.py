import sys, time
sys.path.insert(0,'.')
import numpy as np
from oracle import oracle as O
import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver
for seed in (0,):
    for ref in (1,2):
        n=3; dim=3
        width = 2 * (driver.compute_box_radius(0, n) + driver.compute_boundary_layer(1.0, n))
        sgrid = driver.generate_conductivity(dim, width, seed)
        t0=time.time()
        sigma,hist = O.checkerboard_homogenization(n=n, dim=dim, refinements=ref, tolerance=1e-4, sigma_grid=sgrid, seed=seed)
        print(seed, ref, width, sigma, len(hist), round(time.time()-t0,1), flush=True)
