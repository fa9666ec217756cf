import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from oracle import oracle as O
import homogenization_jl_amd as hmg
import _docs_example as D
ctx = hmg.Context(0)
want, xs_o, _ = D.run_oracle(O)
got, xs_d = D.run_device(hmg, ctx, O)
for i in range(0, 100, 3):
    print(i, f"{want[i]:.6e} {abs(want[i]-got[i])/want[i]:.2e} x: {np.abs(xs_d[i]-xs_o[i]).max()/np.abs(xs_o[i]).max():.2e}")
ctx.set_option("coarse_rtol", 1e-15)
got2, xs_d2 = D.run_device(hmg, ctx, O)
for i in range(0, 100, 9):
    print("rtol1e-15", i, f"{abs(want[i]-got2[i])/want[i]:.2e} dev-dev {abs(got[i]-got2[i])/want[i]:.2e}")
