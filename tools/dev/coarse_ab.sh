#!/bin/bash
# level-1 solve: plain Jacobi-PCG (coarse_poly=1) against the Chebyshev-preconditioned one, and the tests that police it
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_coarse.py tests/test_gpu_parity.py tests/test_gpu_parity_l6.py -x -q > gpurun_out/t_coarse.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/t_coarse.log
for opt in "coarse_poly=1" "coarse_poly=3" "coarse_poly=4" "coarse_poly=5"; do
  echo "== $opt"; HMG_OPTIONS=$opt timeout -k 10 300 python3 tools/dev/coarse_64.py 2>/dev/null
done
