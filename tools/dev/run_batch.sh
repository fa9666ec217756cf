R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 300 python3 tools/dev/coarse_64.py 2>/dev/null
timeout -k 10 600 python -m pytest tests/test_gpu_coarse.py tests/test_gpu_parity.py tests/test_gpu_dist.py -x -q 2>&1 | tail -3
