R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_l6.py tests/test_gpu_fullsize.py tests/test_gpu_dist.py -x -q 2>&1 | tail -2
bash tools/dev/apply_sequence.sh 2>&1 | tail -3
