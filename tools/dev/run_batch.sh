R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_dist.py tests/test_gpu_wave.py -x -q > gpurun_out/t6.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/t6.log
timeout -k 10 400 python3 tools/ab_options.py lazy_top=0 lazy_top=1 --rounds 8 2>&1 | tail -4
