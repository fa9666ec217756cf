R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/t14.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/t14.log
