R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/t12.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/t12.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
