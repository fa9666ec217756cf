R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_small.py tests/test_gpu_dist.py tests/test_gpu_parity.py -x -q > gpurun_out/t8.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/t8.log
echo "== sequence, lazy_top=2"; bash tools/dev/apply_sequence.sh 2>&1 | tail -3
echo "== sequence, lazy_top=1"; EXTRA_OPTIONS=lazy_top=1 bash tools/dev/apply_sequence.sh 2>&1 | tail -3
timeout -k 10 400 python3 tools/ab_options.py apply_pack=0 apply_pack=1 --rounds 6 2>&1 | tail -4
timeout -k 10 300 python3 tools/level_bench.py 2>/dev/null | grep vcycle_from
