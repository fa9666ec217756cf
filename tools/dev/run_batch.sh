R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for lib in prev new prev new; do
  if [ $lib = prev ]; then export HMG_LIB_PATH=$R/homogenization.jl_amd/ab/libhmg_prev.so HMG_LIB_AB=1; else unset HMG_LIB_PATH HMG_LIB_AB; fi
  timeout -k 10 300 python3 bench.py --levels 7 --width 16 --sigma-high 100 --no-cpu-baseline --no-time-to-tolerance --steps 5 --warmup 2 --no-level-report 2>/dev/null | python3 -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$lib', round(d['ms_per_step'],3), d['config']['residual_norm_after'])"
done
