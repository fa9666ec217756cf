#!/bin/bash
# config 5's per-GPU share, previous build (homogenization.jl_amd/ab/libhmg_prev.so) against the current one, alternating
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x -k "level7 or config5 or large_levels or refinements_6 or l6_transfer or transfer" > gpurun_out/t_l7.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/t_l7.log
for rep in 1 2 3; do for lib in prev new; do
  if [ $lib = prev ]; then export HMG_LIB_PATH=$R/homogenization.jl_amd/ab/libhmg_prev.so HMG_LIB_AB=1; else unset HMG_LIB_PATH HMG_LIB_AB; fi
  timeout -k 10 300 python3 bench.py --levels 7 --width 16 --sigma-high 100 --no-cpu-baseline --no-time-to-tolerance --steps 5 --warmup 2 --no-level-report --tune-placement 0 > gpurun_out/l7_$lib.log 2>/dev/null
  python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/l7_$lib.log') if l.startswith('{')][-1])
print('$lib', round(d['ms_per_step'],3), '%.4e' % d['value'], round(d['roofline']['avg_launch_ms'],3), d['config']['residual_norm_after'])"
done; done
