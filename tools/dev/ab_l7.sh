#!/bin/bash
# A/B of builds on config 5's per-GPU share, same box, alternating: homogenization.jl_amd/libhmg_hip_<name>.so for every name in
# AB_LIBS (default: base = built from the commit compared against) and "new" = the working build.  Prints ms per V-cycle and the mean finest-level apply of each run.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
B="python3 bench.py --levels 7 --width 16 --sigma-high 100 --steps 10 --warmup 3 --no-cpu-baseline --no-time-to-tolerance --no-level-report --tune-placement 0"
for rep in 1 2; do
  for which in ${AB_LIBS:-base new}; do
    if [ $which != new ]; then export HMG_LIB_PATH=$R/homogenization.jl_amd/libhmg_hip_$which.so HMG_LIB_AB=1; else unset HMG_LIB_PATH HMG_LIB_AB; fi
    timeout -k 10 200 $B $AB_EXTRA 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$which', 'ms_per_step %.2f' % j['ms_per_step'], 'apply avg %.3f ms' % j['roofline']['avg_launch_ms'], 'frac %.3f' % j['roofline']['frac'], 'resid', j['config']['residual_norm_after'])
" || exit 1
  done
done
