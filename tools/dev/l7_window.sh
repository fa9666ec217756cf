#!/bin/bash
# level 7 (config 5's per-GPU share): the LDS window of the slab kernel (HMG_SLAB_LDS_KB; default 70 = two workgroups per CU)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for kb in 70 50 100 150 70; do
  HMG_SLAB_LDS_KB=$kb timeout -k 10 300 python3 bench.py --levels 7 --width 16 --sigma-high 100 --no-cpu-baseline --no-time-to-tolerance --steps 4 --warmup 1 --no-level-report --tune-placement 0 > gpurun_out/l7_kb.log 2>/dev/null
  python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/l7_kb.log') if l.startswith('{')][-1])
print('window $kb KB:', round(d['ms_per_step'],2), 'ms, mean L7 apply', round(d['roofline']['avg_launch_ms'],3), 'ms', d['config']['residual_norm_after'])"
done
