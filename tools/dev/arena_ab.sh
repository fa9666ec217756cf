#!/bin/bash
# Placement of the level vectors under control: all of them carved out of ONE device allocation (HMG_VEC_ARENA_GB), the n-th block shifted by n x stagger.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/arena; rm -rf $O; mkdir -p $O
cd $R
B="--no-cpu-baseline --no-time-to-tolerance --steps 6 --warmup 2"
run() { python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],3))" | tee -a $O/log.txt; }
run plain
for kb in 0 4 36 68 260 1028 4100 16388 66052 263172; do
  HMG_VEC_ARENA_GB=84 HMG_VEC_ARENA_STAGGER_KB=$kb run "arena stagger ${kb} KB"
done
run plain
