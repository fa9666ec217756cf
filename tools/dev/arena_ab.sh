#!/bin/bash
# Which address bits of the spacing between the level-6 vectors decide between the fast and the slow mode of the V-cycle?
# The five level-6 vectors at n x (2^34 + S), S = one bit at a time.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/arena; rm -rf $O; mkdir -p $O
cd $R
B="--no-cpu-baseline --no-time-to-tolerance --steps 5 --warmup 2"
run() { python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],3))" | tee -a $O/log.txt; }
run plain
export HMG_VEC_ARENA_GB=112 HMG_VEC_ARENA_SPACING_LOG2=34
HMG_VEC_ARENA_STAGGER_KB=0 run "S=0"
for j in 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 21; do
  HMG_VEC_ARENA_STAGGER_KB=$((1 << j)) run "S=2^$((j + 10))"
done
