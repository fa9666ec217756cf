#!/bin/bash
# Does carving the level vectors out of ONE device allocation tame the placement scatter?  Alternating runs, one box.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/arena; rm -rf $O; mkdir -p $O
cd $R
B="--no-cpu-baseline --no-time-to-tolerance --steps 6 --warmup 2"
for i in 1 2 3; do
  python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain ', round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],3))" | tee -a $O/log.txt
  HMG_VEC_ARENA_GB=80 python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('arena ', round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],3))" | tee -a $O/log.txt
done
