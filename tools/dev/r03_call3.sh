#!/bin/bash
# Round 3, GPU call 3: the whole GPU suite on the sharers-only exchange build, partition rehearsals in both exchange forms.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03c3; rm -rf $O; mkdir -p $O
cd $R
step() {  # step <seconds> <name> <cmd...>
    local t=$1 n=$2; shift 2
    echo "== $n" | tee -a $O/steps.log
    timeout -k 10 $t "$@" > $O/$n.log 2>&1
    local rc=$?
    echo "   rc=$rc" | tee -a $O/steps.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $n: stopping" | tee -a $O/steps.log; exit 1; fi
    return 0
}
step 1100 tests python3 -m pytest tests -q -m gpu -x
tail -3 $O/tests.log
B="--no-cpu-baseline --no-time-to-tolerance --steps 10 --warmup 2"
HMG_FORCE_PARTITIONED=1 step 300 part_nocut python3 bench.py $B
HMG_SYNTHETIC_CUT=planes step 300 part_syn_overlap python3 bench.py $B
HMG_SYNTHETIC_CUT=planes HMG_OVERLAP=0 step 300 part_syn_sync python3 bench.py $B
HMG_REHEARSE_WORLD=8 step 600 part_r8_overlap python3 bench.py $B
HMG_REHEARSE_WORLD=8 HMG_OVERLAP=0 step 600 part_r8_sync python3 bench.py $B
HMG_EXCHANGE=allreduce HMG_REHEARSE_WORLD=8 step 600 part_r8_allreduce_overlap python3 bench.py $B
HMG_EXCHANGE=allreduce HMG_REHEARSE_WORLD=8 HMG_OVERLAP=0 step 600 part_r8_allreduce_sync python3 bench.py $B
cd /tmp; export TMPDIR=/tmp
T="--no-cpu-baseline --no-time-to-tolerance --steps 2 --warmup 1"
HMG_REHEARSE_WORLD=8 step 600 trace_r8 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_r8 -- python3 $R/bench.py $T
for t in trace_r8; do
  F=$(ls -t $O/$t/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$F" ] && cp $F $O/$t.kernel_stats.csv
  rm -rf $O/$t
done
grep -h '"metric"' $O/part_*.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],3), d['config']['residual_norm_after'], d['config']['workload'][:100])
" | tee $O/summary.txt
echo done
