#!/bin/bash
# Round 3, GPU call 2 (call 1 with its bugs fixed): new tests, fold A/B (round-2 build vs in-image staging), partition rehearsals, traces.
# Every step logs under gpurun_out/r03c2/; a step that is killed by its timeout ends the script (no further GPU step).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03c2; rm -rf $O; mkdir -p $O
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
step 900 tests python3 -m pytest tests/test_gpu_coarse.py tests/test_gpu_dist.py tests/test_gpu_parity_l6.py -q -m gpu
tail -3 $O/tests.log
B="--no-cpu-baseline --no-time-to-tolerance --steps 10 --warmup 2"
step 300 bench_new python3 bench.py $B
HMG_LIB_PATH=$R/homogenization.jl_amd/ab/libhmg_r02.so step 300 bench_r02 python3 bench.py $B
step 300 bench_new2 python3 bench.py $B
HMG_LIB_PATH=$R/homogenization.jl_amd/ab/libhmg_r02.so step 300 ab_r02_fold01 python3 tools/ab_options.py fold_prolong=0 fold_prolong=1
HMG_SYNTHETIC_CUT=planes step 300 part_syn_overlap python3 bench.py $B
HMG_SYNTHETIC_CUT=planes HMG_OVERLAP=0 step 300 part_syn_sync python3 bench.py $B
HMG_FORCE_PARTITIONED=1 step 300 part_nocut python3 bench.py $B
HMG_REHEARSE_WORLD=8 step 600 part_r8_overlap python3 bench.py $B
HMG_REHEARSE_WORLD=8 HMG_OVERLAP=0 step 600 part_r8_sync python3 bench.py $B
step 300 mfma_probe $R/tools/dev/mfma_probe
step 600 coarse_64 python3 tools/dev/coarse_64.py
cd /tmp; export TMPDIR=/tmp
T="--no-cpu-baseline --no-time-to-tolerance --steps 2 --warmup 1"
step 600 trace_new rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_new -- python3 $R/bench.py $T
HMG_SYNTHETIC_CUT=planes step 600 trace_syn rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_syn -- python3 $R/bench.py $T
HMG_REHEARSE_WORLD=8 step 600 trace_r8 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_r8 -- python3 $R/bench.py $T
for t in trace_new trace_syn trace_r8; do
  F=$(ls -t $O/$t/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$F" ] && cp $F $O/$t.kernel_stats.csv
  rm -rf $O/$t
done
grep -h '"metric"' $O/bench_*.log $O/part_*.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],3), d['config']['residual_norm_after'], d['config']['workload'][:90])
" | tee $O/summary.txt
tail -4 $O/ab_*.log; cat $O/mfma_probe.log $O/coarse_64.log
echo done
