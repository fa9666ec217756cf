#!/bin/bash
# Round 3, GPU call 4: whole GPU suite (sharers-only exchange, epilogue restriction), fold_restrict A/B, bench, trace.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03c4; rm -rf $O; mkdir -p $O
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
step 600 tests_l6 python3 -m pytest tests/test_gpu_parity_l6.py tests/test_gpu_coarse.py -q -m gpu -x
tail -3 $O/tests_l6.log
step 300 ab_restrict01 python3 tools/ab_options.py fold_restrict=0 fold_restrict=1
step 300 ab_restrict10 python3 tools/ab_options.py fold_restrict=1 fold_restrict=0
B="--no-cpu-baseline --no-time-to-tolerance --steps 10 --warmup 2"
step 300 bench_new python3 bench.py $B
HMG_REHEARSE_WORLD=8 step 600 part_r8_overlap python3 bench.py $B
HMG_REHEARSE_WORLD=8 HMG_OVERLAP=0 step 600 part_r8_sync python3 bench.py $B
HMG_SYNTHETIC_CUT=planes step 300 part_syn_overlap python3 bench.py $B
step 1100 tests_all python3 -m pytest tests -q -m gpu
tail -5 $O/tests_all.log
grep -h '"metric"' $O/bench_*.log $O/part_*.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],3), round(d['roofline']['frac'],3), d['config']['residual_norm_after'], d['config']['workload'][:100])
" | tee $O/summary.txt
tail -4 $O/ab_*.log
echo done
