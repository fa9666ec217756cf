#!/bin/bash
# One GPU call that validates a build (used at the end of round 3): whole GPU suite, smoke(), the default bench line, a 2-rank rehearsal of bench.py's N > 1 path over gloo, profiles.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/validate_${1:-main}; rm -rf $O; mkdir -p $O
cd $R
step() {  # step <seconds> <name> <cmd...>
    local t=$1 n=$2; shift 2
    echo "== $n" | tee -a $O/steps.log
    timeout -k 10 $t "$@" > $O/$n.log 2> $O/$n.err
    local rc=$?
    echo "   rc=$rc" | tee -a $O/steps.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $n: stopping" | tee -a $O/steps.log; exit 1; fi
    return 0
}
PART=${1:-main}
if [ "$PART" = main ]; then
step 1100 tests_all python3 -m pytest tests -q -m gpu
tail -n 4 $O/tests_all.log
step 300 smoke python3 __graft_entry__.py smoke
tail -n 2 $O/smoke.log
step 600 bench_default python3 bench.py
cut -c1-600 $O/bench_default.log
HMG_SINGLE_DEVICE=1 HMG_DIST_BACKEND=gloo step 600 bench_2rank_gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --width 8 --levels 5 --steps 3 --warmup 1 --no-cpu-baseline --no-time-to-tolerance
cut -c1-400 $O/bench_2rank_gloo.log
HMG_REHEARSE_WORLD=8 step 600 part_r8_overlap python3 bench.py --no-cpu-baseline --no-time-to-tolerance --steps 10 --warmup 2
HMG_SYNTHETIC_CUT=planes step 600 part_syn_overlap python3 bench.py --no-cpu-baseline --no-time-to-tolerance --steps 10 --warmup 2
step 300 bench_l7 python3 bench.py --levels 7 --width 16 --sigma-high 100 --no-cpu-baseline --no-time-to-tolerance --steps 5 --warmup 2
grep -h '"metric"' $O/bench_default.log $O/part_*.log $O/bench_l7.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['ms_per_step'],2), '%.3e' % d['value'], round(d['roofline']['avg_launch_ms'],3), round(d['roofline']['frac'],3), d['config']['residual_norm_after'], d['config']['workload'][:90])
" | tee $O/summary.txt
echo main part done
exit 0
fi
bash $R/tools/collect_profiles.sh r03 > $O/collect.log 2>&1 && bash $R/tools/dev/apply_sequence.sh > $O/sequence.log 2>&1
# kernel tables of the partition rehearsals on this build (profiles/r03_partitioned_kernel_stats_final.csv is made from them)
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
echo traces done
