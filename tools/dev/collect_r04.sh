#!/bin/bash
# GPU box, one call: the round-4 profile set -> gpurun_out/prof_r04 (tools/summarize_profiles.py r04 condenses it), level sequences,
# level split, config 5's per-GPU share, level-7 kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash tools/collect_profiles.sh r04 > gpurun_out/collect_r04.log 2>&1; echo "collect rc=$?"
bash tools/level5_sequence.sh l5_r04 > /dev/null 2>&1; cat gpurun_out/l5_r04/sequence.txt
EXTRA_OPTIONS="apply_wave=0" LAST_B=16 bash tools/level5_sequence.sh l5_r04_wg > /dev/null 2>&1; cat gpurun_out/l5_r04_wg/sequence.txt
bash tools/dev/apply_sequence.sh > gpurun_out/seq_r04.txt 2>&1; tail -2 gpurun_out/seq_r04.txt
timeout -k 10 300 python3 tools/level_bench.py > gpurun_out/level_bench_r04.txt 2>&1; cat gpurun_out/level_bench_r04.txt
timeout -k 10 300 python3 bench.py --levels 7 --width 16 --sigma-high 100 --no-cpu-baseline --no-time-to-tolerance --steps 5 --warmup 2 > gpurun_out/bench_l7_r04.log 2> gpurun_out/bench_l7_r04.err
python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bench_l7_r04.log") if l.startswith("{")][-1])
print("config 5 share:", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
for r in d["roofline"]["levels"]: print("  L", r["level"], round(r["apply_ms_per_vcycle"],3), r["apply_TBps"] and round(r["apply_TBps"],2), round(r["level_share_ms"],3))
PY
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/l7trace -- python3 $R/bench.py --levels 7 --width 16 --sigma-high 100 --no-cpu-baseline --no-time-to-tolerance --steps 2 --warmup 1 --no-level-report > $R/gpurun_out/l7trace.log 2>&1
cp $(ls -t $R/gpurun_out/l7trace/*/*kernel_stats.csv | head -1) $R/gpurun_out/r04_level7_kernel_stats.csv; rm -rf $R/gpurun_out/l7trace
head -8 $R/gpurun_out/r04_level7_kernel_stats.csv | cut -c1-200
