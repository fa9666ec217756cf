#!/bin/bash
# GPU box, one call: the round-5 profile set -> gpurun_out/prof_r05 (tools/summarize_profiles.py r05 condenses it), the driver's own
# bench command, level sequences (levels 5, 6, 7), level split, config 5's per-GPU share + its kernel stats, the oracle's time for
# config 3 at full size
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
t0=$(date +%s); timeout -k 10 900 python3 bench.py > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err; echo "default bench rc=$? seconds=$(( $(date +%s) - t0 ))"
t0=$(date +%s); timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_bench_driver_shape.json 2> gpurun_out/r5_bench_driver_shape.err; echo "driver-shape bench rc=$? seconds=$(( $(date +%s) - t0 ))"
python3 - <<PY
import json
for f in ("gpurun_out/r5_bench_default.json", "gpurun_out/r5_bench_driver_shape.json"):
    d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], d["config"]["residual_norm_after"], d["config"]["placement"].get("ms_per_step_tuned"), d.get("time_to_tolerance", {}).get("config3", {}).get("seconds"))
PY
bash tools/collect_profiles.sh r05 > gpurun_out/collect_r05.log 2>&1; echo "collect rc=$?"
bash tools/level5_sequence.sh l5_r05 > /dev/null 2>&1; cat gpurun_out/l5_r05/sequence.txt
bash tools/dev/apply_sequence.sh > gpurun_out/seq_r05.txt 2>&1; tail -3 gpurun_out/seq_r05.txt
bash tools/level7_sequence.sh l7_r05 > /dev/null 2>&1; cat gpurun_out/l7_r05/sequence.txt
timeout -k 10 300 python3 tools/level_bench.py > gpurun_out/level_bench_r05.txt 2>&1; cat gpurun_out/level_bench_r05.txt
timeout -k 10 300 python3 bench.py --levels 7 --width 16 --sigma-high 100 --no-cpu-baseline --no-time-to-tolerance --steps 10 --warmup 3 > gpurun_out/r5_bench_l7.json 2> gpurun_out/r5_bench_l7.err
python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r5_bench_l7.json") if l.startswith("{")][-1])
print("config 5 share:", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["config"]["placement"].get("ms_per_step_tuned"))
for r in d["roofline"]["levels"]: print("  L", r["level"], round(r["apply_ms_per_vcycle"],3), r["apply_TBps"] and round(r["apply_TBps"],2), round(r["level_share_ms"],3))
PY
HMG_REHEARSE_WORLD=8 timeout -k 10 300 python3 bench.py --width 32 --steps 10 --warmup 3 --no-cpu-baseline --no-time-to-tolerance --no-level-report --tune-placement 0 > gpurun_out/r5_bench_rank0of8.json 2> gpurun_out/r5_bench_rank0of8.err; grep -o '"ms_per_step": [0-9.]*' gpurun_out/r5_bench_rank0of8.json | head -1
(cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/l7trace -- python3 $R/bench.py --levels 7 --width 16 --sigma-high 100 --no-cpu-baseline --no-time-to-tolerance --steps 2 --warmup 1 --no-level-report --tune-placement 0 > $R/gpurun_out/l7trace.log 2>&1
cp $(ls -t $R/gpurun_out/l7trace/*/*kernel_stats.csv | head -1) $R/gpurun_out/r05_level7_kernel_stats.csv; rm -rf $R/gpurun_out/l7trace)
head -6 gpurun_out/r05_level7_kernel_stats.csv | cut -c1-200
timeout -k 10 600 python3 -m pytest tests/test_gpu_fullsize.py -q -s -k "vcycle_matches_oracle_full_size" 2>&1 | grep -E "oracle V-cycle|device vs oracle|passed|failed|skipped" | tee gpurun_out/r05_config3_oracle_fullsize.txt
