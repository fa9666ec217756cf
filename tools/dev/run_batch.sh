R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash tools/dev/bench_rehearsal.sh 2>&1 | grep -v "^   level\|placement" | cut -c1-400
HMG_REHEARSE_WORLD=8 timeout -k 10 400 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-time-to-tolerance --no-level-report > gpurun_out/bench_rank0of8.json 2> gpurun_out/bench_rank0of8.err; echo "rank0of8 rc=$?"
HMG_SYNTHETIC_CUT=planes timeout -k 10 400 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-time-to-tolerance --no-level-report > gpurun_out/bench_syncut.json 2> gpurun_out/bench_syncut.err; echo "syncut rc=$?"
python3 - <<PY
import json
for f in ("bench_rank0of8", "bench_syncut"):
    d = json.loads([x for x in open("gpurun_out/" + f + ".json").read().splitlines() if x.startswith("{")][-1])
    print("%.2f %.3e %.3f %.3f %s %s" % (d["ms_per_step"], d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["config"]["residual_norm_after"], d["config"]["workload"]))
    print("   comm", json.dumps(d.get("comm")))
PY
