#!/bin/bash
# GPU box: the default bench line, and the N = 2 path of bench.py rehearsed on the one GPU (gloo moves the data, every launched
# rank supervises its child, preflight parity on a small brick): gpurun_out/rehearsal/
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/rehearsal; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 500 python3 bench.py > $O/bench_default.log 2> $O/bench_default.err; echo "default rc=$?"
HMG_SINGLE_DEVICE=1 HMG_DIST_BACKEND=gloo HMG_PREFLIGHT_WIDTH=4 HMG_BENCH_ATTEMPT_SECONDS=300 timeout -k 10 700 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --width 8 --levels 5 --steps 3 --warmup 1 --no-cpu-baseline --no-time-to-tolerance > $O/bench_2rank_gloo.log 2> $O/bench_2rank_gloo.err; echo "2rank rc=$?"
python3 - <<PY
import json
for f in ("bench_default", "bench_2rank_gloo"):
    try:
        l = [x for x in open("$O/" + f + ".log").read().splitlines() if x.startswith("{")][-1]
        d = json.loads(l)
        print(f, "ms_per_step", round(d["ms_per_step"], 3), "value %.3e" % d["value"], "frac", round(d["roofline"]["frac"], 4))
        for k in ("launcher", "comm", "preflight", "ms_per_step_ranks"):
            if k in d: print("  ", k, json.dumps(d[k]))
        print("   placement", json.dumps(d["config"]["placement"]))
        print("   traffic", d["roofline"]["traffic"], d["roofline"]["traffic_source"])
        for r in d["roofline"]["levels"]:
            print("   level", r["level"], "apply %.3f ms, %.2f GB, %s TB/s, share %.3f ms" % (r["apply_ms_per_vcycle"], r["apply_algorithmic_GB_per_vcycle"], "%.2f" % r["apply_TBps"] if r["apply_TBps"] else "-", r["level_share_ms"]))
    except Exception as e:
        print(f, "no line:", e)
PY
tail -5 $O/bench_2rank_gloo.err
