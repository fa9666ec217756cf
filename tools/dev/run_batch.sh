R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
t0=$(date +%s); timeout -k 10 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$? seconds=$(( $(date +%s) - t0 ))"
python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bench_default.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], d["config"]["residual_norm_after"], d["time_to_tolerance"]["config3"]["seconds"], d["config"]["coarse_solver"])
PY
