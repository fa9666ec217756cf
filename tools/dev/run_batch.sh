R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
t0=$(date +%s); timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_shape.json 2> gpurun_out/bench_driver_shape.err; echo "bench rc=$? seconds=$(( $(date +%s) - t0 ))"
python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bench_driver_shape.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], d["config"]["residual_norm_after"], d["time_to_tolerance"]["config3"]["seconds"])
PY
