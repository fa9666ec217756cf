#!/bin/bash
# placement tuner: its tests, then bench.py with 0 / 8 / 16 candidates (two processes each: the draw differs per process)
O=gpurun_out/tune; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_placement.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for t in 0 8 0 8 16; do
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-time-to-tolerance --tune-placement $t > $O/bench_t$t.json 2>$O/bench_t$t.err || { echo "bench $t failed"; tail -5 $O/bench_t$t.err; exit 1; }
  python3 - "$O/bench_t$t.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d["ms_per_step"],2), "%.3e"%d["value"], round(d["roofline"]["avg_launch_ms"],3), round(d["roofline"]["frac"],3), d["config"]["placement"])
PY
done
