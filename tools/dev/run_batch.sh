timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wave.py tests/test_gpu_parity_l6.py -x -q > gpurun_out/t4.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/t4.log
timeout -k 10 300 python tools/ab_toggle.py "apply_small=0" "apply_small=1" --rounds 6 > gpurun_out/ab_small.txt 2>&1; cat gpurun_out/ab_small.txt
timeout -k 10 200 python bench.py --no-cpu-baseline --no-time-to-tolerance > gpurun_out/b4.log 2>gpurun_out/b4.err
python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/b4.log") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"])
for r in d["roofline"]["levels"]: print(r["level"], round(r["apply_ms_per_vcycle"],3), round(r["level_share_ms"],3))
PY
