#!/bin/bash
# level 7 (config 5's per-GPU share): parity tests of the slab levels, bench line, kernel table
O=gpurun_out/l7; mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "level7 or config5 or l7 or slab" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1; fi
timeout -k 10 300 python3 bench.py --levels 7 --width 16 --sigma-high 100 --no-cpu-baseline --no-time-to-tolerance --steps 5 --warmup 2 > $O/bench_l7.json 2>$O/bench_l7.err || { tail -5 $O/bench_l7.err; exit 1; }
python3 - $O/bench_l7.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d["ms_per_step"],2), "%.3e"%d["value"], round(d["roofline"]["avg_launch_ms"],3), round(d["roofline"]["frac"],3), d["config"]["placement"])
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o l7 -- python3 $GRAFT_REPO_ROOT/bench.py --levels 7 --width 16 --sigma-high 100 --no-cpu-baseline --no-time-to-tolerance --steps 2 --warmup 1 --tune-placement 0 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/$O/trace.log; exit 1; }
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/l7.kernel_stats.csv
python3 - $O/l7.kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print("%-100s %5s %9.3f ms %9.2f us"%(r['Name'][:100], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
