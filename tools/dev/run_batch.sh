R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -x -q 2>&1 | tail -2
for rep in 1 2; do for lib in prev new; do
  if [ $lib = prev ]; then export HMG_LIB_PATH=$R/homogenization.jl_amd/ab/libhmg_prev.so HMG_LIB_AB=1; else unset HMG_LIB_PATH HMG_LIB_AB; fi
  timeout -k 10 300 python3 tools/level_bench.py 2>/dev/null | python3 -c "
import sys, json
r=[json.loads(l) for l in sys.stdin if l.startswith('{')]
print('$lib', ' '.join('%d:%.3f' % (x['vcycle_from_level'], x['ms']) for x in r))"
done; done
unset HMG_LIB_PATH HMG_LIB_AB
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ictrace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-time-to-tolerance --no-level-report > $R/gpurun_out/ictrace.log 2>&1
grep "k_iface" $(ls -t $R/gpurun_out/ictrace/*/*kernel_stats.csv | head -1) | cut -c1-60,120-200; rm -rf $R/gpurun_out/ictrace
