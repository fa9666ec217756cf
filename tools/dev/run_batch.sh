R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2; do for lib in any off; do
  if [ $lib = off ]; then export HMG_LIB_PATH=$R/homogenization.jl_amd/ab/libhmg_xnackoff.so HMG_LIB_AB=1; else unset HMG_LIB_PATH HMG_LIB_AB; fi
  timeout -k 10 300 python3 tools/level_bench.py 2>/dev/null | python3 -c "
import sys, json
r=[json.loads(l) for l in sys.stdin if l.startswith('{')]
print('xnack-$lib', ' '.join('%d:%.3f' % (x['vcycle_from_level'], x['ms']) for x in r))"
done; done
export HMG_LIB_PATH=$R/homogenization.jl_amd/ab/libhmg_xnackoff.so HMG_LIB_AB=1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
