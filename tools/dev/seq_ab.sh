#!/bin/bash
# per-launch durations of the finest-level applies, previous build (homogenization.jl_amd/ab/libhmg_prev.so) against the current one, alternating
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity_l6.py "tests/test_gpu_parity.py::test_exact_savings_are_exact" -q -x > gpurun_out/t_ahead.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/t_ahead.log
for rep in 1 2; do for lib in prev new; do
  if [ $lib = prev ]; then export HMG_LIB_PATH=$R/homogenization.jl_amd/ab/libhmg_prev.so HMG_LIB_AB=1; else unset HMG_LIB_PATH HMG_LIB_AB; fi
  bash tools/dev/apply_sequence.sh > /dev/null 2>&1; echo "$lib: $(tail -2 gpurun_out/seq/sequence.txt | head -1)"
done; done
