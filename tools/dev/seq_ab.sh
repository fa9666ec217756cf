#!/bin/bash
# Per-launch apply durations of two builds, alternating (HMG_LIB_PATH selects the other build): gpurun_out/seq_ab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  for lib in prev new; do
    if [ $lib = prev ]; then export HMG_LIB_PATH=$R/homogenization.jl_amd/ab/libhmg_prev.so HMG_LIB_AB=1; else unset HMG_LIB_PATH HMG_LIB_AB; fi
    bash $R/tools/dev/apply_sequence.sh > /dev/null 2>&1
    echo "$lib: $(head -n 1 $R/gpurun_out/seq/sequence.txt)" | tee -a $R/gpurun_out/seq_ab.txt
  done
done
