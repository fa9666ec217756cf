#!/bin/bash
O=gpurun_out/small; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 tools/dev/small_levels_ab.py > $O/ab.log 2>&1; echo "ab rc=$?"; cat $O/ab.log | tail -8
