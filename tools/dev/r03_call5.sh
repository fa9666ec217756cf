#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/collect_profiles.sh r03 && bash $R/tools/dev/apply_sequence.sh && ls $R/gpurun_out/prof_r03
