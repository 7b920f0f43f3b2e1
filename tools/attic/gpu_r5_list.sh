#!/bin/bash
# GPU box: the whole GPU suite on the round's final library; the walker-list kernels beside the fill; what the intervals of `--gpus 4 --interval` hold
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r5_list_tests.txt
STEPS=3 bash tools/gpu_ab_env.sh "" "RB3GPU_LIST_BESIDE=0" "" "RB3GPU_LIST_BESIDE=0" 2>&1 | tee gpurun_out/r5_ab_list.txt
bash tools/gpu_interval_peaks.sh 1000000 60m 2>&1 | tee gpurun_out/r5_ivpeak.txt
