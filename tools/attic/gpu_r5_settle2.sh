#!/bin/bash
# GPU box: the whole GPU suite with the round-5 defaults (age 16, two stretches per octet in k_events), then k_events' unroll and the settle kernels' launch widths on the headline
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | tail -6 | tee gpurun_out/r5_settle2_tests.txt
REPS=2 bash tools/gpu_ab_lib.sh release ev1 2>&1 | tee gpurun_out/r5_ab_settle2.txt
STEPS=3 bash tools/gpu_ab_env.sh "" "RB3GPU_CUM_BLOCKS=2048" "RB3GPU_CUM_BLOCKS=4096" "RB3GPU_RESW_BLOCKS=2048" "RB3GPU_EV_BLOCKS=4096" "RB3GPU_SFIN_BLOCKS=4096" "RB3GPU_CUM_BLOCKS=2048 RB3GPU_RESW_BLOCKS=2048 RB3GPU_EV_BLOCKS=4096" 2>&1 | tee -a gpurun_out/r5_ab_settle2.txt
