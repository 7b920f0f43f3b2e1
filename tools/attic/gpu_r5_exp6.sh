#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_engine.py -m gpu -x -q > gpurun_out/r5_engine_tests.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r5_engine_tests.txt | tail -3
timeout 300 python tools/soak.py ${SOAK:-60} 2>&1 | tail -2
bash tools/gpu_ab.sh "" "RB3GPU_JUNCTION_CHECK=0" "RB3GPU_JUNCTION_CHECK=1" "" "RB3GPU_JUNCTION_CHECK=0"
