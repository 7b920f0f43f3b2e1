#!/bin/bash
# GPU box: rows per thread of the validation pass (k_pos_finalize_check_rowsN): merge tests, then the headline per variant
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_parity.py -q -x 2>&1 | tail -4 | tee gpurun_out/r5_fin_tests.txt
REPS=3 bash tools/gpu_ab_lib.sh release fin2 2>&1 | tee gpurun_out/r5_ab_fin.txt
