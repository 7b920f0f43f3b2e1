#!/bin/bash
# round 5: the interval path (no gather, sharded batch): engine + CLI tests that touch it, then everything
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_cli.py tests/test_gpu_multi.py -m gpu -x -q -k "shard or interval or balanced or rccl or multi_gpu or device" > gpurun_out/r5_iv_tests.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r5_iv_tests.txt | tail -3; grep -E "^E  |Error" gpurun_out/r5_iv_tests.txt | head -20
