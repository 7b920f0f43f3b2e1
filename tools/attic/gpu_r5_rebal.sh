#!/bin/bash
# GPU box: the shard object's rebalancing + the interval tests with it on by default, then the pre-roll/age variants on the headline
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_multi.py tests/test_gpu_cli.py -q -x -k "shard or interval or sharded or balanced or rebalance or device_count" 2>&1 | tail -15 > gpurun_out/r5_rebal_tests.txt
cat gpurun_out/r5_rebal_tests.txt
REPS=2 bash tools/gpu_ab_lib.sh release age24 age16 2>&1 | tee gpurun_out/r5_ab_age.txt
