#!/bin/bash
# round 5: rb3gpu_sh_merge with one interval, rounds back to back on the device: tests, then the per-round cost against the host-driven loop
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_cli.py -m gpu -x -q -k "shard or interval or balanced or rccl" > gpurun_out/r5_iv_tests.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r5_iv_tests.txt | tail -3; grep -E "^E  |Error" gpurun_out/r5_iv_tests.txt | head -20
echo "== rounds on the device"; timeout 600 python tools/probe_sh_round.py 1000 100000 1000000 2000000
echo "== rounds driven by the host (RB3GPU_SH_HOST_ROUNDS=1)"; RB3GPU_SH_HOST_ROUNDS=1 timeout 600 python tools/probe_sh_round.py 1000 100000 1000000 2000000
