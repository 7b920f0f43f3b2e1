#!/bin/bash
# GPU box: engine tests with the final round-5 kernels, then walker spacing and text-order records (trec) on the headline with the pre-roll of 16
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_engine.py -q -x 2>&1 | tail -3 | tee gpurun_out/r5_step_tests.txt
for a in "" "--walker-step 192" "--walker-step 260" "--walker-step 300"; do
	BENCH_ARGS="$a" STEPS=3 bash tools/gpu_ab_env.sh "" 2>&1 | sed "s/(default)/step:$a /" | tee -a gpurun_out/r5_ab_step.txt
done
STEPS=3 bash tools/gpu_ab_env.sh "RB3GPU_TREC=1" "RB3GPU_TREC=0" 2>&1 | tee -a gpurun_out/r5_ab_step.txt
