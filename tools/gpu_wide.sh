#!/bin/bash
# GPU box: the wide-mask settle (more than 255 relatives): forced widths through the whole engine suite and the soak, the 320-relative golden, a 400-genome build
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
echo "== engine tests, default"; timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -x -q 2>&1 | tail -3
for q in 2 8; do echo "== engine tests, RB3GPU_TENT_Q=$q"; RB3GPU_TENT_Q=$q timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -x -q 2>&1 | tail -3; done
echo "== soak default"; timeout 300 python tools/soak.py 40 2>&1 | tail -1
for q in 2 4 8; do echo "== soak RB3GPU_TENT_Q=$q"; RB3GPU_TENT_Q=$q timeout 600 python tools/soak.py 60 $((2000 + q)) 2>&1 | tail -1; done
echo "== 320 relatives"; timeout 600 python -m pytest tests/test_gpu_cli.py -m gpu -x -q -k "relatives or family" 2>&1 | tail -3
echo "== 400 genomes"; timeout 900 python tools/probe_mtb.py 400 4400000 RB3GPU_TENT_Q=1 2>&1 | grep -v "^    round" | tail -12
