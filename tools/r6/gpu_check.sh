#!/bin/bash
# GPU box, round 6: the standard check of a kernel change -- GPU suite (stop at the first failure), then the headline leg twice
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout ${SUITE_TIMEOUT:-1500} python -m pytest tests -m gpu -x -q ${PYTEST_ARGS} 2>&1 | tail -${TAIL:-8}
STEPS=${STEPS:-3} bash tools/gpu_ab2.sh "" ""
