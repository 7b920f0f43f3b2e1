#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -x -q -k "stretch or thin or fewer_walkers or merge_index or whole_index" 2>&1 | tail -3
NS="2 4" bash tools/gpu_exp9.sh
