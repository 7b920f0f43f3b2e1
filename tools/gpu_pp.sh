#!/bin/bash
# GPU box: check of a rebuild change -- engine + CLI tests, per-round rebuild times of the headline build, the headline leg
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -8
timeout 300 python tools/probe_reb_rounds.py 152 2>&1 | tail -45
timeout 300 python bench.py --only headline --steps 3 --warmup 1 > gpurun_out/quick.json 2> gpurun_out/quick.err; tail -2 gpurun_out/quick.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/quick.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["phases_ms_per_step"], d["config"]["fmd_identical_to_reference"], d["config"]["rank_phase_fallbacks"])
PY
