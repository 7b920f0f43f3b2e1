#!/bin/bash
# GPU box: a quick check of a kernel change -- engine tests, a short soak, the headline leg of the bench
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python tools/soak.py ${SOAK:-60} 2>&1 | tail -2
timeout 300 python bench.py --only headline --steps 3 --warmup 1 > gpurun_out/quick.json 2> gpurun_out/quick.err; tail -2 gpurun_out/quick.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/quick.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["phases_ms_per_step"], d["config"]["fmd_identical_to_reference"], d["config"]["rank_phase_fallbacks"])
PY
