#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 600 python bench.py --only headline --steps 2 --warmup 1 > gpurun_out/exp8.json 2>/dev/null
python - <<'PY'
import json
d = json.loads(open("gpurun_out/exp8.json").read().strip().splitlines()[-1])
print("N=1 value", d["value"], "ms", d["ms_per_step"], d["phases_ms_per_step"], "kchain/launch", d["roofline"]["ms_per_launch"], d["config"]["fmd_identical_to_reference"], d["config"]["rank_phase_fallbacks"])
PY
for n in 2 4; do
timeout 900 python bench.py --gpus $n --steps 1 --warmup 0 > gpurun_out/exp8_n$n.json 2> gpurun_out/exp8_n$n.err; echo "N=$n rc=$?"
python - $n <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/exp8_n%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
    print("N", d["n_gpus"], "value", d["value"], "ms", d["ms_per_step"], d["phases_ms_per_step"], d["config"]["fmd_identical_to_reference"])
except Exception as e:
    print("no line", e); print(open("gpurun_out/exp8_n%s.err" % sys.argv[1]).read()[-1500:])
PY
done
