#!/bin/bash
# GPU box: the round's standard check -- GPU tests, the default bench line, the N>1 path with ranks sharing the GPU, kernel statistics of the headline
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out
WHAT=${1:-all}
if [ "$WHAT" = all ] || [ "$WHAT" = tests ]; then
	timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
	tail -3 gpurun_out/pytest_gpu.log
fi
if [ "$WHAT" = all ] || [ "$WHAT" = bench ]; then
	timeout 900 python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err; echo "bench rc=$?"; tail -3 gpurun_out/bench_n1.err
	python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench_n1.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step", "phases_ms_per_step", "not_counted_ms_per_step", "h2d")})
    print(d["config"]["fmd_identical_to_reference"], d["roofline"]["frac"], d["roofline"]["ms_per_launch"], d["roofline_path"]["frac"], d["roofline_rebuild"]["frac"])
    print(d.get("aux_cli_build"))
except Exception as e:
    print("no bench line:", e)
PY
fi
if [ "$WHAT" = all ] || [ "$WHAT" = multi ]; then
	timeout 600 python bench.py --gpus 2 --steps 1 --warmup 0 --mtb 24 > gpurun_out/bench_n2_mtb24.json 2> gpurun_out/bench_n2_mtb24.err; echo "bench --gpus 2 rc=$?"; tail -5 gpurun_out/bench_n2_mtb24.err; cat gpurun_out/bench_n2_mtb24.json | cut -c1-1500
fi
if [ "$WHAT" = all ] || [ "$WHAT" = prof ]; then
	bash tools/prof_bench.sh r5_mtb152 --only headline --steps 1 --warmup 1
fi
