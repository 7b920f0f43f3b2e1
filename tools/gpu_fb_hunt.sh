#!/bin/bash
# GPU box: repeat the headline leg and report every merge that was redone (the engine's warnings say why):  tools/gpu_fb_hunt.sh [runs] [lib.so]
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
[ -n "$2" ] && export RB3GPU_LIB=$R/ropebwt3_amd/prof/$2.so
for i in $(seq 1 ${1:-20}); do
	RB3_BENCH_VERBOSE=2 timeout 300 python bench.py --only headline --steps 3 --warmup 1 > gpurun_out/fb.json 2> gpurun_out/fb.err
	python - <<'PY'
import json
d = json.loads(open("gpurun_out/fb.json").read().strip().splitlines()[-1])
print("ms %.1f fb %s md5 %s" % (d["ms_per_step"], d["config"]["rank_phase_fallbacks"], d["config"]["fmd_identical_to_reference"]))
PY
	grep -A40 "\[W" gpurun_out/fb.err | grep -v "^\[M" | head -60
done
