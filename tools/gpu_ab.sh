#!/bin/bash
# GPU box: A/B of library variants on the headline leg:  tools/gpu_ab.sh name1 name2 ...   (ropebwt3_amd/prof/NAME.so; "release" = the in-tree library)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for rep in $(seq 1 ${REPS:-2}); do
for v in "$@"; do
	if [ "$v" = release ]; then unset RB3GPU_LIB; else export RB3GPU_LIB=$R/ropebwt3_amd/prof/$v.so; fi
	RB3_BENCH_VERBOSE=2 timeout 300 python bench.py --only headline --steps ${STEPS:-3} --warmup 1 ${BENCH_ARGS:-} > gpurun_out/ab.json 2> gpurun_out/ab.err || tail -3 gpurun_out/ab.err
	python - "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
p = d["phases_ms_per_step"]
print("%-10s ms %.1f  h2d %.1f lf %.1f rank %.1f k_chain %.1f rebuild %.1f host %.1f  md5ok %s fb %s" % (sys.argv[1], d["ms_per_step"], p["h2d"], p["lf"], p["rank"], p["k_chain"], p["rebuild"], p["host_and_sync_inside_merge_calls"], d["config"]["fmd_identical_to_reference"], d["config"]["rank_phase_fallbacks"]))
PY
	grep "\[W" gpurun_out/ab.err | head -3
done; done
