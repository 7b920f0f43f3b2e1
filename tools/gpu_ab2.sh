#!/bin/bash
# GPU box: A/B of tune switches + bench arguments on the headline leg:  bash tools/gpu_ab2.sh "ENV=.. -- --walker-step N" ...
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for cfg in "$@"; do
	envs="${cfg%%--*}"; args="${cfg#*--}"; [ "$args" = "$cfg" ] && args=""
	env $envs timeout 600 python bench.py --only headline --no-aux --steps ${STEPS:-2} --warmup 1 $args > gpurun_out/ab.json 2>/dev/null
	python - "$cfg" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1]); p = d["phases_ms_per_step"]
    print("%-44s ms %.1f  k_chain %.1f rank %.1f rebuild %.1f lf %.1f steps %.2fM fallbacks %s md5ok %s" % (sys.argv[1] or "(defaults)", d["ms_per_step"], p["k_chain"], p["rank"], p["rebuild"], p["lf"], d["config"]["lf_steps_per_step"] / 151e6, d["config"]["rank_phase_fallbacks"], d["config"]["fmd_identical_to_reference"]))
except Exception as e:
    print("%-44s FAILED %r" % (sys.argv[1], e))
PY
done
