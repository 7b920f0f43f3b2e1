#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for v in "" age16 age8; do
	if [ -n "$v" ]; then export RB3GPU_LIB=$R/ropebwt3_amd/prof/$v.so; else unset RB3GPU_LIB; fi
	timeout 300 python bench.py --only headline --steps 2 --warmup 1 > gpurun_out/exp14_$v.json 2>/dev/null
	python - "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/exp14_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("variant %-8s" % (sys.argv[1] or "default"), "ms", d["ms_per_step"], "k_chain", d["phases_ms_per_step"]["k_chain"], "rank", d["phases_ms_per_step"]["rank"], "steps", d["config"]["lf_steps_per_step"], d["config"]["fmd_identical_to_reference"], "fallbacks", d["config"]["rank_phase_fallbacks"], d["config"]["long_settles"])
PY
	[ -n "$v" ] && (SOAK_TEXT=1 timeout 600 python tools/soak.py 120 20000 2>&1 | tail -1)
done
