#!/bin/bash
# k_chain at 6 waves per SIMD (80 registers, a few spills) against 5 (88): library variant x walker spacing
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for rep in 1 2; do
for cfg in "release 0" "wpe6 186" "wpe6 192" "wpe6 220" "release 220"; do
	set -- $cfg
	if [ "$1" = release ]; then unset RB3GPU_LIB; else export RB3GPU_LIB=$R/ropebwt3_amd/prof/$1.so; fi
	timeout 300 python bench.py --only headline --steps 2 --warmup 1 --walker-step $2 > gpurun_out/exp21.json 2>/dev/null
	python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/exp21.json").read().strip().splitlines()[-1]); p = d["phases_ms_per_step"]
print("%s walker-step %s: ms %.1f k_chain %.1f rank %.1f rebuild %.1f host %.1f md5ok %s fb %s" % (sys.argv[1], sys.argv[2], d["ms_per_step"], p["k_chain"], p["rank"], p["rebuild"], p["host_and_sync_inside_merge_calls"], d["config"]["fmd_identical_to_reference"], d["config"]["rank_phase_fallbacks"]))
PY
done; done
