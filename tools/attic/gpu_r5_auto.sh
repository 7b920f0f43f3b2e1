#!/bin/bash
# GPU box: the age from which walkers of the device-made list (BWT-only signature) record tentatively: the reference-signature leg per variant
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for rep in 1 2; do for v in "$@"; do
	if [ "$v" = release ]; then unset RB3GPU_LIB; else export RB3GPU_LIB=$R/ropebwt3_amd/prof/$v.so; fi
	RB3_BENCH_VERBOSE=2 timeout 600 python bench.py --only headline --steps 3 --warmup 1 > gpurun_out/ab.json 2>gpurun_out/ab.err
	python - "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1]); a = d["aux_mtb152_reference_signature"]; p = a["phases_ms_per_step"]
print("%-10s refsig ms %.1f lf %.1f rank %.1f k_chain %.1f rebuild %.1f fallbacks %s md5ok %s (headline %.1f)" % (sys.argv[1], a["ms_per_step"], p["lf"], p["rank"], p["k_chain"], p["rebuild"], a["rank_phase_fallbacks"], a["fmd_identical_to_reference"], d["ms_per_step"]))
PY
	grep -c "redoing" gpurun_out/ab.err
done; done
