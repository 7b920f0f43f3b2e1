#!/bin/bash
# GPU box: A/B of library variants on the auxiliary legs (large index in HBM, reads regime):  tools/gpu_ab_aux.sh name1 name2 ...
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for rep in 1 2; do for v in "$@"; do
	if [ "$v" = release ]; then unset RB3GPU_LIB; else export RB3GPU_LIB=$R/ropebwt3_amd/prof/$v.so; fi
	for leg in large reads; do
		timeout 600 python bench.py --only $leg > gpurun_out/abx.json 2> gpurun_out/abx.err || tail -3 gpurun_out/abx.err
		python - "$v" "$leg" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/abx.json").read().strip().splitlines()[-1])
print("%-8s %-6s value %.3f Gbp/s  ms %.3f  phases %s  k_chain ms %.3f frac %.3f" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["phases_ms_per_step"], d["roofline"]["ms_per_launch"], d["roofline"]["frac"]))
PY
	done
done; done
