#!/bin/bash
# experiment: marginal cost of vector instructions in k_chain<mixed>; fast step on/off  (K genomes of the mtb star)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for v in "" nofast valu0 valu64 ${EXTRA_VARIANTS:-}; do
	if [ -n "$v" ]; then export RB3GPU_LIB=$R/ropebwt3_amd/prof/$v.so; else unset RB3GPU_LIB; fi
	timeout 300 python bench.py --only headline --steps 2 --warmup 1 --mtb ${K:-152} > gpurun_out/exp2_$v.json 2>/dev/null
	python - "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/exp2_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("variant %-8s" % (sys.argv[1] or "default"), "ms", d["ms_per_step"], "k_chain", d["phases_ms_per_step"]["k_chain"], "rank", d["phases_ms_per_step"]["rank"], "rebuild", d["phases_ms_per_step"]["rebuild"], "kchain/launch", d["roofline"]["ms_per_launch"], "steps", d["config"]["lf_steps_per_step"], d["config"]["fmd_identical_to_reference"], d["config"]["rank_phase_fallbacks"])
PY
done
