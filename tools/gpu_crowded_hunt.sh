#!/bin/bash
# GPU box: the headline build over and over on a deliberately CROWDED GPU -- a second process runs a memory-latency-bound kernel
# (tools/ubench/lfchase: random 128-byte lines + written-through stores) all the time, so that waves of k_chain start late and
# records become visible late: does any merge leave tentative records unsettled (rank phase redone)?   tools/gpu_crowded_hunt.sh [runs] [steps]
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
[ -x $R/tools/ubench/lfchase ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $R/tools/ubench/lfchase $R/tools/ubench/lfchase.hip
( while true; do $R/tools/ubench/lfchase 2048 162000 2 131072 3000 1 > /dev/null 2>&1; done ) &
HAMMER=$!
trap "kill $HAMMER 2>/dev/null; wait $HAMMER 2>/dev/null" EXIT
sleep 1
TOT=0; FB=0; BAD=0
for i in $(seq 1 ${1:-20}); do
	RB3_BENCH_VERBOSE=2 timeout 600 python bench.py --only headline --no-aux --steps ${2:-12} --warmup 1 > gpurun_out/ch.json 2> gpurun_out/ch.err
	read n f ok ms <<< $(python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/ch.json").read().strip().splitlines()[-1])
    print((d["steps"] + 2) * 151, d["config"]["rank_phase_fallbacks"], int(bool(d["config"]["fmd_identical_to_reference"])), d["ms_per_step"])
except Exception:
    print(0, 0, 0, 0)
PY
)
	TOT=$((TOT + n)); FB=$((FB + f)); [ "$ok" = 1 ] || BAD=$((BAD + 1))
	echo "run $i: merges so far $TOT, redone (timed steps) $FB, runs with a wrong md5 $BAD, ms per build on the crowded GPU $ms"
	grep -h "\[W" gpurun_out/ch.err | head -5
done
kill $HAMMER 2>/dev/null
echo "TOTAL: $TOT merges on a crowded GPU, $FB redone in the timed steps, $BAD runs with a wrong md5"
