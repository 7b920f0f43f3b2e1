#!/bin/bash
# GPU box, the final library of round 6: soak, CLI fuzz against the reference binary, idle and crowded hunts (does any merge leave tentative records unsettled; is any .fmd wrong?) -> gpurun_out/prof/r6_*
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; P=gpurun_out/prof; mkdir -p $P
timeout 900 python tools/soak.py 120 51000 2>&1 | tail -2 | tee $P/r6_soak.txt
timeout 1500 python tools/fuzz_cli.py ${FUZZ:-240} 5000 2>&1 | tail -8 | tee $P/r6_fuzz_cli.txt
( for i in 1 2 3 4; do RB3_BENCH_VERBOSE=2 timeout 600 python bench.py --only headline --no-aux --steps 40 --warmup 1 2> $P/hunt.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('idle run: %d merges, ms per build %.1f, redone %s, md5 ok %s' % (42*151, d['ms_per_step'], d['config']['rank_phase_fallbacks'], d['config']['fmd_identical_to_reference']))"; grep -h "\[W" $P/hunt.err | head -3; done ) | tee $P/r6_hunt_idle.txt
RB3GPU_JUNCTION_CHECK=1 bash tools/gpu_crowded_hunt.sh ${HUNT:-6} 40 2>&1 | grep -v "^\[W" | tail -4 | tee $P/r6_hunt_crowded.txt
rm -f $P/hunt.err
