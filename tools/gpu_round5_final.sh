#!/bin/bash
# GPU box, the final library of round 5 in one call: the GPU suite, the evidence of tools/gpu_profiles_r5.sh, the N = 2 path with the ranks sharing
# the device, the build merge by merge, soak, CLI fuzz against the reference binary, idle and crowded hunts.  Everything lands in gpurun_out/prof/r5_*.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/prof
P=gpurun_out/prof
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $P/r5_gpu_tests.txt
bash tools/gpu_profiles_r5.sh all > $P/r5_profiles.log 2>&1; tail -3 $P/r5_profiles.log | cut -c1-200
timeout 900 python bench.py --gpus 2 --steps 1 --warmup 0 --mtb 24 > $P/r5_bench_n2_shared_gpu.json 2> $P/r5_bench_n2_shared_gpu.err; echo "bench --gpus 2 rc=$?"
bash tools/gpu_per_merge.sh > /dev/null 2>&1; cp gpurun_out/r5_series.err $P/r5_series.err 2>/dev/null
timeout 900 python tools/soak.py 120 51000 2>&1 | tail -2 | tee $P/r5_soak.txt
timeout 1500 python tools/fuzz_cli.py ${FUZZ:-240} 5000 2>&1 | tail -8 | tee $P/r5_fuzz_cli.txt
( for i in 1 2 3 4 5 6; do RB3_BENCH_VERBOSE=2 timeout 600 python bench.py --only headline --no-aux --steps 40 --warmup 1 2> $P/hunt.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('idle run: %d merges, ms per build %.1f, redone %s, md5 ok %s' % (42*151, d['ms_per_step'], d['config']['rank_phase_fallbacks'], d['config']['fmd_identical_to_reference']))"; grep -h "\[W" $P/hunt.err | head -3; done ) | tee $P/r5_hunt_idle.txt
RB3GPU_JUNCTION_CHECK=1 bash tools/gpu_crowded_hunt.sh ${HUNT:-10} 40 2>&1 | grep -v "^\[W" | tail -4 | tee $P/r5_hunt_crowded_age16.txt
