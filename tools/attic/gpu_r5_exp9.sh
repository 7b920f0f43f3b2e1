#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cli.py -m gpu -x -q > gpurun_out/r5_cli_tests.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r5_cli_tests.txt | tail -3; grep -E "^E  " gpurun_out/r5_cli_tests.txt | head
timeout 600 python -m pytest tests/test_gpu_engine.py -m gpu -x -q -k "walker_list or junction or shard or interval" 2>&1 | grep -E "passed|failed|^E  " | head
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
	rm -rf /tmp/prof_sh$m
	RB3GPU_SH_HOST_ROUNDS=$m timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_sh$m -o sh -- python $R/tools/probe_sh_round.py 1000000 > /tmp/prof_sh$m.log 2>&1
	echo "== RB3GPU_SH_HOST_ROUNDS=$m"; tail -1 /tmp/prof_sh$m.log | cut -c1-200
	python $R/tools/prof_summary.py stats $(ls /tmp/prof_sh$m/*_results.db /tmp/prof_sh$m/*/*_results.db 2>/dev/null | head -1) /tmp/prof_sh$m.txt > /dev/null 2>&1; grep -i "k_sh_round\|^kernel" /tmp/prof_sh$m.txt | head -5
done
cp /tmp/prof_sh0.txt $R/gpurun_out/r5_sh_round_device_stats.txt; cp /tmp/prof_sh1.txt $R/gpurun_out/r5_sh_round_host_stats.txt
