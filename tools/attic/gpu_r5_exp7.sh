#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_engine.py -m gpu -x -q > gpurun_out/r5_engine_tests.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r5_engine_tests.txt | tail -3; grep -E "^E  " gpurun_out/r5_engine_tests.txt | head
timeout 300 python tools/soak.py ${SOAK:-60} 2>&1 | tail -2
for a in "" "--host-walkers"; do
timeout 600 python bench.py --only headline --steps 3 --warmup 1 $a > gpurun_out/ab.json 2>gpurun_out/ab.err
python - "$a" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1]); p = d["phases_ms_per_step"]
    print("%-16s ms %.1f  k_chain %.1f (%.4f ms/launch) rank %.1f rebuild %.1f lf %.1f host %.1f fallbacks %s md5ok %s not-counted %s" % (sys.argv[1] or "(device list)", d["ms_per_step"], p["k_chain"], d["roofline"]["ms_per_launch"], p["rank"], p["rebuild"], p["lf"], p["host_and_sync_inside_merge_calls"], d["config"]["rank_phase_fallbacks"], d["config"]["fmd_identical_to_reference"], {k: v for k, v in d["not_counted_ms_per_step"].items() if k != "note"}))
except Exception as e:
    print("FAILED", e); print(open("gpurun_out/ab.err").read()[-1500:])
PY
done
echo "== rounds on the device"; timeout 600 python tools/probe_sh_round.py 1000 100000 1000000 2000000
timeout 900 python -m pytest tests/test_gpu_cli.py -m gpu -x -q > gpurun_out/r5_cli_tests.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r5_cli_tests.txt | tail -3; grep -E "^E  " gpurun_out/r5_cli_tests.txt | head
