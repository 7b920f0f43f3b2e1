#!/bin/bash
# GPU box: counters and kernel statistics of the large-index leg (1.07 G symbols of bit-plane slots in HBM, 1 M reads per batch)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/prof
LINES_OUT=16 bash tools/prof_bench.sh r3_large_index --only large | cut -c1-130
LEG=large BENCH_STEPS=" " bash tools/pmc_headline.sh r3_large "k_chain|k_pass1w|k_pass2w" k_chain_large "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" > /dev/null 2>&1
python tools/pmc_json.py gpurun_out/prof/r3_large_counters.txt gpurun_out/prof/r3_pmc_k_chain_large.json "k_chain<list,dense,plain>" "k_chain<list,dense,plain,text> on the large-index leg (bench.py --only large: 302 M LF steps per launch into 546 MB of bit-plane slots; %d launches)"
cat gpurun_out/prof/r3_large_counters.txt | cut -c1-120
