#!/bin/bash
# GPU box: the round's committed evidence (copy gpurun_out/prof/r6_* into profiles/)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/prof
WHAT=${1:-all}
if [ "$WHAT" = all ] || [ "$WHAT" = bench ]; then
	timeout 900 python bench.py > gpurun_out/prof/r6_bench_default.json 2> gpurun_out/prof/r6_bench_default.err; echo "bench rc=$?"
fi
if [ "$WHAT" = all ] || [ "$WHAT" = stats ]; then
	# kernel statistics: the headline (warm-up build + timed build + the one with serial H2D), the large-index leg, the index beyond 2^32 symbols
	LINES_OUT=48 bash tools/prof_bench.sh r6_mtb152 --only headline --no-aux --steps 1 --warmup 1 | cut -c1-130
	LINES_OUT=14 bash tools/prof_bench.sh r6_large_index --only large | cut -c1-130
	LINES_OUT=24 bash tools/prof_bench.sh r6_index_8g --only 8g | cut -c1-130
fi
if [ "$WHAT" = all ] || [ "$WHAT" = pmc ]; then
	BENCH_ARGS="--no-aux" bash tools/pmc_headline.sh r6_mtb152 "k_chain|k_reb_group|k_plane_group|k_events|k_pos_finalize|k_place" k_chain_mtb152 "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
		"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
		"SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH" > /dev/null 2>&1
	python tools/pmc_json.py gpurun_out/prof/r6_mtb152_counters.txt gpurun_out/prof/r6_pmc_k_chain_mtb152.json
	LEG=large BENCH_STEPS=" " bash tools/pmc_headline.sh r6_large "k_chain|k_plane_group|k_pos_finalize|k_place_pg" k_chain_large "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
		"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" > /dev/null 2>&1
	python tools/pmc_json.py gpurun_out/prof/r6_large_counters.txt gpurun_out/prof/r6_pmc_k_chain_large.json "k_chain<list,dense,plain>" "k_chain<list,dense,plain,text> on the large-index leg (bench.py --only large: 302 M LF steps per launch into 546 MB of bit-plane slots, records in text order; %d launches)" "The index (546 MB of slots) is far larger than L2 + Infinity Cache: this is DRAM traffic."
	head -60 gpurun_out/prof/r6_mtb152_counters.txt | cut -c1-120
	cat gpurun_out/prof/r6_pmc_k_chain_mtb152.json gpurun_out/prof/r6_pmc_k_chain_large.json
fi
