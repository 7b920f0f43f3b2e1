#!/bin/bash
# GPU box: the round's committed evidence (copy gpurun_out/prof/r3_* into profiles/)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/prof
# 1. the default bench line
timeout 900 python bench.py > gpurun_out/prof/r3_bench_default.json 2> gpurun_out/prof/r3_bench_default.err; echo "bench rc=$?"
# 2. kernel statistics of the headline (one warm-up build + one timed build)
LINES_OUT=45 bash tools/prof_bench.sh r3_mtb152 --only headline --steps 1 --warmup 1 | cut -c1-130
# 3. counters of the headline's kernels: HBM traffic of k_chain, issue/stall counters of k_chain and the run-space rebuild
bash tools/pmc_headline.sh r3_mtb152 "k_chain|k_reb_group|k_events|k_pos_finalize" k_chain_mtb152 "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
	"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
	"SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH" > /dev/null 2>&1
python tools/pmc_json.py $R/gpurun_out/prof/r3_mtb152_counters.txt $R/gpurun_out/prof/r3_mtb152_pmc_k_chain_mtb152.json
head -50 gpurun_out/prof/r3_mtb152_counters.txt | cut -c1-120
cat gpurun_out/prof/r3_mtb152_pmc_k_chain_mtb152.json
