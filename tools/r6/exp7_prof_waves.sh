#!/bin/bash
# GPU box, round 6: where does the tail of a k_chain launch come from?  -DRB3_PROF_WAVES: every wave notes when it began and ended and where it ran (HW_ID,
# XCC_ID); the host groups the waves of merge 140 of the 152-genome build by the number of waves their SIMD / compute unit held.
#   tools/build_variant.sh profwaves -DRB3_PROF_WAVES; tools/build_variant.sh stride -DRB3_EXP_STRIDE     (in the container, before gpurun)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
STEPS=1 bash tools/gpu_ab_lib.sh profwaves; grep "prof waves" gpurun_out/ab.err | head -16
REPS=2 bash tools/gpu_ab_lib.sh release stride
STEPS=2 bash tools/gpu_ab2.sh "" " -- --walker-step 222" "RB3GPU_LF_AFTER=1 -- --walker-step 222" "RB3GPU_LF_AFTER=1 -- --walker-step 217" " -- --walker-step 217" ""
