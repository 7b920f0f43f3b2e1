#!/bin/bash
# GPU box, round 6, experiment 1: the headline leg with an octet per walker (release) against a QUAD per walker (RB3GPU_LPW=4: 16 walkers per wave,
# k_chain<..., 4, true>) at several walker spacings -- does more walkers in flight per SIMD pay now that the kernel runs at resident capacity?
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
STEPS=2 bash tools/gpu_ab2.sh "" "RB3GPU_LPW=4" "RB3GPU_LPW=4 -- --walker-step 180" "RB3GPU_LPW=4 -- --walker-step 140" "RB3GPU_LPW=4 RB3GPU_BLKCAP=1024 -- --walker-step 140" ""
