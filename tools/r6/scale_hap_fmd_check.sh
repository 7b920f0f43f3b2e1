#!/bin/bash
# GPU box, round 6: the .fmd of the 4 x 3.1 Gbp build (4.99 G runs: more than 2^32) packed on the GPU against the same index encoded by the host's encoder (--host-fmd): md5 of both
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/prof
NH=${1:-4}; LEN=${2:-3100000000}; D=${SCALE_TMP:-/dev/shm}/rb3_hapchk_$$
FILES=$(python - <<PY
import sys
sys.path.insert(0, "$R")
from tools import gen_family
print(" ".join(gen_family.big_haplotype_files($NH, $LEN, "$D", 40000000, 135000000)))
PY
)
t0=$(date +%s); RB3GPU_LF_CHECK=0 timeout 600 ropebwt3_amd/ropebwt3-amd build -d -o $D/gpu.fmd $FILES 2> gpurun_out/prof/r6_fmdchk_gpu.err; t1=$(date +%s)
a=$(md5sum $D/gpu.fmd | cut -c1-32); sa=$(stat -c %s $D/gpu.fmd); rm -f $D/gpu.fmd
RB3GPU_LF_CHECK=0 timeout 1500 ropebwt3_amd/ropebwt3-amd build -d --host-fmd -o $D/host.fmd $FILES 2> gpurun_out/prof/r6_fmdchk_host.err; t2=$(date +%s)
b=$(md5sum $D/host.fmd | cut -c1-32); sb=$(stat -c %s $D/host.fmd)
echo "$NH haplotypes x $LEN bp: .fmd packed on the GPU $a ($sa bytes, build $((t1 - t0)) s); encoded on the host $b ($sb bytes, build $((t2 - t1)) s): $([ "$a" = "$b" ] && echo IDENTICAL || echo DIFFERENT)" | tee gpurun_out/prof/r6_fmd_packer_check.txt
grep -h "packed\|W::" gpurun_out/prof/r6_fmdchk_gpu.err | tail -3 >> gpurun_out/prof/r6_fmd_packer_check.txt
rm -rf $D
