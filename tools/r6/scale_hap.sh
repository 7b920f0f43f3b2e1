#!/bin/bash
# GPU box, round 6: BASELINE configs[4] in shape at real contig sizes: NH haplotypes (default 4) of a LEN-bp genome (default 3.1 Gbp; 0.1 % substitutions each,
# contigs of 40-135 Mbp), one FASTA file per haplotype, `ropebwt3-amd build -d` (each file's batch cut into GPU sub-batches at record boundaries), every 64th row of every merge
# LF-checked (the reference cannot make a golden of this size in the container) -> gpurun_out/prof/r6_scale_hap.json
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/prof
NH=${1:-4}; LEN=${2:-3100000000}; D=${SCALE_TMP:-/dev/shm}/rb3_hap_$$
t0=$(date +%s.%N)
FILES=$(python - <<PY
import sys
sys.path.insert(0, "$R")
from tools import gen_family
print(" ".join(gen_family.big_haplotype_files($NH, $LEN, "$D", 40000000, 135000000)))
PY
)
t1=$(date +%s.%N)
echo "generated $NH haplotypes of $LEN bp in $(python -c "print(round($t1 - $t0, 1))") s: $(du -sh $D | cut -f1)" >&2
RB3GPU_LF_CHECK=${LF_CHECK:-64} RB3_VERBOSE=4 timeout ${BUILD_TIMEOUT:-2400} ropebwt3_amd/ropebwt3-amd build -d ${EXTRA} -o $D/out.fmd $FILES 2> gpurun_out/prof/r6_scale_hap.err; rc=$?
t2=$(date +%s.%N)
ls -la $D/out.fmd >&2; md5sum $D/out.fmd | cut -c1-32 > gpurun_out/prof/r6_scale_hap_md5.txt; cat gpurun_out/prof/r6_scale_hap_md5.txt >&2
python tools/r6/scale_summary.py "cfg5-shape: $NH haplotypes x $LEN bp in contigs of 40-135 Mbp, build -d, rc=$rc, fmd $(stat -c %s $D/out.fmd 2>/dev/null) bytes" gpurun_out/prof/r6_scale_hap.err $(python -c "print(round($t2 - $t1, 2))") | tee gpurun_out/prof/r6_scale_hap.json | cut -c1-1500
grep -v "merge of \|\[prof\]" gpurun_out/prof/r6_scale_hap.err | tail -25 | cut -c1-260 > gpurun_out/prof/r6_scale_hap_tail.txt
gzip -f gpurun_out/prof/r6_scale_hap.err
rm -rf $D
