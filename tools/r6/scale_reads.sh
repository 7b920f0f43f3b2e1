#!/bin/bash
# GPU box, round 6: BASELINE configs[3] in shape at 1/10: N synthetic 150-bp reads (default 60 M = 18.1 G symbols with both strands, ~30x of a 300 Mbp
# genome, 1 % errors), `ropebwt3-amd build -L -d -m7g` (three -m7g batches, each cut into GPU sub-batches), every 64th row of every merge LF-checked -> gpurun_out/prof/r6_scale_reads.json
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/prof
N=${1:-60000000}; D=${SCALE_TMP:-/dev/shm}; F=$D/rb3_reads_$N.txt
t0=$(date +%s.%N); python tools/gen_reads.py $N $F > /dev/null; t1=$(date +%s.%N)
echo "generated $N reads in $(python -c "print(round($t1 - $t0, 1))") s: $(ls -la $F | awk '{print $5}') bytes" >&2
RB3GPU_LF_CHECK=${LF_CHECK:-64} RB3_VERBOSE=4 timeout ${BUILD_TIMEOUT:-1500} ropebwt3_amd/ropebwt3-amd build -L -d -m7g ${EXTRA} -o $D/rb3_reads_$N.fmd $F 2> gpurun_out/prof/r6_scale_reads.err; rc=$?
t2=$(date +%s.%N)
ls -la $D/rb3_reads_$N.fmd >&2; md5sum $D/rb3_reads_$N.fmd | cut -c1-32 > gpurun_out/prof/r6_scale_reads.md5
python tools/r6/scale_summary.py "cfg4-shape: $N reads x 150 bp, build -L -d -m7g, rc=$rc, fmd $(stat -c %s $D/rb3_reads_$N.fmd 2>/dev/null) bytes" gpurun_out/prof/r6_scale_reads.err $(python -c "print(round($t2 - $t1, 2))") | tee gpurun_out/prof/r6_scale_reads.json | cut -c1-1500
grep -v "merge of \|\[prof\]" gpurun_out/prof/r6_scale_reads.err | tail -25 | cut -c1-260 > gpurun_out/prof/r6_scale_reads_tail.txt
gzip -f gpurun_out/prof/r6_scale_reads.err
rm -f $F $D/rb3_reads_$N.fmd
