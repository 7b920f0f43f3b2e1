#!/bin/bash
# GPU box: `build --gpus 4 --interval` on N synthetic reads (every handle on the one device): what each interval's handle held at its peak
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
N=${1:-1000000}
python tools/gen_reads.py $N /tmp/reads_iv.txt > /dev/null
for g in 4; do
	$R/ropebwt3_amd/ropebwt3-amd build -L -d -m${2:-60m} --gpus $g --interval /tmp/reads_iv.txt 2> gpurun_out/ivpeak.err | md5sum
	grep "interval [0-9] on device\|intervals\|rebalanc\|Real time\|device memory of" gpurun_out/ivpeak.err | tail -12
done
$R/ropebwt3_amd/ropebwt3-amd build -L -d -m${2:-60m} /tmp/reads_iv.txt 2> gpurun_out/ivpeak1.err | md5sum
grep "Real time\|device memory of" gpurun_out/ivpeak1.err | tail -3
