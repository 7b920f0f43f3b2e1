#!/bin/bash
# the tree merge of `build --gpus N` on one device: how long do the whole-index merges take?
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
python tools/gen_mtb.py 152 4400000 /dev/shm/mtb > /dev/null
for n in ${NS:-2 8}; do
	./ropebwt3_amd/ropebwt3-amd build -d --gpus $n -o /dev/shm/out$n.fmd /dev/shm/mtb/g*.fa 2> gpurun_out/exp9_n$n.err
	echo "== --gpus $n: md5 $(md5sum < /dev/shm/out$n.fmd)"
	grep -E "merged [0-9]{8,} symbols|tree merge|table of tentative|wall|Real time|redoing|other slices" gpurun_out/exp9_n$n.err | cut -c1-220
done
