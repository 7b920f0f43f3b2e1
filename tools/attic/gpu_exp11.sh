#!/bin/bash
# rank phase per round against the number of indexed relatives (K up to 400 genomes of the mtb star)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
python tools/gen_mtb.py 400 4400000 /dev/shm/mtb400 > /dev/null
./ropebwt3_amd/ropebwt3-amd build -d ${EXTRA:-} -o /dev/shm/out400.fmd /dev/shm/mtb400/g*.fa 2> gpurun_out/exp11.err
md5sum /dev/shm/out400.fmd
grep "merge_core" gpurun_out/exp11.err | awk '{n++; if (n%25==0 || n>=395) print n, $0}' | cut -c1-200
grep -E "GPU merge path|k_chain|run-space|Real time" gpurun_out/exp11.err | cut -c1-250
