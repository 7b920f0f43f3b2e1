#!/bin/bash
# config-5 shape: haplotypes of long contigs, one file per haplotype; walker spacing from rb3gpu_walker_step against fixed ones
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
python tools/gen_family.py haplotypes ${N:-6} ${L:-50000000} 5000000 /dev/shm/hap > /dev/null
for opt in "" "${FIX1:--k9}" "${FIX2:--k11}"; do
	for rep in 1 2; do
	./ropebwt3_amd/ropebwt3-amd build -d $opt -o /dev/shm/hap.fmd /dev/shm/hap/hap*.fa 2> /tmp/hap.err
	echo "opt '$opt': md5 $(md5sum < /dev/shm/hap.fmd | cut -c1-12) $(grep -o 'GPU merge path.*rebuild [0-9.]*)' /tmp/hap.err) $(grep -o 'Real time: [0-9.]* sec' /tmp/hap.err) fb=$(grep -o '[0-9]* merges redone' /tmp/hap.err)"
	done
done
