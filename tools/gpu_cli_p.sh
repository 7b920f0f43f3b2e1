#!/bin/bash
# GPU box: wall time of `ropebwt3-amd build -d` on the 152 genomes against the number of sorter threads (-p)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
python tools/gen_mtb.py 152 4400000 /tmp/mtb_star_4400000 > /dev/null
for p in 1 2 3 1 2 3; do
	./ropebwt3_amd/ropebwt3-amd build -d -p$p -o /tmp/out.fmd /tmp/mtb_star_4400000/g*.fa 2> /tmp/cli.err
	echo "-p$p: $(grep 'Real time' /tmp/cli.err | cut -c1-80)  $(md5sum /tmp/out.fmd | cut -c1-32)  $(grep 'GPU merge path' /tmp/cli.err | cut -c20-120)"
done
