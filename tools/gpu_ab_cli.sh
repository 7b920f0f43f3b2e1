#!/bin/bash
# GPU box: A/B of library variants on the CLI build of the 152 genomes (wall seconds):  tools/gpu_ab_cli.sh name1 name2 ...
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
python tools/gen_mtb.py 152 4400000 /tmp/mtb_star_4400000 > /dev/null
nproc
for rep in $(seq 1 ${REPS:-3}); do
for v in "$@"; do
	if [ "$v" = release ]; then PRE=; else PRE=$R/ropebwt3_amd/prof/$v.so; fi
	LD_PRELOAD=$PRE ./ropebwt3_amd/ropebwt3-amd build -d -o /tmp/out.fmd /tmp/mtb_star_4400000/g*.fa 2> /tmp/cli.err
	echo "$v $(grep 'Real time' /tmp/cli.err | cut -c1-80) | $(grep 'GPU merge path' /tmp/cli.err | cut -c20-110)"
done; done
