#!/bin/bash
# where does the wall time of `ropebwt3-amd build` on the 152 genomes go?  (its own timestamps)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
python tools/gen_mtb.py 152 4400000 /tmp/mtb_star_4400000 > /dev/null
for rep in 1 2; do
	S=$(date +%s.%N)
	./ropebwt3_amd/ropebwt3-amd build -d -o /tmp/out.fmd /tmp/mtb_star_4400000/g*.fa 2> /tmp/cli.err
	E=$(date +%s.%N)
	echo "wall $(echo "$E - $S" | bc) s"
done
grep -c "merged the partial" /tmp/cli.err
grep "constructed partial BWT\|merged the partial\|encoded the partial" /tmp/cli.err | sed -n '1,6p;145,152p;298,306p' | cut -c1-120
grep -v "constructed partial BWT\|merged the partial\|encoded the partial" /tmp/cli.err | tail -14 | cut -c1-220
