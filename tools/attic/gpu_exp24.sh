#!/bin/bash
# rebuild time per round of the first 70 rounds of the star build (which kernels: the CLI's per-merge lines)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
python tools/gen_mtb.py 70 4400000 /tmp/mtb_star_4400000 > /dev/null
./ropebwt3_amd/ropebwt3-amd build -d --host-sort -o /tmp/out.fmd $(ls /tmp/mtb_star_4400000/g*.fa | head -70) 2> /tmp/cli.err
grep "::merge_core" /tmp/cli.err | awk '{n++; printf "%d:%s ", n, $(NF-1)} END{print ""}'
grep "run-space rebuild:" /tmp/cli.err | cut -c1-200
