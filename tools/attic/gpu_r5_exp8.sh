#!/bin/bash
# round 5: why does the 320-relatives build redo 5 merges?  (list on the device / on the host) x (records nt / as before), 3 runs each
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from tools import gen_family
import json
from tests import util
man = json.load(open(os.path.join(util.GOLDEN, "MANIFEST.json")))
ent = man["family"]["relatives_320x200k"]
print(gen_family.relatives(*ent["spec"][1:], "/tmp/rel320.fa"))
PY
for cfg in "dev nt" "host nt" "dev old" "host old"; do
	set -- $cfg
	for rep in 1 2 3; do
		unset RB3_HOST_WALKERS LD_PRELOAD
		[ "$1" = host ] && export RB3_HOST_WALKERS=1
		[ "$2" = old ] && export LD_PRELOAD=$R/ropebwt3_amd/prof/norecnt.so
		RB3GPU_TENT_Q=${TQ:-1} ropebwt3_amd/ropebwt3-amd build -d -m300k /tmp/rel320.fa 2> gpurun_out/rel320.err | md5sum | cut -c1-8 | tr '\n' ' '
		unset LD_PRELOAD
		echo "$cfg: $(grep -o '[0-9]* merges redone without tentative records, [0-9]* needed the long settle pass' gpurun_out/rel320.err)"
	done
done
