#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for a in "" "--full-upload"; do
timeout 600 python bench.py --only headline --steps 2 --warmup 1 $a > gpurun_out/exp12.json 2>/dev/null
python - <<'PY'
import json
d = json.loads(open("gpurun_out/exp12.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], d["phases_ms_per_step"], d["h2d"], d["config"]["fmd_identical_to_reference"])
PY
done
./ropebwt3_amd/ropebwt3-amd build -d -o /dev/shm/o.fmd $(python -c "
import sys; sys.path.insert(0,'.')
from tools import gen_mtb; print(' '.join(gen_mtb.generate(152, 4400000, '/dev/shm/mtbx')))") 2>&1 | grep -E "GPU sorter threads|Real time|GPU merge path"; md5sum /dev/shm/o.fmd
