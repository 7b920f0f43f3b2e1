#!/bin/bash
# GPU box: what do FETCH_SIZE / WRITE_SIZE report for the ACCESS PATTERN of k_chain?  (calibrate instead of a blanket x2)
# tools/ubench/lfchase: every octet reads one random 128-byte line per step (16 B per lane) out of an array far larger than
# L2 + Infinity Cache, optionally with one written-through 8-byte store per step to a random row: the bytes are known exactly.
# -> gpurun_out/prof/r4_fetch_size_calibration.json
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; export TMPDIR=/tmp; P=$R/gpurun_out/prof; mkdir -p $P; cd /tmp
[ -x $R/tools/ubench/lfchase ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $R/tools/ubench/lfchase $R/tools/ubench/lfchase.hip
MB=${MB:-8192}; W=${W:-262144}; ST=${ST:-200}
run() { # counter mode store
	rm -rf $P/cal.$1.$2.$3
	timeout 300 rocprofv3 --pmc $1 --kernel-trace -d $P/cal.$1.$2.$3 -o cal -- $R/tools/ubench/lfchase $MB 162000 $2 $W $ST $3 > $P/cal.$1.$2.$3.log 2>&1
	DB=$(ls $P/cal.$1.$2.$3/*_results.db $P/cal.$1.$2.$3/*/*_results.db 2>/dev/null | head -1)
	python - "$DB" "$1" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
r = con.execute("select avg(value), count(*) from counters_collection where counter_name = ? and kernel_name like '%k_walk%'", (sys.argv[2],)).fetchone()
print(r[0], r[1])
PY
	rm -rf $P/cal.$1.$2.$3
}
F0=$(run FETCH_SIZE 0 0); F2=$(run FETCH_SIZE 2 0); FS=$(run FETCH_SIZE 0 1); WS=$(run WRITE_SIZE 0 1); RQ=$(run TCC_EA0_RDREQ_sum 0 0)
python - "$MB" "$W" "$ST" "$F0" "$F2" "$FS" "$WS" "$RQ" > $P/r4_fetch_size_calibration.json <<'PY'
import json, sys
mb, w, st = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
f = lambda s: float(s.split()[0]) if s.split() and s.split()[0] != "None" else None
f0, f2, fs, ws, rq = map(f, sys.argv[4:9])
lines = w * st
out = {"tool": "tools/ubench/lfchase (tools/pmc_calibrate.sh): %d octets x %d steps, one random 128-byte line per step out of %d MB (>> L2 + Infinity Cache), per launch" % (w, st, mb),
       "known_read_bytes_per_launch": lines * 128, "FETCH_SIZE_KB_per_launch_slot_only": f0,
       "fetch_factor": round(lines * 128 / (f0 * 1024), 4) if f0 else None,
       "FETCH_SIZE_KB_per_launch_with_a_compact_directory_word": f2, "fetch_factor_with_directory_word": round(lines * 128 / (f2 * 1024), 4) if f2 else None,
       "TCC_EA0_RDREQ_per_launch": rq, "bytes_per_RDREQ": round(lines * 128 / rq, 2) if rq else None,
       "known_store_bytes_per_launch": lines * 8, "WRITE_SIZE_KB_per_launch_with_one_8B_store_per_step": ws,
       "write_amplification_of_random_8B_stores": round(ws * 1024 / (lines * 8), 3) if ws else None, "FETCH_SIZE_KB_per_launch_with_stores": fs,
       "how_to_use": "hbm bytes of a kernel with this access pattern = FETCH_SIZE_KB x 1024 x fetch_factor + WRITE_SIZE_KB x 1024"}
print(json.dumps(out, indent=1))
PY
cat $P/r4_fetch_size_calibration.json
