export TMPDIR=/tmp; R=$PWD; mkdir -p $R/gpurun_out/prof
python tools/probe_rebuild.py prep ${1:-40} > /dev/null 2>&1
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  i=$((i+1)); rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/prof/p1sq$i -o p1 -- python $R/tools/probe_rebuild.py time > /dev/null 2>&1
done
cd $R
for i in 1 2 3; do python - gpurun_out/prof/p1sq$i/p1_results.db <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cols = [d[0] for d in con.execute("select * from counters_collection limit 1").description]
ncol = "counter_name" if "counter_name" in cols else "name"
kcol = "kernel_name" if "kernel_name" in cols else "name"
for k, c, n, a in con.execute("select %s, %s, count(*), avg(value) from counters_collection group by %s, %s" % (kcol, ncol, kcol, ncol)):
    if "k_pass1w" in k and "Lb0" in k or ("k_pass1w<false" in k): print("%-22s %-22s calls %4d avg %16.1f" % ("k_pass1w<merge>", c, n, a))
PY
done
