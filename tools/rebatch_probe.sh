export TMPDIR=/tmp; R=$PWD; mkdir -p $R/gpurun_out/prof
python tools/e2e_mtb.py 12 4400000 --no-ref /tmp/e2e12 2>&1 | tail -5
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/reb -o reb -- $R/ropebwt3_amd/ropebwt3-amd build -d --rebatch -m40m -p4 /tmp/e2e12/g0*.fa > /tmp/reb.fmd 2> $R/gpurun_out/prof/reb.log
cd $R; python tools/prof_summary.py stats gpurun_out/prof/reb/reb_results.db gpurun_out/prof/reb_kernel_stats.txt | head -14; grep "GPU merge path" gpurun_out/prof/reb.log
