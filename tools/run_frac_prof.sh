set -eu
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun does)}
O=$R/gpurun_out
mkdir -p $O
rm -rf $O/fprof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fprof -- python3 $R/tools/frac_step.py 200 > $O/fprof.log 2>&1
cd $R
head -12 $(ls gpurun_out/fprof/*/*kernel_stats.csv | head -1) | cut -c1-160 > gpurun_out/fprof_stats_head.csv
