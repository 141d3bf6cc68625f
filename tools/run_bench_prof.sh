set -eu
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun does)}
O=$R/gpurun_out
mkdir -p $O
rm -rf $O/bprof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bprof -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --no-multi-iteration > $O/bprof.json 2> $O/bprof.err
cd $R
f=$(ls gpurun_out/bprof/*/*kernel_stats.csv | head -1)
head -15 $f > gpurun_out/bprof_stats_head.csv
python tools/trace_gaps.py gpurun_out/bprof > gpurun_out/bprof_gaps.txt 2>&1 || true
