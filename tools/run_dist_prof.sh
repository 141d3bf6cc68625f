# one rank through the row-block driver (FV_BENCH_FORCE_DIST=1) under rocprofv3: which kernels a step of the block launches
set -eu
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun does)}
O=$R/gpurun_out
mkdir -p $O
rm -rf $O/dprof
export FV_BENCH_FORCE_DIST=1
export MASTER_PORT=29655
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dprof -- python3 $R/bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-other-configs --no-multi-iteration --no-hetero $DPROF_EXTRA > $O/dprof.json 2> $O/dprof.err
cd $R
f=$(ls gpurun_out/dprof/*/*kernel_stats.csv | head -1)
head -24 $f > gpurun_out/dprof_stats_head.csv
python tools/trace_gaps.py gpurun_out/dprof > gpurun_out/dprof_gaps.txt 2>&1 || true
