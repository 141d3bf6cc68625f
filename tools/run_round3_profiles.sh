# the round's evidence in one call: default bench line, the driver's command, kernel-trace stats of the headline loop, PMC passes
set -eu
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun does)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python bench.py > $O/r03_bench464_default_run.json 2> $O/r03_bench464_default_run.err
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r03_bench464_driver_command.json 2> $O/r03_bench464_driver_command.err
cd /tmp
rm -rf $O/r03_kstats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_kstats -- python3 $R/bench.py --no-other-configs --no-multi-iteration --no-hetero --no-cpu-baseline > $O/r03_bench464_headline_only_under_rocprof.json 2> $O/r03_kstats.err
cd $R
cp $(ls gpurun_out/r03_kstats/*/*kernel_stats.csv | head -1) gpurun_out/r03_bench464_kernel_stats_headline_only.csv
python tools/trace_gaps.py gpurun_out/r03_kstats > gpurun_out/r03_bench464_trace_gaps.txt 2>&1 || true
bash tools/run_bench_pmc.sh
