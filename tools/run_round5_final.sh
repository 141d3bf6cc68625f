# round 5, the final library: the driver's command, then the SAME command under rocprofv3 --kernel-trace --stats on the same box (the kernels'
# average durations in the csv against the avg_launch_ms the bench line measures with HIP events)
set -eu
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun does)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_final_driver_command.json 2> $O/r05_final_driver_command.err
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r05_final_kstats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05_final_kstats -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_final_driver_command_under_rocprof.json 2> $O/r05_final_kstats.err
cd $R
cp $(ls gpurun_out/r05_final_kstats/*/*kernel_stats.csv | head -1) gpurun_out/r05_final_driver_command_kernel_stats.csv
head -12 gpurun_out/r05_final_driver_command_kernel_stats.csv | cut -c1-160
