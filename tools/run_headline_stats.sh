# kernel-trace stats of the headline loop alone (the final library of a round): bash tools/run_headline_stats.sh TAG (inside gpurun)
set -eu
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun does)}
O=$R/gpurun_out
T=${1:-final}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${T}_kstats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_kstats -- python3 $R/bench.py --no-other-configs --no-multi-iteration --no-hetero --no-cpu-baseline > $O/${T}_headline_only_under_rocprof.json 2> $O/${T}_kstats.err
cd $R
cp $(ls gpurun_out/${T}_kstats/*/*kernel_stats.csv | head -1) gpurun_out/${T}_kernel_stats_headline_only.csv
head -4 gpurun_out/${T}_kernel_stats_headline_only.csv
