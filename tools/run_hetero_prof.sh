# kernel-trace stats of the bench's heterogeneous-conductivity block (sigma = 1 field on the 464^3 box: 2-3 PCG iterations per step at dt = 60 s)
set -eu
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun does)}
O=$R/gpurun_out
mkdir -p $O
rm -rf $O/hprof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hprof -- python3 $R/bench.py --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-other-configs --no-multi-iteration --no-profile > $O/hprof.json 2> $O/hprof.err
cd $R
python - <<PY
import csv, glob
f = glob.glob('gpurun_out/hprof/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
with open('gpurun_out/hprof_stats.txt', 'w') as o:
    for r in rows[:24]:
        o.write("%-90s %6s calls %9.1f us avg %8.2f ms total\n" % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
