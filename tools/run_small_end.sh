# the small end: timings, then the kernel-trace stats of the device stepper on Theis (launches per solve)
set -eu
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun does)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
python3 $R/tools/small_end.py > $O/small_end.json 2> $O/small_end.err
rm -rf $O/small_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/small_prof -- python3 $R/tools/small_end.py theis > $O/small_end_prof.json 2> $O/small_end_prof.err
cd $R
python - <<PY
import csv, glob
f = glob.glob('gpurun_out/small_prof/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r['Calls']) for r in rows)
with open('gpurun_out/small_end_kernel_stats.txt', 'w') as o:
    o.write("total kernel launches %d (three Theis integrations + a warm-up: ~4 x 3 280 solves)\n" % tot)
    for r in rows[:20]:
        o.write("%-90s %7s calls %8.1f us avg %8.2f ms total\n" % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
