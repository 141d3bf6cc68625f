set -eu
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun does)}
O=$R/gpurun_out
mkdir -p $O
rm -rf $O/aprof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/aprof -- python3 $R/tools/amg_box.py 256 3.0 0 > $O/aprof.log 2>&1
FV_AMG_VERBOSE=1 python3 $R/tools/amg_box.py 256 3.0 0 > $O/aprof_verbose.log 2>&1
cd $R
python - <<PY
import csv, glob
f = glob.glob('gpurun_out/aprof/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
with open('gpurun_out/aprof_stats.txt', 'w') as o:
    for r in rows[:40]:
        o.write("%-70s %6s calls %9.1f us avg %8.2f ms total %5.1f %%\n" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, 100*float(r['TotalDurationNs'])/tot))
PY
