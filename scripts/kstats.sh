# usage: bash scripts/kstats.sh <tag> <python script + args...>: rocprofv3 kernel stats of a script, printed compactly
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG -- python3 "$@" > $GRAFT_REPO_ROOT/gpurun_out/$TAG.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/$TAG.log | cut -c1-200
python3 - <<PY
import csv,glob
for r in csv.DictReader(open(glob.glob("$GRAFT_REPO_ROOT/gpurun_out/$TAG/*/*kernel_stats.csv")[0])):
    print("%-64s %5s %12.1f %10s" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]), r["MinNs"]))
PY
