# Calibrates FETCH_SIZE on known byte counts: 17.18 GB streamed with dword and dwordx4 loads.
OUT=$GRAFT_REPO_ROOT/gpurun_out/calib; mkdir -p $OUT
[ -x $GRAFT_REPO_ROOT/scripts/micro/stream ] || hipcc --offload-arch=gfx950 -O3 -o $GRAFT_REPO_ROOT/scripts/micro/stream $GRAFT_REPO_ROOT/scripts/micro/stream.hip
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $GRAFT_REPO_ROOT/scripts/micro/stream > $OUT/fetch.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("$OUT/fetch/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:50]][0] += 1; agg[r["Kernel_Name"][:50]][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items()):
    print("%-52s calls=%3d FETCH_SIZE avg %.6g KB -> x1024 = %.4g B (true 1.718e10)  ratio true/counted = %.3f" % (k, n, v / n, v / n * 1024, 17179869184.0 / (v / n * 1024)))
PY
