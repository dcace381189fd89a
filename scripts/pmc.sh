# usage: bash scripts/pmc.sh <outdir> <bench args...>   -- separate rocprofv3 --pmc passes (gfx950 slot limits)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/$tag.log 2>&1 || echo "pass $tag failed"
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/*/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:60], r["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for (kn, cn), (n, v) in sorted(agg.items()):
        if "carve" in kn or "emit" in kn or "build_lut" in kn:
            print("%-62s %-22s calls=%3d avg=%.6g" % (kn, cn, n, v / n))
PY
