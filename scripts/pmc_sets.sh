# usage: bash scripts/pmc_sets.sh <tag> <kernel substring> <script args...>: SQ / TA / TCP counter passes of scripts/exp_pmc.py
TAG=$1; KERN=$2; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
      "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES" \
      "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUSY_avr" \
      "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
      "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
      "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum")
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/scripts/exp_pmc.py "$@" > $OUT/p$i.log 2>&1 || echo "pass $i ($set) failed: $(tail -2 $OUT/p$i.log)"
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: [0, 0.0])
for f in sorted(glob.glob("$OUT/p*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "$KERN" in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
            agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
rows = {}
for (kn, cn), (n, v) in sorted(agg.items()):
    print("%-40s %-36s calls=%3d avg=%.6g" % (kn, cn, n, v / n))
    rows.setdefault(kn, {})[cn] = v / n
json.dump(rows, open("$OUT/summary.json", "w"), indent=1)
PY
