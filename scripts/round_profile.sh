# Round profile: bench line, rocprofv3 kernel stats, PMC traffic passes.  Outputs under gpurun_out/<tag>/.
#   bash scripts/round_profile.sh <tag> [bench args, e.g. --workload config5]
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
WORKLOAD=real; MODE=lut
ARGS=("$@")
for ((i = 0; i < ${#ARGS[@]}; i++)); do
  [ "${ARGS[$i]}" = "--workload" ] && WORKLOAD=${ARGS[$((i + 1))]}
  [ "${ARGS[$i]}" = "--mode" ] && MODE=${ARGS[$((i + 1))]}
done
cd /tmp && export TMPDIR=/tmp
if [ -z "$PMC_ONLY" ]; then
cd $GRAFT_REPO_ROOT && python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_headline -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline --only-headline --e2e-steps 0 --prewarm-seconds 0 > $OUT/stats_headline.log 2>&1
fi
SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" \
      "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
      "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA")
for set in "${SETS[@]}"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$tag -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_$tag.log 2>&1 || echo "pass $tag failed"
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections, json, os
out, workload, mode = "$OUT", "$WORKLOAD", "$MODE"
if os.path.exists(out + "/bench.json"): print(open(out + "/bench.json").read()[:3000])
for d in ("stats", "stats_headline"):
    for f in glob.glob(out + "/%s/*/*kernel_stats.csv" % d):
        print(d); print(open(f).read())
agg = collections.defaultdict(lambda: [0, 0.0])
for f in sorted(glob.glob(out + "/pmc_*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][-48:], r["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
rows = {}
for (kn, cn), (n, v) in sorted(agg.items()):
    if any(t in kn for t in ("k_lut", "k_carve", "k_emit", "k_scan", "k_build", "k_prep", "k_finish", "k_cull", "k_brick", "k_voxel", "k_assemble")):
        print("%-42s %-18s calls=%3d avg=%.6g" % (kn, cn, n, v / n))
        rows.setdefault(kn, {})[cn] = v / n
json.dump(rows, open(out + "/pmc_summary.json", "w"), indent=1)
# HBM bytes per launch of the kernels bench.py may build its roofline objects on (FETCH_SIZE doubled: profiles/r01_fetch_size_calibration.txt);
# keys as bench.py looks them up: <kernel>_<mode>_<workload>_<grid>_g<ranks>
grid = {"real": "1024x1024x1024", "config5": "512x512x512", "big2048": "2048x2048x1023"}[workload]
traffic = {}
needles = {"k_emit_busy": "k_emit_busy<true, true", "k_brick_words": "k_brick_words", "k_voxel_words": "k_voxel_words<true", "k_cull_bricks": "k_cull_bricks"}
if mode == "fused":
    needles.update({"k_emit_busy": "k_emit_busy<", "k_voxel_words": "k_voxel_words<false"})
for name, needle in needles.items():
    for kn, d in rows.items():
        if needle in kn and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            rd, wr = d["FETCH_SIZE"] * 1024 * 2, d["WRITE_SIZE"] * 1024
            traffic["%s_%s_%s_%s_g1" % (name, mode, workload, grid)] = {
                "kernel": kn.strip(), "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr,
                "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, KB per launch; "
                          "FETCH_SIZE x2 (gfx950 correction, calibrated in profiles/r01_fetch_size_calibration.txt)"}
            break
if workload == "real" and mode == "lut":
    for kn, d in rows.items():
        if "k_lut_first" in kn and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            rd, wr = d["FETCH_SIZE"] * 1024 * 2, d["WRITE_SIZE"] * 1024
            traffic["lut_stream_real_1024x1024x1024_g1"] = {"kernel": kn.strip(), "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr,
                "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, KB per launch; FETCH_SIZE x2 (gfx950 correction)"}
json.dump(traffic, open(out + "/traffic.json", "w"), indent=1)
print(json.dumps(traffic, indent=1))
PY
# the traces themselves are large (gpurun copies back at most 64 MiB): keep a steady-state timeline and the summaries only
python3 scripts/timeline.py $OUT/stats_headline 400 60 > $OUT/timeline_headline.txt 2>/dev/null
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
du -sh $OUT
