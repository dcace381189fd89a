"""Derived per-kernel figures from a round profile's pmc_summary.json (scripts/round_profile.sh): python scripts/pmc_derive.py <file>."""
import json, sys
d = json.load(open(sys.argv[1]))
print("# derived from %s (rocprofv3 --pmc, one counter group per pass, averages per launch; 1024^3 x 4 cameras)" % sys.argv[1].split("/")[-1])
print("# cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs); SQ_* cycle counters are in quad-cycles (x4); 1024 SIMDs, 8192 wave slots")
print("%-46s %9s %9s %9s %9s %9s %9s %9s %10s %10s" % ("kernel", "waves", "cycles", "resident", "VALUbusy", "SALUbusy", "wait_any", "wait_inst", "HBM_rd_MB", "HBM_wr_MB"))
for k in sorted(d):
    v = d[k]
    need = ("GRBM_GUI_ACTIVE", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "FETCH_SIZE", "WRITE_SIZE")
    if any(n not in v for n in need):
        continue
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    print("%-46s %9.0f %9.0f %9.0f %9.2f %9.2f %9.2f %9.2f %10.1f %10.1f" % (
        k.strip()[:46], v["SQ_WAVES"], cyc, v["SQ_WAVE_CYCLES"] * 4 / cyc, v["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024),
        v["SQ_ACTIVE_INST_SCA"] * 4 / (cyc * 1024), v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"],
        v["FETCH_SIZE"] * 1024 * 2 / 1e6, v["WRITE_SIZE"] * 1024 / 1e6))
