#!/usr/bin/env python3
"""Summarise the counter passes of tools/attn_pmc.sh: per attention kernel, mean counter values per launch and the
ratios that say where the time goes (SQ_* cycle counters are in quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES in cycles =
32 per v_mfma_f32_32x32x16 -- MI355X_MICROARCH.md, cycle-constants table)."""
import collections
import csv
import glob
import json
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "attention" not in n:
                continue
            key = n.split("(")[0].replace("void ", "")
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "attention" in n:
                dur[n.split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {}
for k, cs in agg.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    o = {"counters_per_launch": {c: round(x) for c, x in sorted(m.items())}}
    if k in dur:
        o["avg_launch_us_under_pmc"] = round(sum(dur[k]) / len(dur[k]), 1)
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        o["share_of_wave_cycles"] = {
            "issuing (SQ_ACTIVE_INST_ANY)": round(m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
            "parked on s_waitcnt / barrier (SQ_WAIT_ANY)": round(m.get("SQ_WAIT_ANY", 0) / wc, 3),
            "issue-stalled (SQ_WAIT_INST_ANY)": round(m.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
            "VALU active (SQ_ACTIVE_INST_VALU)": round(m.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3),
        }
    if m.get("GRBM_GUI_ACTIVE") and m.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD (32 per v_mfma_f32_32x32x16), summed over the chip's SIMDs;
        # the cycles available to them are (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8) x 1024 SIMDs.  (Round 2's
        # version of this script divided by SQ_BUSY_CU_CYCLES x 16 and printed 0.089 for a pipe that was 33 % busy.)
        cycles = m["GRBM_GUI_ACTIVE"] / 8.0
        o["shader_cycles_per_launch"] = round(cycles)
        if o.get("avg_launch_us_under_pmc"):
            o["clock_ghz_under_pmc"] = round(cycles / o["avg_launch_us_under_pmc"] / 1e3, 3)
        o["mfma_pipe_busy_fraction"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024.0), 3)
        o["mfma_coexec_with_valu_fraction_of_mfma_busy"] = round(
            m.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0) / m["SQ_VALU_MFMA_BUSY_CYCLES"], 3)
    if m.get("SQ_INSTS_MFMA"):
        o["valu_instructions_per_mfma"] = round(m.get("SQ_INSTS_VALU", 0) / m["SQ_INSTS_MFMA"], 2)
    out[k] = o
print(json.dumps(out, indent=1))
