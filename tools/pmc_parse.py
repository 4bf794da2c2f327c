#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/pmc_run.py into
profiles/traffic.json: HBM-side bytes per launch of each merge-path kernel, averaged over the launches of
the 12-layer chain.  Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section)
prescribes for gfx950: both counters are in KiB; FETCH_SIZE reports exactly half of the bytes of a wide
(16 B/lane) coalesced read, so it is doubled; WRITE_SIZE is exact for 16 B/lane streaming stores.
The calibration row is a PyTorch fp32->bf16 copy of known size that runs in the same process."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(d):
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
    agg = {}
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        key = None
        for k in ("k_merge_rows", "k_scores_rowmax", "k_scores_filter", "k_exact_rows", "k_unit_rows",
                  "k_unit_rows_heads", "k_unit_rows_f", "k_rank_select", "k_add_ln_rows", "bfloat16_copy_kernel"):
            if k in name:
                key = k
        if key is None:
            continue
        agg.setdefault(key, []).append((float(row["Counter_Value"]), int(row["Grid_Size"])))
    return agg


def main():
    fetch = load(sys.argv[1])
    write = load(sys.argv[2])
    sys.path.insert(0, ROOT)
    import bench
    batch = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] else bench.DEFAULT_BATCH
    out = {"_units": f"bytes per launch (mean over the launches of one 12-layer chain, batch {batch})",
           "_correction": "FETCH_SIZE KiB x2 (gfx950 wide-read under-count) + WRITE_SIZE KiB"}
    for k in fetch:
        f = sum(v for v, _ in fetch[k]) / len(fetch[k]) * 1024 * 2
        w = sum(v for v, _ in write.get(k, [(0, 0)])) / max(1, len(write.get(k, []))) * 1024
        if k == "bfloat16_copy_kernel":
            # calibration: the largest copy reads B*T*C fp32 and writes bf16
            big = max(fetch[k], key=lambda t: t[0])
            out["_calibration_copy_fetch_bytes_x2"] = big[0] * 1024 * 2
            continue
        out[k] = {"fetch": round(f), "write": round(w), "total": round(f + w), "launches": len(fetch[k])}
    flat = {k: v["total"] for k, v in out.items() if isinstance(v, dict)}
    flat["_batch"] = batch
    import datetime
    flat["_collected"] = sys.argv[4] if len(sys.argv) > 4 else datetime.date.today().isoformat()
    flat["_detail"] = out
    with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as fh:
        json.dump(flat, fh, indent=1)
    print(json.dumps(flat, indent=1))


if __name__ == "__main__":
    main()
