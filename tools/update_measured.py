#!/usr/bin/env python3
"""Fold what GPU runs of the model tests measured (TOME_RECORD_MEASURED=1 -> gpurun_out/measured_seen.json) into the
tracked tests/golden/measured.json: per fixture the LARGEST logit errors and the SMALLEST per-layer group agreement seen
over all folded runs.  The tests then hold a run to 1.5x / minus 2 points of these instead of one global tolerance.
    python tools/update_measured.py [gpurun_out/measured_seen.json ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "measured.json")


def main():
    paths = sys.argv[1:] or [os.path.join(ROOT, "gpurun_out", "measured_seen.json")]
    cur = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for p in paths:
        for name, vals in json.load(open(p)).items():
            rec = cur.setdefault(name, {})
            for k, v in vals.items():
                if k == "agree":
                    rec[k] = [min(a, b) for a, b in zip(rec[k], v)] if k in rec and len(rec[k]) == len(v) else v
                else:
                    rec[k] = max(rec.get(k, 0.0), v)
            rec["runs"] = rec.get("runs", 0) + 1
    with open(OUT, "w") as f:
        json.dump(cur, f, indent=1, sort_keys=True)
    print(f"{OUT}: {len(cur)} fixtures")


if __name__ == "__main__":
    main()
