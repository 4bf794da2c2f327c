#!/bin/bash
# SQ counters of the merge-path kernels (one walk of the 12-layer chain, tools/pmc_run.py).  On the GPU box:
#   bash tools/merge_pmc.sh <tag>  -> gpurun_out/<tag>_sq.txt  (per kernel: waves, vector / scalar instructions per wave,
#   share of the SIMD-cycles in which a vector instruction was issuing)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-merge}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${tag}_sq
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY \
    --output-format csv -d $R/gpurun_out/${tag}_sq -- python3 $R/tools/pmc_run.py ${2:-128} > $R/gpurun_out/${tag}_sq.log 2>&1 || { echo "pass failed"; tail -5 $R/gpurun_out/${tag}_sq.log; exit 1; }
python3 - $R/gpurun_out/${tag}_sq > $R/gpurun_out/${tag}_sq.txt <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:60]
    if not k.startswith(("void k_", "k_")):
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
for k, c in agg.items():
    w = c["SQ_WAVES"] or 1
    print(f"{k:62s} dispatches {len(disp[k]):4d} waves/dispatch {w / len(disp[k]):9.0f}  VALU/wave {c['SQ_INSTS_VALU'] / w:7.1f}  "
          f"SALU/wave {c['SQ_INSTS_SALU'] / w:7.1f}  wave-cycles/wave {c['SQ_WAVE_CYCLES'] / w:8.0f}  "
          f"active VALU / busy {c['SQ_ACTIVE_INST_VALU'] / max(c['SQ_BUSY_CYCLES'], 1):.3f}  "
          f"active any / busy {c['SQ_ACTIVE_INST_ANY'] / max(c['SQ_BUSY_CYCLES'], 1):.3f}")
PY
cat $R/gpurun_out/${tag}_sq.txt
