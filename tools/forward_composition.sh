#!/bin/bash
# Device-time composition of N patched forwards of one family (tools/family_forward.py) from a rocprofv3 kernel trace.
#   bash tools/forward_composition.sh <family> <r> <batch> <tag>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
fam=$1; r=$2; batch=$3; tag=$4
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/tools/family_forward.py $fam $r $batch 5 > $R/gpurun_out/prof_$tag.log 2>&1
f=$(ls $R/gpurun_out/prof_$tag/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/${tag}_kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/${tag}_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("$tag: total device ms over 5 forwards (+ clip generation)", round(tot/1e6,1))
for r in rows[:30]:
    print("%6.2f%% %9.1f us x%5s  total %8.2f ms  %s" % (float(r["Percentage"]), float(r["AverageNs"])/1e3, r["Calls"], float(r["TotalDurationNs"])/1e6, r["Name"][:110]))
PY
