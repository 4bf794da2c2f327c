#!/bin/bash
# rocprofv3 kernel stats of one model family through the reference-protocol harness (tools/model_benchmark.py).
#   bash tools/profile_family.sh <config name without .yaml> <tag> [--opts overrides...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
fam=$1; tag=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/tools/model_benchmark.py --cfg $R/configs/$fam.yaml --opts TRAIN.ENABLE False MODEL_BENCHMARK.WARMUP_ITERATIONS 3 MODEL_BENCHMARK.ITERATIONS 10 "$@" > $R/gpurun_out/prof_$tag.log 2>&1
f=$(ls $R/gpurun_out/prof_$tag/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/${tag}_kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/${tag}_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("$tag total device ms", round(tot/1e6,1))
for r in rows[:14]:
    print("%6.2f%% %8.1f us x%5s  %s" % (float(r["Percentage"]), float(r["AverageNs"])/1e3, r["Calls"], r["Name"][:100]))
PY
tail -1 $R/gpurun_out/prof_$tag.log | cut -c1-200
