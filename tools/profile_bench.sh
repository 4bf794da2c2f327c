#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command (headline workload only), summaries copied next to the
# bench line it printed.  Usage (on the GPU box, through gpurun):  bash tools/profile_bench.sh <tag>
# -> gpurun_out/<tag>_kernel_stats.csv, gpurun_out/<tag>_bench_line_under_rocprof.json
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$tag
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- \
    python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-also > $R/gpurun_out/${tag}_bench_line_under_rocprof.json 2> $R/gpurun_out/prof_$tag.err
rc=$?
f=$(ls $R/gpurun_out/prof_$tag/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/${tag}_kernel_stats.csv && head -12 $R/gpurun_out/${tag}_kernel_stats.csv | cut -c1-200
exit $rc
