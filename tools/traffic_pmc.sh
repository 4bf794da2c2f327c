#!/bin/bash
# HBM traffic of the merge-path kernels: two counter passes (FETCH_SIZE, WRITE_SIZE; kernel trace only) over
# tools/pmc_run.py at bench.py's default batch, parsed into gpurun_out/<tag>_traffic.json (copy to profiles/traffic.json).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-traffic}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/${tag}_$c
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${tag}_$c -- python3 $R/tools/pmc_run.py \
        > $R/gpurun_out/${tag}_$c.log 2>&1 || { echo "$c pass failed"; tail -5 $R/gpurun_out/${tag}_$c.log; exit 1; }
done
cd $R && python3 tools/pmc_parse.py gpurun_out/${tag}_FETCH_SIZE gpurun_out/${tag}_WRITE_SIZE "" "${2:-round 4}" > gpurun_out/${tag}_parse.log 2>&1; tail -3 gpurun_out/${tag}_parse.log
cp $R/profiles/traffic.json $R/gpurun_out/${tag}_traffic.json
