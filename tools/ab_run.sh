#!/bin/bash
# On the GPU box: run a command once per A/B library, alternating, `reps` times.
#   bash tools/ab_run.sh "<names>" <reps> <command...>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
names=$1; reps=$2; shift 2
for i in $(seq 1 $reps); do
    for n in $names; do
        echo "== $n (round $i)"
        TOME_HIP_LIB=$R/video-how-do-your-tokens-merge_amd/lib/ab_$n.so "$@" 2>/dev/null
    done
done
