#!/bin/bash
# Build variants of libtome_hip.so with different -D flags (in the build container) for A/B runs on one GPU box:
#   bash tools/ab_lib.sh name1 "-DFOO=1" name2 "-DFOO=2" ...   ->  video-how-do-your-tokens-merge_amd/lib/ab_<name>.so
# On the box: TOME_HIP_LIB=$GRAFT_REPO_ROOT/video-how-do-your-tokens-merge_amd/lib/ab_<name>.so python tools/attn_bench.py
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/video-how-do-your-tokens-merge_amd
while [ $# -ge 2 ]; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math \
        -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function $2 -o $P/lib/ab_$1.so $P/csrc/tome_kernels.hip &
    shift 2
done
wait
ls -la $P/lib/
