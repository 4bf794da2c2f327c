#!/bin/bash
# The model-level GPU tests under every measurement switch of tome/patch/*.py and hosts/timesformer.py (README table):
# each must pass.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
rc=0
ALL="TOME_FUSE_FC2 TOME_FUSE_NEXT TOME_FUSE_LN TOME_FUSE_ADD TOME_ATTN_KERNEL TOME_GELU_KERNEL TOME_SKIP_FIRST TOME_TRAJ_JOIN TOME_TRAJ_KEYS_ONLY TOME_SHORT_ATTN TOME_VIVIT_QKV TOME_ATTN_RESIDENT TOME_MERGE_XCD TOME_MERGE_EAGER TOME_MATCH_STREAM TOME_ROWS_KERNEL"
# (a subset: bash tools/test_switches.sh TOME_FUSE_LN TOME_MATCH_STREAM ...)
for sw in ${@:-$ALL}; do
    out=$(env $sw=0 python -m pytest tests/test_models_gpu.py tests/test_embed_gpu.py -q -m gpu -x 2>&1 | tail -1)
    echo "$sw=0: $out"
    case "$out" in *passed*) ;; *) rc=1 ;; esac
done
exit $rc
