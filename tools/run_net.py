#!/usr/bin/env python3
"""Multi-view test loop with the reference's command line (tools/run_net.py there), on synthetic videos:
    python tools/run_net.py --cfg <configs/.../tome_VideoMAE_B_16_224_K400.yaml> --opts TRAIN.ENABLE False \
        TOME.ENABLE True TOME.R_VALUE 150 TOME.PROP_ATTN False"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]

from hosts.harness import main_run_net  # noqa: E402

if __name__ == "__main__":
    main_run_net()
