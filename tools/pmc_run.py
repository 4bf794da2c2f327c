#!/usr/bin/env python3
"""Workload for the PMC passes: one walk of the 12-layer merge-path chain of the bench workload
(VideoMAE-B 16x224, r=16, bf16, bench.py's default batch), each kernel launched a few times.  Run under
  rocprofv3 --kernel-trace --pmc FETCH_SIZE  --output-format csv -d <dir> -- python3 tools/pmc_run.py
  rocprofv3 --kernel-trace --pmc WRITE_SIZE  --output-format csv -d <dir> -- python3 tools/pmc_run.py
(separate passes: FETCH_SIZE takes 3 of the 4 TCC slots), then tools/pmc_parse.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch  # noqa: E402

import bench  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else bench.DEFAULT_BATCH
with torch.no_grad():
    stats = bench.measure_kernels(batch, 1568, 16, torch.device("cuda", 0), reps=2)
torch.cuda.synchronize()
print({k: round(v["ms"], 4) for k, v in stats.items()})
