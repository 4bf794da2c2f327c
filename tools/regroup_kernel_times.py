#!/usr/bin/env python3
"""Kernel-only per-layer times of the regrouped merge chain: runs tools/regroup_layers.py under rocprofv3
--kernel-trace and pairs every layer's 13 launches (3 warm-up + 10 timed) of k_merge_rows* with the bytes the tool prints;
per layer the MEDIAN of the 10 timed launches (device timestamps: no host effects, no event overhead).
    python tools/regroup_kernel_times.py [batch] [r] [frames] [tokens]        (starts rocprofv3 itself; GPU box)"""
import os
import re
import shutil
import statistics
import subprocess
import sys
import tempfile

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import rocpd_kernels  # noqa: E402

args = sys.argv[1:]
out = tempfile.mkdtemp(prefix="regroup_", dir="/tmp")
env = dict(os.environ, TMPDIR="/tmp")
p = subprocess.run(["rocprofv3", "--kernel-trace", "-d", out, "--", sys.executable,
                    os.path.join(ROOT, "tools", "regroup_layers.py")] + args, cwd="/tmp", env=env, capture_output=True, text=True)
layers = [(int(m.group(1)), int(m.group(2)), int(m.group(3)), float(m.group(4)), float(m.group(5)))
          for m in re.finditer(r"layer\s+(\d+):\s+(\d+) ->\s+(\d+) tokens per frame group,\s+([\d.]+) MB,\s+([\d.]+) us", p.stdout)]
if not layers:
    sys.exit(p.stdout[-2000:] + p.stderr[-2000:])
con, sym, disp = rocpd_kernels.open_db(out)
rows = [(en - st) / 1e3 for st, en in con.execute(
    f"select d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id where s.kernel_name like '%k_merge_rows%' "
    f"or s.kernel_name like '%k_merge_group%' order by d.start")]
per = len(rows) // len(layers)
tb = tt = 0.0
for i, (layer, t0, t1, mb, host_us) in enumerate(layers):
    mine = rows[i * per:(i + 1) * per][-10:]
    us = statistics.median(mine)
    tb += mb
    tt += us
    print(f"layer {layer:2d}: {t0:4d} -> {t1:4d}  {mb:7.1f} MB  kernel {us:7.2f} us (min {min(mine):7.2f})  {mb / us:5.2f} TB/s   "
          f"[back-to-back with host: {host_us:6.1f} us]")
print(f"total {tb:.1f} MB in {tt:.1f} us kernel time: {tb / tt:.2f} TB/s = {tb / tt / 8:.3f} of 8 TB/s  ({' '.join(args)})")
shutil.rmtree(out, ignore_errors=True)
