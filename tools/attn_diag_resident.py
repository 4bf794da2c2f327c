#!/usr/bin/env python3
"""Phase timeline of k_resident_attention's workgroups from the DIAGNOSTIC build (-DATT_DIAG: s_memrealtime stamps, 100 MHz;
never the shipped library):  bash tools/ab_lib.sh diag "-DATT_DIAG";
TOME_HIP_LIB=.../lib/ab_diag.so python tools/attn_diag_resident.py [B H N]      (spatial form, N = keys = queries)"""
import ctypes
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tome import _abi  # noqa: E402

B, H, N = [int(v) for v in (sys.argv[1:4] + ["512", "12", "197"][len(sys.argv) - 1:])]
L = _abi.lib()
if not hasattr(L, "tome_attn_diag_read"):
    raise SystemExit("this libtome_hip.so is not the diagnostic build (-DATT_DIAG)")
dev = torch.device("cuda", 0)
qkv = torch.randn(B, N, 3, H, 64, device=dev).bfloat16()
q, k, v = qkv.permute(2, 0, 3, 1, 4)
for _ in range(3):
    _abi.prop_attention(q, k, v, None, 0.125)
torch.cuda.synchronize()
WGS, NST = 8192, 8
buf = (ctypes.c_ulonglong * (WGS * 8 * NST))()
L.tome_attn_diag_read.argtypes = [ctypes.c_void_p, ctypes.c_int64]
assert L.tome_attn_diag_read(buf, WGS * 8 * NST) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(WGS, 8, NST).astype(np.int64)
n = min(WGS, (B * H + 7) // 8 * 8)
st = st[:n]
t0 = st[:, :, 0].min()
names = ["entry", "staged(LDS written)", "barrier passed", "Q~ ready", "block loop done", "stores issued"]
w0 = st[:, 0, :]  # wave 0 of each workgroup
print(f"{B}x{H}x{N}: {n} workgroups; phase durations of wave 0, 10 ns ticks -> us (median over workgroups)")
for i in range(1, 6):
    d = (w0[:, i] - w0[:, i - 1]) / 100.0
    print(f"  {names[i - 1]:>22} -> {names[i]:<22} median {np.median(d):6.2f} us  p90 {np.percentile(d, 90):6.2f}")
life = (st[:, :, 5].max(1) - st[:, :, 0].min(1)) / 100.0
print(f"  workgroup lifetime (entry of first wave .. stores of last): median {np.median(life):.2f} us, p90 {np.percentile(life, 90):.2f}")
span = (st[:, :, 5].max() - t0) / 100.0
print(f"  launch span by the stamps: {span:.1f} us; sum of lifetimes / span = {life.sum() / span:.1f} workgroups in flight on average "
      f"({life.sum() / span / 256:.2f} per CU)")
