#!/usr/bin/env python3
"""Phase times of k_prop_attention_stream's third item per workgroup (DIAGNOSTIC build, -DATT_DIAG).
   TOME_HIP_LIB=.../lib/ab_diag.so python tools/attn_diag_stream.py [B H N]"""
import ctypes
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tome import _abi  # noqa: E402

B, H, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (128, 12, 1536)
L = _abi.lib()
if not hasattr(L, "tome_attn_diag_read"):
    raise SystemExit("this libtome_hip.so is not the diagnostic build (-DATT_DIAG)")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
qkv = torch.randn(B, N, 3, H, 64, device=dev).bfloat16()
q, k, v = qkv.permute(2, 0, 3, 1, 4)
t_begin = time.perf_counter()
while time.perf_counter() - t_begin < 1.5:
    for _ in range(20):
        _abi.prop_attention(q, k, v, None, 0.125)
    torch.cuda.synchronize()
WGS, NS = 8192, 8
buf = np.zeros(WGS * 8 * NS, dtype=np.uint64)
L.tome_attn_diag_read.restype = ctypes.c_int
L.tome_attn_diag_read.argtypes = [ctypes.c_void_p, ctypes.c_int64]
assert L.tome_attn_diag_read(buf.ctypes.data, buf.size) == 0
st = buf.reshape(WGS, 8, NS).astype(np.int64)[:256]
t = st[:, :, :6] * 0.01
names = ["tile loop", "loop done->last product, stores issued", "->next tile staged, behind the barrier",
         "->Q fragment ready", "->first tile's weights"]
ntile = (N + 63) // 64
for w in (0, 7):
    print(f"wave {w}:   (per step: wait for the tile requested one step earlier {np.median(st[:, w, 6]) * 0.01 / (ntile - 1):.3f} us, "
          f"request + barrier {np.median(st[:, w, 7]) * 0.01 / (ntile - 1):.3f} us; instrumented steps are slower)")
    for i, nm in enumerate(names):
        d = t[:, w, i + 1] - t[:, w, i]
        print(f"   {nm:38s} median {np.median(d):7.2f} us  mean {d.mean():7.2f}")
