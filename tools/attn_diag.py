#!/usr/bin/env python3
"""Phase timeline of k_prop_attention's query blocks from the DIAGNOSTIC build (-DATT_DIAG: s_memrealtime stamps; never
the shipped library):   bash tools/ab_lib.sh diag "-DATT_DIAG";  TOME_HIP_LIB=.../lib/ab_diag.so python tools/attn_diag.py [B H N]
Prints, per phase, the median / mean time over the recorded workgroups (wave 0), the skew between the first and the
last wave of a workgroup, and how many CU slots were busy on average (sum of workgroup durations / recorded span:
256 = no gap between consecutive workgroups of a CU).
    TOME_ATTN_STREAM=0 python tools/attn_diag.py ...   # k_prop_attention (one workgroup per query block)
    python tools/attn_diag_stream.py ...                # k_prop_attention_stream (persistent workgroups)"""
import collections
import ctypes
import os
import sys

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
import time  # noqa: E402
t_begin = time.perf_counter()
while time.perf_counter() - t_begin < 1.5:  # the clock the chip holds under this load, not its idle state
    for _ in range(20):
        _abi.prop_attention(q, k, v, None, 0.125)
    torch.cuda.synchronize()
WGS, NS = 8192, 8
buf = np.zeros(WGS * 8 * NS, dtype=np.uint64)
L.tome_attn_diag_read.restype = ctypes.c_int
L.tome_attn_diag_read.argtypes = [ctypes.c_void_p, ctypes.c_int64]
assert L.tome_attn_diag_read(buf.ctypes.data, buf.size) == 0
st = buf.reshape(WGS, 8, NS).astype(np.int64)
nwg = min(WGS, (B * H + 7) // 8 * 8 * ((N + 255) // 256))
st = st[:nwg]
t = st[:, :, :7] * 0.01  # 100 MHz ticks -> us
names = ["entry->Q ready", "Q ready->tile 0 in LDS", "tile 0 in LDS->first weights", "first weights->loop done",
         "loop done->stores issued", "stores issued->stores done"]
w0 = t[:, 0, :]
full = (t[:, 7, 4] > 0)  # workgroups whose last wave is active
print(f"B {B} H {H} N {N}: {nwg} workgroups recorded, {int(full.sum())} with all eight waves active")
for i, nm in enumerate(names):
    d = (w0[:, i + 1] - w0[:, i])[full]
    print(f"  wave 0  {nm:34s} median {np.median(d):7.2f} us  mean {d.mean():7.2f}")
tot = (w0[:, 6] - w0[:, 0])[full]
print(f"  wave 0  entry->stores done                 median {np.median(tot):7.2f} us  mean {tot.mean():7.2f}")
ent = t[:, :, 0]
end = t[:, :, 6]
print(f"  entry skew over the 8 waves of a workgroup  median {np.median(ent.max(1) - ent.min(1)):6.2f} us;"
      f" end skew {np.median((end.max(1) - end.min(1))[full]):6.2f} us")
wg_start, wg_end = ent.min(1), end.max(1)
span = wg_end.max() - wg_start.min()
print(f"  recorded span {span:8.1f} us; sum of workgroup durations / span = {float((wg_end - wg_start).sum() / span):6.1f} (CU slots busy)")
