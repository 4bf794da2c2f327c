#!/usr/bin/env python3
"""In-kernel clock of k_scores_rowmax (the one MFMA kernel of the path) on the bench workload.
Needs the DIAGNOSTIC build of the library (stamps around the tile loop; never shipped):
    hipcc <flags of csrc/build.py> -DTOME_DIAG_CLOCK -o scratch/libs/diag.so csrc/tome_kernels.hip
    TOME_HIP_LIB=scratch/libs/diag.so python tools/diag_clock.py
Method (MI355X_MICROARCH.md, DVFS give-back item 6): >= 2 s of back-to-back launches on random data, then
clock = sum(s_memtime deltas) / sum(s_memrealtime deltas) x 100 MHz over the waves of the last launch."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from tome import _abi  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64  # 64 clips: 64*25*8 = 12800 waves fit the stamp buffer
dev = torch.device("cuda", 0)
L = _abi.lib()
if not hasattr(L, "tome_diag_clock"):
    raise SystemExit("this libtome_hip.so is not the diagnostic build (-DTOME_DIAG_CLOCK)")
L.tome_diag_clock.restype = ctypes.c_int
L.tome_diag_clock.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                              ctypes.POINTER(ctypes.c_int64)]
g = torch.Generator(device=dev).manual_seed(0)
t = 1568
qkv = torch.randn(batch, t, 3, bench.HEADS, bench.HEAD_DIM, device=dev, generator=g).bfloat16()
keys = qkv[:, :, 1].permute(0, 2, 1, 3)
t0 = time.perf_counter()
launches = 0
while time.perf_counter() - t0 < 2.5:
    for _ in range(50):
        _abi.match_keys(keys, 16, False, False)
    torch.cuda.synchronize()
    launches += 50
ghz, us, nw = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
assert L.tome_diag_clock(ctypes.byref(ghz), ctypes.byref(us), ctypes.byref(nw)) == 0
flops_per_cycle_cu = 256.0  # v_mfma_f32_32x32x2_f32: 157.3 TFLOP/s / (256 CUs x 2.4 GHz)
peak_at_clock = flops_per_cycle_cu * 256 * ghz.value * 1e9 / 1e12
print(f"batch {batch}: {launches} matchings in {time.perf_counter() - t0:.1f} s; last k_scores_rowmax launch: "
      f"{nw.value} waves, mean wave {us.value:.2f} us, in-kernel clock {ghz.value:.3f} GHz "
      f"-> fp32 MFMA peak at that clock {peak_at_clock:.1f} TFLOP/s (157.3 at 2.4 GHz)")
