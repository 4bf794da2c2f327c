#!/usr/bin/env python3
"""Compile the HIP kernels + C ABI into lib/libtome_hip.so for gfx950 (MI355X).

hipcc cross-compiles without a GPU, so this runs in the build container; the .so is git-ignored
but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "tome_kernels.hip")]
OUT_DIR = os.path.join(PKG, "lib")
OUT = os.path.join(OUT_DIR, "libtome_hip.so")
# measurement build (bench.py's stage timing only): the same kernels + the tome_profile_* hooks of include/tome_hip.h
OUT_PROF = os.path.join(OUT_DIR, "libtome_hip_prof.so")

FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-shared",
    # the arithmetic contract needs separate IEEE mul/add/div; fused ops are written explicitly
    "-ffp-contract=off",
    "-fno-fast-math",
    "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-Wall",
    "-Wno-unused-function",
]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm; set HIPCC=/path/to/hipcc)")


def needs_build() -> bool:
    if not os.path.exists(OUT) or not os.path.exists(OUT_PROF):
        return True
    # the OLDER of the two outputs decides: a measurement build left behind by an earlier, interrupted or failed run
    # must not be taken for current because the product library next to it is
    t = min(os.path.getmtime(OUT), os.path.getmtime(OUT_PROF))
    import glob
    deps = SRC + sorted(glob.glob(os.path.join(HERE, "*.h")))  # every kernel header of this directory
    deps += [os.path.join(PKG, "..", "include", "tome_hip.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, extra=()) -> str:
    if not force and not needs_build():
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    # each compiler writes to a temporary name; the outputs are put in place (os.replace: atomic) only when BOTH
    # builds have succeeded, so no half-written or stale library is ever bound
    tmp = [OUT + f".tmp{os.getpid()}", OUT_PROF + f".tmp{os.getpid()}"]
    cmds = [[hipcc(), *FLAGS, *extra, "-o", tmp[0], *SRC],
            [hipcc(), *FLAGS, *extra, "-DTOME_PROFILE_HOOKS", "-o", tmp[1], *SRC]]
    if verbose:
        for cmd, final in zip(cmds, (OUT, OUT_PROF)):
            print(" ".join(cmd[:-2] + [final] + cmd[-1:]), flush=True)
    procs = []
    try:
        for cmd in cmds:  # the two builds side by side
            procs.append(subprocess.Popen(cmd))
        codes = [proc.wait() for proc in procs]  # every compiler is waited for before anything is decided
        for code, cmd in zip(codes, cmds):
            if code != 0:
                raise subprocess.CalledProcessError(code, cmd)
        for t, final in zip(tmp, (OUT, OUT_PROF)):
            os.replace(t, final)
    finally:
        for proc in procs:  # (an exception above, e.g. KeyboardInterrupt: no compiler is left running)
            if proc.poll() is None:
                proc.kill()
                proc.wait()
        for t in tmp:
            if os.path.exists(t):
                os.remove(t)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True,
                extra=["-Rpass-analysis=kernel-resource-usage"] if "--usage" in sys.argv else ()))
