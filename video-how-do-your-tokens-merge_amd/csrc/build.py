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
    t = os.path.getmtime(OUT)
    import glob
    deps = SRC + sorted(glob.glob(os.path.join(HERE, "*.h")))  # every kernel header of this directory
    deps += [os.path.join(PKG, "..", "include", "tome_hip.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, extra=()) -> str:
    if not force and not needs_build():
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    cmds = [[hipcc(), *FLAGS, *extra, "-o", OUT, *SRC],
            [hipcc(), *FLAGS, *extra, "-DTOME_PROFILE_HOOKS", "-o", OUT_PROF, *SRC]]
    if verbose:
        for cmd in cmds:
            print(" ".join(cmd), flush=True)
    procs = [subprocess.Popen(cmd) for cmd in cmds]  # the two builds side by side
    for proc, cmd in zip(procs, cmds):
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True,
                extra=["-Rpass-analysis=kernel-resource-usage"] if "--usage" in sys.argv else ()))
