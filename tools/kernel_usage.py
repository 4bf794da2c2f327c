#!/usr/bin/env python3
"""Print VGPR/SGPR/LDS/scratch and a few instruction counts of every kernel in csrc/tome_kernels.hip
(compiles to assembly with hipcc; no GPU needed)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "video-how-do-your-tokens-merge_amd", "csrc", "tome_kernels.hip")  # includes tome_*.h


def main():
    pats = [a for a in sys.argv[1:] if not a.startswith("-")]
    pat = pats[0] if pats else ""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                        "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt", "-S", "--cuda-device-only", "-o", out, SRC]
                       + [a for a in sys.argv[1:] if a.startswith("-D")], check=True,
                       stderr=subprocess.DEVNULL)
        s = open(out).read()
        if "--keep" in sys.argv:
            open("/tmp/tome_kernels.s", "w").write(s)
    demangle = subprocess.run(["c++filt"], input="\n".join(
        re.findall(r"\.amdhsa_kernel (\S+)", s)), capture_output=True, text=True).stdout.split("\n")
    names = re.findall(r"\.amdhsa_kernel (\S+)", s)
    pretty = dict(zip(names, demangle))
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
        name, body = m.group(1), m.group(2)
        short = re.sub(r"\(.*", "", pretty.get(name, name))
        if pat and pat not in short:
            continue
        g = lambda k: re.search(r"\.amdhsa_%s (\d+)" % k, body).group(1)
        i = s.index(name + ":")
        code = s[i:s.index("s_endpgm", i)]
        print(f"{short[:70]:70s} vgpr {g('next_free_vgpr'):>3} sgpr {g('next_free_sgpr'):>3} lds {g('group_segment_fixed_size'):>6} "
              f"scratch {g('private_segment_fixed_size'):>4} | mfma {code.count('v_mfma'):3d} gload {len(re.findall(r'global_load', code)):3d} "
              f"gstore {len(re.findall(r'global_store', code)):3d} lines {code.count(chr(10)):5d}")


if __name__ == "__main__":
    main()
