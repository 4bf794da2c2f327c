#!/usr/bin/env python3
"""Compressed view of a kernel's ISA from /tmp/tome_kernels.s (made by kernel_usage.py --keep): MFMA runs,
waits, memory ops, branches."""
import re
import sys

s = open('/tmp/tome_kernels.s').read()
pat = sys.argv[1]
name = [n for n in re.findall(r'\.amdhsa_kernel (\S+)', s) if pat in n][0]
i = s.index(name + ':')
j = s.index('s_endpgm', i)
out, run = [], 0
for l in s[i:j].split('\n'):
    t = l.strip()
    if t.startswith('v_mfma'):
        run += 1
        continue
    if run:
        out.append(f'   [mfma x{run}]')
        run = 0
    if re.match(r'(s_waitcnt|scratch_|global_load|global_store|s_cbranch|s_branch|\.LBB|s_barrier|buffer_|ds_)', t):
        out.append(t[:90])
print('\n'.join(out[:int(sys.argv[2]) if len(sys.argv) > 2 else 150]))
