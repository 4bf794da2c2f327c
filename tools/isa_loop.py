#!/usr/bin/env python3
"""Print the instructions of the n-th inner loop of a kernel from /tmp/tome_kernels.s (tools/kernel_usage.py --keep),
one per line, with run-length compression of repeated opcodes.   python tools/isa_loop.py <kernel substring> [loop#] [--raw]"""
import re
import sys

s = open('/tmp/tome_kernels.s').read()
pat = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 0
name = [n for n in re.findall(r'\.amdhsa_kernel (\S+)', s) if all(p in n for p in pat.split(','))][0]
i = s.index(name + ':')
j = s.index('s_endpgm', i)
body = s[i:j].split('\n')
hdr = [k for k, l in enumerate(body) if 'Loop Header' in l and '=>' in l]
k0 = hdr[which]
label = body[k0].strip().split(':')[0]
# loop end: last branch back to the header label
end = max(k for k, l in enumerate(body) if re.search(r's_c?branch\w* ' + re.escape(label) + r'\b', l))
print(name, 'loop', which, 'lines', k0, end)
prev, cnt = None, 0
for l in body[k0:end + 1]:
    t = l.strip()
    if not t or t.startswith(';'):
        continue
    if '--raw' in sys.argv:
        print(t[:110])
        continue
    op = t.split(' ')[0]
    if op == prev:
        cnt += 1
        continue
    if prev:
        print(f'{prev} x{cnt}' if cnt > 1 else prev)
    prev, cnt = op, 1
print(f'{prev} x{cnt}' if cnt > 1 else prev)
